// Bank behaviour of ds_read_b128 on gfx950 for the fragment layouts of the fused GraphConvolution kernels: a [16 rows][64 B] tile per K step,
// lane (li = lane & 15, lq = lane >> 4) reads the 16 bytes of row li, k group lq, at row * 64 + 16 * swz(row, lq).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/lds_b128_probe.hip -o tools/probe/lds_b128_probe && tools/probe/lds_b128_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int P>
__device__ __forceinline__ int off_of(int lane) {
    const int li = lane & 15, lq = lane >> 4;
    if (P == 0) return li * 64 + (((lq + 2 * (li >> 3)) & 3) << 4);      // gf_lds_off as it is
    if (P == 1) return li * 64 + (((lq + (li >> 1)) & 3) << 4);          // rows 2 apart rotate by one group
    if (P == 2) return lane * 16;                                        // contiguous
    if (P == 3) return li * 64 + (lq << 4);                              // no swizzle
    return li * 64 + (((lq + li) & 3) << 4);                             // rows 1 apart rotate
}
template <int P>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int iters) {
    extern __shared__ unsigned char sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 1024) reinterpret_cast<uint32_t*>(sm)[i] = i;
    __syncthreads();
    const int o = off_of<P>(lane);
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 7; ++c) {                                    // seven fragments per step, as a wave of the stack kernel reads
            const u32x4 v = *reinterpret_cast<const u32x4*>(sm + ((w & 3) * 5 + c) * 1024 % (36 * 1024) + o);
            acc += v;
        }
        asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    if (lane == 0) { out[2 * w] = static_cast<uint32_t>(t1 - t0); out[2 * w + 1] = acc.x + acc.y + acc.z + acc.w; }
}
template <int P>
void run(const char* name, int waves) {
    uint32_t* d; hipMalloc(&d, 256);
    const int iters = 2000;
    k<P><<<1, 64 * waves, 40 * 1024>>>(d, iters);
    hipDeviceSynchronize();
    k<P><<<1, 64 * waves, 40 * 1024>>>(d, iters);
    hipDeviceSynchronize();
    uint32_t h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    uint32_t mx = 0; for (int w = 0; w < waves; ++w) mx = h[2 * w] > mx ? h[2 * w] : mx;
    const double bytes = 1024.0 * 7 * iters * waves;
    printf("%-34s %2d waves: %8u cycles (s_memtime units x 1), %.1f B per cycle of the slowest wave's clock\n", name, waves, mx, bytes / mx);
    hipFree(d);
}
int main() {
    for (int waves : {1, 4, 16}) {
        if (waves == 1) { run<0>("gf_lds_off (lq + 2 (li >> 3))", 1); run<1>("(lq + (li >> 1)) & 3", 1); run<2>("contiguous lane * 16", 1); run<3>("no swizzle", 1); run<4>("(lq + li) & 3", 1); }
        if (waves == 4) { run<0>("gf_lds_off (lq + 2 (li >> 3))", 4); run<1>("(lq + (li >> 1)) & 3", 4); run<2>("contiguous lane * 16", 4); run<3>("no swizzle", 4); run<4>("(lq + li) & 3", 4); }
        if (waves == 16) { run<0>("gf_lds_off (lq + 2 (li >> 3))", 16); run<1>("(lq + (li >> 1)) & 3", 16); run<2>("contiguous lane * 16", 16); run<3>("no swizzle", 16); run<4>("(lq + li) & 3", 16); }
    }
    return 0;
}
