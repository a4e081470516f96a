// probe: raw buffer stores with a scalar offset and out-of-range lane offsets (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int nbytes, int soff) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x;
    const int vo = lane < 50 ? lane * 16 : 0x7ffffff0;
    const uint32_t v = 1000 + lane;
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v, v, v}, rsrc, vo, soff, 0);
}
int main() {
    const int n = 4096;
    uint32_t* d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
    for (int soff : {0, 2400, 4800}) {
        hipMemset(d, 0, n * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, reinterpret_cast<float*>(d), n * 4, soff);
        std::vector<uint32_t> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int cnt = 0, first = -1, last = -1;
        for (int i = 0; i < n; ++i) if (h[i]) { ++cnt; if (first < 0) first = i; last = i; }
        printf("soff %d: %d words written, first word %d (=%u), last word %d (=%u)\n", soff, cnt, first, first >= 0 ? h[first] : 0, last, last >= 0 ? h[last] : 0);
    }
    return 0;
}
