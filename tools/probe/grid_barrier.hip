// probe: cost of a software grid barrier (agent-scope release / acquire around a counter) among G co-resident workgroups on gfx950,
// against the same phases as separate dependent launches.   hipcc --offload-arch=gfx950 -O3 -o /tmp/gb tools/probe/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ unsigned g_ctr[2];
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            if (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}
// each phase: block b writes buf[phase][b*256+t] = f(buf[phase-1][((b+1)%G)*256+t])  (cross-block dependency)
__global__ void __launch_bounds__(256) k_fused(int* buf, int G, int phases) {
    const int b = blockIdx.x, t = threadIdx.x;
    buf[b * 256 + t] = b + t;
    for (int p = 1; p < phases; ++p) {
        if (!grid_barrier(&g_ctr[0], static_cast<unsigned>(p) * G)) return;
        buf[p * G * 256 + b * 256 + t] = buf[(p - 1) * G * 256 + ((b + 1) % G) * 256 + t] + 1;
    }
    __syncthreads();
    if (t == 0) {
        const unsigned e = __hip_atomic_fetch_add(&g_ctr[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (e == static_cast<unsigned>(G) - 1) { g_ctr[0] = 0; g_ctr[1] = 0; __threadfence(); }
    }
}
__global__ void __launch_bounds__(256) k_phase(int* buf, int G, int p) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (p == 0) buf[b * 256 + t] = b + t;
    else buf[p * G * 256 + b * 256 + t] = buf[(p - 1) * G * 256 + ((b + 1) % G) * 256 + t] + 1;
}
int main(int argc, char** argv) {
    const int phases = 7, iters = 200;
    for (int G : {32, 128, 256}) {
        int* buf; hipMalloc(&buf, sizeof(int) * phases * G * 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_fused, dim3(G), dim3(256), 0, 0, buf, G, phases);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_fused, dim3(G), dim3(256), 0, 0, buf, G, phases);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        int* h = (int*)malloc(sizeof(int) * G * 256);
        hipMemcpy(h, buf + (phases - 1) * G * 256, sizeof(int) * G * 256, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int b = 0; b < G; ++b) for (int t = 0; t < 256; ++t) if (h[b * 256 + t] != (b + phases - 1) % G + t + phases - 1) ++bad;
        printf("G=%d fused (%d phases, %d barriers): %.2f us per launch, wrong=%d\n", G, phases, phases - 1, 1e3 * ms / iters, bad);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(G), dim3(256), 0, 0, buf, G, p);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("G=%d as %d dependent launches: %.2f us per chain\n", G, phases, 1e3 * ms / iters);
        hipFree(buf); free(h);
    }
    return 0;
}
