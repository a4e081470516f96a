// probe: data movement of v_permlane32_swap / v_permlane16_swap (gfx950) — prints, per lane, which (register, lane) each result holds
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r32[0]; out[64 + threadIdx.x] = r32[1]; out[128 + threadIdx.x] = r16[0]; out[192 + threadIdx.x] = r16[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
    for (int q = 0; q < 4; ++q) { printf("%s:", names[q]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * q + l]); printf("\n"); }
    return 0;
}
