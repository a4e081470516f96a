// Probe of ds_read_b64_tr_b16 semantics (build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe; run on the GPU box): every lane passes the address of 4 contiguous 16-bit elements
// (row i'>>2, column quad i'&3 of a [4][16] block with row stride S); prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int S = 40;   // row stride in elements
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short L[8192];
    const int t = threadIdx.x;
    for (int i = t; i < 8192; i += 64) L[i] = i;
    __syncthreads();
    const int ip = t & 15, g = t >> 4;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(&L[g * 1000 + (ip >> 2) * S + 4 * (ip & 3)]));
    for (int j = 0; j < 4; ++j) out[t * 4 + j] = r[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int t = 0; t < 64; ++t) printf("lane %2d: %5d %5d %5d %5d   expect col %d: %d %d %d %d\n", t, h[4*t], h[4*t+1], h[4*t+2], h[4*t+3],
        t & 15, (t>>4)*1000 + (t&15), (t>>4)*1000 + S + (t&15), (t>>4)*1000 + 2*S + (t&15), (t>>4)*1000 + 3*S + (t&15));
    return 0;
}
