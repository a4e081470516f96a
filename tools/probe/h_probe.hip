// probe: rows_max / wave_max (DPP + permlane swaps) and out-of-range buffer loads, as prop_h.hip uses them
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../recon_amd/csrc/recon_common.h"
using namespace recon;
// __builtin_bit_cast applied directly to a vector ELEMENT (a[1], q.w) reads element 0 on this compiler: go through a scalar copy
__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float rows_max(float m) {
    const uint32_t u = __builtin_bit_cast(uint32_t, m);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    m = fmaxf(as_f(a[0]), as_f(a[1]));
    const uint32_t w = __builtin_bit_cast(uint32_t, m);
    auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(as_f(c[0]), as_f(c[1]));
}
template <int CTRL> __device__ __forceinline__ float dpp_max(float v) { return fmaxf(v, dpp_mov<CTRL>(v)); }
__device__ __forceinline__ float wave_max(float m) {
    m = dpp_max<0xB1>(m); m = dpp_max<0x4E>(m); m = dpp_max<0x141>(m); m = dpp_max<0x140>(m);
    return rows_max(m);
}
__global__ void k(const float* in, float* out, const float* buf, int nbytes) {
    const int l = threadIdx.x;
    out[l] = rows_max(in[l]);
    out[64 + l] = wave_max(in[l]);
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, nbytes, 0x00020000);
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, l < 32 ? 16u * l : 0xfffffff0u, 0, 0);
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, 16u * l + 64, 0, 0);          // lanes past nbytes: zeros
    out[128 + l] = as_f(a.x) + as_f(a.w);
    out[192 + l] = as_f(b.x);
}
int main() {
    float h[64], o[256], hb[4096];
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64);
    for (int i = 0; i < 4096; ++i) hb[i] = 1.0f + i;
    float *d, *dout, *dbuf; hipMalloc(&d, 256); hipMalloc(&dout, 1024); hipMalloc(&dbuf, sizeof(hb));
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice); hipMemcpy(dbuf, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout, dbuf, 40 * 16);
    hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float rm = 0, wm = 0;
        for (int q = 0; q < 4; ++q) rm = fmaxf(rm, h[(l & 15) + 16 * q]);
        for (int q = 0; q < 64; ++q) wm = fmaxf(wm, h[q]);
        if (o[l] != rm) { if (bad++ < 8) printf("rows_max lane %d: got %g want %g\n", l, o[l], rm); }
        if (o[64 + l] != wm) { if (bad++ < 8) printf("wave_max lane %d: got %g want %g\n", l, o[64 + l], wm); }
        float ea = l < 32 ? hb[4 * l] + hb[4 * l + 3] : 0.f;
        float eb = (16 * l + 64 + 16 <= 40 * 16) ? hb[4 * l + 16] : 0.f;
        if (o[128 + l] != ea) { if (bad++ < 16) printf("oob-a lane %d: got %g want %g\n", l, o[128 + l], ea); }
        if (o[192 + l] != eb) { if (bad++ < 16) printf("oob-b lane %d: got %g want %g\n", l, o[192 + l], eb); }
    }
    printf("bad = %d\n", bad);
    return 0;
}
