// probe: group_sum<G> / multi_sum<NV> of recon_common.h against a host reference (integer-valued floats: exact), and the raw data
// movement of v_permlane32_swap / v_permlane16_swap.   hipcc --offload-arch=gfx950 -O3 -I recon_amd/csrc -I include -o tools/probe/multi_sum_probe tools/probe/multi_sum_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "recon_common.h"
using namespace recon;
__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[j * 64 + lane];
    out[0 * 64 + lane] = group_sum<64>(v[0]);
    out[1 * 64 + lane] = group_sum<32>(v[0]);
    out[2 * 64 + lane] = group_sum<16>(v[0]);
    out[3 * 64 + lane] = group_sum<8>(v[0]);
    float a2[2] = {v[0], v[1]}, a4[4] = {v[0], v[1], v[2], v[3]}, a8[8];
    for (int j = 0; j < 8; ++j) a8[j] = v[j];
    out[4 * 64 + lane] = multi_sum<2>(a2, lane);
    out[5 * 64 + lane] = multi_sum<4>(a4, lane);
    out[6 * 64 + lane] = multi_sum<8>(a8, lane);
    const auto r32 = __builtin_amdgcn_permlane32_swap(1000u + lane, 2000u + lane, false, false);
    const auto r16 = __builtin_amdgcn_permlane16_swap(1000u + lane, 2000u + lane, false, false);
    const unsigned x0 = r32[0], x1 = r32[1], y0 = r16[0], y1 = r16[1];
    out[7 * 64 + lane] = x0; out[8 * 64 + lane] = x1; out[9 * 64 + lane] = y0; out[10 * 64 + lane] = y1;
}
int main() {
    float h[8 * 64], o[11 * 64];
    for (int i = 0; i < 8 * 64; ++i) h[i] = (i * 7919 % 101) - 50;
    float *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    auto gsum = [&](int j, int l, int G) { float s = 0; for (int t = 0; t < G; ++t) s += h[j * 64 + (l / G) * G + t]; return s; };
    int bad = 0;
    const int Gs[4] = {64, 32, 16, 8};
    for (int q = 0; q < 4; ++q) for (int l = 0; l < 64; ++l) if (o[q * 64 + l] != gsum(0, l, Gs[q])) { if (bad < 8) printf("group_sum<%d> lane %d: %g != %g\n", Gs[q], l, o[q * 64 + l], gsum(0, l, Gs[q])); ++bad; }
    const int NVs[3] = {2, 4, 8};
    for (int q = 0; q < 3; ++q) for (int l = 0; l < 64; ++l) { const int nv = NVs[q], j = l / (64 / nv); if (o[(4 + q) * 64 + l] != gsum(j, 0, 64)) { if (bad < 16) printf("multi_sum<%d> lane %d: %g != %g\n", nv, l, o[(4 + q) * 64 + l], gsum(j, 0, 64)); ++bad; } }
    const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
    for (int q = 0; q < 4; ++q) { printf("%s:", names[q]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%g", l, o[(7 + q) * 64 + l]); printf("\n"); }
    printf("%s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
