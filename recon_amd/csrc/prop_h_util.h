// Device helpers shared by the two-term f16 propagation kernels (prop_h.hip: whole graph per workgroup; prop_hl.hip: wide states,
// channel chunks per workgroup).
#pragma once
#include "prop_common.h"

namespace recon {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// COMPILER HAZARD (hipcc, ROCm 7.2): __builtin_bit_cast applied directly to a vector ELEMENT (q.w, a[1]) yields element 0 — e.g.
// fmaxf(bit_cast(a[0]), bit_cast(a[1])) folds to a[0].  Every element goes through this by-value helper (tools/probe/h_probe.hip).
__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }

__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS traffic is done; vmcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_wait() {      // this wave's LDS writes have landed (same-wave hand-over, no barrier)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("" ::: "memory");
}
// max over lanes l, l ^ 16, l ^ 32, l ^ 48 (the four 16-lane rows) on the VALU: v_permlane{16,32}_swap exchange rows between two
// registers, no LDS round trip as __shfl_xor (ds_bpermute) would take
__device__ __forceinline__ float rows_max(float m) {
    const uint32_t u = __builtin_bit_cast(uint32_t, m);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    m = fmaxf(as_f(a[0]), as_f(a[1]));
    const uint32_t w = __builtin_bit_cast(uint32_t, m);
    auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(as_f(c[0]), as_f(c[1]));
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) { return fmaxf(v, dpp_mov<CTRL>(v)); }
// max over all 64 lanes: DPP inside the rows, row exchange across them
__device__ __forceinline__ float wave_max(float m) {
    m = dpp_max<0xB1>(m);                         // quad_perm [1,0,3,2]
    m = dpp_max<0x4E>(m);                         // quad_perm [2,3,0,1]
    m = dpp_max<0x141>(m);                        // row_half_mirror
    m = dpp_max<0x140>(m);                        // row_mirror
    return rows_max(m);
}

// tanh to ~3e-7 relative without ocml's branchy tanhf (20 activations per lane and hop): odd polynomial below 0.1 (truncation
// error 2e-11 there), (1 - t) / (1 + t) with t = exp(-2|x|) above (1 - t >= 0.18: no cancellation)
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float poly = x * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    const float t = __expf(-2.f * ax);
    const float big = copysignf((1.f - t) * __frcp_rn(1.f + t), x);
    return ax < 0.1f ? poly : big;
}

constexpr uint32_t kOOB = 0xfffffff0u;            // a buffer offset past every num_records: the load returns zeros, the store is dropped

}  // namespace
}  // namespace recon
