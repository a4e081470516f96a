// Shared by the GEMM translation units: argument block and operand address maps.
#pragma once
#include "recon_common.h"

namespace recon {

struct GemmArgs {
    OperandDesc A, B;
    OutputDesc C;
    int32_t M, N, K;
    int32_t k_per_split;     // multiple of the kernel's K tile
    float* partial;          // non-null => write plain [batch][z][M][N]
    int32_t nsplit;          // blockIdx.z = batch_index * nsplit + split_index
    int32_t epilogue;        // GEMM_EPI_*
    int32_t c_vec4;          // 1 => float4 stores of 4 consecutive output columns are legal (alignment, segments)
    int32_t xcd_remap;       // 1 => XCD-aware tile order (xcd_tile in gemm_f32.hip)
    int64_t a_bs, b_bs, c_bs;   // per-batch element offsets of A, B, C
};
enum { GEMM_EPI_NONE = 0, GEMM_EPI_ELU = 1 };

// exp(v) - 1 for v <= 0 as v_exp_f32 on v * log2(e), minus one: three instructions.  Absolute error <= ~1e-7 everywhere (exp(v) <= 1,
// one ulp of it), which is what the layer's 1e-4 parity and the backward's h = log1p(y) need; the RELATIVE error near zero is that of
// the cancellation (a degree-7 polynomial for v > -0.25 kept it at 2e-9 for 9 more instructions per value: s_memtime stamps put the
// projection's epilogue — 104 values per lane, two waves per SIMD in lockstep — at 18 k of a wave's 117 k cycles).  expm1f is a
// ~40-instruction library routine.
__device__ __forceinline__ float expm1_nonpos(float v) { return __expf(v) - 1.f; }
// The exact-fp32 family keeps expm1's RELATIVE accuracy near zero as well (F.elu of the reference is expm1): below 1/16 the series
// v + v^2/2 + v^3/6 + v^4/24 (truncation <= v^4/120 relative = 1.3e-7), the fast form elsewhere (no cancellation there).
__device__ __forceinline__ float expm1_nonpos_exact(float v) {
    const float poly = v * fmaf(v, fmaf(v, fmaf(v, 1.f / 24.f, 1.f / 6.f), 0.5f), 1.f);
    return v > -0.0625f ? poly : __expf(v) - 1.f;
}
template <bool EXACT = false>
__device__ __forceinline__ float gemm_epilogue(float v, int epi) {
    if (epi == GEMM_EPI_ELU) return v > 0.f ? v : (EXACT ? expm1_nonpos_exact(v) : expm1_nonpos(v));
    return v;
}

__device__ __forceinline__ int64_t major_off(const OperandDesc& d, int32_t i) {
    if (d.gather) return static_cast<int64_t>(d.gather[i]) * d.S1;
    if (i < d.P) return static_cast<int64_t>(i) * d.S1;
    return static_cast<int64_t>(i % d.P) * d.S1 + static_cast<int64_t>(i / d.P) * d.S2;
}
__device__ __forceinline__ int64_t minor_off(int32_t Dseg, int64_t Sseg, int32_t j) {
    if (j < Dseg) return j;
    return static_cast<int64_t>(j % Dseg) + static_cast<int64_t>(j / Dseg) * Sseg;
}

// row offset of output row `row` / epilogue store shared by both GEMM kernels
__device__ __forceinline__ int64_t out_row_off(const OutputDesc& C, int32_t row) {
    if (C.scatter) return static_cast<int64_t>(C.scatter[row]) * C.S1;
    if (row < C.P) return static_cast<int64_t>(row) * C.S1;
    return static_cast<int64_t>(row % C.P) * C.S1 + static_cast<int64_t>(row / C.P) * C.S2;
}

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs by linear id, and each XCD has its own L2:
// with the natural order the N tiles that share one A tile (and the M tiles that share one B slab) land on 8
// different L2s and each re-fetches the operand from HBM (PMC: 349 MB for the g_V GEMM whose operands are 56 MB).
// Give XCD c the c-th CONTIGUOUS chunk of the (n fastest, m, batch*split) tile order instead.
struct TileId { int x, y, z; };
__device__ __forceinline__ TileId xcd_tile(int remap) {
    if (!remap) return {static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y), static_cast<int>(blockIdx.z)};
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned c = L & 7u, pos = L >> 3, q = total >> 3, r = total & 7u;
    const unsigned logical = c * q + min(c, r) + pos;
    TileId id;
    id.x = static_cast<int>(logical % gx);
    const unsigned rest = logical / gx;
    id.y = static_cast<int>(rest % gy);
    id.z = static_cast<int>(rest / gy);
    return id;
}



}  // namespace recon
