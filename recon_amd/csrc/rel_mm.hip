// Per-row matrix selected by a relation id — the GAT_sep_space variant's entity -> relation-space map (GAT_sep_space/main.py:359-364,
// :372-377; GAT_sep_space/models.py:316-320):  out[t] = x[t] . W[rel[t]],  x [T, D], W [R, D, Dout], rel [T].  The reference gathers
// W[rel] as a [T, D, Dout] tensor (160 KB per triple at D = 200) and calls torch.bmm; here rows are walked in relation order (a stable
// argsort done by the caller on the device, `order`), 16 sorted rows per workgroup: the rows of a tile that share a relation share ONE
// pass over that relation's matrix (a thread owns an output column, the matrix element it loads feeds up to 16 rows), so W is read
// about T / 16 times instead of T times and nothing of size T x D x D exists.  fp32 FMA arithmetic (the products are tens to hundreds of
// MFLOP per call: latency bound, like recon_sgemm_small).  Three launches cover forward and backward:
//   k_rel_rows_mm<false>    out[t]  = x[t] . W[rel[t]]
//   k_rel_rows_mm<true>     g_x[t]  = g[t] . W[rel[t]]^T
//   k_rel_rows_wgrad        g_W[r]  = sum over the rows t of relation r of x[t]^T g[t]    (segments of the sorted order; fixed order: deterministic)
#include "recon_common.h"

namespace recon {
namespace {

constexpr int kRows = 16;

// IN = length of a row of `x` (the contraction), OUT = length of a row of `out`.  TRANS: W[r] is [OUT][IN] and is read transposed.
template <bool TRANS>
__global__ void __launch_bounds__(256) k_rel_rows_mm(const float* __restrict__ x, const int32_t* __restrict__ order, const int64_t* __restrict__ rel,
                                                     const float* __restrict__ W, int32_t T, int32_t IN, int32_t OUT, float* __restrict__ out) {
    extern __shared__ float xs[];                                       // [kRows][IN]
    __shared__ int32_t row_s[kRows], rel_s[kRows];
    const int p0 = blockIdx.x * kRows, nrows = min(kRows, T - p0);
    if (threadIdx.x < kRows) {
        const int t = threadIdx.x < nrows ? order[p0 + threadIdx.x] : -1;
        row_s[threadIdx.x] = t;
        rel_s[threadIdx.x] = t >= 0 ? static_cast<int32_t>(rel[t]) : -1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kRows * IN; i += blockDim.x) {
        const int r = i / IN, k = i - r * IN;
        xs[i] = r < nrows ? x[static_cast<int64_t>(row_s[r]) * IN + k] : 0.f;
    }
    __syncthreads();
    const int64_t wsz = static_cast<int64_t>(IN) * OUT;
    for (int j = threadIdx.x; j < OUT; j += blockDim.x) {
        float acc[kRows];
#pragma unroll
        for (int i = 0; i < kRows; ++i) acc[i] = 0.f;
        int i0 = 0;
        while (i0 < nrows) {                                            // runs of equal relation (block-uniform)
            const int r = rel_s[i0];
            int i1 = i0 + 1;
            while (i1 < nrows && rel_s[i1] == r) ++i1;
            const float* Wr = W + r * wsz;
            float m[kRows];                                             // 1 for the rows of this run
#pragma unroll
            for (int i = 0; i < kRows; ++i) m[i] = (i >= i0 && i < i1) ? 1.f : 0.f;
#pragma unroll 8
            for (int k = 0; k < IN; ++k) {
                const float w = TRANS ? Wr[static_cast<int64_t>(j) * IN + k] : Wr[static_cast<int64_t>(k) * OUT + j];
#pragma unroll
                for (int i = 0; i < kRows; ++i) acc[i] = fmaf(xs[i * IN + k] * m[i], w, acc[i]);
            }
            i0 = i1;
        }
#pragma unroll
        for (int i = 0; i < kRows; ++i)
            if (i < nrows) out[static_cast<int64_t>(row_s[i]) * OUT + j] = acc[i];
    }
}

// block (relation r, 16 rows k0 .. k0 + 15 of W[r]); thread = output column j
__global__ void __launch_bounds__(256) k_rel_rows_wgrad(const float* __restrict__ x, const float* __restrict__ g, const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ seg_ptr, int32_t D, int32_t OUT, float* __restrict__ gW) {
    const int r = blockIdx.x, k0 = blockIdx.y * kRows;
    const int lo = seg_ptr[r], hi = seg_ptr[r + 1];
    __shared__ float xk[8][kRows];
    __shared__ int32_t rows[8];
    float acc[(512 / 256)][kRows];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < kRows; ++i) acc[q][i] = 0.f;
    for (int p = lo; p < hi; p += 8) {                                  // eight rows of the segment per round
        __syncthreads();
        if (threadIdx.x < 8 * kRows) {
            const int pr = threadIdx.x / kRows, i = threadIdx.x % kRows;
            const int t = p + pr < hi ? order[p + pr] : -1;
            if (i == 0) rows[pr] = t;
            xk[pr][i] = (t >= 0 && k0 + i < D) ? x[static_cast<int64_t>(t) * D + k0 + i] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = threadIdx.x + 256 * q;
            if (j < OUT) {
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    const int t = rows[pr];
                    const float gj = t >= 0 ? g[static_cast<int64_t>(t) * OUT + j] : 0.f;
#pragma unroll
                    for (int i = 0; i < kRows; ++i) acc[q][i] = fmaf(xk[pr][i], gj, acc[q][i]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = threadIdx.x + 256 * q;
        if (j < OUT)
#pragma unroll
            for (int i = 0; i < kRows; ++i)
                if (k0 + i < D) gW[(static_cast<int64_t>(r) * D + k0 + i) * OUT + j] = acc[q][i];
    }
}

}  // namespace
}  // namespace recon

extern "C" int recon_rel_rows_mm(const float* x, const int32_t* order, const int64_t* rel, const float* W, int32_t T, int32_t in_dim, int32_t out_dim,
                                 int32_t transpose_w, float* out, recon_stream_t stream) {
    if (T < 0 || in_dim <= 0 || out_dim <= 0 || in_dim > 1024) return RECON_ERR_INVALID;
    if (T == 0) return RECON_OK;
    if (!x || !order || !rel || !W || !out) return RECON_ERR_INVALID;
    const dim3 grid(static_cast<unsigned>((T + recon::kRows - 1) / recon::kRows));
    const size_t lds = sizeof(float) * recon::kRows * in_dim;
    hipStream_t st = as_stream(stream);
    if (lds > 48 * 1024) {                                               // in_dim > 768: above the default dynamic-LDS limit
        const void* kern = transpose_w ? reinterpret_cast<const void*>(recon::k_rel_rows_mm<true>) : reinterpret_cast<const void*>(recon::k_rel_rows_mm<false>);
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return RECON_ERR_LAUNCH;
    }
    if (transpose_w) hipLaunchKernelGGL(recon::k_rel_rows_mm<true>, grid, dim3(256), lds, st, x, order, rel, W, T, in_dim, out_dim, out);
    else hipLaunchKernelGGL(recon::k_rel_rows_mm<false>, grid, dim3(256), lds, st, x, order, rel, W, T, in_dim, out_dim, out);
    return hipGetLastError() == hipSuccess ? RECON_OK : RECON_ERR_LAUNCH;
}

extern "C" int recon_rel_rows_mm_wgrad(const float* x, const float* g, const int32_t* order, const int32_t* seg_ptr, int32_t R, int32_t in_dim,
                                       int32_t out_dim, float* gW, recon_stream_t stream) {
    if (R < 0 || in_dim <= 0 || out_dim <= 0) return RECON_ERR_INVALID;
    if (out_dim > 512) return RECON_ERR_UNSUPPORTED;
    if (R == 0) return RECON_OK;
    if (!x || !g || !order || !seg_ptr || !gW) return RECON_ERR_INVALID;
    const dim3 grid(static_cast<unsigned>(R), static_cast<unsigned>((in_dim + recon::kRows - 1) / recon::kRows));
    hipLaunchKernelGGL(recon::k_rel_rows_wgrad, grid, dim3(256), 0, as_stream(stream), x, g, order, seg_ptr, in_dim, out_dim, gW);
    return hipGetLastError() == hipSuccess ? RECON_OK : RECON_ERR_LAUNCH;
}
