// K4 — fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak),
// with operand addressing general enough for every projection of the GAT layer and its
// backward (gathered rows, head-major segmented layouts, permuted output rows).  Replaces the
// `self.a.mm(edge_h)` GEMM of GAT/layers.py:129-137 (split algebraically into node and edge parts,
// SURVEY.md 8a/G4) and the autograd GEMMs derived from it.
//
// Block tile 128x128x16, 256 threads = 4 waves (2x2), each wave 2x2 MFMA tiles of 32x32.
// Global -> registers -> LDS staging with register prefetch of the next K tile; LDS tiles are
// k-major ([16][128+4]) so both MFMA operand reads are conflict-free ds_read_b32.
#include "recon_common.h"

namespace recon {
namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132, NT = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
    OperandDesc A, B;
    OutputDesc C;
    int32_t M, N, K;
    int32_t k_per_split;     // multiple of BK
    float* partial;          // non-null => write plain [z][M][N]
};

__device__ __forceinline__ int64_t major_off(const OperandDesc& d, int32_t i) {
    if (d.gather) return static_cast<int64_t>(d.gather[i]) * d.S1;
    if (i < d.P) return static_cast<int64_t>(i) * d.S1;
    return static_cast<int64_t>(i % d.P) * d.S1 + static_cast<int64_t>(i / d.P) * d.S2;
}
__device__ __forceinline__ int64_t minor_off(int32_t Dseg, int64_t Sseg, int32_t j) {
    if (j < Dseg) return j;
    return static_cast<int64_t>(j % Dseg) + static_cast<int64_t>(j / Dseg) * Sseg;
}

// ---- tile loaders: fill r[2][4] (two passes of four floats per thread) ---------------------
// k-minor operand: tile is 128 (mn) x 16 (k); thread -> row t>>2 (+64), k quad (t&3)*4
template <int VEC>
__device__ __forceinline__ void load_kminor(const OperandDesc& d, int32_t mn0, int32_t mn_ext, int32_t k0, int32_t k_end,
                                            const int64_t* rowoff, float (&r)[2][4]) {
    const int t = threadIdx.x;
    const int kq = k0 + (t & 3) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int mn = mn0 + (t >> 2) + 64 * p;
        const bool rv = mn < mn_ext;
        if constexpr (VEC == 4) {
            if (rv && kq < k_end) {
                const float4 v = *reinterpret_cast<const float4*>(d.base + rowoff[p] + minor_off(d.Dseg, d.Sseg, kq));
                r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
            } else {
                r[p][0] = r[p][1] = r[p][2] = r[p][3] = 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[p][j] = (rv && kq + j < k_end) ? d.base[rowoff[p] + minor_off(d.Dseg, d.Sseg, kq + j)] : 0.f;
        }
    }
}
__device__ __forceinline__ void store_kminor(float (*T)[LDT], const float (&r)[2][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) T[(t & 3) * 4 + j][(t >> 2) + 64 * p] = r[p][j];
}
// mn-minor operand: tile is 16 (k) x 128 (mn); thread -> k row t>>5 (+8), mn quad (t&31)*4
template <int VEC>
__device__ __forceinline__ void load_mnminor(const OperandDesc& d, int32_t mn0, int32_t mn_ext, int32_t k0, int32_t k_end,
                                             const int64_t* coloff, float (&r)[2][4]) {
    const int t = threadIdx.x;
    const int mn = mn0 + (t & 31) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int k = k0 + (t >> 5) + 8 * p;
        if (k < k_end) {
            const int64_t ro = major_off(d, k);
            if constexpr (VEC == 4) {
                if (mn < mn_ext) {
                    const float4 v = *reinterpret_cast<const float4*>(d.base + ro + coloff[0]);
                    r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
                } else {
                    r[p][0] = r[p][1] = r[p][2] = r[p][3] = 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[p][j] = (mn + j < mn_ext) ? d.base[ro + coloff[j]] : 0.f;
            }
        } else {
            r[p][0] = r[p][1] = r[p][2] = r[p][3] = 0.f;
        }
    }
}
__device__ __forceinline__ void store_mnminor(float (*T)[LDT], const float (&r)[2][4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; ++p)
        *reinterpret_cast<float4*>(&T[(t >> 5) + 8 * p][(t & 31) * 4]) = make_float4(r[p][0], r[p][1], r[p][2], r[p][3]);
}

template <bool A_KMINOR, bool B_KMINOR, int VEC>
__global__ void __launch_bounds__(NT) k_gemm_f32(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float As[BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);

    // per-thread address pieces that do not change along K
    int64_t a_fix[4], b_fix[4];
    if constexpr (A_KMINOR) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int m = m0 + (t >> 2) + 64 * q; a_fix[q] = (m < p.M) ? major_off(p.A, m) : 0; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int m = m0 + (t & 31) * 4 + j; a_fix[j] = (m < p.M) ? minor_off(p.A.Dseg, p.A.Sseg, m) : 0; }
    }
    if constexpr (B_KMINOR) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int n = n0 + (t >> 2) + 64 * q; b_fix[q] = (n < p.N) ? major_off(p.B, n) : 0; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int n = n0 + (t & 31) * 4 + j; b_fix[j] = (n < p.N) ? minor_off(p.B.Dseg, p.B.Sseg, n) : 0; }
    }

    float ra[2][4], rb[2][4];
    auto load_tiles = [&](int k0) {
        if constexpr (A_KMINOR) load_kminor<VEC>(p.A, m0, p.M, k0, k_end, a_fix, ra);
        else load_mnminor<VEC>(p.A, m0, p.M, k0, k_end, a_fix, ra);
        if constexpr (B_KMINOR) load_kminor<VEC>(p.B, n0, p.N, k0, k_end, b_fix, rb);
        else load_mnminor<VEC>(p.B, n0, p.N, k0, k_end, b_fix, rb);
    };
    auto store_tiles = [&]() {
        if constexpr (A_KMINOR) store_kminor(As, ra); else store_mnminor(As, ra);
        if constexpr (B_KMINOR) store_kminor(Bs, rb); else store_mnminor(Bs, rb);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int mb = (wid >> 1) * 64, nb = (wid & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;

    if (k_begin < k_end) {
        load_tiles(k_begin);
        store_tiles();
    }
    __syncthreads();
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) load_tiles(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kk = 2 * ks + lk;
            const float a0 = As[kk][mb + lr], a1 = As[kk][mb + 32 + lr];
            const float b0 = Bs[kk][nb + lr], b1 = Bs[kk][nb + 32 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (more) { store_tiles(); }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + nb + j * 32 + lr;
        if (col >= p.N) continue;
        int64_t coff;
        float* base;
        if (p.partial) { base = p.partial + static_cast<int64_t>(blockIdx.z) * p.M * p.N; coff = col; }
        else { base = p.C.base; coff = minor_off(p.C.Dseg, p.C.Sseg, col); }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + mb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row >= p.M) continue;
                int64_t roff;
                if (p.partial) roff = static_cast<int64_t>(row) * p.N;
                else if (p.C.scatter) roff = static_cast<int64_t>(p.C.scatter[row]) * p.C.S1;
                else if (row < p.C.P) roff = static_cast<int64_t>(row) * p.C.S1;
                else roff = static_cast<int64_t>(row % p.C.P) * p.C.S1 + static_cast<int64_t>(row / p.C.P) * p.C.S2;
                base[roff + coff] = acc[i][j][r];
            }
        }
    }
}

// deterministic second pass of split-K: C = sum_z partial[z] (fixed order), written through C's addressing
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ partial, int32_t splits, int32_t M,
                                                       int32_t N, const OutputDesc C) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t MN = static_cast<int64_t>(M) * N;
    if (idx >= MN) return;
    const int row = static_cast<int>(idx / N), col = static_cast<int>(idx % N);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[z * MN + idx];
    int64_t roff;
    if (C.scatter) roff = static_cast<int64_t>(C.scatter[row]) * C.S1;
    else if (row < C.P) roff = static_cast<int64_t>(row) * C.S1;
    else roff = static_cast<int64_t>(row % C.P) * C.S1 + static_cast<int64_t>(row / C.P) * C.S2;
    C.base[roff + minor_off(C.Dseg, C.Sseg, col)] = s;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool operand_vec4(const OperandDesc& d, int32_t minor_extent) {
    if (!aligned16(d.base)) return false;
    if ((d.S1 & 3) || (d.S2 & 3) || (d.Sseg & 3)) return false;
    if (d.Dseg < minor_extent && (d.Dseg & 3)) return false;
    return (minor_extent & 3) == 0;
}

template <bool AK, bool BK_, int VEC>
void launch(const GemmArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((k_gemm_f32<AK, BK_, VEC>), grid, dim3(NT), 0, st, a);
}

}  // namespace

int gemm_pick_split_k(int32_t M, int32_t N, int32_t K) {
    const int64_t tiles = ceil_div64(M, BM) * ceil_div64(N, BN);
    if (tiles >= 192) return 1;
    int64_t s = ceil_div64(512, tiles);
    const int64_t max_s = (K / (8 * BK)) > 0 ? K / (8 * BK) : 1;   // at least 8 K tiles per split
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    return static_cast<int>(s < 1 ? 1 : s);
}

int gemm_f32(int32_t M, int32_t N, int32_t K, const OperandDesc& A, bool a_k_minor, const OperandDesc& B, bool b_k_minor,
             const OutputDesc& C, int32_t split_k, float* partial, hipStream_t st) {
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    if (!A.base || !B.base || !C.base) return RECON_ERR_INVALID;
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && !partial) return RECON_ERR_INVALID;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K;
    int64_t kps = ceil_div64(K > 0 ? K : 1, split_k);
    kps = ceil_div64(kps, BK) * BK;
    a.k_per_split = static_cast<int32_t>(kps);
    split_k = static_cast<int32_t>(ceil_div64(K > 0 ? K : 1, kps));
    a.partial = split_k > 1 ? partial : nullptr;
    const bool v4 = operand_vec4(A, a_k_minor ? K : M) && operand_vec4(B, b_k_minor ? K : N);
    dim3 grid(static_cast<unsigned>(ceil_div64(N, BN)), static_cast<unsigned>(ceil_div64(M, BM)), static_cast<unsigned>(split_k));
    if (a_k_minor && b_k_minor) { if (v4) launch<true, true, 4>(a, grid, st); else launch<true, true, 1>(a, grid, st); }
    else if (a_k_minor && !b_k_minor) { if (v4) launch<true, false, 4>(a, grid, st); else launch<true, false, 1>(a, grid, st); }
    else if (!a_k_minor && !b_k_minor) { if (v4) launch<false, false, 4>(a, grid, st); else launch<false, false, 1>(a, grid, st); }
    else return RECON_ERR_UNSUPPORTED;
    if (split_k > 1) {
        const int64_t MN = static_cast<int64_t>(M) * N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3(static_cast<unsigned>(ceil_div64(MN, 256))), dim3(256), 0, st, partial, split_k,
                           M, N, C);
    }
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace recon

extern "C" int recon_sgemm(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                           int32_t b_is_nk, float* C, int32_t ldc, recon_stream_t stream) {
    using namespace recon;
    return gemm_f32(M, N, K, plain_operand(A, lda), true, plain_operand(B, ldb), b_is_nk != 0, plain_output(C, ldc), 1,
                    nullptr, as_stream(stream));
}
