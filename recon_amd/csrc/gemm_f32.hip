// K4 — fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak),
// with operand addressing general enough for every projection of the GAT layer and its
// backward (gathered rows, head-major segmented layouts, permuted output rows).  Replaces the
// `self.a.mm(edge_h)` GEMM of GAT/layers.py:129-137 (split algebraically into node and edge parts,
// SURVEY.md 8a/G4) and the autograd GEMMs derived from it.
//
// Block tile 128x128x16 (4 waves as 2x2, each 2x2 MFMA tiles of 32x32) or, for outputs 129..208
// columns wide, 128x208x16 on the 16x16x4 MFMA (4 waves stacked along M, each 2x13 tiles, wide LDS
// fragment reads).  Global -> registers -> LDS staging with register prefetch of the next K tile;
// LDS tiles are k-major.  The layer's three large products normally run on the split-precision
// kernels of gemm_bx3.hip; this file serves layout (B), GraphConvolution's weight gradient, operand
// layouts those kernels do not take, and RECON_GEMM_BX3=0.
#include <stdlib.h>
#include "gemm_common.h"

namespace recon {
namespace {

constexpr int NT = 256, BK16 = 16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// One operand tile: W (m or n extent, 128 or 224) x BK, staged global -> registers -> LDS [BK][W+4].
//   K_MINOR : memory is contiguous along k   -> thread item = (row, k quad): 4 floats along k
//   !K_MINOR: memory is contiguous along m/n -> thread item = (k row, mn quad): 4 floats along mn
// SWZ: column index XORed with ((k >> 2) & 3) << 3 (see k_gemm_f32_n208) — makes the transposed K_MINOR stores
// conflict-free and keeps 2/4-float alignment for the wide fragment reads.
// LIN (VEC == 4 only; the host checks that the operand is linear along k: K_MINOR with one k segment, or k-major
// without gather / row wrap): the loader keeps ONE pointer per item and advances it by a constant per K tile, so a
// full tile costs NP unpredicated global_load_dwordx4 + NP 64-bit adds instead of the general address arithmetic
// (~270 non-MFMA instructions per 32 MFMAs in the general form).  Rows beyond the operand's extent read a clamped
// (valid) row: they only reach output rows/columns that are never stored.  The K tail takes the predicated path.
template <int W, bool K_MINOR, int VEC, int BK, int LD_ = W + 4, bool SWZ = false, bool LIN = false>
struct TileLoader {
    static_assert(!LIN || VEC == 4, "linear loader is float4 only");
    static constexpr int LD = LD_;
    static constexpr int QPR = K_MINOR ? BK / 4 : W / 4;          // quads per tile row
    static constexpr int ITEMS = K_MINOR ? W * (BK / 4) : BK * (W / 4);
    static constexpr int NP = (ITEMS + NT - 1) / NT;
    float r[NP][4];
    int64_t fix[LIN ? 1 : NP][K_MINOR || VEC == 4 ? 1 : 4];       // K-invariant address part per item (general form)
    const float* ptr[LIN ? NP : 1];                               // LIN: this item's float4 in the current K tile
    int64_t kstep;

    __device__ __forceinline__ void init(const OperandDesc& d, int32_t mn0, int32_t mn_ext, int32_t k_begin) {
        if constexpr (LIN) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int idx = threadIdx.x + NT * p;
                if constexpr (K_MINOR) {
                    const int mn = min(mn0 + idx / QPR, mn_ext - 1);
                    ptr[p] = d.base + major_off(d, mn) + k_begin + (idx % QPR) * 4;
                } else {
                    const int mn = min(mn0 + (idx % QPR) * 4, mn_ext - 4);
                    ptr[p] = d.base + static_cast<int64_t>(k_begin + min(idx / QPR, BK - 1)) * d.S1 + minor_off(d.Dseg, d.Sseg, mn);
                }
            }
            kstep = K_MINOR ? BK : BK * d.S1;
        } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = threadIdx.x + NT * p;
            if constexpr (K_MINOR) {
                const int mn = mn0 + idx / QPR;
                fix[p][0] = (idx < ITEMS && mn < mn_ext) ? major_off(d, mn) : -1;
            } else {
                const int mn = mn0 + (idx % QPR) * 4;
                if constexpr (VEC == 4) fix[p][0] = (idx < ITEMS && mn < mn_ext) ? minor_off(d.Dseg, d.Sseg, mn) : -1;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fix[p][j] = (idx < ITEMS && mn + j < mn_ext) ? minor_off(d.Dseg, d.Sseg, mn + j) : -1;
                }
            }
        }
        }
    }
    __device__ __forceinline__ void load(const OperandDesc& d, int32_t k0, int32_t k_end) {
        if constexpr (LIN) {
            if (k0 + BK <= k_end) {                                   // uniform: full K tile
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float4 v = *reinterpret_cast<const float4*>(ptr[p]);
                    r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
                    ptr[p] += kstep;
                }
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int idx = threadIdx.x + NT * p;
                    const int k = K_MINOR ? k0 + (idx % QPR) * 4 : k0 + idx / QPR;
                    r[p][0] = r[p][1] = r[p][2] = r[p][3] = 0.f;
                    if (k < k_end) {
                        const float4 v = *reinterpret_cast<const float4*>(ptr[p]);
                        r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
                    }
                    ptr[p] += kstep;
                }
            }
        } else {
        // K_MINOR: NT is a multiple of QPR, so every pass of this thread has the same k quad
        const int kq_t = k0 + (threadIdx.x % QPR) * 4;
        const int64_t koff_t = (K_MINOR && VEC == 4) ? minor_off(d.Dseg, d.Sseg, kq_t) : 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = threadIdx.x + NT * p;
            r[p][0] = r[p][1] = r[p][2] = r[p][3] = 0.f;
            if constexpr (K_MINOR) {
                const int kq = kq_t;
                if (fix[p][0] >= 0) {
                    if constexpr (VEC == 4) {
                        if (kq < k_end) {
                            const float4 v = *reinterpret_cast<const float4*>(d.base + fix[p][0] + koff_t);
                            r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (kq + j < k_end) r[p][j] = d.base[fix[p][0] + minor_off(d.Dseg, d.Sseg, kq + j)];
                    }
                }
            } else {
                const int k = k0 + idx / QPR;
                if (idx < ITEMS && k < k_end) {
                    const int64_t ro = major_off(d, k);
                    if constexpr (VEC == 4) {
                        if (fix[p][0] >= 0) {
                            const float4 v = *reinterpret_cast<const float4*>(d.base + ro + fix[p][0]);
                            r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (fix[p][j] >= 0) r[p][j] = d.base[ro + fix[p][j]];
                    }
                }
            }
        }
        }
    }
    __device__ __forceinline__ void store(float (*T)[LD]) const {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = threadIdx.x + NT * p;
            if (idx < ITEMS) {
                if constexpr (K_MINOR) {
                    const int col = SWZ ? (idx / QPR) ^ (((idx % QPR) & 3) << 3) : idx / QPR;
#pragma unroll
                    for (int j = 0; j < 4; ++j) T[(idx % QPR) * 4 + j][col] = r[p][j];
                } else {
                    const int k = idx / QPR;
                    const int col = SWZ ? ((idx % QPR) * 4) ^ (((k >> 2) & 3) << 3) : (idx % QPR) * 4;
                    *reinterpret_cast<float4*>(&T[k][col]) = make_float4(r[p][0], r[p][1], r[p][2], r[p][3]);
                }
            }
        }
    }
};

// Block tile (WM*TM*32) x (WN*TN*32) x 16; 4 waves arranged WM x WN, each wave TM x TN MFMA tiles.
template <bool A_KMINOR, bool B_KMINOR, int VEC, int WM, int WN, int TM, int TN, int BK, bool LIN>
__global__ void __launch_bounds__(NT, (TM * TN <= 4) ? (VEC == 1 ? 3 : 4) : 1) k_gemm_f32(const GemmArgs p) {      // the scalar-load form spills at 4
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    using LA = TileLoader<BM, A_KMINOR, VEC, BK, BM + 4, false, LIN>;
    using LB = TileLoader<BN, B_KMINOR, VEC, BK, BN + 4, false, LIN>;
    __shared__ __attribute__((aligned(16))) float As2[2][BK][LA::LD];      // double buffered: one barrier per K tile
    __shared__ __attribute__((aligned(16))) float Bs2[2][BK][LB::LD];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const TileId tile = xcd_tile(p.xcd_remap);
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int bz = tile.z / p.nsplit, zs = tile.z % p.nsplit;
    const int k_begin = zs * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    OperandDesc dA = p.A, dB = p.B;
    dA.base += bz * p.a_bs;
    dB.base += bz * p.b_bs;

    LA la; LB lb;
    la.init(dA, m0, p.M, k_begin);
    lb.init(dB, n0, p.N, k_begin);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int mb = (wid / WN) * TM * 32, nb = (wid % WN) * TN * 32;
    const int lr = lane & 31, lk = lane >> 5;

    if (k_begin < k_end) {
        la.load(dA, k_begin, k_end);
        lb.load(dB, k_begin, k_end);
        la.store(As2[0]);
        lb.store(Bs2[0]);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        float (*As)[LA::LD] = As2[cur];
        float (*Bs)[LB::LD] = Bs2[cur];
        if (more) { la.load(dA, k0 + BK, k_end); lb.load(dB, k0 + BK, k_end); }
        // fragments of k-step ks+1 are read from LDS before the MFMAs of k-step ks are issued
        float a[TM], b[TN], an[TM], bn[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[lk][mb + i * 32 + lr];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[lk][nb + j * 32 + lr];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            if (ks + 1 < BK / 2) {
                const int kk = 2 * (ks + 1) + lk;
#pragma unroll
                for (int i = 0; i < TM; ++i) an[i] = As[kk][mb + i * 32 + lr];
#pragma unroll
                for (int j = 0; j < TN; ++j) bn[j] = Bs[kk][nb + j * 32 + lr];
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the next fragments' ds_reads ahead of this step's MFMAs
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = an[i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bn[j];
        }
        if (more) { la.store(As2[cur ^ 1]); lb.store(Bs2[cur ^ 1]); }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (!p.partial && !p.C.scatter && p.M <= p.C.P && p.N <= p.C.Dseg) {          // plain row-major C: no index maps
        float* cb = p.C.base + bz * p.c_bs;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + nb + j * 32 + lr;
            if (col < p.N) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int rbase = m0 + mb + i * 32 + 4 * lk;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < p.M) cb[static_cast<int64_t>(row) * p.C.S1 + col] = gemm_epilogue<true>(acc[i][j][r], p.epilogue);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + nb + j * 32 + lr;
        if (col >= p.N) continue;
        int64_t coff;
        float* base;
        if (p.partial) { base = p.partial + static_cast<int64_t>(tile.z) * p.M * p.N; coff = col; }
        else { base = p.C.base + bz * p.c_bs; coff = minor_off(p.C.Dseg, p.C.Sseg, col); }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + mb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row >= p.M) continue;
                if (p.partial) base[static_cast<int64_t>(row) * p.N + coff] = acc[i][j][r];
                else base[out_row_off(p.C, row) + coff] = gemm_epilogue<true>(acc[i][j][r], p.epilogue);
            }
        }
    }
}


// 128 x 208 x 16 block tile on v_mfma_f32_16x16x4_f32 for outputs 129..208 columns wide (N = D = 200 per head at
// cfg 2): 4 waves stacked along M, each 32 rows x 208 columns = 2 x 13 tiles (104 accumulator registers, three
// waves per SIMD).  The LDS images are k-major like k_gemm_f32's, but one lane reads the operands of SEVERAL tiles
// with one wide LDS instruction: lane (i = lane&15, g = lane>>4) reads rows mb + 2i, mb + 2i + 1 of k row 4s+g as
// one ds_read_b64 — M tile t therefore owns rows {mb + 2i + t} — and float4s of columns 64q + 4i .. (N tiles
// 4q..4q+3 own columns {64q + 4i + t}); tile 12 = columns 192..207 is a b32 read.  5 LDS reads feed 26 MFMAs per
// k step (15 with one b32 read per fragment), and the accumulators of 4 neighbouring N tiles are 4 consecutive
// output columns, so the epilogue stores float4s.  Row lengths 160 (A) and 256 (B) put the lane groups of each
// wide read on disjoint banks; the XOR swizzle of TileLoader keeps that and removes the 4-way conflict of the
// transposed k-minor stores (PMC before: half of all LDS cycles were bank-conflict cycles).
// Measured (tools/gemm_bench.py, M=65536 N=200 K=600): 105 TF vs 97 TF for the b32-read form; a 256 x 208
// variant with 208 AGPR accumulators at one wave per SIMD reached 97 TF (its epilogue and barriers have no
// second workgroup to hide behind) and the (T,F)/(F,F) operand layouts spill at 168 VGPRs, so only (T,T) runs here.
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <bool A_KMINOR, bool B_KMINOR, int VEC, bool LIN>
__global__ void __launch_bounds__(NT, 2) k_gemm_f32_n208(const GemmArgs p) {
    constexpr int BM = 128, BN = 208, BK = 16, TN = 13, LDA = 160, LDB = 256;
    using LA = TileLoader<BM, A_KMINOR, VEC, BK, LDA, true, LIN>;
    using LB = TileLoader<BN, B_KMINOR, VEC, BK, LDB, true, LIN>;
    __shared__ __attribute__((aligned(16))) float As2[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs2[2][BK][LDB];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const TileId tile = xcd_tile(p.xcd_remap);
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int bz = tile.z / p.nsplit, zs = tile.z % p.nsplit;
    const int k_begin = zs * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    OperandDesc dA = p.A, dB = p.B;
    dA.base += bz * p.a_bs;
    dB.base += bz * p.b_bs;
    LA la; LB lb;
    la.init(dA, m0, p.M, k_begin);
    lb.init(dB, n0, p.N, k_begin);
    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int mb = wid * 32;
    const int li = lane & 15, lq = lane >> 4;
    if (k_begin < k_end) {
        la.load(dA, k_begin, k_end);
        lb.load(dB, k_begin, k_end);
        la.store(As2[0]);
        lb.store(Bs2[0]);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        float (*As)[LDA] = As2[cur];
        float (*Bs)[LDB] = Bs2[cur];
        if (more) { la.load(dA, k0 + BK, k_end); lb.load(dB, k0 + BK, k_end); }
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int kk = 4 * ks + lq, swz = ks << 3;
            float b[TN];
            const float2 a = *reinterpret_cast<const float2*>(&As[kk][(mb + 2 * li) ^ swz]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&Bs[kk][(64 * q + 4 * li) ^ swz]);
                b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
            }
            b[12] = Bs[kk][(192 + li) ^ swz];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[j], acc[1][j], 0, 0, 0);
            }
        }
        if (more) { la.store(As2[cur ^ 1]); lb.store(Bs2[cur ^ 1]); }
        __syncthreads();
        cur ^= 1;
    }
    // epilogue: MFMA C layout col = lane&15, row = (lane>>4)*4 + r, mapped back through the row/column ownership above
    float* base = p.partial ? p.partial + static_cast<int64_t>(tile.z) * p.M * p.N : p.C.base + bz * p.c_bs;
    const int epi = p.partial ? GEMM_EPI_NONE : p.epilogue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mb + 2 * (4 * lq + r) + i;
            if (row >= p.M) continue;
            float* crow = base + (p.partial ? static_cast<int64_t>(row) * p.N : out_row_off(p.C, row));
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int col = n0 + 64 * q + 4 * li;
                if (p.c_vec4) {
                    if (col < p.N) {
                        const int64_t coff = p.partial ? col : minor_off(p.C.Dseg, p.C.Sseg, col);
                        *reinterpret_cast<float4*>(crow + coff) =
                            make_float4(gemm_epilogue<true>(acc[i][4 * q][r], epi), gemm_epilogue<true>(acc[i][4 * q + 1][r], epi),
                                        gemm_epilogue<true>(acc[i][4 * q + 2][r], epi), gemm_epilogue<true>(acc[i][4 * q + 3][r], epi));
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (col + jj < p.N)
                            crow[p.partial ? col + jj : minor_off(p.C.Dseg, p.C.Sseg, col + jj)] = gemm_epilogue<true>(acc[i][4 * q + jj][r], epi);
                }
            }
            const int col = n0 + 192 + li;
            if (col < p.N) crow[p.partial ? col : minor_off(p.C.Dseg, p.C.Sseg, col)] = gemm_epilogue<true>(acc[i][12][r], epi);
        }
}

// deterministic second pass of split-K: C = epilogue(sum_z partial[batch][z]) (fixed order), written through C's addressing
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ partial, int32_t splits, int32_t M,
                                                       int32_t N, const OutputDesc C, int64_t c_bs, int32_t epilogue) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t MN = static_cast<int64_t>(M) * N;
    if (idx >= MN) return;
    const int bz = blockIdx.y;
    const int row = static_cast<int>(idx / N), col = static_cast<int>(idx % N);
    const float* pz = partial + static_cast<int64_t>(bz) * splits * MN;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                      // four interleaved chains, combined in a fixed order
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
        s0 += pz[z * MN + idx]; s1 += pz[(z + 1) * MN + idx]; s2 += pz[(z + 2) * MN + idx]; s3 += pz[(z + 3) * MN + idx];
    }
    for (; z < splits; ++z) s0 += pz[z * MN + idx];
    C.base[bz * c_bs + out_row_off(C, row) + minor_off(C.Dseg, C.Sseg, col)] = gemm_epilogue<true>((s0 + s1) + (s2 + s3), epilogue);
}

// transposing form: out(row n, col m) = epilogue(sum_z partial[z][m][n]); 32 x 32 tiles through LDS so that both the
// partial reads (along n) and the output writes (along m) are coalesced
__global__ void __launch_bounds__(256) k_splitk_reduce_t(const float* __restrict__ partial, int32_t splits, int32_t M,
                                                         int32_t N, const OutputDesc C, int64_t c_bs, int32_t epilogue) {
    __shared__ float tile[32][33];
    const int bz = blockIdx.z, m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t MN = static_cast<int64_t>(M) * N;
    const float* pz = partial + static_cast<int64_t>(bz) * splits * MN;
    // split index outermost: the four rows of a thread are four independent load / add chains (fixed order per element)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t off[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 8 * i, n = n0 + tx;
        ok[i] = m < M && n < N;
        off[i] = ok[i] ? static_cast<int64_t>(m) * N + n : 0;
    }
    for (int z = 0; z < splits; ++z) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += pz[z * MN + off[i]];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = ok[i] ? acc[i] : 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, m = m0 + tx;
        if (m < M && n < N)
            C.base[bz * c_bs + out_row_off(C, n) + minor_off(C.Dseg, C.Sseg, m)] = gemm_epilogue<true>(tile[tx][ty + 8 * i], epilogue);
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool operand_vec4(const OperandDesc& d, int32_t minor_extent) {
    if (!aligned16(d.base)) return false;
    if ((d.S1 & 3) || (d.S2 & 3) || (d.Sseg & 3)) return false;
    if (d.Dseg < minor_extent && (d.Dseg & 3)) return false;
    return (minor_extent & 3) == 0;
}

template <bool AK, bool BK_, int VEC>
void launch(const GemmArgs& a, bool narrow, bool lin, dim3 grid, hipStream_t st) {
    if constexpr (VEC == 4) {
        if (narrow) {                                                                                   // 128 x 208
            if (lin) hipLaunchKernelGGL((k_gemm_f32_n208<AK, BK_, 4, true>), grid, dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((k_gemm_f32_n208<AK, BK_, 4, false>), grid, dim3(NT), 0, st, a);
            return;
        }
        if (lin) { hipLaunchKernelGGL((k_gemm_f32<AK, BK_, 4, 2, 2, 2, 2, 16, true>), grid, dim3(NT), 0, st, a); return; }
    }
    hipLaunchKernelGGL((k_gemm_f32<AK, BK_, VEC, 2, 2, 2, 2, 16, false>), grid, dim3(NT), 0, st, a);      // 128 x 128
}

// operand linear along k (TileLoader LIN): one k segment for k-contiguous operands, no gather / row wrap for k-major ones
bool operand_linear(const OperandDesc& d, bool k_minor, int32_t K) {
    return k_minor ? d.Dseg >= K : (d.gather == nullptr && K <= d.P);
}

// Outputs 129..208 columns wide take ONE 208-wide column tile of 13 16x16 MFMA tiles (4 % padding at N = 200)
// instead of two 128-wide tiles (28 % padding).  RECON_GEMM_CFG=1 forces 128x128 (config.hip).
bool use_narrow(int32_t N, bool a_k_minor, bool b_k_minor, bool v4) {
    const int force = cfg_int(CFG_GEMM_CFG, 0);
    if (force == 1) return false;
    (void)a_k_minor; (void)b_k_minor;
    return v4 && N > 128 && N <= 208;
}

bool output_vec4(const OutputDesc& C, int32_t N, int64_t c_bs, const float* partial) {
    if ((N & 3) || (c_bs & 3)) return false;
    if (partial) return (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
    if ((reinterpret_cast<uintptr_t>(C.base) & 15) || (C.S1 & 3) || (C.S2 & 3) || (C.Sseg & 3)) return false;
    return C.Dseg >= N || (C.Dseg & 3) == 0;
}

}  // namespace

int gemm_pick_split_k(int32_t M, int32_t N, int32_t K, int32_t batch) {
    const int force = cfg_int(CFG_GEMM_SPLITK, 0);   // tuning knob
    if (force > 0) return force;
    const bool narrow = N > 128 && N <= 208;                       // 128 x 208 kernel (when the operands are float4-able)
    const int bn = narrow ? 208 : 128;
    const int64_t tiles = ceil_div64(M, 128) * ceil_div64(N, bn) * (batch > 0 ? batch : 1);
    const int64_t kResident = 256 * (narrow ? 2 : 4);              // CUs x workgroups per CU (2 / 4 waves per SIMD)
    if (tiles >= kResident / 2) return 1;
    // largest split that keeps every workgroup resident in one round and >= 128 of K per split
    int64_t s = kResident / tiles;
    const int64_t max_s = (K / 128) > 0 ? K / 128 : 1;
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    return static_cast<int>(s < 1 ? 1 : s);
}

int gemm_f32_batched(int32_t M, int32_t N, int32_t K, const OperandDesc& A, bool a_k_minor, const OperandDesc& B,
                     bool b_k_minor, const OutputDesc& C, const GemmBatch& bt, int32_t split_k, float* partial, hipStream_t st) {
    if (M < 0 || N < 0 || K < 0 || bt.batch < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0 || bt.batch == 0) return RECON_OK;
    if (!A.base || !B.base || !C.base) return RECON_ERR_INVALID;
    if (split_k < 1) split_k = 1;
    if ((split_k > 1 || bt.c_transpose) && !partial) return RECON_ERR_INVALID;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K;
    a.a_bs = bt.a_bs; a.b_bs = bt.b_bs; a.c_bs = bt.c_bs; a.epilogue = bt.epilogue;
    a.xcd_remap = cfg_int(CFG_GEMM_XCD, 1) == 0 ? 0 : 1;
    const bool v4 = operand_vec4(A, a_k_minor ? K : M) && operand_vec4(B, b_k_minor ? K : N) && (bt.batch == 1 || (!(bt.a_bs & 3) && !(bt.b_bs & 3)));
    const int bk = BK16;
    int64_t kps = ceil_div64(K > 0 ? K : 1, split_k);
    kps = ceil_div64(kps, bk) * bk;
    a.k_per_split = static_cast<int32_t>(kps);
    split_k = static_cast<int32_t>(ceil_div64(K > 0 ? K : 1, kps));
    a.nsplit = split_k;
    a.partial = (split_k > 1 || bt.c_transpose) ? partial : nullptr;
    if (static_cast<int64_t>(bt.batch) * split_k > 65535) return RECON_ERR_UNSUPPORTED;
    {
        const bool narrow = use_narrow(N, a_k_minor, b_k_minor, v4);
        const bool lin = v4 && operand_linear(A, a_k_minor, K) && operand_linear(B, b_k_minor, K) &&
                         cfg_int(CFG_GEMM_LIN, 1) != 0;
        a.c_vec4 = output_vec4(C, N, bt.c_bs, a.partial) ? 1 : 0;
        dim3 grid(static_cast<unsigned>(ceil_div64(N, narrow ? 208 : 128)), static_cast<unsigned>(ceil_div64(M, 128)),
                  static_cast<unsigned>(split_k * bt.batch));
        if (a_k_minor && b_k_minor) { if (v4) launch<true, true, 4>(a, narrow, lin, grid, st); else launch<true, true, 1>(a, narrow, lin, grid, st); }
        else if (a_k_minor && !b_k_minor) { if (v4) launch<true, false, 4>(a, narrow, lin, grid, st); else launch<true, false, 1>(a, narrow, lin, grid, st); }
        else if (!a_k_minor && !b_k_minor) { if (v4) launch<false, false, 4>(a, narrow, lin, grid, st); else launch<false, false, 1>(a, narrow, lin, grid, st); }
        else return RECON_ERR_UNSUPPORTED;
    }
    if (bt.c_transpose) {
        hipLaunchKernelGGL(k_splitk_reduce_t, dim3(static_cast<unsigned>(ceil_div64(N, 32)), static_cast<unsigned>(ceil_div64(M, 32)),
                           static_cast<unsigned>(bt.batch)), dim3(256), 0, st, partial, split_k, M, N, C, bt.c_bs, bt.epilogue);
    } else if (split_k > 1) {
        const int64_t MN = static_cast<int64_t>(M) * N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3(static_cast<unsigned>(ceil_div64(MN, 256)), static_cast<unsigned>(bt.batch)), dim3(256), 0,
                           st, partial, split_k, M, N, C, bt.c_bs, bt.epilogue);
    }
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

int splitk_reduce(const float* partial, int32_t splits, int32_t M, int32_t N, const OutputDesc& C, int64_t c_bs, int32_t batch,
                  int32_t epilogue, bool transpose, hipStream_t st) {
    if (!partial || !C.base || splits < 1 || M < 0 || N < 0 || batch < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0 || batch == 0) return RECON_OK;
    if (transpose) {
        hipLaunchKernelGGL(k_splitk_reduce_t, dim3(static_cast<unsigned>(ceil_div64(N, 32)), static_cast<unsigned>(ceil_div64(M, 32)),
                           static_cast<unsigned>(batch)), dim3(256), 0, st, partial, splits, M, N, C, c_bs, epilogue);
    } else {
        const int64_t MN = static_cast<int64_t>(M) * N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3(static_cast<unsigned>(ceil_div64(MN, 256)), static_cast<unsigned>(batch)), dim3(256), 0,
                           st, partial, splits, M, N, C, c_bs, epilogue);
    }
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

int gemm_f32(int32_t M, int32_t N, int32_t K, const OperandDesc& A, bool a_k_minor, const OperandDesc& B, bool b_k_minor,
             const OutputDesc& C, int32_t split_k, float* partial, hipStream_t st) {
    GemmBatch bt;
    bt.batch = 1; bt.a_bs = bt.b_bs = bt.c_bs = 0; bt.epilogue = GEMM_EPI_NONE;
    return gemm_f32_batched(M, N, K, A, a_k_minor, B, b_k_minor, C, bt, split_k, partial, st);
}

// Small dense products (tens of MFLOP: relation_embed.mm(W) and its two gradients, GAT/models.py:75).  Tile GEMMs — this file's and the
// library's behind torch.mm — take 18 - 36 us for them on MI355X (one or two workgroups walking K step by step, every step a
// dependent round trip; 200 us here at K = 1 600): the arithmetic is nothing, the serial K walk is everything.  Here a workgroup owns a
// 16 x 16 output tile and its 16 groups of 16 threads each take 1/16 of K (4 x 4 outputs per thread, plain FMAs), then the slices are
// added in fixed order through LDS: hundreds of workgroups, 1/16 of the walk each.
// gridDim.z > 1: K is also cut over workgroups (weight-gradient shapes: a 50 x 200 output over K = 14 541 rows is 52 tiles);
// workgroup z writes its tile of partial[z][M][N] (C = partial, ldc = N) and k_gemm_small_combine adds the z slices in order.
template <bool A_KM, bool B_NK, bool VEC4>
__global__ void __launch_bounds__(256) k_gemm_small(const float* __restrict__ A, int32_t lda, const float* __restrict__ B, int32_t ldb,
                                                    float* __restrict__ C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t kz) {
    __shared__ float red[16][16][17];
    const int tid = threadIdx.x, slice = tid >> 4, tm = (tid >> 2) & 3, tn = tid & 3;
    const int m0 = blockIdx.y * 16 + tm * 4, n0 = blockIdx.x * 16 + tn * 4;
    const int z0 = blockIdx.z * kz, z1 = min(K, z0 + kz);               // this workgroup's part of K (kz: a multiple of 4; all of K when gridDim.z == 1)
    C += static_cast<int64_t>(blockIdx.z) * M * ldc;
    const int ks = ((z1 - z0 + 15) / 16 + 3) & ~3;                       // K per slice, a multiple of the 4-deep step
    const int k_begin = min(z1, z0 + slice * ks), k_end = min(z1, k_begin + ks);
    int64_t arow[4], bcol[4];
    float am[4], bm[4];                                                   // 0 for rows / columns past the edge (their loads are clamped)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + i, M - 1), n = min(n0 + i, N - 1);
        arow[i] = A_KM ? m : static_cast<int64_t>(m) * lda;
        bcol[i] = B_NK ? static_cast<int64_t>(n) * ldb : n;
        am[i] = m0 + i < M ? 1.f : 0.f;
        bm[i] = n0 + i < N ? 1.f : 0.f;
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k = k_begin; k < k_end; k += 4) {
        float a[4][4], b[4][4];
        if constexpr (VEC4) {
            // 16-byte loads along whichever index is contiguous (the host checked the divisibility and alignment this needs): a
            // group of four rows / columns is inside or outside as a whole and is clamped as a whole
            const int m4 = min(m0, M - 4), n4 = min(n0, N - 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int kc = min(k + kk, k_end - 1);
                const float km = k + kk < k_end ? 1.f : 0.f;
                if constexpr (A_KM) {
                    const float4 t = *reinterpret_cast<const float4*>(A + static_cast<int64_t>(kc) * lda + m4);
                    a[0][kk] = t.x * km; a[1][kk] = t.y * km; a[2][kk] = t.z * km; a[3][kk] = t.w * km;
                }
                if constexpr (!B_NK) {
                    const float4 t = *reinterpret_cast<const float4*>(B + static_cast<int64_t>(kc) * ldb + n4);
                    b[kk][0] = t.x; b[kk][1] = t.y; b[kk][2] = t.z; b[kk][3] = t.w;
                    if constexpr (!A_KM) { (void)km; }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (!A_KM) {                                        // K % 4 == 0: k .. k + 3 < k_end
                    const float4 t = *reinterpret_cast<const float4*>(A + arow[i] + k);
                    a[i][0] = t.x; a[i][1] = t.y; a[i][2] = t.z; a[i][3] = t.w;
                }
                if constexpr (B_NK) {
                    const float4 t = *reinterpret_cast<const float4*>(B + bcol[i] + k);
                    b[0][i] = t.x; b[1][i] = t.y; b[2][i] = t.z; b[3][i] = t.w;
                }
            }
            if constexpr (!A_KM && !B_NK) {                                   // the k mask rides on a in the other forms
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i][kk] *= (k + kk < k_end ? 1.f : 0.f);
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kc = min(k + kk, k_end - 1);
            const float km = k + kk < k_end ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i][kk] = A[arow[i] + (A_KM ? static_cast<int64_t>(kc) * lda : kc)] * km;
                b[kk][i] = B[bcol[i] + (B_NK ? kc : static_cast<int64_t>(kc) * ldb)];
            }
        }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i][kk], b[kk][j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[slice][tm * 4 + i][tn * 4 + j] = acc[i][j] * am[i] * bm[j];
    __syncthreads();
    const int om = tid >> 4, on = tid & 15;
    float sum = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) sum += red[sl][om][on];
    const int m = blockIdx.y * 16 + om, n = blockIdx.x * 16 + on;
    if (m < M && n < N) C[static_cast<int64_t>(m) * ldc + n] = sum;
}

__global__ void __launch_bounds__(256) k_gemm_small_combine(const float* __restrict__ partial, int32_t nz, int64_t MN, int32_t N, float* __restrict__ C,
                                                            int32_t ldc) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= MN) return;
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += partial[z * MN + i];
    C[(i / N) * ldc + i % N] = s;
}

}  // namespace recon

// K cut over workgroups when the output has few tiles and K is long: (splits, K per split)
static void small_split(int32_t M, int32_t N, int32_t K, int* nz, int* kz) {
    const int64_t tiles = static_cast<int64_t>((M + 15) / 16) * ((N + 15) / 16);
    int z = 1;
    if (tiles < 512 && K >= 1024) {
        z = static_cast<int>((1024 + tiles - 1) / tiles);
        const int zmax = K / 512;                                        // at least 32 k per 16-thread slice
        if (z > zmax) z = zmax;
        if (z > 64) z = 64;
        if (z < 1) z = 1;
    }
    int per = (K + z - 1) / z;
    per = (per + 3) & ~3;
    *kz = per; *nz = (K + per - 1) / per;
}
extern "C" size_t recon_sgemm_small_workspace_floats(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int nz, kz;
    small_split(M, N, K, &nz, &kz);
    return nz > 1 ? static_cast<size_t>(nz) * M * N : 0;
}

extern "C" int recon_sgemm_small(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, int32_t a_is_km, const float* B, int32_t ldb,
                                 int32_t b_is_nk, float* C, int32_t ldc, float* workspace, recon_stream_t stream) {
    using namespace recon;
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    if (!C || (K > 0 && (!A || !B))) return RECON_ERR_INVALID;
    if (lda < (a_is_km ? M : K) || ldb < (b_is_nk ? K : N) || ldc < N) return RECON_ERR_INVALID;
    if ((N + 15) / 16 > 65535 * 16 || (M + 15) / 16 > 65535) return RECON_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (K == 0) {
        if (hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, st) != hipSuccess) return RECON_ERR_LAUNCH;
        return RECON_OK;
    }
    int nz = 1, kz = K;
    if (workspace) small_split(M, N, K, &nz, &kz);                       // NULL: one workgroup per tile walks all of K
    if (nz == 1) kz = (K + 3) & ~3;
    float* const Cfinal = C;
    const int32_t ldc_final = ldc;
    if (nz > 1) { C = workspace; ldc = N; }
    const dim3 grid(static_cast<unsigned>((N + 15) / 16), static_cast<unsigned>((M + 15) / 16), static_cast<unsigned>(nz));
    const bool k4 = !a_is_km || b_is_nk;                                  // some operand is read 4 k at a time
    const bool vec4 = !((lda | ldb) & 3) && !((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) && (!k4 || !(K & 3)) &&
                      (!a_is_km || !(M & 3)) && (b_is_nk || !(N & 3));
#define SMALL_CALL(AK, BN)                                                                                                      \
    do {                                                                                                                        \
        if (vec4) hipLaunchKernelGGL((k_gemm_small<AK, BN, true>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, kz);   \
        else hipLaunchKernelGGL((k_gemm_small<AK, BN, false>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, kz);       \
    } while (0)
    if (a_is_km) { if (b_is_nk) SMALL_CALL(true, true); else SMALL_CALL(true, false); }
    else { if (b_is_nk) SMALL_CALL(false, true); else SMALL_CALL(false, false); }
#undef SMALL_CALL
    if (nz > 1) {
        const int64_t MN = static_cast<int64_t>(M) * N;
        hipLaunchKernelGGL(k_gemm_small_combine, dim3(static_cast<unsigned>((MN + 255) / 256)), dim3(256), 0, st, workspace, nz, MN, N, Cfinal, ldc_final);
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" size_t recon_sgemm_ex_workspace_floats(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int sk = recon::gemm_pick_split_k(M, N, K);
    return sk > 1 ? static_cast<size_t>(sk) * M * N : 0;
}

extern "C" int recon_sgemm_ex(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, int32_t a_is_km, const float* B, int32_t ldb,
                              int32_t b_is_nk, float* C, int32_t ldc, float* workspace, recon_stream_t stream) {
    using namespace recon;
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    const int sk = workspace ? gemm_pick_split_k(M, N, K) : 1;
    return gemm_f32(M, N, K, plain_operand(A, lda), a_is_km == 0, plain_operand(B, ldb), b_is_nk != 0, plain_output(C, ldc), sk,
                    sk > 1 ? workspace : nullptr, as_stream(stream));
}

extern "C" int recon_sgemm(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                           int32_t b_is_nk, float* C, int32_t ldc, recon_stream_t stream) {
    using namespace recon;
    return gemm_f32(M, N, K, plain_operand(A, lda), true, plain_operand(B, ldb), b_is_nk != 0, plain_output(C, ldc), 1,
                    nullptr, as_stream(stream));
}
