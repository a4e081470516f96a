// K4' — fp32-accurate GEMM on the bf16 matrix cores by operand splitting ("bf16 x 3").
//
// Every fp32 operand element is written as three bfloat16 terms
//     x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)   (3 x 8 = 24 mantissa bits;
//     bfloat16 keeps fp32's exponent range, so no scaling is needed)
// and a*b is accumulated in fp32 from the six term pairs of weight >= 2^-16:
//     a0b2 + a2b0 + a1b1 + a0b1 + a1b0 + a0b0        (dropped: a1b2, a2b1, a2b2 <= 2^-24 relative, the size of an
//     fp32 rounding error) with v_mfma_f32_16x16x32_bf16 (2.5 PF dense): 6 MFMAs of 16 cycles replace the
//     8 fp32 MFMAs of 32 cycles of one 16x16x32 block => 2.67x the fp32-MFMA ceiling (419 TF fp32-equivalent).
//
// C[M,N] = act(A[M,K] . B[N,K]^T), both operands k-contiguous:
//   A  fp32, split ON THE FLY while it is staged into LDS (in the GAT layer every A element is consumed by one
//      to three column tiles, so a pre-split copy would cost more HBM traffic than it saves VALU work);
//   B  PRE-SPLIT bf16 planes [3][N][Kp] (Kp = K rounded up to 32, zero padded) made once per step by
//      k_split_planes: B is the small parameter matrix `a` that every row tile re-reads.
// Block tile 128 x 208 x 32, 4 waves stacked along M, each 32 rows x 208 columns = 2 x 13 MFMA tiles (104
// accumulators), two workgroups per CU.  LDS holds the three planes of both operands k-contiguous (64-byte rows,
// the 16-byte slot of k group kq rotated by 2*(row>>3) so that the four lane groups of every ds_read_b128 and the
// 8-lane groups of every ds_write_b128 hit disjoint banks).  B rows are permuted on the way into LDS (tile 4q+t
// owns columns {64q + 4i + t}) so that the accumulators of four neighbouring tiles are four consecutive output
// columns and the epilogue stores float4s.
#include <stdlib.h>
#include <hip/hip_bf16.h>
#include "gemm_common.h"

namespace recon {
namespace {

constexpr int BM = 128, BN = 208, BK = 32, NT = 256, TN = 13;
constexpr int B_ITEMS = 3 * BN * 4;                              // 16-byte slots of one B tile (3 planes x 208 rows x 4)
constexpr int B_NP = (B_ITEMS + NT - 1) / NT;                    // 10
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

struct Bx3Args {
    OperandDesc A;                 // fp32, k-contiguous (Dseg >= K), rows through major_off
    const __bf16* Bp;              // planes [3][rows][Kp]
    int64_t b_plane, b_row;        // element strides between planes / rows (b_row = Kp)
    OutputDesc C;
    int32_t M, N, K;
    int32_t epilogue, c_vec4, xcd_remap;
    int64_t a_bs, b_bs, c_bs;      // per-batch element offsets (b_bs in bf16 elements)
};

// byte offset of (row, k group kq of 8 bf16) inside one plane of an LDS tile
__device__ __forceinline__ int lds_off(int row, int kq) { return row * 64 + (((kq + 2 * (row >> 3)) & 3) << 4); }

// 8 fp32 -> three packed bf16x8 terms
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[3]) {
    float r[8];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = q == 0 ? v[2 * j] : r[2 * j], x1 = q == 0 ? v[2 * j + 1] : r[2 * j + 1];
            const __bf16 h0 = static_cast<__bf16>(x0), h1 = static_cast<__bf16>(x1);
            const uint32_t b0 = __builtin_bit_cast(uint16_t, h0), b1 = __builtin_bit_cast(uint16_t, h1);
            out[q][j] = b0 | (b1 << 16);
            if (q < 2) {
                r[2 * j] = x0 - __builtin_bit_cast(float, b0 << 16);
                r[2 * j + 1] = x1 - __builtin_bit_cast(float, b1 << 16);
            }
        }
    }
}

__global__ void __launch_bounds__(NT, 2) k_gemm_bx3(const Bx3Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char As[3][BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3][BN * 64];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const TileId tile = xcd_tile(p.xcd_remap);
    const int m0 = tile.y * BM, n0 = tile.x * BN, bz = tile.z;

    // ---- A items: item i (0, 1) = row t/4 + 64 i, 8 floats at k0 + 8 (t & 3)
    const int a_kq = t & 3;
    const float* aptr[2];
    int a_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (t >> 2) + 64 * i;
        aptr[i] = p.A.base + bz * p.a_bs + major_off(p.A, min(m0 + row, p.M - 1)) + 8 * a_kq;
        a_lds[i] = lds_off(row, a_kq);
    }
    // ---- B items: slot index idx = t + 256 i over (plane, LDS row, k group); LDS row <-> output column permutation
    int b_goff[B_NP], b_lds[B_NP];
    const __bf16* bbase = p.Bp + bz * p.b_bs;
#pragma unroll
    for (int i = 0; i < B_NP; ++i) {
        const int idx = min(t + NT * i, B_ITEMS - 1);
        const int plane = idx / (BN * 4), rem = idx % (BN * 4), rowL = rem >> 2, kq = rem & 3;
        const int j = rowL >> 4, rho = rowL & 15;
        const int col = j < 12 ? 64 * (j >> 2) + 4 * rho + (j & 3) : 192 + rho;
        b_goff[i] = static_cast<int>(plane * p.b_plane + static_cast<int64_t>(min(n0 + col, p.N - 1)) * p.b_row + 8 * kq);
        b_lds[i] = plane * (BN * 64) + lds_off(rowL, kq);
    }

    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float av[2][8];
    u32x4 bv[B_NP];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + 8 * a_kq + 4 * h;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) v = *reinterpret_cast<const float4*>(aptr[i] + k0 + 4 * h);
                av[i][4 * h] = v.x; av[i][4 * h + 1] = v.y; av[i][4 * h + 2] = v.z; av[i][4 * h + 3] = v.w;
            }
#pragma unroll
        for (int i = 0; i < B_NP; ++i) bv[i] = *reinterpret_cast<const u32x4*>(bbase + b_goff[i] + k0);   // planes are zero padded to Kp
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 s[3];
            split8(av[i], s);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(&As[q][a_lds[i]]) = s[q];
        }
#pragma unroll
        for (int i = 0; i < B_NP; ++i)
            if (t + NT * i < B_ITEMS) *reinterpret_cast<u32x4*>(&Bs[0][0] + b_lds[i]) = bv[i];
    };

    const int mb = wid * 32;
    const int li = lane & 15, lq = lane >> 4;
    int a_rd[2], b_rd;
#pragma unroll
    for (int i = 0; i < 2; ++i) a_rd[i] = lds_off(mb + 16 * i + li, lq);
    b_rd = lds_off(li, lq);                                           // + j * 16 rows * 64 B (the rotation depends on row & 8 only)

    load_tile(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < p.K) load_tile(k0 + BK);
        bf16x8 a[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(&As[q][a_rd[i]]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf16x8 b[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) b[q] = *reinterpret_cast<const bf16x8*>(&Bs[q][b_rd + j * 1024]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[2], acc[i][j], 0, 0, 0);      // small terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[0], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: MFMA C layout col = lane&15, row = (lane>>4)*4 + r; columns through the B row permutation
    float* base = p.C.base + bz * p.c_bs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mb + 16 * i + 4 * lq + r;
            if (row >= p.M) continue;
            float* crow = base + out_row_off(p.C, row);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int col = n0 + 64 * q + 4 * li;
                if (p.c_vec4) {
                    if (col < p.N)
                        *reinterpret_cast<float4*>(crow + minor_off(p.C.Dseg, p.C.Sseg, col)) =
                            make_float4(gemm_epilogue(acc[i][4 * q][r], p.epilogue), gemm_epilogue(acc[i][4 * q + 1][r], p.epilogue),
                                        gemm_epilogue(acc[i][4 * q + 2][r], p.epilogue), gemm_epilogue(acc[i][4 * q + 3][r], p.epilogue));
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (col + jj < p.N) crow[minor_off(p.C.Dseg, p.C.Sseg, col + jj)] = gemm_epilogue(acc[i][4 * q + jj][r], p.epilogue);
                }
            }
            const int col = n0 + 192 + li;
            if (col < p.N) crow[minor_off(p.C.Dseg, p.C.Sseg, col)] = gemm_epilogue(acc[i][12][r], p.epilogue);
        }
}

// planes[q][r][k] = q-th bf16 term of src[r][k] (row stride ld), k < Kp zero padded; one thread per 8 k values.
// TRANS: src is [K][rows] (element (r, k) at src[k*ld + r]) — used for the a^T planes of the g_V product.
template <bool TRANS>
__global__ void __launch_bounds__(256) k_split_planes(const float* __restrict__ src, int64_t ld, int64_t src_bs, int32_t rows, int32_t K,
                                                      int32_t Kp, __bf16* __restrict__ dst, int64_t plane, int64_t dst_bs) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int kq = static_cast<int>(idx % (Kp / 8));
    const int r = static_cast<int>(idx / (Kp / 8));
    if (r >= rows) return;
    const float* s = src + blockIdx.y * src_bs;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * kq + j;
        v[j] = k < K ? (TRANS ? s[static_cast<int64_t>(k) * ld + r] : s[static_cast<int64_t>(r) * ld + k]) : 0.f;
    }
    u32x4 out[3];
    split8(v, out);
    __bf16* d = dst + blockIdx.y * dst_bs + static_cast<int64_t>(r) * Kp + 8 * kq;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(d + q * plane) = out[q];
}

}  // namespace

int32_t bx3_kp(int32_t K) { return (K + BK - 1) / BK * BK; }

int bx3_split_planes(const float* src, int64_t ld, int64_t src_bs, bool transposed, int32_t rows, int32_t K, int32_t batch, void* dst,
                     hipStream_t st) {
    if (rows <= 0 || K <= 0 || batch <= 0) return RECON_OK;
    if (!src || !dst || (reinterpret_cast<uintptr_t>(dst) & 15)) return RECON_ERR_INVALID;
    const int32_t Kp = bx3_kp(K);
    const int64_t per = static_cast<int64_t>(rows) * Kp;              // elements of one plane of one batch entry
    // layout [3][batch][rows][Kp]: plane stride batch*rows*Kp, batch stride rows*Kp
    const dim3 grid(static_cast<unsigned>(ceil_div64(per / 8, 256)), static_cast<unsigned>(batch));
    if (transposed) hipLaunchKernelGGL((k_split_planes<true>), grid, dim3(256), 0, st, src, ld, src_bs, rows, K, Kp, static_cast<__bf16*>(dst), per * batch, per);
    else hipLaunchKernelGGL((k_split_planes<false>), grid, dim3(256), 0, st, src, ld, src_bs, rows, K, Kp, static_cast<__bf16*>(dst), per * batch, per);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

bool bx3_supported(const OperandDesc& A, int32_t K, const GemmBatch& bt) {
    if (K <= 0 || (K & 3) || A.Dseg < K) return false;
    if ((reinterpret_cast<uintptr_t>(A.base) & 15) || (A.S1 & 3) || (A.S2 & 3) || (bt.a_bs & 3)) return false;
    return bt.c_transpose == 0;
}

// planes: [3][batch][N][Kp] as written by bx3_split_planes
int gemm_bx3_batched(int32_t M, int32_t N, int32_t K, const OperandDesc& A, const void* planes, const OutputDesc& C,
                     const GemmBatch& bt, hipStream_t st) {
    if (M < 0 || N < 0 || K < 0 || bt.batch < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0 || bt.batch == 0) return RECON_OK;
    if (!A.base || !planes || !C.base) return RECON_ERR_INVALID;
    if (!bx3_supported(A, K, bt) || bt.batch > 65535) return RECON_ERR_UNSUPPORTED;
    Bx3Args a;
    a.A = A; a.C = C; a.M = M; a.N = N; a.K = K;
    const int32_t Kp = bx3_kp(K);
    a.Bp = static_cast<const __bf16*>(planes);
    a.b_row = Kp;
    a.b_bs = static_cast<int64_t>(N) * Kp;
    a.b_plane = a.b_bs * bt.batch;
    if (3 * a.b_plane >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;    // 32-bit element offsets inside the planes
    a.a_bs = bt.a_bs; a.c_bs = bt.c_bs; a.epilogue = bt.epilogue;
    a.c_vec4 = (!(N & 3) && !(bt.c_bs & 3) && !(reinterpret_cast<uintptr_t>(C.base) & 15) && !(C.S1 & 3) && !(C.S2 & 3) && !(C.Sseg & 3) &&
                (C.Dseg >= N || !(C.Dseg & 3))) ? 1 : 0;
    a.xcd_remap = 1;
    const dim3 grid(static_cast<unsigned>(ceil_div64(N, BN)), static_cast<unsigned>(ceil_div64(M, BM)), static_cast<unsigned>(bt.batch));
    hipLaunchKernelGGL(k_gemm_bx3, grid, dim3(NT), 0, st, a);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace recon

// Stand-alone entry (tests, tools/gemm_bench.py): C[M,N] = A[M,K] . B[N,K]^T with B split into `workspace`.
extern "C" size_t recon_sgemm_bx3_workspace_bytes(int32_t N, int32_t K) {
    return static_cast<size_t>(3) * (N > 0 ? N : 0) * recon::bx3_kp(K > 0 ? K : 0) * 2 + 16;
}

extern "C" int recon_sgemm_bx3(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb, float* C_,
                               int32_t ldc, void* workspace, recon_stream_t stream) {
    using namespace recon;
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    if (!A || !B || !C_ || !workspace) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    GemmBatch bt;
    bt.batch = 1; bt.a_bs = bt.b_bs = bt.c_bs = 0; bt.epilogue = GEMM_EPI_NONE;
    const OperandDesc Ad = plain_operand(A, lda);
    if (!bx3_supported(Ad, K, bt)) return RECON_ERR_UNSUPPORTED;
    int rc = bx3_split_planes(B, ldb, 0, false, N, K, 1, workspace, st);
    if (rc != RECON_OK) return rc;
    return gemm_bx3_batched(M, N, K, Ad, workspace, plain_output(C_, ldc), bt, st);
}
