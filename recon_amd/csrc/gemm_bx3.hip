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

// the six term products of one accumulator, issued for a GROUP of independent accumulators term by term: six
// back-to-back MFMAs into the same accumulator would each wait for the previous result (dependent-issue latency
// of the 4-pass MFMA), a group of 4 (2 row tiles x 2 column tiles) keeps 3 independent MFMAs between dependent ones
template <int NJ>
__device__ __forceinline__ void bx3_products(f32x4 (&acc)[2][TN], const bf16x8 (&a)[2][3], const bf16x8 (&b)[2][3], int j0) {
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // small terms first
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][TA[t]], b[jj][TB[t]], acc[i][j0 + jj], 0, 0, 0);
}

__device__ __forceinline__ void bx3_mma_reg(f32x4 (&acc)[2][TN], const bf16x8 (&a)[2][3], const unsigned char* Bs, int b_rd) {
    // column tiles in pairs; the fragments of the next pair are read while the 24 MFMAs of this one run (two register
    // sets pinned with sched_barrier: left alone the scheduler reads into one set and waits for every read)
    bf16x8 b[2][2][3];
    auto read_pair = [&](int j0, bf16x8 (&dst)[2][3]) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            if (j0 + jj < TN) {
#pragma unroll
                for (int q = 0; q < 3; ++q) dst[jj][q] = *reinterpret_cast<const bf16x8*>(Bs + q * (BN * 64) + b_rd + (j0 + jj) * 1024);
            }
    };
    read_pair(0, b[0]);
#pragma unroll
    for (int g = 0; g < (TN + 1) / 2; ++g) {
        if (2 * g + 2 < TN) read_pair(2 * g + 2, b[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (2 * g + 1 < TN) bx3_products<2>(acc, a, b[g & 1], 2 * g);
        else bx3_products<1>(acc, a, b[g & 1], 2 * g);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one K tile (32) of the wave's 32 x 208 block with both operands in LDS (k-major kernel): 6 + 39 ds_read_b128, 156 MFMAs
__device__ __forceinline__ void bx3_mma(f32x4 (&acc)[2][TN], const unsigned char (*As)[BM * 64], const unsigned char (*Bs)[BN * 64],
                                        const int (&a_rd)[2], int b_rd) {
    bf16x8 a[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8*>(&As[q][a_rd[i]]);
    bx3_mma_reg(acc, a, &Bs[0][0], b_rd);
}

// MFMA C layout col = lane&15, row = (lane>>4)*4 + r; columns through the B row permutation (tile 4q+t <-> columns 64q+4i+t)
__device__ __forceinline__ void bx3_store(const f32x4 (&acc)[2][TN], const OutputDesc& C, float* base, int M, int N, int m0, int n0,
                                          int mb, int li, int lq, int epi, int c_vec4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mb + 16 * i + 4 * lq + r;
            if (row >= M) continue;
            float* crow = base + out_row_off(C, row);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int col = n0 + 64 * q + 4 * li;
                if (c_vec4) {
                    if (col < N)
                        *reinterpret_cast<float4*>(crow + minor_off(C.Dseg, C.Sseg, col)) =
                            make_float4(gemm_epilogue(acc[i][4 * q][r], epi), gemm_epilogue(acc[i][4 * q + 1][r], epi),
                                        gemm_epilogue(acc[i][4 * q + 2][r], epi), gemm_epilogue(acc[i][4 * q + 3][r], epi));
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (col + jj < N) crow[minor_off(C.Dseg, C.Sseg, col + jj)] = gemm_epilogue(acc[i][4 * q + jj][r], epi);
                }
            }
            const int col = n0 + 192 + li;
            if (col < N) crow[minor_off(C.Dseg, C.Sseg, col)] = gemm_epilogue(acc[i][12][r], epi);
        }
}

// A never touches LDS: the rows of a wave's 32 x 208 block are private to that wave, so every lane loads its own MFMA
// fragment (row lane&15, 8 consecutive k) straight from global memory, splits it in registers and keeps the three
// bf16x8 terms as MFMA operands.  B (pre-split planes) is copied global -> LDS by the LDS-DMA path
// (global_load_lds_dwordx4: no staging registers, no ds_write) into a DOUBLE-buffered image, so a K tile costs one
// barrier and both operands of tile t+1 are in flight during the MFMAs of tile t.  The DMA writes lane-linear
// (wave base + 16 B x lane), so the bank rotation of the image is applied on the SOURCE address: lane -> (plane, row,
// physical slot) -> the k group that lives there.
constexpr int B_TILE_BYTES = 3 * BN * 64;                        // 39936: one buffer of the B image
constexpr int B_DMA = (B_TILE_BYTES / 16 + NT - 1) / NT;         // 10 DMA instructions per thread and tile

__global__ void __launch_bounds__(NT, 2) k_gemm_bx3(const Bx3Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][B_TILE_BYTES];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const TileId tile = xcd_tile(p.xcd_remap);
    const int m0 = tile.y * BM, n0 = tile.x * BN, bz = tile.z;
    const int mb = wid * 32;
    const int li = lane & 15, lq = lane >> 4;

    // ---- A: fragment-shaped loads, M tile i = rows mb + 16 i + li, this lane's 8 k values start at 8 lq
    const float* aptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aptr[i] = p.A.base + bz * p.a_bs + major_off(p.A, min(m0 + mb + 16 * i + li, p.M - 1)) + 8 * lq;
    // ---- B: DMA i of wave w fills the 1 KiB piece (4 i + w) of the image; slot s = 64 (4 i + w) + lane
    int b_goff[B_DMA];
    const __bf16* bbase = p.Bp + bz * p.b_bs;
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int s = min(64 * (4 * i + wid) + lane, B_TILE_BYTES / 16 - 1);
        const int plane = s / (BN * 4), rem = s % (BN * 4), rowL = rem >> 2, pslot = rem & 3;
        const int kq = (pslot - 2 * (rowL >> 3)) & 3;                 // inverse of lds_off's rotation
        const int j = rowL >> 4, rho = rowL & 15;
        const int col = j < 12 ? 64 * (j >> 2) + 4 * rho + (j & 3) : 192 + rho;
        b_goff[i] = static_cast<int>(plane * p.b_plane + static_cast<int64_t>(min(n0 + col, p.N - 1)) * p.b_row + 8 * kq);
    }
    auto dma_b = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < B_DMA; ++i)
            if (64 * (4 * i + wid) < B_TILE_BYTES / 16)                // wave-uniform: the last round is 3 waves wide
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(bbase + b_goff[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(&Bs[buf][1024 * (4 * i + wid)]), 16, 0, 0);
    };

    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float av[2][8];
    bf16x8 af[2][3];
    // The A loads are branch free (a guarded load gets its own basic block and the join serialises the tile): lanes
    // past K read the start of their row instead and are zeroed when the tile is split.  They are ordinary loads —
    // inline-asm loads would let the register allocator reuse the destination registers while the data is in flight.
    bool a_ok[2];
    f32x4 araw[2][2];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) a_ok[h] = k0 + 8 * lq + 4 * h < p.K;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int off = -8 * lq + ((k0 + 4 * h + 8 * lq) & -static_cast<int>(a_ok[h]));
                const float4 v = *reinterpret_cast<const float4*>(aptr[i] + off);
                araw[i][h] = f32x4{v.x, v.y, v.z, v.w};
            }
    };
    auto wait_a = [&]() {                                             // first use: the compiler's wait also drains the DMA of the same tile
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) av[i][e] = a_ok[e >> 2] ? araw[i][e >> 2][e & 3] : 0.f;
    };
    auto split_a = [&]() {
        wait_a();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 s[3];
            split8(av[i], s);
#pragma unroll
            for (int q = 0; q < 3; ++q) af[i][q] = __builtin_bit_cast(bf16x8, s[q]);
        }
    };
    const int b_rd = lds_off(li, lq);                                 // + j * 16 rows * 64 B (the rotation depends on row & 8 only)

    dma_b(0, 0);
    load_a(0);
    split_a();
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        if (k0 + BK < p.K) dma_b(k0 + BK, buf ^ 1);
        load_a(k0 + BK);                                              // unconditional (past K every lane re-reads its row start): no
        __builtin_amdgcn_sched_barrier(0);                            // control-flow join may sit between a load and its use, and
        bx3_mma_reg(acc, af, Bs[buf], b_rd);                          // the first use (with its s_waitcnt) stays behind the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        split_a();
        __syncthreads();
        buf ^= 1;
    }
    bx3_store(acc, p.C, p.C.base + bz * p.c_bs, p.M, p.N, m0, n0, mb, li, lq, p.epilogue, p.c_vec4);
}

// The same product for k-MAJOR operands — the weight gradient g_a^T = V^T g_h, whose K is the node dimension:
//   A  fp32 [K][M] (m contiguous), split on the fly;   B  pre-split bf16 planes [3][K][ldb] (n contiguous).
// Both tiles are staged (A through registers, B by LDS-DMA) into ROW-MAJOR LDS images [k][m] / [k][n] as they lie in memory
// (A: four float4 -> 3 x ds_write_b64 each; B: ten 16-byte copies), and the MFMA fragments — 8 consecutive k for one
// m — come out of LDS through the transposing read ds_read_b64_tr_b16 (16 lanes read a [4 k][16 m] block, lane i
// receives column i), two reads per fragment.  Bank layout: A rows are 256 B = 8 chunks of 32 B, chunk index XORed
// with (k&3 | (k>>3&1)<<2); B rows are 28 slots of 16 B, shifted by 2 slots when k & 8 — in both images the
// 8 rows one transposing read touches per 32 lanes land on disjoint bank groups (B: 28 slots per row — two of them
// padding — rotated by 2 slots when k & 8: found conflict-free for every column tile by exhaustive check; the unpadded
// 26-slot row with a wrapping rotation left 17 % of the LDS cycles as conflicts).
// Split-K: every (batch, split) writes its tile to partial[z][M][N]; the caller reduces (and here transposes).
struct Bx3KmArgs {
    const float* A; const __bf16* Bp;
    int64_t lda, ldb, b_plane, a_bs, b_bs;
    float* partial;
    int32_t M, N, K, k_per_split, nsplit;
    int32_t n_ld;                  // columns present in the planes (multiple of 8, >= N; the excess is zero padding)
};

using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using i16x4 = __attribute__((ext_vector_type(4))) short;
constexpr int KB_SLOTS = 28;                                       // 16-byte slots per B image row: 26 of data + 2 of padding
constexpr int KA_PLANE = BK * 256, KB_PLANE = BK * KB_SLOTS * 16;  // bytes per plane of the A / B image

__device__ __forceinline__ int ka_h(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base, int off_lo, int off_hi) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base + off_lo));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base + off_hi));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ void __launch_bounds__(NT, 2) k_gemm_bx3_kmajor(const Bx3KmArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char S[3 * KA_PLANE + 3 * KB_PLANE];      // A image | B image
    unsigned char* const As = S;
    unsigned char* const Bs = S + 3 * KA_PLANE;
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const TileId tile = xcd_tile(1);
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int bz = tile.z / p.nsplit, zs = tile.z % p.nsplit;
    const int k_begin = zs * p.k_per_split, k_end = min(p.K, k_begin + p.k_per_split);

    // ---- A items: idx = t + 256 i -> (k = idx >> 5, m quad = idx & 31); one float4 each
    const float* aptr[4];
    int a_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + NT * i, k = idx >> 5, mq = idx & 31;
        aptr[i] = p.A + bz * p.a_bs + min(m0 + 4 * mq, p.M - 4);
        a_lds[i] = k * 256 + (((mq >> 2) ^ ka_h(k)) << 5) + ((mq & 3) << 3);
    }
    // ---- B: LDS-DMA, no staging registers.  Piece pc = 4 i + wave (42 pieces of 64 slots: 14 per plane) lands lane-linear
    //      at slot s = 64 (pc % 14) + lane of its plane = (k = s / 28, physical slot s % 28); the rotation of the image is
    //      applied on the source side: the physical slot holds logical slot phys - 2 [k & 8] (padding slots read slot 0).
    constexpr int KB_PIECES = 3 * (BK * KB_SLOTS / 64), KB_DMA = (KB_PIECES + 3) / 4;     // 42, 11
    int b_goff[KB_DMA], b_k[KB_DMA];
#pragma unroll
    for (int i = 0; i < KB_DMA; ++i) {
        const int pc = min(4 * i + wid, KB_PIECES - 1);
        const int s = 64 * (pc % (KB_PIECES / 3)) + lane, k = s / KB_SLOTS, phys = s % KB_SLOTS;
        int slot = phys - ((k & 8) ? 2 : 0);
        if (slot < 0 || slot >= 26) slot = 0;
        b_k[i] = k;
        b_goff[i] = min(n0 + 8 * slot, p.n_ld - 8);
    }
    const __bf16* bbase = p.Bp + bz * p.b_bs;
    auto dma_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KB_DMA; ++i)
            if (4 * i + wid < KB_PIECES) {                                // wave-uniform
                const int pc = 4 * i + wid;
                const __bf16* q = bbase + (pc / (KB_PIECES / 3)) * p.b_plane + static_cast<int64_t>(min(k0 + b_k[i], k_end - 1)) * p.ldb + b_goff[i];
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(q),
                                                 (__attribute__((address_space(3))) void*)(Bs + (pc / (KB_PIECES / 3)) * KB_PLANE + 1024 * (pc % (KB_PIECES / 3))), 16, 0, 0);
            }
    };

    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 araw[4];
    bool a_ok[4];
    // loads are branch free: rows past the split's K range re-read its last row (finite values); A's are zeroed at the split
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ((t + NT * i) >> 5);
            a_ok[i] = k < k_end;
            const float4 v = *reinterpret_cast<const float4*>(aptr[i] + static_cast<int64_t>(min(k, k_end - 1)) * p.lda);
            araw[i] = f32x4{v.x, v.y, v.z, v.w};
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = a_ok[i] ? araw[i][e] : 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                uint32_t w[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const __bf16 h0 = static_cast<__bf16>(x[2 * h]), h1 = static_cast<__bf16>(x[2 * h + 1]);
                    const uint32_t b0 = __builtin_bit_cast(uint16_t, h0), b1 = __builtin_bit_cast(uint16_t, h1);
                    w[h] = b0 | (b1 << 16);
                    if (q < 2) {
                        x[2 * h] -= __builtin_bit_cast(float, b0 << 16);
                        x[2 * h + 1] -= __builtin_bit_cast(float, b1 << 16);
                    }
                }
                *reinterpret_cast<uint2*>(As + q * KA_PLANE + a_lds[i]) = make_uint2(w[0], w[1]);
            }
        }
    };

    // ---- fragment addresses: lane (ip = lane & 15, g = lane >> 4); half hh covers k = 8 g + 4 hh + (ip >> 2)
    const int mb = wid * 32;
    const int ip = lane & 15, g = lane >> 4;
    int a_off[2][2], b_row[2];
    bool b_rot[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int k = 8 * g + 4 * hh + (ip >> 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) a_off[i][hh] = k * 256 + (((2 * wid + i) ^ ka_h(k)) << 5) + ((ip & 3) << 3);
        b_row[hh] = k * (KB_SLOTS * 16) + (((ip & 3) & 1) << 3);
        b_rot[hh] = (k & 8) != 0;
    }
    auto b_off = [&](int j, int hh) {                                 // column tile j: slot 2 j + ((ip & 3) >> 1), shifted by 2 when k & 8
        return b_row[hh] + (2 * j + ((ip & 3) >> 1) + (b_rot[hh] ? 2 : 0)) * 16;
    };
    auto mma_tile = [&]() {
        bf16x8 a[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[i][q] = tr_frag(As + q * KA_PLANE, a_off[i][0], a_off[i][1]);
        bf16x8 b[2][2][3];                                                // column tile j+1 is read while the 12 MFMAs of tile j run ([.][0] only)
        auto read_one = [&](int j, bf16x8 (&dst)[2][3]) {
            const int o0 = b_off(j, 0), o1 = b_off(j, 1);
#pragma unroll
            for (int q = 0; q < 3; ++q) dst[0][q] = tr_frag(Bs + q * KB_PLANE, o0, o1);
        };
        read_one(0, b[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) read_one(j + 1, b[(j + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            bx3_products<1>(acc, a, b[j & 1], j);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // B has ONE LDS buffer: its DMA for tile t+1 can only start once every wave is done with tile t (second barrier);
    // it then runs under the split/store of A, and the co-resident workgroup's MFMA phase covers what is left.
    if (k_begin < k_end) { load_tile(k_begin); dma_b(k_begin); }
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        store_tile();                                                 // s_waitcnt vmcnt(0): A registers and the B DMA
        __syncthreads();
        load_tile(k0 + BK);                                           // past the range: clamped re-reads, never stored
        __builtin_amdgcn_sched_barrier(0);
        mma_tile();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (k0 + BK < k_end) dma_b(k0 + BK);
    }
    // epilogue: MFMA C layout col = lane & 15, row = (lane >> 4) * 4 + r; plain column order in this kernel
    float* base = p.partial + static_cast<int64_t>(tile.z) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mb + 16 * i + 4 * g + r;
            if (row >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + 16 * j + ip;
                if (col < p.N) base[static_cast<int64_t>(row) * p.N + col] = acc[i][j][r];
            }
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

// N itself may be any size as long as every plane row holds (N rounded up to 8) columns, zero padded: b_bs + that <= ldb
bool bx3_kmajor_supported(const float* A, int64_t lda, int64_t a_bs, int64_t ldb, int64_t b_bs, int32_t M, int32_t N) {
    const int64_t n_ld = (static_cast<int64_t>(N) + 7) / 8 * 8;
    if ((M & 3) || M < 4 || N < 1 || (lda & 3) || (a_bs & 3) || (ldb & 7) || (b_bs & 7) || n_ld > ldb) return false;
    if ((N & 7) && b_bs != 0) return false;                          // batched heads sit side by side in a row: no room for padding
    return !(reinterpret_cast<uintptr_t>(A) & 15);
}

// split-K choice of the k-major product: 512 workgroups are resident at a time (2 per CU).  Few tiles: as many splits as
// fill them once (>= 4 K tiles per split).  Half full or more: the split count (<= 8) whose workgroup total fills its LAST
// round best, minus 1 % per extra M x N partial — out_att's 4800 x 1600 weight gradient (304 tiles) takes 5 splits = 2.97
// rounds and SpGAT's step drops from 3.74 to 3.45 ms against one 59 %-full round.
int bx3_kmajor_split_k(int32_t M, int32_t N, int32_t K, int32_t batch) {
    const int64_t tiles = ceil_div64(M, BM) * ceil_div64(N, BN) * (batch > 0 ? batch : 1);
    int64_t max_s = K / (4 * BK) > 0 ? K / (4 * BK) : 1;
    if (max_s > 64) max_s = 64;
    if (tiles >= 256) {
        int best = 1;
        double best_score = -1.0;
        for (int64_t s = 1; s <= (max_s < 8 ? max_s : 8); ++s) {
            const int64_t wg = tiles * s, rounds = ceil_div64(wg, 512);
            const double score = static_cast<double>(wg) / static_cast<double>(rounds * 512) - 0.01 * static_cast<double>(s - 1);
            if (score > best_score + 1e-9) { best_score = score; best = static_cast<int>(s); }
        }
        return best;
    }
    int64_t s = 512 / (tiles > 0 ? tiles : 1);
    if (s > max_s) s = max_s;
    return static_cast<int>(s < 1 ? 1 : s);
}

// partial[batch][split][M][N] = A_slice^T . B_slice  with A = fp32 [K][M] (row stride lda), B = bf16 planes [3][K][ldb]
// (plane stride b_plane elements); a_bs / b_bs = per-batch column offsets
int gemm_bx3_kmajor_batched(int32_t M, int32_t N, int32_t K, const float* A, int64_t lda, int64_t a_bs, const void* Bplanes, int64_t ldb,
                            int64_t b_plane, int64_t b_bs, int32_t batch, int32_t split_k, float* partial, hipStream_t st) {
    if (M < 0 || N < 0 || K < 0 || batch < 0 || split_k < 1) return RECON_ERR_INVALID;
    if (M == 0 || N == 0 || batch == 0) return RECON_OK;
    if (!A || !Bplanes || !partial) return RECON_ERR_INVALID;
    if (!bx3_kmajor_supported(A, lda, a_bs, ldb, b_bs, M, N) || (reinterpret_cast<uintptr_t>(Bplanes) & 15) || (b_plane & 7))
        return RECON_ERR_UNSUPPORTED;
    Bx3KmArgs a;
    a.A = A; a.Bp = static_cast<const __bf16*>(Bplanes); a.lda = lda; a.ldb = ldb; a.b_plane = b_plane; a.a_bs = a_bs; a.b_bs = b_bs;
    a.partial = partial; a.M = M; a.N = N; a.K = K;
    a.n_ld = (N + 7) / 8 * 8;
    int64_t kps = ceil_div64(K > 0 ? K : 1, split_k);
    kps = ceil_div64(kps, BK) * BK;
    a.k_per_split = static_cast<int32_t>(kps);
    a.nsplit = static_cast<int32_t>(ceil_div64(K > 0 ? K : 1, kps));
    if (a.nsplit != split_k) return RECON_ERR_INVALID;                // callers size `partial` with the same rounding (bx3_kmajor_splits)
    if (static_cast<int64_t>(batch) * split_k > 65535) return RECON_ERR_UNSUPPORTED;
    const dim3 grid(static_cast<unsigned>(ceil_div64(N, BN)), static_cast<unsigned>(ceil_div64(M, BM)), static_cast<unsigned>(batch * split_k));
    hipLaunchKernelGGL(k_gemm_bx3_kmajor, grid, dim3(NT), 0, st, a);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// number of splits actually used for a requested split count (K tiles of 32 per split, rounded up)
int bx3_kmajor_splits(int32_t K, int32_t split_k) {
    if (split_k < 1) split_k = 1;
    int64_t kps = ceil_div64(K > 0 ? K : 1, split_k);
    kps = ceil_div64(kps, BK) * BK;
    return static_cast<int>(ceil_div64(K > 0 ? K : 1, kps));
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

// C[M,N] = A^T . B for k-major operands A = [K][M], B = [K][N] (the weight-gradient form), same accuracy.
// workspace = bf16 planes of B ([3][K][kp(N)]) followed by the split-K partials.
static size_t tn_planes_bytes(int32_t N, int32_t K) { return align_up(static_cast<size_t>(3) * K * recon::bx3_kp(N) * 2, 256); }
extern "C" size_t recon_sgemm_bx3_tn_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    using namespace recon;
    if (M <= 0 || N <= 0 || K <= 0) return 256;
    const int sk = bx3_kmajor_splits(K, bx3_kmajor_split_k(M, N, K, 1));
    return tn_planes_bytes(N, K) + static_cast<size_t>(sk) * M * N * sizeof(float) + 256;
}

extern "C" int recon_sgemm_bx3_tn(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb, float* C_,
                                  int32_t ldc, void* workspace, recon_stream_t stream) {
    using namespace recon;
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    if (!A || !B || !C_ || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    const int32_t Np = bx3_kp(N);
    if (!bx3_kmajor_supported(A, lda, 0, Np, 0, M, N)) return RECON_ERR_UNSUPPORTED;
    int rc = bx3_split_planes(B, ldb, 0, false, K, N, 1, workspace, st);             // rows = k, minor = n (zero padded to Np)
    if (rc != RECON_OK) return rc;
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + tn_planes_bytes(N, K));
    const int sk = bx3_kmajor_splits(K, bx3_kmajor_split_k(M, N, K, 1));
    rc = gemm_bx3_kmajor_batched(M, N, K, A, lda, 0, workspace, Np, static_cast<int64_t>(K) * Np, 0, 1, sk, partial, st);
    if (rc != RECON_OK) return rc;
    return splitk_reduce(partial, sk, M, N, plain_output(C_, ldc), 0, 1, GEMM_EPI_NONE, false, st);
}
