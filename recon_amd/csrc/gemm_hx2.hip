// K4'' — fp32-accurate GEMM on the fp16 matrix cores by operand splitting ("f16 x 2"), both operands PRE-SPLIT.
//
// Every fp32 operand element is scaled by a power of two s (one per tensor, chosen from the tensor's max magnitude so that
// s*amax lies in [2^14, 2^15)) and written as two IEEE half terms
//     s x = x0 + x1 + r,   x0 = f16(s x),  x1 = f16(s x - x0),  |r| <= 2^-22 |s x|      (11 + 11 significant bits)
// and a*b is accumulated in fp32 from the three term pairs of weight >= 2^-11:
//     a0 b1 + a1 b0 + a0 b0          (dropped: a1 b1 <= 2^-22 relative)
// with v_mfma_f32_16x16x32_f16 (2.5 PF dense): 3 MFMAs per 16x16x32 block where the bf16 x 3 scheme of gemm_bx3.hip
// needs 6 and exact fp32 MFMA the time of 32.  Representation + dropped-term error <= 3 * 2^-22 relative per product,
// i.e. of the size of fp32's own accumulation error (tests bound |err| <= 16 * 2^-24 * sum_k |a_k b_k| against fp64,
// the bound the bf16 x 3 kernels are held to).  Half has 5 exponent bits, hence the scale: elements down to 2^-18 of the
// tensor's maximum keep all 22 bits, smaller ones keep an ABSOLUTE error of 2^-40 of the maximum (their low term goes
// subnormal) — below fp32's 2^-24 for anything that matters in a dot product with the large elements.
//
// Why pre-split: the producers of the layer's GEMM operands are HBM-bound kernels with idle VALU (the edge aggregation
// writes V, the ELU-gradient pass writes g_h), two half planes take exactly the bytes of the fp32 tensor they replace,
// and the GEMM main loops then contain no conversion work at all — A fragments (k-contiguous form) are 16-byte global
// loads straight into MFMA operand registers, everything else arrives by LDS-DMA.
//
// Tiling is the one of gemm_bx3.hip: block 128 x 208 x 32, 4 waves x (2 x 13) MFMA tiles, two workgroups per CU.
#include <stdlib.h>
#include "gemm_common.h"

namespace recon {
namespace {

constexpr int BM = 128, BN = 208, BK = 32, NT = 256, TN = 13, T = 2;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using i16x4 = __attribute__((ext_vector_type(4))) short;

struct Hx2Args {
    const _Float16* Ap; const _Float16* Bp;
    int64_t a_plane, a_row, a_bs;  // element strides of A: between planes / rows (m) / batch entries
    int64_t b_plane, b_row, b_bs;
    OutputDesc C;
    int64_t c_bs;
    int32_t M, N, K;               // K: multiple of 8; B planes are zero padded to hx2_kp(K), A is masked here
    int32_t epilogue, c_vec4, xcd_remap, c_plain;
    int32_t a_shared_k;            // A's columns k < a_shared_k are the same for every batch entry and are read from entry 0 (0: none)
    Hx2Scale sa, sb;
    const float* row_inv; int64_t row_inv_bs;      // per-row inverse scales of A (see gemm_hx2_batched) or null
    int32_t a_span_bytes;          // k_gemm_hx2_r3: bytes from Ap to the end of the last fragment any lane reads (0: not below 2^31, the ring form is not used)
};

// byte offset of (row, k group kq of 8 halves) inside one plane of the B tile image (see gemm_bx3.hip)
__device__ __forceinline__ int lds_off(int row, int kq) { return row * 64 + (((kq + 2 * (row >> 3)) & 3) << 4); }

// the three term products for a GROUP of independent accumulators, term by term (small terms first): NJ column tiles x 2
// row tiles keep 2 NJ - 1 independent MFMAs between two that hit the same accumulator
template <int NJ>
__device__ __forceinline__ void hx2_products(f32x4 (&acc)[2][TN], const f16x8 (&a)[2][T], const f16x8 (&b)[2][T], int j0) {
    constexpr int TA[3] = {0, 1, 0}, TB[3] = {1, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][j0 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][TA[t]], b[jj][TB[t]], acc[i][j0 + jj], 0, 0, 0);
}

// MFMA C layout col = lane&15, row = (lane>>4)*4 + r; columns through the B row permutation (tile 4q+t <-> columns 64q+4i+t)
__device__ __forceinline__ void hx2_store(const f32x4 (&acc)[2][TN], const OutputDesc& C, float* base, int M, int N, int m0, int n0,
                                          int mb, int li, int lq, int epi, int c_vec4, const float (&rsc)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float scale = rsc[i][r];
            auto fin = [&](float v) { return gemm_epilogue(v * scale, epi); };
            const int row = m0 + mb + 16 * i + 4 * lq + r;
            if (row >= M) continue;
            float* crow = base + out_row_off(C, row);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int col = n0 + 64 * q + 4 * li;
                if (c_vec4) {
                    if (col < N)
                        *reinterpret_cast<float4*>(crow + minor_off(C.Dseg, C.Sseg, col)) =
                            make_float4(fin(acc[i][4 * q][r]), fin(acc[i][4 * q + 1][r]), fin(acc[i][4 * q + 2][r]), fin(acc[i][4 * q + 3][r]));
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (col + jj < N) crow[minor_off(C.Dseg, C.Sseg, col + jj)] = fin(acc[i][4 * q + jj][r]);
                }
            }
            const int col = n0 + 192 + li;
            if (col < N) crow[minor_off(C.Dseg, C.Sseg, col)] = fin(acc[i][12][r]);
        }
}

// The same for the layer's own outputs — plain rows (no scatter, no segments), float4-aligned, the region below 2 GiB — without any
// branch: rows past M and columns past N become out-of-range buffer offsets.  The general form above decides scatter / segments / vector width / column bounds per
// store, a few branches each; a wave spends several hundred cycles per store instruction in it.
template <int EPI>
__device__ __forceinline__ void hx2_store_plain(const f32x4 (&acc)[2][TN], float* base, int64_t ld, int M, int N, int m0, int n0, int mb,
                                                int li, int lq, const float (&rsc)[2][4]) {
    // buffer stores: a lane whose offset lies outside the region is dropped by the hardware — no exec mask, no branch per store
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, static_cast<int>((static_cast<int64_t>(M - 1) * ld + N) * 4), 0x00020000);
    constexpr int kOut = 0x7ffffff0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float scale = rsc[i][r];
            auto fin = [&](float v) { return __builtin_bit_cast(uint32_t, gemm_epilogue(v * scale, EPI)); };
            const int row = m0 + mb + 16 * i + 4 * lq + r;
            const int rowoff = static_cast<int>(row * ld) + n0;          // < 2^29 (checked on the host)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int col = 64 * q + 4 * li;
                const int off = (row < M && n0 + col < N) ? (rowoff + col) * 4 : kOut;     // N % 4 == 0: a float4 is in or out as a whole
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{fin(acc[i][4 * q][r]), fin(acc[i][4 * q + 1][r]), fin(acc[i][4 * q + 2][r]),
                                                             fin(acc[i][4 * q + 3][r])}, rsrc, off, 0, 0);
            }
            const int off = (row < M && n0 + 192 + li < N) ? (rowoff + 192 + li) * 4 : kOut;
            __builtin_amdgcn_raw_buffer_store_b32(fin(acc[i][12][r]), rsrc, off, 0, 0);
        }
}

constexpr int B_TILE_BYTES = T * BN * 64;                        // 26624: one buffer of the B image
constexpr int B_PIECES = B_TILE_BYTES / 1024;                    // 26 pieces of 1 KiB (one LDS-DMA instruction of one wave each)

// C = act(A . B^T / (s_a s_b)), both operands k-contiguous half planes.  A never touches LDS: the rows of a wave's
// 32 x 208 block are private to that wave, so every lane loads its own MFMA fragments (row lane & 15, 8 consecutive k,
// one 16-byte load per term).  B is copied global -> LDS by the LDS-DMA path into a double-buffered image whose bank
// rotation is applied on the source address; one barrier per K tile.
// (Measured and not kept: three K tiles in flight behind a counted vmcnt and a raw barrier, with the B copies issued from
// inline asm so that the compiler does not drain them before every ds_read, and 256-row workgroups of 8 waves — 5-10 % faster
// on out_att-sized products in isolation, 3-8 % slower inside the layer's step at K = 200 / 600.  By elimination on the
// projection shape: 59.5 us in full, 49.6 without the epilogue stores, 48.1 without A loads, 50.4 without B copies, 33.0 for
// MFMAs + fragment reads + barriers alone.)
// Round 5, measured and not kept: A's fragments requested TWO K tiles ahead (a second register set; the A stream is latency bound by
// Little's law — 16 KB of requests in flight per workgroup, 2 per CU, 256 CUs = 8 MB per ~3 us round trip = 2.7 TB/s) with the B copies
// hidden from the compiler and one counted wait + raw barrier per tile: the unrolled loop needs 256 registers + 252 bytes of scratch
// (scratch loads inside the K loop) against 198 for this form — the 2 x 13 accumulator tiles (104) leave no room for a second A set at two
// waves per SIMD.
// s_setprio 1 around the MFMA groups (what bought 3 % in the wide-state propagation kernels): 0.409 ms per cfg 2 step with and without.
// NW waves per workgroup: 4 (128 rows, two workgroups per CU: the layer's K = 200 / 600 products) or 8 (256 rows, one workgroup per
// CU: every B tile copied into LDS serves twice the rows — out_att-sized products, K >= 1024, where the copies of B and the re-reads
// of A through L2 are what the loop waits for).
template <int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) k_gemm_hx2(const Hx2Args p) {
    constexpr int BMW = 32 * NW, B_DMAW = (B_PIECES + NW - 1) / NW;
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][B_TILE_BYTES];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const TileId tile = xcd_tile(p.xcd_remap);
    const int m0 = tile.y * BMW, n0 = tile.x * BN, bz = tile.z;
    const int mb = wid * 32;
    const int li = lane & 15, lq = lane >> 4;

    const _Float16* aptr[2][T];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < T; ++q)
            aptr[i][q] = p.Ap + bz * p.a_bs + q * p.a_plane + static_cast<int64_t>(min(m0 + mb + 16 * i + li, p.M - 1)) * p.a_row + 8 * lq;
    int b_goff[B_DMAW];
    const _Float16* bbase = p.Bp + bz * p.b_bs;
#pragma unroll
    for (int i = 0; i < B_DMAW; ++i) {
        const int s = min(64 * (NW * i + wid) + lane, B_TILE_BYTES / 16 - 1);
        const int plane = s / (BN * 4), rem = s % (BN * 4), rowL = rem >> 2, pslot = rem & 3;
        const int kq = (pslot - 2 * (rowL >> 3)) & 3;                 // inverse of lds_off's rotation
        const int j = rowL >> 4, rho = rowL & 15;
        const int col = j < 12 ? 64 * (j >> 2) + 4 * rho + (j & 3) : 192 + rho;
        b_goff[i] = static_cast<int>(plane * p.b_plane + static_cast<int64_t>(min(n0 + col, p.N - 1)) * p.b_row + 8 * kq);
    }
    auto dma_b = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < B_DMAW; ++i)
            if (NW * i + wid < B_PIECES)                               // wave-uniform
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(bbase + b_goff[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(&Bs[buf][1024 * (NW * i + wid)]), 16, 0, 0);
    };

    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A loads are branch free: lanes past K re-read the start of their row and are zeroed when the fragment is taken
    u32x4 araw[2][T];
    f16x8 af[2][T];
    bool a_ok = true;
    const int a_back = static_cast<int>(bz * p.a_bs);                 // < 2^31 (checked on the host when a_shared_k is set)
    auto load_a = [&](int k0) {
        a_ok = k0 + 8 * lq < p.K;
        int off = -8 * lq + ((k0 + 8 * lq) & -static_cast<int>(a_ok));
        if (k0 + 8 * lq < p.a_shared_k) off -= a_back;                 // shared columns: batch entry 0's copy (lane predicate, no branch)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < T; ++q) araw[i][q] = *reinterpret_cast<const u32x4*>(aptr[i][q] + off);
    };
    auto take_a = [&]() {                                            // first use: the wait also covers the DMA of the same tile
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < T; ++q) {
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = a_ok ? araw[i][q][e] : 0u;
                af[i][q] = __builtin_bit_cast(f16x8, v);
            }
    };
    const int b_rd = lds_off(li, lq);

    auto mma = [&](const unsigned char* Bt) {
        f16x8 b[2][2][T];
        auto read_pair = [&](int j0, f16x8 (&dst)[2][T]) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                if (j0 + jj < TN) {
#pragma unroll
                    for (int q = 0; q < T; ++q) dst[jj][q] = *reinterpret_cast<const f16x8*>(Bt + q * (BN * 64) + b_rd + (j0 + jj) * 1024);
                }
        };
        read_pair(0, b[0]);
#pragma unroll
        for (int g = 0; g < (TN + 1) / 2; ++g) {
            if (2 * g + 2 < TN) read_pair(2 * g + 2, b[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * g + 1 < TN) hx2_products<2>(acc, af, b[g & 1], 2 * g);
            else hx2_products<1>(acc, af, b[g & 1], 2 * g);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    dma_b(0, 0);
    load_a(0);
    // the operands' scales (32 amax slots each): read here, under the first tile's round trip, not in front of the stores
    const float ia = hx2_inv(hx2_scale_wave(p.sa)), ib = hx2_inv(hx2_scale_wave(p.sb));

    take_a();
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        if (k0 + BK < p.K) dma_b(k0 + BK, buf ^ 1);
        load_a(k0 + BK);
        __builtin_amdgcn_sched_barrier(0);
        mma(Bs[buf]);
        __builtin_amdgcn_sched_barrier(0);
        take_a();
        __syncthreads();
        buf ^= 1;
    }
    // per-row scales of A (rows past M re-read the last row; they are never stored).  Requested here, behind the loop: eight more live
    // registers across the K loop cost 15 of the loop's 256 (213 against 198)
    float rsc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            rsc[i][r] = (p.row_inv ? p.row_inv[bz * p.row_inv_bs + min(m0 + mb + 16 * i + 4 * lq + r, p.M - 1)] : 1.f) * (ia * ib);
    if (p.c_plain) {                                                 // uniform
        if (p.epilogue == GEMM_EPI_ELU) hx2_store_plain<GEMM_EPI_ELU>(acc, p.C.base + bz * p.c_bs, p.C.S1, p.M, p.N, m0, n0, mb, li, lq, rsc);
        else hx2_store_plain<GEMM_EPI_NONE>(acc, p.C.base + bz * p.c_bs, p.C.S1, p.M, p.N, m0, n0, mb, li, lq, rsc);
    } else {
        hx2_store(acc, p.C, p.C.base + bz * p.c_bs, p.M, p.N, m0, n0, mb, li, lq, p.epilogue, p.c_vec4, rsc);
    }
}

// ---- round 6: the same product with A THREE tiles deep -------------------------------------------------------------------------------
// The kernel above asks for the A fragments of tile k + 1 at the top of tile k and waits for them (and for B's copies: vmcnt(0)) behind
// tile k's 78 MFMAs: requests are one MMA phase old when they are needed.  With K = 200 / 600 there are 7 / 19 K tiles per workgroup,
// the 512 workgroups of the projection are ONE round over 256 CUs x 2, and the kernel's time is 19 x (a tile's step): 3.2 us per step
// where the MFMAs of the two co-resident workgroups take 1.7 — every step ends in a wait for HBM (A = V is streamed from HBM, 157 MB;
// B = the layer's a sits in L2).  Here A lives in THREE register sets used in rotation WITHOUT a masked copy (fragments go straight from
// the load's registers into the MFMAs: rows past K are dropped by the buffer descriptor's range check, not by a select), A of tile
// k + 2 is requested inside tile k, B's copies are issued from inline asm (the compiler does not see them: through the builtin it
// drains the request counter in front of the next LDS read) and ONE counted wait per tile — s_waitcnt vmcnt(4): everything but the four
// youngest requests, which are A(k + 1) — stands in front of a raw s_barrier.  Requests are two MMA phases old when they are needed.
// The K loop runs in whole triples (register sets are named statically) with the remainder peeled behind it; every wave issues the
// same number of requests per tile (missing B pieces copy into a scratch KiB), so the wait's immediate holds.
// Host condition: A's span below 2^31 bytes (32-bit buffer offsets), 128-row workgroups.
template <int NW>
__global__ void __launch_bounds__(64 * NW, 2) k_gemm_hx2_r3(const Hx2Args p) {
    static_assert(NW == 4, "128-row workgroups");
    constexpr int BMW = 32 * NW, B_DMAW = (B_PIECES + NW - 1) / NW;       // 7 copies per wave and tile
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * B_TILE_BYTES + 1024];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const TileId tile = xcd_tile(p.xcd_remap);
    const int m0 = tile.y * BMW, n0 = tile.x * BN, bz = tile.z;
    const int mb = wid * 32;
    const int li = lane & 15, lq = lane >> 4;
    const int KT = (p.K + BK - 1) / BK, Kp = KT * BK;

    // ---- A: one descriptor over the tensor, byte offsets per (row tile, term); the k part is added per tile
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Ap), 0, p.a_span_bytes, 0x00020000);
    uint32_t a_off[2][T];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < T; ++q)
            a_off[i][q] = static_cast<uint32_t>((bz * p.a_bs + q * p.a_plane + static_cast<int64_t>(min(m0 + mb + 16 * i + li, p.M - 1)) * p.a_row + 8 * lq) * 2);
    const uint32_t a_back = static_cast<uint32_t>(bz * p.a_bs * 2);
    u32x4 araw[3][2][T];
    auto load_a = [&](u32x4 (&dst)[2][T], int kt) {                  // tile kt (any value: tiles past K read nothing and come back as zeros)
        const int k = kt * BK + 8 * lq;
        uint32_t ko = k < p.K ? static_cast<uint32_t>(kt * BK * 2) : 0x7ffffff0u;
        if (k < p.a_shared_k) ko -= a_back;                           // shared columns: batch entry 0's copy (lane predicate, no branch)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < T; ++q) dst[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[i][q] + ko, 0, 0);
    };
    // ---- B: the copy plan of gemm_hx2 (piece pc = NW i + wave of the tile image), missing pieces go to the scratch KiB
    int b_goff[B_DMAW];
    const _Float16* bbase = p.Bp + bz * p.b_bs;
#pragma unroll
    for (int i = 0; i < B_DMAW; ++i) {
        const int s = min(64 * (NW * i + wid) + lane, B_TILE_BYTES / 16 - 1);
        const int plane = s / (BN * 4), rem = s % (BN * 4), rowL = rem >> 2, pslot = rem & 3;
        const int kq = (pslot - 2 * (rowL >> 3)) & 3;                 // inverse of lds_off's rotation
        const int j = rowL >> 4, rho = rowL & 15;
        const int col = j < 12 ? 64 * (j >> 2) + 4 * rho + (j & 3) : 192 + rho;
        b_goff[i] = static_cast<int>(plane * p.b_plane + static_cast<int64_t>(min(n0 + col, p.N - 1)) * p.b_row + 8 * kq);
    }
    auto dma_b = [&](int kt, int buf) {                              // kt past the end: the last tile again, into scratch
        const bool real = kt < KT;
        const int k0 = min(kt, KT - 1) * BK;
#pragma unroll
        for (int i = 0; i < B_DMAW; ++i) {
            const bool piece = real && NW * i + wid < B_PIECES;       // wave-uniform
            dma16_to_lds(bbase + b_goff[i] + k0, Bs + (piece ? buf * B_TILE_BYTES + 1024 * (NW * i + wid) : 2 * B_TILE_BYTES));
        }
    };

    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int b_rd = lds_off(li, lq);

    // one tile: wait, barrier, first MFMA group, the tile's requests, the other groups
    auto step = [&](int kt, const u32x4 (&acur)[2][T], u32x4 (&anext2)[2][T]) {
        dma_wait<4>();                                               // A(kt) and B(kt) have landed; A(kt + 1) may be in flight
        __builtin_amdgcn_s_barrier();                                 // everybody's pieces of B(kt) are there; everybody is past tile kt - 1
        asm volatile("" ::: "memory");
        const unsigned char* Bt = Bs + (kt & 1) * B_TILE_BYTES;
        f16x8 af[2][T];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < T; ++q) af[i][q] = __builtin_bit_cast(f16x8, acur[i][q]);
        f16x8 b[2][2][T];
        auto read_pair = [&](int j0, f16x8 (&dst)[2][T]) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                if (j0 + jj < TN) {
#pragma unroll
                    for (int q = 0; q < T; ++q) dst[jj][q] = *reinterpret_cast<const f16x8*>(Bt + q * (BN * 64) + b_rd + (j0 + jj) * 1024);
                }
        };
        read_pair(0, b[0]);
#pragma unroll
        for (int g = 0; g < (TN + 1) / 2; ++g) {
            if (2 * g + 2 < TN) read_pair(2 * g + 2, b[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * g + 1 < TN) hx2_products<2>(acc, af, b[g & 1], 2 * g);
            else hx2_products<1>(acc, af, b[g & 1], 2 * g);
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) {                                             // behind the first group: the tile's requests (B first: it is needed first)
                dma_b(kt + 1, (kt + 1) & 1);
                load_a(anext2, kt + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    dma_b(0, 0);
    load_a(araw[0], 0);
    load_a(araw[1], 1);
    // the operands' scales (32 amax slots each): read here, under the first tile's round trip, not in front of the stores
    const float ia = hx2_inv(hx2_scale_wave(p.sa)), ib = hx2_inv(hx2_scale_wave(p.sb));
    int kt = 0;
#pragma unroll 1
    for (; kt + 3 <= KT; kt += 3) {
        step(kt, araw[0], araw[2]);
        step(kt + 1, araw[1], araw[0]);
        step(kt + 2, araw[2], araw[1]);
    }
    if (kt < KT) step(kt, araw[0], araw[2]);
    if (kt + 1 < KT) step(kt + 1, araw[1], araw[0]);
    (void)Kp;
    dma_wait<0>();                                                   // the last tiles' look-ahead copies (scratch KiB) must not outlive the workgroup's LDS
    float rsc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            rsc[i][r] = (p.row_inv ? p.row_inv[bz * p.row_inv_bs + min(m0 + mb + 16 * i + 4 * lq + r, p.M - 1)] : 1.f) * (ia * ib);
    if (p.c_plain) {                                                 // uniform
        if (p.epilogue == GEMM_EPI_ELU) hx2_store_plain<GEMM_EPI_ELU>(acc, p.C.base + bz * p.c_bs, p.C.S1, p.M, p.N, m0, n0, mb, li, lq, rsc);
        else hx2_store_plain<GEMM_EPI_NONE>(acc, p.C.base + bz * p.c_bs, p.C.S1, p.M, p.N, m0, n0, mb, li, lq, rsc);
    } else {
        hx2_store(acc, p.C, p.C.base + bz * p.c_bs, p.M, p.N, m0, n0, mb, li, lq, p.epilogue, p.c_vec4, rsc);
    }
}

// The same product for k-MAJOR operands — the weight gradient g_a^T = V^T g_h, whose K is the node dimension:
//   A half planes [2][K][lda] (m contiguous),  B half planes [2][K][ldb] (n contiguous).
// Both tiles arrive by LDS-DMA in ROW-MAJOR images [k][m] / [k][n] as they lie in memory and the MFMA fragments — 8
// consecutive k for one m — come out of LDS through the transposing read ds_read_b64_tr_b16 (two per fragment).  Bank
// layout as in gemm_bx3.hip: A rows are 256 B = 8 chunks of 32 B, chunk index XORed with (k&3 | (k>>3&1)<<2); B rows
// are 28 slots of 16 B (26 of data), rotated by 2 slots when k & 8; both rotations are applied on the DMA's SOURCE
// address.  Rows past the split's K range are read from a page of zeros (there is no register stage to mask them in).
// Split-K: every (batch, split) writes its scaled-back tile to partial[z][M][N]; the caller reduces (and transposes).
struct Hx2KmArgs {
    const _Float16* Ap; const _Float16* Bp; const _Float16* zeros;    // zeros: >= 1 KiB of zero bytes, 16-byte aligned
    int64_t lda, ldb, a_plane, b_plane, a_bs, b_bs;
    float* partial;
    int32_t M, N, K, k_per_split, nsplit;
    int32_t m_ld, n_ld;            // columns present in the planes (multiples of 8, >= M / N; the excess is zero padding)
    int32_t a_shared_m;            // A's columns m < a_shared_m are the same for every batch entry and are read from entry 0 (0: none)
    Hx2Scale sa, sb;
    const float* k_inv; int64_t k_inv_bs;          // per-k inverse scales of B's rows (see gemm_hx2_kmajor_batched) or null
};

constexpr int KB_SLOTS = 28;
constexpr int KA_PLANE = BK * 256, KB_PLANE = BK * KB_SLOTS * 16;
constexpr int KA_PIECES = T * (KA_PLANE / 1024), KA_DMA = KA_PIECES / 4;               // 16, 4
constexpr int KB_PIECES = T * (KB_PLANE / 1024), KB_DMA = (KB_PIECES + 3) / 4;         // 28, 7

__device__ __forceinline__ int ka_h(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

__device__ __forceinline__ f16x8 tr_frag(const unsigned char* base, int off_lo, int off_hi) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base + off_lo));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base + off_hi));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int OCC, bool KSC>
__global__ void __launch_bounds__(NT, OCC) k_gemm_hx2_kmajor(const Hx2KmArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char S[T * KA_PLANE + T * KB_PLANE];      // A image | B image: 45056 B
    unsigned char* const As = S;
    unsigned char* const Bs = S + T * KA_PLANE;
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const TileId tile = xcd_tile(1);
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int bz = tile.z / p.nsplit, zs = tile.z % p.nsplit;
    const int k_begin = zs * p.k_per_split, k_end = min(p.K, k_begin + p.k_per_split);

    // ---- A: piece pc = 4 i + wave (8 per plane) lands lane-linear at slot s = 64 (pc & 7) + lane of its plane
    //      = (k = s >> 4, physical 16-byte slot s & 15); that slot holds logical chunk ((phys >> 1) ^ ka_h(k)), half phys & 1
    int a_k[KA_DMA];
    int64_t a_col[KA_DMA];                                           // with the batch entry's offset: 2 N W halves per head pass 2^31 at a few million nodes
#pragma unroll
    for (int i = 0; i < KA_DMA; ++i) {
        const int pc = 4 * i + wid, s = 64 * (pc & 7) + lane, k = s >> 4, phys = s & 15;
        a_k[i] = k;
        a_col[i] = min(m0 + 16 * ((phys >> 1) ^ ka_h(k)) + 8 * (phys & 1), p.m_ld - 8);
        if (a_col[i] >= p.a_shared_m) a_col[i] += bz * p.a_bs;        // its batch entry's columns; shared ones stay at entry 0
    }
    // ---- B: piece pc = 4 i + wave (14 per plane), slot s = 64 (pc % 14) + lane = (k = s / 28, physical slot s % 28);
    //      the physical slot holds logical slot phys - 2 [k & 8] (the two padding slots re-read slot 0)
    int b_k[KB_DMA], b_col[KB_DMA];
#pragma unroll
    for (int i = 0; i < KB_DMA; ++i) {
        const int pc = min(4 * i + wid, KB_PIECES - 1);
        const int s = 64 * (pc % (KB_PIECES / T)) + lane, k = s / KB_SLOTS, phys = s % KB_SLOTS;
        int slot = phys - ((k & 8) ? 2 : 0);
        if (slot < 0 || slot >= 26) slot = 0;
        b_k[i] = k;
        b_col[i] = min(n0 + 8 * slot, p.n_ld - 8);
    }
    const _Float16* abase = p.Ap;                                    // the batch entry's offset rides in a_col (see a_shared_m)
    const _Float16* bbase = p.Bp + bz * p.b_bs;
    const _Float16* zlane = p.zeros + 8 * lane;
    auto dma = [&](int k0) {
#pragma unroll
        for (int i = 0; i < KA_DMA; ++i) {
            const int pc = 4 * i + wid, k = k0 + a_k[i];
            const _Float16* q = k < k_end ? abase + (pc >> 3) * p.a_plane + static_cast<int64_t>(k) * p.lda + a_col[i] : zlane;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(q),
                                             (__attribute__((address_space(3))) void*)(As + (pc >> 3) * KA_PLANE + 1024 * (pc & 7)), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < KB_DMA; ++i)
            if (4 * i + wid < KB_PIECES) {                                // wave-uniform
                const int pc = 4 * i + wid, pl = pc / (KB_PIECES / T), k = k0 + b_k[i];
                const _Float16* q = k < k_end ? bbase + pl * p.b_plane + static_cast<int64_t>(k) * p.ldb + b_col[i] : zlane;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(q),
                                                 (__attribute__((address_space(3))) void*)(Bs + pl * KB_PLANE + 1024 * (pc % (KB_PIECES / T))), 16, 0, 0);
            }
    };

    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
    auto b_off = [&](int j, int hh) { return b_row[hh] + (2 * j + ((ip & 3) >> 1) + (b_rot[hh] ? 2 : 0)) * 16; };
    // KSC: B's rows k carry their own scales; lane group g's fragments hold k = k0 + 8 g .. + 7, whose factors s_g / s_k ride in kf (requested
    // one step ahead, next to the copies)
    float4 kin[2];
    float sg = 1.f;
    auto load_kinv = [&](int k0) {
        if constexpr (KSC) {
            const float* q = p.k_inv + bz * p.k_inv_bs + min(k0 + 8 * g, ((p.K + 7) & ~7) - 8);   // the table is padded to a multiple of 8 rows; rows past the split are zeros anyway
            kin[0] = *reinterpret_cast<const float4*>(q);
            kin[1] = *reinterpret_cast<const float4*>(q + 4);
        }
    };
    auto mma_tile = [&]() {
        f16x8 a[2][T];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < T; ++q) a[i][q] = tr_frag(As + q * KA_PLANE, a_off[i][0], a_off[i][1]);
        if constexpr (KSC) {
            const float f[8] = {kin[0].x, kin[0].y, kin[0].z, kin[0].w, kin[1].x, kin[1].y, kin[1].z, kin[1].w};
            f16x8 kf;
#pragma unroll
            for (int e = 0; e < 8; ++e) kf[e] = static_cast<_Float16>(fminf(f[e] * sg, 1.f));      // a power of two <= 1 (zero rows: s_k = 1)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < T; ++q) a[i][q] = a[i][q] * kf;
        }
        f16x8 b[2][2][T];
        auto read_pair = [&](int j0, f16x8 (&dst)[2][T]) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                if (j0 + jj < TN) {
                    const int o0 = b_off(j0 + jj, 0), o1 = b_off(j0 + jj, 1);
#pragma unroll
                    for (int q = 0; q < T; ++q) dst[jj][q] = tr_frag(Bs + q * KB_PLANE, o0, o1);
                }
        };
        read_pair(0, b[0]);
#pragma unroll
        for (int gp = 0; gp < (TN + 1) / 2; ++gp) {
            if (2 * gp + 2 < TN) read_pair(2 * gp + 2, b[(gp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (2 * gp + 1 < TN) hx2_products<2>(acc, a, b[gp & 1], 2 * gp);
            else hx2_products<1>(acc, a, b[gp & 1], 2 * gp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ONE buffer per operand: the DMA of tile t+1 starts once every wave is done with tile t; the co-resident workgroups'
    // MFMA phases cover its flight
    if (k_begin < k_end) { dma(k_begin); load_kinv(k_begin); }
    const float ia = hx2_inv(hx2_scale_wave(p.sa));
    sg = hx2_scale_wave(p.sb);                                        // under the first tile's round trip
    const float ib = hx2_inv(sg);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mma_tile();
        __syncthreads();
        if (k0 + BK < k_end) { dma(k0 + BK); load_kinv(k0 + BK); }
    }
    // 104 four-byte stores per wave, rows past M and columns past N masked out: as buffer stores whose offset lies outside the tile's
    // M x N region for a masked lane (the hardware drops those), i.e. without an exec mask and a branch around every one of them
    float* base = p.partial + static_cast<int64_t>(tile.z) * p.M * p.N;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, p.M * p.N * 4, 0x00020000);
    const float sc = ia * ib;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mb + 16 * i + 4 * g + r;
            const int rowoff = row < p.M ? row * p.N : 0x1ffffff0;        // (rowoff + col) * 4 stays below 2^32 and outside the region
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + 16 * j + ip;
                const int off = col < p.N ? (rowoff + col) * 4 : 0x7ffffff0;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, acc[i][j][r] * sc), rsrc, off, 0, 0);
            }
        }
}

// planes[q][r][k] = q-th half term of s * src[r][k] (row stride ld), k < Kp zero padded; one thread per 8 k values.
// TRANS: src is [K][rows] (element (r, k) at src[k*ld + r]).
template <bool TRANS>
__global__ void __launch_bounds__(256) k_hx2_split_planes(const float* __restrict__ src, int64_t ld, int64_t src_bs, int32_t rows, int32_t K,
                                                          int32_t Kp, _Float16* __restrict__ dst, int64_t plane, int64_t dst_bs, const Hx2Scale sc) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int kq = static_cast<int>(idx % (Kp / 8));
    const int r = static_cast<int>(idx / (Kp / 8));
    if (r >= rows) return;
    const float s = hx2_scale(sc);
    const float* sp = src + blockIdx.y * src_bs;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * kq + j;
        v[j] = k < K ? (TRANS ? sp[static_cast<int64_t>(k) * ld + r] : sp[static_cast<int64_t>(r) * ld + k]) : 0.f;
    }
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hx2_split2(v[2 * j] * s, v[2 * j + 1] * s, lo[j], hi[j]);
    _Float16* d = dst + blockIdx.y * dst_bs + static_cast<int64_t>(r) * Kp + 8 * kq;
    *reinterpret_cast<u32x4*>(d) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    *reinterpret_cast<u32x4*>(d + plane) = u32x4{hi[0], hi[1], hi[2], hi[3]};
}

// Both orientations of ONE batched matrix in a single launch: hx2_split_both_block (recon_common.h) per block
__global__ void __launch_bounds__(256) k_hx2_split_both(const Hx2SplitBoth p) { hx2_split_both_block(p, blockIdx.x, blockIdx.y, blockIdx.z); }

// *slot = max(*slot, max |src[r][c]|) as fp32 bit pattern (non-negative floats order like unsigned integers); the slot
// must have been zeroed.  max is exact and order independent, so the atomics do not cost determinism.
template <int NTHREADS>
__global__ void __launch_bounds__(NTHREADS) k_hx2_amax(const float* __restrict__ src, int64_t rows, int32_t cols, int64_t ld, uint32_t* __restrict__ slot) {
    float m = 0.f;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * NTHREADS;
    if (ld == cols && !(reinterpret_cast<uintptr_t>(src) & 15)) {
        const int64_t n = rows * cols, n4 = n >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        int64_t i = static_cast<int64_t>(blockIdx.x) * NTHREADS + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {                  // four independent 16-byte loads in flight per lane
            const float4 v0 = s4[i], v1 = s4[i + stride], v2 = s4[i + 2 * stride], v3 = s4[i + 3 * stride];
            const float a = fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w)));
            const float b = fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)));
            const float c = fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w)));
            const float d = fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)));
            m = fmaxf(m, fmaxf(fmaxf(a, b), fmaxf(c, d)));
        }
        for (; i < n4; i += stride) {
            const float4 v = s4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(src[(n4 << 2) + threadIdx.x]));
    } else {
        const int64_t n = rows * cols;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * NTHREADS + threadIdx.x; i < n; i += stride)
            m = fmaxf(m, fabsf(src[(i / cols) * ld + (i % cols)]));
    }
    // one commit per workgroup: wave maxima through LDS first
    __shared__ float wm[NTHREADS / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < NTHREADS / 64 ? wm[threadIdx.x] : 0.f;
        hx2_amax_commit(v, slot);
    }
}

}  // namespace

int32_t hx2_kp(int32_t K) { return (K + BK - 1) / BK * BK; }

int hx2_amax(const float* src, int64_t rows, int32_t cols, int64_t ld, uint32_t* slot, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return RECON_OK;
    if (!src || !slot) return RECON_ERR_INVALID;
    const int64_t n = rows * cols;
    // 512 threads x <= 1024 blocks (32 waves per CU), one commit per block: 13.5 us for 52 MB at cfg 2 (1024 x 512: 14.1, 256 x 2048: 14.8;
    // one commit per WAVE: 17.7)
    int64_t blocks = ceil_div64(n, 512 * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL((k_hx2_amax<512>), dim3(static_cast<unsigned>(blocks)), dim3(512), 0, st, src, rows, cols, ld, slot);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// layout [2][batch][rows][Kp]: plane stride batch*rows*Kp, batch stride rows*Kp
int hx2_split_planes(const float* src, int64_t ld, int64_t src_bs, bool transposed, int32_t rows, int32_t K, int32_t batch, void* dst,
                     const Hx2Scale& sc, hipStream_t st) {
    if (rows <= 0 || K <= 0 || batch <= 0) return RECON_OK;
    if (!src || !dst || (reinterpret_cast<uintptr_t>(dst) & 15)) return RECON_ERR_INVALID;
    const int32_t Kp = hx2_kp(K);
    const int64_t per = static_cast<int64_t>(rows) * Kp;
    const dim3 grid(static_cast<unsigned>(ceil_div64(per / 8, 256)), static_cast<unsigned>(batch));
    if (transposed) hipLaunchKernelGGL((k_hx2_split_planes<true>), grid, dim3(256), 0, st, src, ld, src_bs, rows, K, Kp, static_cast<_Float16*>(dst), per * batch, per, sc);
    else hipLaunchKernelGGL((k_hx2_split_planes<false>), grid, dim3(256), 0, st, src, ld, src_bs, rows, K, Kp, static_cast<_Float16*>(dst), per * batch, per, sc);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// planes of src[b][R][C] (contiguous rows) and of its transpose in one launch: dst_n [2][batch][R][kp(C)], dst_t [2][batch][C][kp(R)]
bool hx2_split_both_args(const float* src, int64_t src_bs, int32_t R, int32_t C_, int32_t batch, void* dst_n, void* dst_t, Hx2SplitBoth* p) {
    if (!src || !dst_n || !dst_t || ((reinterpret_cast<uintptr_t>(dst_n) | reinterpret_cast<uintptr_t>(dst_t)) & 15)) return false;
    p->src = src; p->src_bs = src_bs; p->R = R; p->C = C_; p->Cp = hx2_kp(C_); p->Rp = hx2_kp(R); p->batch = batch;
    const int64_t n0 = static_cast<int64_t>(R) * (p->Cp / 8), n1 = static_cast<int64_t>(C_) * (p->Rp / 8);
    p->gx = static_cast<int32_t>(ceil_div64(n0 > n1 ? n0 : n1, 256));
    p->dst_n = static_cast<_Float16*>(dst_n); p->dst_t = static_cast<_Float16*>(dst_t);
    p->sc = Hx2Scale{nullptr, nullptr, 1.f}; p->blkmax_quantity = nullptr; p->nblk = 0;
    return true;
}
int hx2_split_planes_both(const float* src, int64_t src_bs, int32_t R, int32_t C_, int32_t batch, void* dst_n, void* dst_t, const Hx2Scale& sc,
                          hipStream_t st, uint32_t* blkmax_quantity, int32_t nblk) {
    if (R <= 0 || C_ <= 0 || batch <= 0) return RECON_OK;
    Hx2SplitBoth p;
    if (!hx2_split_both_args(src, src_bs, R, C_, batch, dst_n, dst_t, &p)) return RECON_ERR_INVALID;
    p.sc = sc; p.blkmax_quantity = blkmax_quantity; p.nblk = nblk;
    hipLaunchKernelGGL(k_hx2_split_both, dim3(static_cast<unsigned>(p.gx), static_cast<unsigned>(batch), 2), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

bool hx2_supported(const void* Ap, int64_t a_plane, int64_t a_row, int64_t a_bs, int32_t K) {
    if (K <= 0 || (K & 7)) return false;
    return !((reinterpret_cast<uintptr_t>(Ap) & 15) || (a_plane & 7) || (a_row & 7) || (a_bs & 7));
}

// A: half planes, element (plane q, batch z, row m, k) at Ap[q*a_plane + z*a_bs + m*a_row + k]; B planes [2][batch][N][hx2_kp(K)]
int gemm_hx2_batched(int32_t M, int32_t N, int32_t K, const void* Ap, int64_t a_plane, int64_t a_row, int64_t a_bs, const void* Bplanes,
                     const OutputDesc& C, const GemmBatch& bt, const Hx2Scale& sa, const Hx2Scale& sb, hipStream_t st, int32_t a_shared_k,
                     const float* row_inv, int64_t row_inv_bs) {
    if (M < 0 || N < 0 || K < 0 || bt.batch < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0 || bt.batch == 0) return RECON_OK;
    if (!Ap || !Bplanes || !C.base) return RECON_ERR_INVALID;
    if (!hx2_supported(Ap, a_plane, a_row, a_bs, K) || bt.c_transpose || bt.batch > 65535 || (reinterpret_cast<uintptr_t>(Bplanes) & 15))
        return RECON_ERR_UNSUPPORTED;
    Hx2Args a;
    a.Ap = static_cast<const _Float16*>(Ap); a.a_plane = a_plane; a.a_row = a_row; a.a_bs = a_bs;
    const int32_t Kp = hx2_kp(K);
    a.Bp = static_cast<const _Float16*>(Bplanes);
    a.b_row = Kp;
    a.b_bs = static_cast<int64_t>(N) * Kp;
    a.b_plane = a.b_bs * bt.batch;
    if (T * a.b_plane >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;    // 32-bit element offsets inside the B planes
    a.C = C; a.c_bs = bt.c_bs; a.M = M; a.N = N; a.K = K; a.epilogue = bt.epilogue;
    a.c_vec4 = (!(N & 3) && !(bt.c_bs & 3) && !(reinterpret_cast<uintptr_t>(C.base) & 15) && !(C.S1 & 3) && !(C.S2 & 3) && !(C.Sseg & 3) &&
                (C.Dseg >= N || !(C.Dseg & 3))) ? 1 : 0;
    a.xcd_remap = 1;
    if (a_shared_k < 0 || (a_shared_k & 7) || (a_shared_k && a_bs * bt.batch >= (1LL << 31))) return RECON_ERR_INVALID;
    a.a_shared_k = a_shared_k;
    a.c_plain = (a.c_vec4 && !C.scatter && C.P >= M && C.Dseg >= N && (static_cast<int64_t>(M) * C.S1 + N) * 4 < (1LL << 31)) ? 1 : 0;   // plain rows: the straight-line store
    a.sa = sa; a.sb = sb;
    a.row_inv = row_inv; a.row_inv_bs = row_inv_bs;
    if (row_inv && sa.p0) return RECON_ERR_INVALID;                      // per-row scales: no tensor scale beside them
    // 256-row workgroups for long K and enough rows to fill the chip with them (see k_gemm_hx2)
    const bool wide = K >= 1024 && ceil_div64(M, 256) * ceil_div64(N, BN) * bt.batch >= 256;
    const dim3 grid(static_cast<unsigned>(ceil_div64(N, BN)), static_cast<unsigned>(ceil_div64(M, wide ? 256 : BM)), static_cast<unsigned>(bt.batch));
    // the three-deep A ring (k_gemm_hx2_r3) wherever 32-bit byte offsets reach all of A; RECON_HX2_RING=0: the two-stage form
    const int64_t a_span = ((T - 1) * a_plane + (bt.batch - 1) * a_bs + static_cast<int64_t>(M - 1) * a_row + hx2_kp(K) + 8) * 2;
    a.a_span_bytes = (a_plane >= 0 && a_bs >= 0 && a_row >= 0 && a_span < (1LL << 31) - 64) ? static_cast<int32_t>(a_span) : 0;
    const bool ring = !wide && a.a_span_bytes > 0 && cfg_char(CFG_HX2_RING) != '0';
    if (wide) hipLaunchKernelGGL((k_gemm_hx2<8>), grid, dim3(512), 0, st, a);
    else if (ring) hipLaunchKernelGGL((k_gemm_hx2_r3<4>), grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_hx2<4>), grid, dim3(NT), 0, st, a);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// every plane row must hold (M resp. N rounded up to 8) columns, zero padded
bool hx2_kmajor_supported(const void* Ap, int64_t lda, int64_t a_plane, int64_t a_bs, const void* Bp, int64_t ldb, int64_t b_plane,
                          int64_t b_bs, int32_t M, int32_t N) {
    const int64_t m_ld = (static_cast<int64_t>(M) + 7) / 8 * 8, n_ld = (static_cast<int64_t>(N) + 7) / 8 * 8;
    if (M < 1 || N < 1 || ((lda | a_plane | a_bs | ldb | b_plane | b_bs) & 7) || m_ld > lda || n_ld > ldb) return false;
    if (((M & 7) && a_bs != 0) || ((N & 7) && b_bs != 0)) return false;     // batched heads sit side by side in a row: no room for padding
    return !((reinterpret_cast<uintptr_t>(Ap) | reinterpret_cast<uintptr_t>(Bp)) & 15);
}

// partial[batch][split][M][N] = A_slice^T . B_slice / (s_a s_b);  a_bs / b_bs = per-batch column offsets;
// split_k must be bx3_kmajor_splits(K, requested) (same K-tile rounding as the bf16 x 3 kernel)
int gemm_hx2_kmajor_batched(int32_t M, int32_t N, int32_t K, const void* Ap, int64_t lda, int64_t a_plane, int64_t a_bs, const void* Bp,
                            int64_t ldb, int64_t b_plane, int64_t b_bs, int32_t batch, int32_t split_k, float* partial, const void* zeros,
                            const Hx2Scale& sa, const Hx2Scale& sb, hipStream_t st, int32_t a_shared_m, const float* k_inv, int64_t k_inv_bs) {
    if (M < 0 || N < 0 || K < 0 || batch < 0 || split_k < 1) return RECON_ERR_INVALID;
    if (M == 0 || N == 0 || batch == 0) return RECON_OK;
    if (!Ap || !Bp || !partial || !zeros || (reinterpret_cast<uintptr_t>(zeros) & 15)) return RECON_ERR_INVALID;
    if (!hx2_kmajor_supported(Ap, lda, a_plane, a_bs, Bp, ldb, b_plane, b_bs, M, N)) return RECON_ERR_UNSUPPORTED;
    Hx2KmArgs a;
    a.Ap = static_cast<const _Float16*>(Ap); a.Bp = static_cast<const _Float16*>(Bp); a.zeros = static_cast<const _Float16*>(zeros);
    a.lda = lda; a.ldb = ldb; a.a_plane = a_plane; a.b_plane = b_plane; a.a_bs = a_bs; a.b_bs = b_bs;
    a.partial = partial; a.M = M; a.N = N; a.K = K;
    a.m_ld = (M + 7) / 8 * 8; a.n_ld = (N + 7) / 8 * 8;
    int64_t kps = ceil_div64(K > 0 ? K : 1, split_k);
    kps = ceil_div64(kps, BK) * BK;
    a.k_per_split = static_cast<int32_t>(kps);
    a.nsplit = static_cast<int32_t>(ceil_div64(K > 0 ? K : 1, kps));
    if (a.nsplit != split_k) return RECON_ERR_INVALID;
    if (static_cast<int64_t>(batch) * split_k > 65535) return RECON_ERR_UNSUPPORTED;
    a.sa = sa; a.sb = sb;
    if (a_shared_m < 0 || (a_shared_m & 7)) return RECON_ERR_UNSUPPORTED;
    a.a_shared_m = a_shared_m;
    const dim3 grid(static_cast<unsigned>(ceil_div64(N, BN)), static_cast<unsigned>(ceil_div64(M, BM)), static_cast<unsigned>(batch * split_k));
    a.k_inv = k_inv; a.k_inv_bs = k_inv_bs;
    if (k_inv && ((k_inv_bs & 3) || k_inv_bs < ((K + 7) & ~7) || (reinterpret_cast<uintptr_t>(k_inv) & 15))) return RECON_ERR_UNSUPPORTED;
    // three workgroups per CU (<= 168 registers) spill: 241 us against 70
    if (k_inv) hipLaunchKernelGGL((k_gemm_hx2_kmajor<2, true>), grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_hx2_kmajor<2, false>), grid, dim3(NT), 0, st, a);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace recon

// ---- stand-alone entries (tests, tools/gemm_bench.py): both operands are measured (amax) and split into `workspace` first
//      workspace layout: kHx2AuxBytes of amax quantities (zeroed here) + page of zeros | A planes | B planes | split-K partials
static constexpr size_t kHdr = (recon::kHx2AuxBytes + 255) / 256 * 256;
extern "C" size_t recon_hx2_aux_bytes(void) { return recon::kHx2AuxBytes; }
static size_t hx2_planes_bytes(int64_t rows, int32_t K) { return align_up(static_cast<size_t>(2) * rows * recon::hx2_kp(K) * 2, 256); }

extern "C" size_t recon_sgemm_hx2_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return kHdr;
    return kHdr + hx2_planes_bytes(M, K) + hx2_planes_bytes(N, K);
}

extern "C" int recon_sgemm_hx2(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb, float* C_,
                               int32_t ldc, void* workspace, recon_stream_t stream) {
    using namespace recon;
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    if (!A || !B || !C_ || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 255)) return RECON_ERR_INVALID;
    if (K & 7) return RECON_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* slots = reinterpret_cast<uint32_t*>(ws);
    if (hipMemsetAsync(ws, 0, kHdr, st) != hipSuccess) return RECON_ERR_LAUNCH;
    int rc = hx2_amax(A, M, K, lda, slots + 0, st);
    if (rc == RECON_OK) rc = hx2_amax(B, N, K, ldb, slots + kHx2QuantityWords, st);
    if (rc != RECON_OK) return rc;
    const Hx2Scale sa{slots + 0, nullptr, 1.f}, sb{slots + kHx2QuantityWords, nullptr, 1.f};
    char* ap = ws + kHdr;
    char* bp = ap + hx2_planes_bytes(M, K);
    rc = hx2_split_planes(A, lda, 0, false, M, K, 1, ap, sa, st);
    if (rc == RECON_OK) rc = hx2_split_planes(B, ldb, 0, false, N, K, 1, bp, sb, st);
    if (rc != RECON_OK) return rc;
    GemmBatch bt;
    bt.batch = 1; bt.a_bs = bt.b_bs = bt.c_bs = 0; bt.epilogue = GEMM_EPI_NONE;
    const int64_t Kp = hx2_kp(K);
    return gemm_hx2_batched(M, N, K, ap, static_cast<int64_t>(M) * Kp, Kp, 0, bp, plain_output(C_, ldc), bt, sa, sb, st);
}

// the GEMM alone on operands already split by a previous recon_sgemm_hx2 call into the same workspace (benchmarks)
extern "C" int recon_sgemm_hx2_presplit(int32_t M, int32_t N, int32_t K, float* C_, int32_t ldc, void* workspace, recon_stream_t stream) {
    using namespace recon;
    if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || !C_ || !workspace) return RECON_ERR_INVALID;
    char* ws = static_cast<char*>(workspace);
    uint32_t* slots = reinterpret_cast<uint32_t*>(ws);
    const Hx2Scale sa{slots + 0, nullptr, 1.f}, sb{slots + kHx2QuantityWords, nullptr, 1.f};
    char* ap = ws + kHdr;
    char* bp = ap + hx2_planes_bytes(M, K);
    GemmBatch bt;
    bt.batch = 1; bt.a_bs = bt.b_bs = bt.c_bs = 0; bt.epilogue = GEMM_EPI_NONE;
    const int64_t Kp = hx2_kp(K);
    return gemm_hx2_batched(M, N, K, ap, static_cast<int64_t>(M) * Kp, Kp, 0, bp, plain_output(C_, ldc), bt, sa, sb, as_stream(stream));
}

// C[M,N] = A^T . B for k-major operands A = [K][M], B = [K][N] (the weight-gradient form)
extern "C" size_t recon_sgemm_hx2_tn_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    using namespace recon;
    if (M <= 0 || N <= 0 || K <= 0) return kHdr;
    const int sk = bx3_kmajor_splits(K, bx3_kmajor_split_k(M, N, K, 1));
    return kHdr + hx2_planes_bytes(K, M) + hx2_planes_bytes(K, N) + static_cast<size_t>(sk) * M * N * sizeof(float) + 256;
}

static int hx2_tn(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb, float* C_, int32_t ldc,
                  void* workspace, bool split, recon_stream_t stream) {
    using namespace recon;
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    if (!C_ || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 255)) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* slots = reinterpret_cast<uint32_t*>(ws);
    const Hx2Scale sa{slots + 0, nullptr, 1.f}, sb{slots + kHx2QuantityWords, nullptr, 1.f};
    char* ap = ws + kHdr;
    char* bp = ap + hx2_planes_bytes(K, M);
    float* partial = reinterpret_cast<float*>(bp + hx2_planes_bytes(K, N));
    const int64_t Mp = hx2_kp(M), Np = hx2_kp(N);
    int rc = RECON_OK;
    if (split) {
        if (!A || !B) return RECON_ERR_INVALID;
        if (hipMemsetAsync(ws, 0, kHdr, st) != hipSuccess) return RECON_ERR_LAUNCH;
        rc = hx2_amax(A, K, M, lda, slots + 0, st);
        if (rc == RECON_OK) rc = hx2_amax(B, K, N, ldb, slots + kHx2QuantityWords, st);
        if (rc == RECON_OK) rc = hx2_split_planes(A, lda, 0, false, K, M, 1, ap, sa, st);      // rows = k, minor = m (zero padded to Mp)
        if (rc == RECON_OK) rc = hx2_split_planes(B, ldb, 0, false, K, N, 1, bp, sb, st);
        if (rc != RECON_OK) return rc;
    }
    const int sk = bx3_kmajor_splits(K, bx3_kmajor_split_k(M, N, K, 1));
    rc = gemm_hx2_kmajor_batched(M, N, K, ap, Mp, static_cast<int64_t>(K) * Mp, 0, bp, Np, static_cast<int64_t>(K) * Np, 0, 1, sk, partial,
                                 ws + kHx2ZeroPageOffset, sa, sb, st);
    if (rc != RECON_OK) return rc;
    return splitk_reduce(partial, sk, M, N, plain_output(C_, ldc), 0, 1, GEMM_EPI_NONE, false, st);
}

extern "C" int recon_sgemm_hx2_tn(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb, float* C_,
                                  int32_t ldc, void* workspace, recon_stream_t stream) {
    return hx2_tn(M, N, K, A, lda, B, ldb, C_, ldc, workspace, true, stream);
}
extern "C" int recon_sgemm_hx2_tn_presplit(int32_t M, int32_t N, int32_t K, float* C_, int32_t ldc, void* workspace, recon_stream_t stream) {
    return hx2_tn(M, N, K, nullptr, 0, nullptr, 0, C_, ldc, workspace, false, stream);
}
