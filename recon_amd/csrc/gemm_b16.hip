// bf16 GEMMs of the bfloat16 GraphConvolution path (BASELINE.json configs[2]: "bf16 + MFMA on W-projection").
//
// Plain bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16, bf16 (or fp32) results: what torch.mm does for bf16
// tensors, i.e. what the reference's layer (models/layers.py:57-63) computes when its tensors are bfloat16.  With ONE MFMA per
// 16x16x32 block these products are memory bound (x @ W at cfg 3a: 6.5 GFLOP = 2.6 us of matrix pipe against 40 MB of
// traffic), so the kernels are the f16 x 2 kernels of gemm_hx2.hip with the second term and its products removed: same
// 128 x 208 x 32 tiles, A fragments loaded straight from global memory, B through LDS-DMA into a double-buffered image.
//
//   k-contiguous form   C[M,N] = A[M,K] . Bp[N,Kp]^T      A rows 16-byte aligned (lda % 8 == 0), Bp zero padded to Kp = 32 k
//   k-major form        C[M,N] = A[K,M]^T . B[K,N]        the weight gradient x^T g_support; split-K partials in fp32
#include <stdlib.h>
#include "gemm_common.h"

namespace recon {
int32_t b16_kp(int32_t K);
namespace {

constexpr int BM = 128, BN = 208, BK = 32, NT = 256, TN = 13;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using i16x4 = __attribute__((ext_vector_type(4))) short;

struct B16Args {
    const uint16_t* A; const uint16_t* Bp;
    int64_t lda, ldb;              // elements; ldb = Kp
    void* C; int64_t ldc;          // bf16 or fp32 rows of ldc elements
    int32_t M, N, K;
};

__device__ __forceinline__ int lds_off(int row, int kq) { return row * 64 + (((kq + 2 * (row >> 3)) & 3) << 4); }
__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(v)); }
typedef __bf16 bf16x2_b __attribute__((ext_vector_type(2)));
typedef float f32x2_b __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_b{a, b}, bf16x2_b)); }   // one v_cvt_pk_bf16_f32

constexpr int B_TILE_BYTES = BN * 64;                            // 13312
constexpr int B_PIECES = B_TILE_BYTES / 1024;                    // 13
constexpr int B_DMA = (B_PIECES + 3) / 4;                        // 4

template <bool OUT_BF16>
__global__ void __launch_bounds__(NT, 2) k_gemm_b16(const B16Args p) {
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][B_TILE_BYTES];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const TileId tile = xcd_tile(1);
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int mb = wid * 32;
    const int li = lane & 15, lq = lane >> 4;

    const uint16_t* aptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aptr[i] = p.A + static_cast<int64_t>(min(m0 + mb + 16 * i + li, p.M - 1)) * p.lda + 8 * lq;
    int b_goff[B_DMA];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int s = min(64 * (4 * i + wid) + lane, B_TILE_BYTES / 16 - 1);
        const int rowL = s >> 2, pslot = s & 3;
        const int kq = (pslot - 2 * (rowL >> 3)) & 3;                 // inverse of lds_off's rotation
        const int j = rowL >> 4, rho = rowL & 15;
        const int col = j < 12 ? 64 * (j >> 2) + 4 * rho + (j & 3) : 192 + rho;      // tile 4q+t <-> columns 64q + 4i + t
        b_goff[i] = static_cast<int>(static_cast<int64_t>(min(n0 + col, p.N - 1)) * p.ldb + 8 * kq);
    }
    auto dma_b = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < B_DMA; ++i)
            if (4 * i + wid < B_PIECES)                                // wave-uniform
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(p.Bp + b_goff[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(&Bs[buf][1024 * (4 * i + wid)]), 16, 0, 0);
    };
    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 araw[2];
    bf16x8 af[2];
    bool a_ok = true;
    auto load_a = [&](int k0) {                                      // branch free: lanes past K re-read the start of their row
        a_ok = k0 + 8 * lq < p.K;
        const int off = -8 * lq + ((k0 + 8 * lq) & -static_cast<int>(a_ok));
#pragma unroll
        for (int i = 0; i < 2; ++i) araw[i] = *reinterpret_cast<const u32x4*>(aptr[i] + off);
    };
    auto take_a = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a_ok ? araw[i][e] : 0u;
            af[i] = __builtin_bit_cast(bf16x8, v);
        }
    };
    const int b_rd = lds_off(li, lq);
    auto mma = [&](const unsigned char* Bt) {
        bf16x8 b[2][2];
        auto read_pair = [&](int j0, bf16x8 (&dst)[2]) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                if (j0 + jj < TN) dst[jj] = *reinterpret_cast<const bf16x8*>(Bt + b_rd + (j0 + jj) * 1024);
        };
        read_pair(0, b[0]);
#pragma unroll
        for (int g = 0; g < (TN + 1) / 2; ++g) {
            if (2 * g + 2 < TN) read_pair(2 * g + 2, b[(g + 1) & 1]);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                if (2 * g + jj < TN)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][2 * g + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], b[g & 1][jj], acc[i][2 * g + jj], 0, 0, 0);
        }
    };

    dma_b(0, 0);
    load_a(0);
    take_a();
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        if (k0 + BK < p.K) dma_b(k0 + BK, buf ^ 1);
        load_a(k0 + BK);
        mma(Bs[buf]);
        take_a();
        __syncthreads();
        buf ^= 1;
    }
    // MFMA C layout col = lane&15, row = (lane>>4)*4 + r; four neighbouring tiles = four consecutive columns
    const bool v4 = !(p.ldc & 3) && !(reinterpret_cast<uintptr_t>(p.C) & 15);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mb + 16 * i + 4 * lq + r;
            if (row >= p.M) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int col = n0 + 64 * q + 4 * li;
                const float v0 = acc[i][4 * q][r], v1 = acc[i][4 * q + 1][r], v2 = acc[i][4 * q + 2][r], v3 = acc[i][4 * q + 3][r];
                if constexpr (OUT_BF16) {
                    uint16_t* crow = static_cast<uint16_t*>(p.C) + static_cast<int64_t>(row) * p.ldc;
                    if (v4 && col + 3 < p.N) {
                        *reinterpret_cast<uint2*>(crow + col) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
                    } else {
                        const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) if (col + jj < p.N) crow[col + jj] = f2bf(vv[jj]);
                    }
                } else {
                    float* crow = static_cast<float*>(p.C) + static_cast<int64_t>(row) * p.ldc;
                    if (v4 && col + 3 < p.N) *reinterpret_cast<float4*>(crow + col) = make_float4(v0, v1, v2, v3);
                    else {
                        const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) if (col + jj < p.N) crow[col + jj] = vv[jj];
                    }
                }
            }
            const int col = n0 + 192 + li;
            if (col < p.N) {
                if constexpr (OUT_BF16) static_cast<uint16_t*>(p.C)[static_cast<int64_t>(row) * p.ldc + col] = f2bf(acc[i][12][r]);
                else static_cast<float*>(p.C)[static_cast<int64_t>(row) * p.ldc + col] = acc[i][12][r];
            }
        }
}

// ---- k-major form: partial[z][M][N] = A[ks..ke, :M]^T . B[ks..ke, :N]; both tiles by LDS-DMA into row-major images, fragments
//      through the transposing read (layouts of gemm_hx2.hip / gemm_bx3.hip); rows past the K range come from a page of zeros
constexpr int kKmJobs = 8;                                           // products of one launch (the layers of a GraphConvolution stack)
struct B16KmArgs {
    const uint16_t* A[kKmJobs]; const uint16_t* B[kKmJobs]; const uint16_t* zeros;
    int64_t lda[kKmJobs], ldb[kKmJobs];
    float* partial[kKmJobs];
    int32_t M[kKmJobs], N[kKmJobs], m_ld[kKmJobs], n_ld[kKmJobs];
    int32_t K, k_per_split, nsplit;                                    // shared: grid.z = jobs x nsplit
};
constexpr int KB_SLOTS = 28;
constexpr int KA_BYTES = BK * 256, KB_BYTES = BK * KB_SLOTS * 16;                      // 8192, 14336
constexpr int KA_PIECES = KA_BYTES / 1024, KB_PIECES = KB_BYTES / 1024;               // 8, 14
__device__ __forceinline__ int ka_h(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base, int off_lo, int off_hi) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base + off_lo));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(base + off_hi));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Three-stage ring of 22-KiB images, copies two K steps ahead, ONE counted wait + a raw barrier per step (round 5; the first form
// double-buffered behind `vmcnt(0)` + __syncthreads and exposed one L2 -> LDS round trip in every step: 37 steps of a cfg 3a weight
// gradient were 29 us for 416 cycles of MFMA per step).  The copies are issued as inline asm (dma16_to_lds): through the builtin the
// compiler drains the request counter in front of the transposing reads.
constexpr int KM_STAGES = 3, KM_STAGE_BYTES = KA_BYTES + KB_BYTES;
__global__ void __launch_bounds__(NT, 2) k_gemm_b16_kmajor(const B16KmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char km_sm[];                  // [KM_STAGES][22 KiB]
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const TileId tile = xcd_tile(1);
    const int job = tile.z / p.nsplit, zs = tile.z - job * p.nsplit;
    const int m0 = tile.y * BM, n0 = tile.x * BN;
    const int pM = p.M[job], pN = p.N[job], p_m_ld = p.m_ld[job], p_n_ld = p.n_ld[job];
    if (m0 >= pM || n0 >= pN) return;                                  // (the grid covers the largest job)
    const uint16_t* const pA = p.A[job];
    const uint16_t* const pB = p.B[job];
    const int64_t p_lda = p.lda[job], p_ldb = p.ldb[job];
    const int k_begin = zs * p.k_per_split, k_end = min(p.K, k_begin + p.k_per_split);
    // pieces 0..7: A image (piece = 4 k rows of 16 slots), pieces 8..21: B image (64 slots each, 28 per k row): 22 pieces over 4 waves
    constexpr int NP = KA_PIECES + KB_PIECES, ND = (NP + 3) / 4;
    int d_k[ND], d_col[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int pc = min(4 * i + wid, NP - 1);
        if (pc < KA_PIECES) {
            const int s = 64 * pc + lane, k = s >> 4, phys = s & 15;
            d_k[i] = k;
            d_col[i] = min(m0 + 16 * ((phys >> 1) ^ ka_h(k)) + 8 * (phys & 1), p_m_ld - 8);
        } else {
            const int s = 64 * (pc - KA_PIECES) + lane, k = s / KB_SLOTS, phys = s % KB_SLOTS;
            int slot = phys - ((k & 8) ? 2 : 0);
            if (slot < 0 || slot >= 26) slot = 0;
            d_k[i] = k;
            d_col[i] = min(n0 + 8 * slot, p_n_ld - 8);
        }
    }
    const uint16_t* zlane = p.zeros + 8 * lane;
    auto dma = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < ND; ++i)
            if (4 * i + wid < NP) {                                       // wave-uniform
                const int pc = 4 * i + wid, k = k0 + d_k[i];
                const bool isa = pc < KA_PIECES;
                const uint16_t* q = k < k_end ? (isa ? pA + static_cast<int64_t>(k) * p_lda : pB + static_cast<int64_t>(k) * p_ldb) + d_col[i] : zlane;
                unsigned char* dst = km_sm + buf * KM_STAGE_BYTES + (isa ? 1024 * pc : KA_BYTES + 1024 * (pc - KA_PIECES));
                dma16_to_lds(q, dst);
            }
    };
    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    if (k_begin < k_end) dma(k_begin, 0);
    if (k_begin + BK < k_end) dma(k_begin + BK, 1);
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        // this wave's copies of this step have landed — those of the next step (ND per wave, ND - 1 for the waves that own one piece less:
        // the smaller count is right for both) may still travel; at the last step nothing newer exists
        if (k0 + BK < k_end) dma_wait<ND - 1>(); else dma_wait<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // ... everyone's have, and nobody still reads the buffer requested next
        asm volatile("" ::: "memory");
        if (k0 + 2 * BK < k_end) dma(k0 + 2 * BK, buf + 2 >= KM_STAGES ? buf + 2 - KM_STAGES : buf + 2);
        const unsigned char* As = km_sm + buf * KM_STAGE_BYTES;
        const unsigned char* Bs = As + KA_BYTES;
        bf16x8 a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = tr_frag(As, a_off[i][0], a_off[i][1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bf16x8 b = tr_frag(Bs, b_off(j, 0), b_off(j, 1));
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][j], 0, 0, 0);
        }
        buf = buf + 1 == KM_STAGES ? 0 : buf + 1;
    }
    float* base = p.partial[job] + static_cast<int64_t>(zs) * pM * pN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mb + 16 * i + 4 * g + r;
            if (row >= pM) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + 16 * j + ip;
                if (col < pN) base[static_cast<int64_t>(row) * pN + col] = acc[i][j][r];
            }
        }
}

// dst[r][k] = bf16 src element, zero padded to Kp columns: TRANS reads src as [K][rows] (dst = src^T)
template <bool TRANS>
__global__ void __launch_bounds__(256) k_b16_pad_planes(const uint16_t* __restrict__ src, int64_t ld, int32_t rows, int32_t K, int32_t Kp,
                                                        uint16_t* __restrict__ dst) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= static_cast<int64_t>(rows) * Kp) return;
    const int r = static_cast<int>(TRANS ? idx % rows : idx / Kp), k = static_cast<int>(TRANS ? idx / rows : idx % Kp);
    dst[static_cast<int64_t>(r) * Kp + k] = k < K ? (TRANS ? src[static_cast<int64_t>(k) * ld + r] : src[static_cast<int64_t>(r) * ld + k]) : 0;
}

// W^T [rows_t = N][kp(K_t = M)] and W [rows_n = M][kp(K_n = N)] of one [M][N] weight in ONE launch (blocks [0, nb_t) the transposed job)
__global__ void __launch_bounds__(256) k_b16_pad_both(const uint16_t* __restrict__ src, int64_t ld, int32_t M, int32_t N, int32_t Kp_t, int32_t Kp_n,
                                                      uint16_t* __restrict__ dst_t, uint16_t* __restrict__ dst_n, int32_t nb_t) {
    if (static_cast<int>(blockIdx.x) < nb_t) {
        const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
        if (idx >= static_cast<int64_t>(N) * Kp_t) return;
        const int r = static_cast<int>(idx % N), k = static_cast<int>(idx / N);
        dst_t[static_cast<int64_t>(r) * Kp_t + k] = k < M ? src[static_cast<int64_t>(k) * ld + r] : 0;
    } else {
        const int64_t idx = static_cast<int64_t>(blockIdx.x - nb_t) * 256 + threadIdx.x;
        if (idx >= static_cast<int64_t>(M) * Kp_n) return;
        const int r = static_cast<int>(idx / Kp_n), k = static_cast<int>(idx % Kp_n);
        dst_n[static_cast<int64_t>(r) * Kp_n + k] = k < N ? src[static_cast<int64_t>(r) * ld + k] : 0;
    }
}

// out (bf16 [M][N], row stride ldo) = sum of `splits` fp32 partials [z][M][N]: 16 elements x 16 split groups per block, fixed order.
// A second job of the same shape (the bias gradient of the GraphConvolution backward: [nb][O] column sums -> [O]) rides in the same launch.
__global__ void __launch_bounds__(256) k_b16_reduce(const B16ReduceJob j0, const B16ReduceJob j1, int32_t nblk0) {
    __shared__ float red[16][17];
    const bool first = static_cast<int>(blockIdx.x) < nblk0;
    const B16ReduceJob& j = first ? j0 : j1;
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t idx = static_cast<int64_t>(first ? blockIdx.x : blockIdx.x - nblk0) * 16 + e, MN = static_cast<int64_t>(j.M) * j.N;
    const int per = (j.splits + 15) / 16;
    const int z0 = grp * per, z1 = min(j.splits, (grp + 1) * per);
    float s = 0.f;
    if (idx < MN) {                                                    // eight independent loads in flight: a chain of 64 dependent round trips (the
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // bias job at B = 1024) would cost more than the launch it saves
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += j.partial[(z + u) * MN + idx];
        }
        for (; z < z1; ++z) a[0] += j.partial[z * MN + idx];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[grp][e] = s;
    __syncthreads();
    if (grp == 0 && idx < MN) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][e];
        j.out[(idx / j.N) * j.ldo + idx % j.N] = f2bf(t);
    }
}

// k_b16_pad_both for the weights of up to kKmJobs layers in one launch: blocks [first[j], first[j + 1]) belong to layer j
struct B16PadMulti { const uint16_t* src[kKmJobs]; uint16_t* dst_t[kKmJobs]; uint16_t* dst_n[kKmJobs]; int32_t M[kKmJobs], N[kKmJobs], nb_t[kKmJobs], first[kKmJobs + 1]; int32_t count; };
__global__ void __launch_bounds__(256) k_b16_pad_both_multi(const B16PadMulti q) {
    int j = 0;
    while (j + 1 < q.count && static_cast<int>(blockIdx.x) >= q.first[j + 1]) ++j;
    const int blk = static_cast<int>(blockIdx.x) - q.first[j];
    const int32_t M = q.M[j], N = q.N[j], Kp_t = (M + 31) & ~31, Kp_n = (N + 31) & ~31;
    const uint16_t* __restrict__ src = q.src[j];
    // (weights of a layer: M * kp(N) < 2^31 — 32-bit index arithmetic, the 64-bit divisions were most of this kernel)
    if (blk < q.nb_t[j]) {
        // W^T: 32 x 32 tiles through LDS, reads along a row of W and writes along a row of W^T both contiguous (element by element the
        // writes had been one cache line per lane)
        __shared__ uint16_t tile[32][33];
        const int ntr = (N + 31) >> 5;                                  // tiles along r (rows of W^T)
        const int kt = blk / ntr, rt = blk - kt * ntr;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 32 * kt + ty + 8 * i, r = 32 * rt + tx;
            tile[ty + 8 * i][tx] = (k < M && r < N) ? src[static_cast<uint32_t>(k) * N + r] : static_cast<uint16_t>(0);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * rt + ty + 8 * i, k = 32 * kt + tx;
            if (r < N) q.dst_t[j][static_cast<uint32_t>(r) * Kp_t + k] = tile[tx][ty + 8 * i];
        }
    } else {
        const uint32_t idx = static_cast<uint32_t>(blk - q.nb_t[j]) * 256u + threadIdx.x;
        if (idx >= static_cast<uint32_t>(M) * Kp_n) return;
        const uint32_t r = idx / static_cast<uint32_t>(Kp_n), k = idx - r * Kp_n;
        q.dst_n[j][r * Kp_n + k] = static_cast<int>(k) < N ? src[r * N + k] : 0;
    }
}

// The same second pass for up to 2 x kKmJobs jobs in one launch (the weight and bias gradients of every layer of a stack): blocks
// [first[j], first[j + 1]) belong to job j; per element the same grouping and order as k_b16_reduce.
struct B16ReduceMulti { B16ReduceJob job[2 * kKmJobs]; int32_t first[2 * kKmJobs + 1]; int32_t count; };
// V = elements per thread (4: 16-byte loads of the partials, one 8-byte store of four bf16 — every job's N, M N and row stride are multiples
// of 4 and its buffers aligned, checked by the host; 1: any shape)
template <int V>
__global__ void __launch_bounds__(1024) k_b16_reduce_multi(const B16ReduceMulti q) {
    // 64 V consecutive elements x the sixteen split groups per block (a wave reads 256 V contiguous bytes of a partial; k_b16_reduce's
    // 16-element blocks read 64): per element the same grouping and order, so the bits are those of k_b16_reduce
    typedef float fv __attribute__((ext_vector_type(V)));
    __shared__ fv red[16][65];
    int ji = 0;
    while (ji + 1 < q.count && static_cast<int>(blockIdx.x) >= q.first[ji + 1]) ++ji;
    const B16ReduceJob& j = q.job[ji];
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t idx = (static_cast<int64_t>(blockIdx.x - q.first[ji]) * 64 + e) * V, MN = static_cast<int64_t>(j.M) * j.N;
    const int per = (j.splits + 15) / 16;
    const int z0 = grp * per, z1 = min(j.splits, (grp + 1) * per);
    fv s = 0.f;
    if (idx < MN) {
        fv a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.f;
        int z = z0;
        // long chains (the bias jobs: 64 partials per group) keep NB x 8 loads in flight and add them in the order of the 8-wide loop
        constexpr int NB = V == 4 ? 2 : 4;
        for (; z + 8 * NB <= z1; z += 8 * NB) {
            fv x[8 * NB];
#pragma unroll
            for (int u = 0; u < 8 * NB; ++u) x[u] = *reinterpret_cast<const fv*>(j.partial + (z + u) * MN + idx);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += x[8 * b + u];
        }
        for (; z + 8 <= z1; z += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += *reinterpret_cast<const fv*>(j.partial + (z + u) * MN + idx);
        }
        for (; z < z1; ++z) a[0] += *reinterpret_cast<const fv*>(j.partial + z * MN + idx);
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[grp][e] = s;
    __syncthreads();
    if (grp == 0 && idx < MN) {
        fv t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][e];
        uint16_t* o = j.out + (idx / j.N) * j.ldo + idx % j.N;
        if constexpr (V == 4) *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]));
        else o[0] = f2bf(t[0]);
    }
}

}  // namespace

int32_t b16_kp(int32_t K) { return (K + BK - 1) / BK * BK; }

int b16_pad_planes(const void* src, int64_t ld, bool transposed, int32_t rows, int32_t K, void* dst, hipStream_t st) {
    if (rows <= 0 || K <= 0) return RECON_OK;
    if (!src || !dst) return RECON_ERR_INVALID;
    const int32_t Kp = b16_kp(K);
    const dim3 grid(static_cast<unsigned>(ceil_div64(static_cast<int64_t>(rows) * Kp, 256)));
    if (transposed) hipLaunchKernelGGL((k_b16_pad_planes<true>), grid, dim3(256), 0, st, static_cast<const uint16_t*>(src), ld, rows, K, Kp, static_cast<uint16_t*>(dst));
    else hipLaunchKernelGGL((k_b16_pad_planes<false>), grid, dim3(256), 0, st, static_cast<const uint16_t*>(src), ld, rows, K, Kp, static_cast<uint16_t*>(dst));
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// both plane sets of a [M][N] bf16 weight (row stride ld): dst_t = W^T [N][kp(M)], dst_n = W [M][kp(N)], zero padded along k
int b16_pad_planes_both(const void* src, int64_t ld, int32_t M, int32_t N, void* dst_t, void* dst_n, hipStream_t st) {
    if (M <= 0 || N <= 0) return RECON_OK;
    if (!src || !dst_t || !dst_n) return RECON_ERR_INVALID;
    const int32_t Kp_t = b16_kp(M), Kp_n = b16_kp(N);
    const int nb_t = static_cast<int>(ceil_div64(static_cast<int64_t>(N) * Kp_t, 256)), nb_n = static_cast<int>(ceil_div64(static_cast<int64_t>(M) * Kp_n, 256));
    hipLaunchKernelGGL(k_b16_pad_both, dim3(static_cast<unsigned>(nb_t + nb_n)), dim3(256), 0, st, static_cast<const uint16_t*>(src), ld, M, N, Kp_t, Kp_n,
                       static_cast<uint16_t*>(dst_t), static_cast<uint16_t*>(dst_n), nb_t);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// W_j^T [N_j][kp(M_j)] and W_j [M_j][kp(N_j)] of `count` <= 8 contiguous weights [M_j][N_j] in one launch
int b16_pad_planes_both_multi(int32_t count, const void* const* src, const int32_t* M, const int32_t* N, void* const* dst_t, void* const* dst_n, hipStream_t st) {
    if (count < 1 || count > kKmJobs) return RECON_ERR_INVALID;
    B16PadMulti q{};
    int nb = 0;
    for (int j = 0; j < count; ++j) {
        if (M[j] <= 0 || N[j] <= 0 || !src[j] || !dst_t[j] || !dst_n[j]) return RECON_ERR_INVALID;
        if (static_cast<int64_t>(M[j]) * b16_kp(N[j]) >= (1LL << 31) || static_cast<int64_t>(N[j]) * b16_kp(M[j]) >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;
        q.src[j] = static_cast<const uint16_t*>(src[j]); q.dst_t[j] = static_cast<uint16_t*>(dst_t[j]); q.dst_n[j] = static_cast<uint16_t*>(dst_n[j]);
        q.M[j] = M[j]; q.N[j] = N[j];
        q.nb_t[j] = (b16_kp(M[j]) / 32) * static_cast<int>(ceil_div64(N[j], 32));          // 32 x 32 tiles of W^T [N][kp(M)]
        q.first[j] = nb;
        nb += q.nb_t[j] + static_cast<int>(ceil_div64(static_cast<int64_t>(M[j]) * b16_kp(N[j]), 256));
    }
    q.first[count] = nb; q.count = count;
    hipLaunchKernelGGL(k_b16_pad_both_multi, dim3(static_cast<unsigned>(nb)), dim3(256), 0, st, q);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// C[M,N] = A[M,K] . Bp[N,kp(K)]^T; A bf16 with lda % 8 == 0 and a 16-byte aligned base; C bf16 (out_bf16) or fp32
int gemm_b16(int32_t M, int32_t N, int32_t K, const void* A, int64_t lda, const void* Bp, void* C, int64_t ldc, bool out_bf16, hipStream_t st) {
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    if (!A || !Bp || !C) return RECON_ERR_INVALID;
    if (K <= 0 || (lda & 7) || lda < ((K + 7) & ~7) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bp)) & 15)) return RECON_ERR_UNSUPPORTED;
    B16Args a;
    a.A = static_cast<const uint16_t*>(A); a.Bp = static_cast<const uint16_t*>(Bp); a.lda = lda; a.ldb = b16_kp(K);
    if (static_cast<int64_t>(N) * a.ldb >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;
    a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    const dim3 grid(static_cast<unsigned>(ceil_div64(N, BN)), static_cast<unsigned>(ceil_div64(M, BM)), 1);
    if (out_bf16) hipLaunchKernelGGL((k_gemm_b16<true>), grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_b16<false>), grid, dim3(NT), 0, st, a);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

int b16_kmajor_splits(int32_t M, int32_t N, int32_t K) { return bx3_kmajor_splits(K, bx3_kmajor_split_k(M, N, K, 1)); }
// splits of a launch of `count` products that share K (M, N: the largest)
int b16_kmajor_splits_multi(int32_t M, int32_t N, int32_t K, int32_t count) { return bx3_kmajor_splits(K, bx3_kmajor_split_k(M, N, K, count)); }

// out_j (bf16 [M_j][N_j], row stride ldo_j) = A_j[K,M_j]^T . B_j[K,N_j] for `count` <= 8 products over the same K rows in ONE launch + one
// second pass (which also runs the `extra` jobs: the bias gradients); lda, ldb % 8 == 0, every row holds (M resp. N rounded up to 8)
// readable columns; partial_j = `splits` * M_j * N_j floats (splits = b16_kmajor_splits for one product, _multi for several)
int gemm_b16_kmajor_multi(int32_t count, const B16KmProduct* pr, int32_t K, const void* zeros, hipStream_t st, const B16ReduceJob* extra, int32_t n_extra) {
    if (count < 1 || count > kKmJobs || !pr || K < 0 || n_extra < 0 || n_extra > kKmJobs || !zeros || (reinterpret_cast<uintptr_t>(zeros) & 15)) return RECON_ERR_INVALID;
    B16KmArgs a{};
    int32_t Mx = 0, Nx = 0;
    for (int j = 0; j < count; ++j) {
        const B16KmProduct& q = pr[j];
        if (q.M <= 0 || q.N <= 0 || !q.A || !q.B || !q.out || !q.partial) return RECON_ERR_INVALID;
        const int32_t m_ld = (q.M + 7) / 8 * 8, n_ld = (q.N + 7) / 8 * 8;
        if ((q.lda & 7) || (q.ldb & 7) || m_ld > q.lda || n_ld > q.ldb || ((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B)) & 15))
            return RECON_ERR_UNSUPPORTED;
        a.A[j] = static_cast<const uint16_t*>(q.A); a.B[j] = static_cast<const uint16_t*>(q.B); a.lda[j] = q.lda; a.ldb[j] = q.ldb;
        a.partial[j] = q.partial; a.M[j] = q.M; a.N[j] = q.N; a.m_ld[j] = m_ld; a.n_ld[j] = n_ld;
        Mx = q.M > Mx ? q.M : Mx; Nx = q.N > Nx ? q.N : Nx;
    }
    a.zeros = static_cast<const uint16_t*>(zeros); a.K = K;
    const int sk = count == 1 ? b16_kmajor_splits(Mx, Nx, K) : b16_kmajor_splits_multi(Mx, Nx, K, count);
    int64_t kps = ceil_div64(K > 0 ? K : 1, sk);
    kps = ceil_div64(kps, BK) * BK;
    a.k_per_split = static_cast<int32_t>(kps);
    a.nsplit = static_cast<int32_t>(ceil_div64(K > 0 ? K : 1, kps));
    if (a.nsplit != sk || static_cast<int64_t>(sk) * count > 65535) return RECON_ERR_INVALID;
    const dim3 grid(static_cast<unsigned>(ceil_div64(Nx, BN)), static_cast<unsigned>(ceil_div64(Mx, BM)), static_cast<unsigned>(sk * count));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_b16_kmajor), hipFuncAttributeMaxDynamicSharedMemorySize, KM_STAGES * KM_STAGE_BYTES);
    hipLaunchKernelGGL(k_gemm_b16_kmajor, grid, dim3(NT), KM_STAGES * KM_STAGE_BYTES, st, a);
    if (count == 1 && n_extra <= 1) {
        const B16ReduceJob j0{pr[0].partial, static_cast<uint16_t*>(pr[0].out), pr[0].ldo, sk, pr[0].M, pr[0].N};
        const B16ReduceJob j1 = n_extra ? extra[0] : B16ReduceJob{nullptr, nullptr, 0, 0, 0, 0};
        const int nblk0 = static_cast<int>(ceil_div64(static_cast<int64_t>(j0.M) * j0.N, 16));
        const int nblk1 = n_extra ? static_cast<int>(ceil_div64(static_cast<int64_t>(j1.M) * j1.N, 16)) : 0;
        hipLaunchKernelGGL(k_b16_reduce, dim3(static_cast<unsigned>(nblk0 + nblk1)), dim3(256), 0, st, j0, j1, nblk0);
    } else {
        B16ReduceMulti q{};
        // the extra jobs first: few blocks with long chains (a bias gradient sums 1 024 per-graph rows) — started last they were the launch's tail
        for (int j = 0; j < n_extra; ++j) q.job[q.count++] = extra[j];
        for (int j = 0; j < count; ++j) q.job[q.count++] = B16ReduceJob{pr[j].partial, static_cast<uint16_t*>(pr[j].out), pr[j].ldo, sk, pr[j].M, pr[j].N};
        bool vec = true;                                                // four elements per thread where every job allows it
        for (int j = 0; j < q.count; ++j) {
            const B16ReduceJob& b = q.job[j];
            vec = vec && (b.N & 3) == 0 && (b.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(b.partial) & 15) == 0 && (reinterpret_cast<uintptr_t>(b.out) & 7) == 0;
        }
        int nb = 0;
        for (int j = 0; j < q.count; ++j) {
            q.first[j] = nb;
            nb += static_cast<int>(ceil_div64(static_cast<int64_t>(q.job[j].M) * q.job[j].N, vec ? 256 : 64));
        }
        q.first[q.count] = nb;
        if (vec) hipLaunchKernelGGL(k_b16_reduce_multi<4>, dim3(static_cast<unsigned>(nb)), dim3(1024), 0, st, q);
        else hipLaunchKernelGGL(k_b16_reduce_multi<1>, dim3(static_cast<unsigned>(nb)), dim3(1024), 0, st, q);
    }
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

int gemm_b16_kmajor(int32_t M, int32_t N, int32_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* out, int64_t ldo, float* partial,
                    const void* zeros, hipStream_t st, const B16ReduceJob* extra) {
    if (M < 0 || N < 0 || K < 0) return RECON_ERR_INVALID;
    if (M == 0 || N == 0) return RECON_OK;
    const B16KmProduct pr{A, lda, B, ldb, out, ldo, partial, M, N};
    return gemm_b16_kmajor_multi(1, &pr, K, zeros, st, extra, extra ? 1 : 0);
}

}  // namespace recon
