// K5 / K6 — GP-GNN gated propagation (block adjacency, L-hop propagation with fused head*tail
// gather, start-entity embeddings) and batched GraphConvolution.
//
// Replaces models/models.py:240-274 (and its copies :450-485, :660-694, :898-932),
// utils/context_utils.py:387-426 and models/layers.py:57-63.
//
// Propagation maths per graph b and channel c (a channel = one ordered entity pair):
//     h^l[c] = act(A_l h^l-1[c])         A_l [S,S], h [S]
// Channels never mix, so a workgroup owns (graph, chunk of <= 80 channels), keeps the channel
// states H^T [channels][S] resident in LDS for all L hops and streams each A_l exactly once from
// HBM straight into MFMA B-operand registers (v_mfma_f32_16x16x4_f32, exact fp32):
//     out^T[c][s] = sum_t H^T[c][t] * A[s][t]      M = channels, N = S, K = S
// Both operands are contiguous along the contraction index, so every lane fetches float4s and the
// K index is permuted consistently (k = 4*(lane>>4) + step) between the A- and B-fragments.
#include <math.h>
#include <stdlib.h>
#include "prop_common.h"

namespace recon {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// logical block index such that XCD x (blocks b with b % 8 == x) owns a contiguous chunk of the logical range
__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ------------------------------------------------------------------------------- P1
__global__ void k_block_adj_fwd(const float* __restrict__ T, const float* __restrict__ I, int32_t B, int32_t n, int32_t dd,
                                float* __restrict__ A) {
    const int64_t S = 1LL * n * dd;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= B * S * S) return;
    const int col = static_cast<int>(idx % S), row = static_cast<int>((idx / S) % S);
    const int64_t b = idx / (S * S);
    const int i = row / dd, r = row % dd, j = col / dd, c = col % dd;
    float v;
    if (i == j) v = I[r * dd + c];
    else {
        const int e = i * (n - 1) + (j < i ? j : j - 1);
        v = T[(b * n * (n - 1) + e) * dd * dd + r * dd + c];
    }
    A[idx] = v;
}
__global__ void k_block_adj_bwd_T(const float* __restrict__ gA, int32_t B, int32_t n, int32_t dd, float* __restrict__ gT) {
    const int64_t S = 1LL * n * dd, Cn = 1LL * n * (n - 1), d2 = 1LL * dd * dd;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= B * Cn * d2) return;
    const int c = static_cast<int>(idx % dd), r = static_cast<int>((idx / dd) % dd);
    const int e = static_cast<int>((idx / d2) % Cn);
    const int64_t b = idx / (d2 * Cn);
    const int i = e / (n - 1);
    int j = e % (n - 1);
    if (j >= i) ++j;
    gT[idx] = gA[(b * S + i * dd + r) * S + j * dd + c];
}

// float4 variants for dd % 4 == 0 (every 4-float chunk lies inside one dd x dd block): grid = (chunks of one graph, B)
__global__ void __launch_bounds__(256) k_block_adj_fwd4(const float* __restrict__ T, const float* __restrict__ I, int32_t n, int32_t dd,
                                                        float* __restrict__ A) {
    const int S = n * dd, S4 = S >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= S * S4) return;
    const int b = blockIdx.y;
    const int row = t / S4, col = (t - row * S4) << 2;
    const int i = row / dd, r = row - i * dd, j = col / dd, c = col - j * dd;
    float4 v;
    if (i == j) v = *reinterpret_cast<const float4*>(I + r * dd + c);
    else {
        const int e = i * (n - 1) + (j < i ? j : j - 1);
        v = *reinterpret_cast<const float4*>(T + (static_cast<int64_t>(b) * n * (n - 1) + e) * dd * dd + r * dd + c);
    }
    *reinterpret_cast<float4*>(A + static_cast<int64_t>(b) * S * S + static_cast<int64_t>(row) * S + col) = v;
}
__global__ void __launch_bounds__(256) k_block_adj_bwd_T4(const float* __restrict__ gA, int32_t n, int32_t dd, float* __restrict__ gT) {
    const int S = n * dd, d2 = dd * dd, Cn = n * (n - 1);
    const int t = blockIdx.x * 256 + threadIdx.x;               // float4 index inside one graph's gT [Cn][dd][dd]
    if (t >= Cn * (d2 >> 2)) return;
    const int b = blockIdx.y;
    const int e = t / (d2 >> 2), rc = (t - e * (d2 >> 2)) << 2;
    const int r = rc / dd, c = rc - r * dd;
    const int i = e / (n - 1);
    int j = e - i * (n - 1);
    if (j >= i) ++j;
    const float4 v = *reinterpret_cast<const float4*>(gA + static_cast<int64_t>(b) * S * S + static_cast<int64_t>(i * dd + r) * S + j * dd + c);
    *reinterpret_cast<float4*>(gT + (static_cast<int64_t>(b) * Cn + e) * d2 + rc) = v;
}
// g_identity[r,c] = sum_{b,i} gA[b, i*dd+r, i*dd+c]; one block per (r,c), fixed-order tree
// d loss / d identity = sum over graphs b and diagonal positions i of the (i, i) block of gA.  Slice `blk` of the (b, i)
// pairs: thread e = (r, c) walks the slice's blocks (a wave reads 64-byte row pieces) with four independent chains and
// writes partial[blk][e]; k_sum_rows adds the slices in a fixed order.  (One workgroup per identity element,
// striding over all 9 216 blocks with 4-byte loads, fetched 300 MB for 9 MB of data.)
constexpr int kIdentSlices = 256;
__global__ void __launch_bounds__(256) k_block_adj_bwd_I(const float* __restrict__ gA, int32_t B, int32_t n, int32_t dd,
                                                         float* __restrict__ partial) {
    const int64_t S = 1LL * n * dd, pairs = 1LL * B * n;
    const int64_t per = (pairs + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = blockIdx.x * per, t1 = min(pairs, t0 + per);
    for (int e = threadIdx.x; e < dd * dd; e += 256) {
        const int r = e / dd, c = e % dd;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        auto at = [&](int64_t t) { const int64_t b = t / n; const int i = static_cast<int>(t % n); return gA[(b * S + i * dd + r) * S + i * dd + c]; };
        int64_t t = t0;
        for (; t + 4 <= t1; t += 4) { s0 += at(t); s1 += at(t + 1); s2 += at(t + 2); s3 += at(t + 3); }
        for (; t < t1; ++t) s0 += at(t);
        partial[static_cast<int64_t>(blockIdx.x) * dd * dd + e] = (s0 + s1) + (s2 + s3);
    }
}
// out[o] = sum_r partial[r][o]: 16 columns x 64 row groups per block, fixed-order LDS combine
__global__ void __launch_bounds__(1024) k_sum_rows(const float* __restrict__ partial, int32_t nrows, int32_t O, float* __restrict__ out) {
    __shared__ float red[64][17];
    const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + c;
    const int per = (nrows + 63) / 64;
    const int r0 = grp * per, r1 = min(nrows, (grp + 1) * per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (o < O) {
        int r = r0;
        for (; r + 4 <= r1; r += 4) {
            s0 += partial[static_cast<int64_t>(r) * O + o];
            s1 += partial[static_cast<int64_t>(r + 1) * O + o];
            s2 += partial[static_cast<int64_t>(r + 2) * O + o];
            s3 += partial[static_cast<int64_t>(r + 3) * O + o];
        }
        for (; r < r1; ++r) s0 += partial[static_cast<int64_t>(r) * O + o];
    }
    red[grp][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && o < O) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 64; ++g) t += red[g][c];
        out[o] = t;
    }
}

// ------------------------------------------------------------------------------- P4
__global__ void k_start_entity(const float* __restrict__ ent, const int64_t* __restrict__ pos, const float* __restrict__ templ,
                               int32_t B, int32_t n, int32_t d, float* __restrict__ out) {
    const int64_t S = 2LL * d * n, C = 1LL * n * (n - 1);
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= B * C * S) return;
    const int s = static_cast<int>(idx % S), c = static_cast<int>((idx / S) % C);
    const int64_t b = idx / (S * C);
    const int i = c / (n - 1);
    int j = c % (n - 1);
    if (j >= i) ++j;
    float v = 0.f;
    if (s >= 2 * d * i && s < 2 * d * i + d) v = ent[pos[(b * C + c) * 2 + 0] * d + (s - 2 * d * i)];
    else if (s >= 2 * d * j + d && s < 2 * d * (j + 1)) v = ent[pos[(b * C + c) * 2 + 1] * d + (s - 2 * d * j - d)];
    out[idx] = v * templ[c * S + s];
}

// rows [2][B C][d]: slot 0 = node i's first half-slot of (g * templ), slot 1 = node j's second half-slot; key [2][2 B C] = pos[..., slot]
__global__ void k_start_entity_bwd(const float* __restrict__ g, const int64_t* __restrict__ pos, const float* __restrict__ templ, int32_t B, int32_t n,
                                   int32_t d, float* __restrict__ rows, int64_t* __restrict__ key) {
    const int64_t S = 2LL * d * n, C = 1LL * n * (n - 1), BC = B * C;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= 2 * BC * d) return;
    const int k = static_cast<int>(idx % d);
    const int64_t r = idx / d;                                     // slot * BC + bc
    const int slot = r >= BC ? 1 : 0;
    const int64_t bc = r - slot * BC;
    const int c = static_cast<int>(bc % C);
    const int i = c / (n - 1);
    int j = c % (n - 1);
    if (j >= i) ++j;
    const int off = slot ? 2 * d * j + d : 2 * d * i;
    rows[idx] = g[bc * S + off + k] * templ[c * S + off + k];
    if (key && k == 0) { const int64_t e = pos[bc * 2 + slot]; key[r] = e; key[2 * BC + r] = e; }
}

// ------------------------------------------------------------------------------- P2 forward
template <bool VEC4>
__device__ __forceinline__ void load_a_row4(float (&v)[4], const float* A, int S, int row, int col, bool row_ok) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (!row_ok) return;
    const float* p = A + static_cast<int64_t>(row) * S + col;
    if constexpr (VEC4) {
        if (col < S) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (col + j < S) v[j] = p[j];
    }
}

// One wave per 16-wide column tile of H^T (up to 16 waves per block), ONE LDS buffer: every wave
// keeps its output tile in MFMA accumulators across the barrier that ends the hop's reads, then
// overwrites its own columns.  A rows are prefetched three 16-float slabs ahead, across hop boundaries.
template <int MT, bool VEC4>
__global__ void __launch_bounds__(1024) k_propagate_fwd(const PropK p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = blockDim.x >> 6;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int c0 = chunk * p.CC;
    const int pitch = p.pitch, S = p.S, CC = p.CC;
    const int NT = p.Sp >> 4;
    const bool dbl = NT > nwaves;                                  // S > 256: several passes, two buffers
    float* H = lds;                                                // state read by this hop
    float* Hn = dbl ? lds + static_cast<int64_t>(CC) * pitch : lds; // state written by this hop
    for (int idx = tid; idx < CC * pitch; idx += nthreads) {       // h^0 chunk, zero padded
        const int cl = idx / pitch, s = idx % pitch;
        const int c = c0 + cl;
        H[idx] = (c < p.C && s < S) ? p.h0[b * p.h0_bs + static_cast<int64_t>(c) * S + s] : 0.f;
    }
    __syncthreads();
    const int li = lane & 15, lq = lane >> 4;
    constexpr int G = 3;                                            // slabs per prefetch group
    const int ngroups = (NT + G - 1) / G;
    const int npass = (NT + nwaves - 1) / nwaves;
    // A rows are fetched one group of G slabs ahead of the MFMAs, ACROSS pass and hop boundaries (the rows of
    // the next hop's adjacency do not depend on the state), so HBM latency is never exposed at a barrier.
    auto fetch = [&](float (&dst)[G][4], int l, int pass, int grp) {
        const int nt = pass * nwaves + wave;
        const int row = 16 * nt + li;
        const bool ok = l < p.L && nt < NT && row < S;
        const float* A = p.adj[l < p.L ? l : 0] + static_cast<int64_t>(b) * S * S;
#pragma unroll
        for (int g = 0; g < G; ++g) load_a_row4<VEC4>(dst[g], A, S, row, 16 * (grp * G + g) + 4 * lq, ok && (grp * G + g) < NT);
    };
    float bq[G][4], bn[G][4];
    fetch(bq, 0, 0, 0);
    for (int l = 0; l < p.L; ++l) {
        for (int pass = 0; pass < npass; ++pass) {
            const int nt = pass * nwaves + wave;
            const bool tile_ok = nt < NT;                           // wave-uniform
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int grp = 0; grp < ngroups; ++grp) {
                int nl = l, np = pass, ng = grp + 1;                // the step after this one
                if (ng == ngroups) { ng = 0; if (++np == npass) { np = 0; ++nl; } }
                fetch(bn, nl, np, ng);
                const int s0 = grp * G;
                if (tile_ok) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (s0 + g < NT) {
                            const int kc = 16 * (s0 + g) + 4 * lq;
                            float4 aq[MT];
#pragma unroll
                            for (int m = 0; m < MT; ++m) aq[m] = *reinterpret_cast<const float4*>(H + (16 * m + li) * pitch + kc);
                            // consecutive MFMAs hit DIFFERENT accumulators (dependent-accumulator latency is 40 cycles, issue 32)
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].x, bq[g][0], acc[m], 0, 0, 0);
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].y, bq[g][1], acc[m], 0, 0, 0);
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].z, bq[g][2], acc[m], 0, 0, 0);
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].w, bq[g][3], acc[m], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bq[g][j] = bn[g][j];
            }
            if (!dbl) __syncthreads();                              // single buffer: all reads of H^{l-1} done
            // C layout: col (s) = lane&15, row (channel) = (lane>>4)*4 + r
            if (tile_ok) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Hn[(16 * m + 4 * lq + r) * pitch + 16 * nt + li] = act_fwd(acc[m][r], p.act);
            }
        }
        __syncthreads();
        if (dbl) { float* t = H; H = Hn; Hn = t; }                  // H now holds h^l
        // relation_l = gather(h, heads) * gather(h, tails)   (models/models.py:270-273)
        for (int idx = tid; idx < CC * p.dd; idx += nthreads) {
            const int cl = idx / p.dd, x = idx % p.dd;
            const int c = c0 + cl;
            if (c < p.C) {
                const int64_t io = b * p.idx_bs + static_cast<int64_t>(c) * p.dd + x;
                const int hi = static_cast<int>(p.head[io]), ti = static_cast<int>(p.tail[io]);
                p.out[(static_cast<int64_t>(b) * p.C + c) * (p.L * p.dd) + l * p.dd + x] = H[cl * pitch + hi] * H[cl * pitch + ti];
            }
        }
        if (p.hsave) {
            float* hs = p.hsave + ((static_cast<int64_t>(l) * p.B + b) * p.C) * S;
            for (int idx = tid; idx < CC * S; idx += nthreads) {
                const int cl = idx / S, s = idx % S;
                if (c0 + cl < p.C) hs[static_cast<int64_t>(c0 + cl) * S + s] = H[cl * pitch + s];
            }
        }
        // the next hop's post-compute barrier orders these reads before its writes
    }
}


// ------------------------------------------------------------------------------- P2 forward on the bf16 matrix cores
// k_propagate_fwd_x: the same L-hop propagation with every fp32 operand split into three bfloat16 terms ON THE FLY and six term
// products accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (the scheme of gemm_bx3.hip: fp32-class accuracy, no scaling — bfloat16
// keeps fp32's exponent range — at 2.67x the fp32-MFMA ceiling).  One workgroup = one graph; wave w owns the 16 state rows
// s = 16 w .. 16 w + 15 of every hop:  Hnew^T [S x C] = A_l [S x S] . H^T [S x C],  M = s, N = channel, K = t.
//   * A_l is read exactly ONCE from HBM: each lane fetches its own MFMA A-fragment (row s = lane & 15, 8 consecutive t) as two
//     float4 and splits it in registers; a hop's rows are requested while the previous hop computes;
//   * the state lives in LDS as three bf16 planes [3][channel][t] (k-contiguous: B fragments are ds_read_b128), split once per
//     hop when it is written, reconstructed exactly (8 + 8 + 8 mantissa bits) for the head (.) tail gather;
//   * two barriers per hop (all reads of H^l-1 done -> write H^l -> visible).
// Opt-in (RECON_PROP_FWD=x), parity-tested like the other forms.  Measured at cfg 3b: 207 us against 167 us for the fp32-MFMA wave
// form below, although its matrix-pipe time is 31 us against 58: one graph = one workgroup of 9 waves (3/2/2/2 over the SIMDs)
// with 77 KB of LDS image and 168 registers, so a CU holds a single workgroup, and the waves spend 62 % of their life parked at
// the two barriers per hop and behind the A loads (PMC: SQ_WAIT_ANY 165 M of 267 M wave cycles, MFMA busy 31 us of 207).  Forcing
// two workgroups per CU (96 registers) spills 420 bytes per lane: 341 us.
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
using u32x4_t = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ void px_split8(const float (&v)[8], bf16x8_t (&out)[3]) {
    float r[8];
    u32x4_t w[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = q == 0 ? v[2 * j] : r[2 * j], x1 = q == 0 ? v[2 * j + 1] : r[2 * j + 1];
            const uint32_t b0 = __builtin_bit_cast(uint16_t, static_cast<__bf16>(x0)), b1 = __builtin_bit_cast(uint16_t, static_cast<__bf16>(x1));
            w[q][j] = b0 | (b1 << 16);
            if (q < 2) { r[2 * j] = x0 - __builtin_bit_cast(float, b0 << 16); r[2 * j + 1] = x1 - __builtin_bit_cast(float, b1 << 16); }
        }
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = __builtin_bit_cast(bf16x8_t, w[q]);
}

// NTC = channel tiles of 16 (C <= 16 NTC), KS = K steps of 32 (S <= 32 KS); blockDim.x = 64 * ceil(S / 16).
// LDS image of the state: [plane 3][K step KS][channel 16 NTC][64 bytes = 32 t], the 16-byte slot of t group kq rotated by
// 2 (channel >> 3) — the B image of gemm_bx3.hip, conflict free for the ds_read_b128 fragment reads.
template <int NTC, int KS>
__global__ void __launch_bounds__(128 * KS, 2) k_propagate_fwd_x(const PropK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xl[];
    constexpr int STEP = NTC * 16 * 64;                               // bytes of one K step of one plane
    constexpr int PLANE = KS * STEP;
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, S = p.S, C = p.C;
    const int li = lane & 15, lq = lane >> 4;
    auto state_off = [](int c, int t) { return (t >> 5) * STEP + c * 64 + ((((t >> 3) + 2 * (c >> 3)) & 3) << 4) + 2 * (t & 7); };
    auto store_state = [&](int c, int t0, const float (&v)[4]) {      // 4 consecutive t (t0 % 4 == 0) of channel c -> the three planes
        float r[4] = {v[0], v[1], v[2], v[3]};
        const int off = state_off(c, t0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            uint32_t w[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t b0 = __builtin_bit_cast(uint16_t, static_cast<__bf16>(r[2 * h])), b1 = __builtin_bit_cast(uint16_t, static_cast<__bf16>(r[2 * h + 1]));
                w[h] = b0 | (b1 << 16);
                r[2 * h] -= __builtin_bit_cast(float, b0 << 16); r[2 * h + 1] -= __builtin_bit_cast(float, b1 << 16);
            }
            *reinterpret_cast<uint2*>(xl + q * PLANE + off) = make_uint2(w[0], w[1]);
        }
    };
    auto state_at = [&](int c, int t) {                               // exact fp32 value of H[c][t]: 8 + 8 + 8 mantissa bits
        const int off = state_off(c, t);
        float v = 0.f;
#pragma unroll
        for (int q = 2; q >= 0; --q) v += __builtin_bit_cast(float, static_cast<uint32_t>(*reinterpret_cast<const uint16_t*>(xl + q * PLANE + off)) << 16);
        return v;
    };
    // ---- h^0 -> planes (zero padded to 16 NTC channels x 32 KS columns)
    for (int idx = tid; idx < NTC * 16 * KS * 8; idx += nthreads) {
        const int c = idx / (KS * 8), t0 = 4 * (idx % (KS * 8));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (c < C && t0 + e < S) ? p.h0[b * p.h0_bs + static_cast<int64_t>(c) * S + t0 + e] : 0.f;
        store_state(c, t0, v);
    }
    // ---- this wave's rows of A_l: fragment (row 16 w + li, columns 32 ks + 8 lq .. + 7) as two float4, two K steps ahead of the MFMAs
    const int row = 16 * wave + li;
    const bool row_ok = row < S;
    const bool vec = (S & 3) == 0;
    const int64_t arow = (static_cast<int64_t>(b) * S + (row_ok ? row : 0)) * S;
    auto load_a = [&](float (&dst)[8], int l, int ks) {
        const float* A = p.adj[l] + arow;
        const int t0 = 32 * ks + 8 * lq;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + 4 * h;
            if (vec) {                                                // S % 4 == 0: a quad is inside the row or outside it
                const float4 q4 = *reinterpret_cast<const float4*>(A + (t < S ? t : 0));
                dst[4 * h] = q4.x; dst[4 * h + 1] = q4.y; dst[4 * h + 2] = q4.z; dst[4 * h + 3] = q4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[4 * h + e] = A[t + e < S ? t + e : 0];
            }
        }
    };
    // gather indices of this thread's first GI output items: the same in every hop, so their (dependent, int64) loads leave the loop
    constexpr int GI = 3;
    int g_hi[GI], g_ti[GI];
#pragma unroll
    for (int i = 0; i < GI; ++i) {
        const int idx = min(tid + i * nthreads, C * p.dd - 1);
        const int64_t io = b * p.idx_bs + idx;
        g_hi[i] = static_cast<int>(p.head[io]); g_ti[i] = static_cast<int>(p.tail[io]);
    }
    constexpr int PFD = KS >= 2 ? 2 : 1;                               // K steps in flight
    float araw[PFD][8];
#pragma unroll
    for (int i = 0; i < PFD; ++i) load_a(araw[i], 0, i);
    __syncthreads();
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // small terms first
    const int b_rd = li * 64 + (((lq + 2 * (li >> 3)) & 3) << 4);     // + 1024 j (the rotation depends on channel & 8 only) + STEP ks
    for (int l = 0; l < p.L; ++l) {
        f32x4 acc[NTC];
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = (row_ok && 32 * ks + 8 * lq + e < S) ? araw[ks % PFD][e] : 0.f;      // rows / columns past S: zero
            // refill the slot just consumed with the step PFD ahead (the next hop's first steps at the end of this one)
            if (ks + PFD < KS) load_a(araw[ks % PFD], l, ks + PFD);
            else if (l + 1 < p.L) load_a(araw[ks % PFD], l + 1, ks % PFD);        // step n of a hop always lives in slot n % PFD
            bf16x8_t a[3];
            px_split8(av, a);
            // channel tiles in pairs: two independent accumulator chains of six products each
#pragma unroll
            for (int j = 0; j < NTC; j += 2) {
                bf16x8_t bfr[2][3];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    if (j + jj < NTC)
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            bfr[jj][q] = *reinterpret_cast<const bf16x8_t*>(xl + q * PLANE + ks * STEP + 1024 * (j + jj) + b_rd);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        if (j + jj < NTC) acc[j + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[TA[t]], bfr[jj][TB[t]], acc[j + jj], 0, 0, 0);
            }
        }
        __syncthreads();                                              // every wave has read H^l-1
        // C layout: column (lane & 15) = channel 16 j + li, rows 4 lq + r = state index s = 16 w + 4 lq + r
        float* hs = p.hsave ? p.hsave + ((static_cast<int64_t>(l) * p.B + b) * C) * S : nullptr;
        const int s0 = 16 * wave + 4 * lq;
#pragma unroll
        for (int j = 0; j < NTC; ++j) {
            const int c = 16 * j + li;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (s0 + r < S) ? act_fwd(acc[j][r], p.act) : 0.f;
            store_state(c, s0, v);                                    // s0 < 16 ceil(S/16) <= 32 KS: inside the padded image
            if (hs && c < C) {
                if (vec && s0 + 3 < S) *reinterpret_cast<float4*>(hs + static_cast<int64_t>(c) * S + s0) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (s0 + r < S) hs[static_cast<int64_t>(c) * S + s0 + r] = v[r];
                }
            }
        }
        __syncthreads();                                              // H^l complete
        // relation_l = heads * tails   (models/models.py:270-273); the first GI items of a thread use the indices fetched before the hop loop
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < C * p.dd) {
                const int c = idx / p.dd, x = idx % p.dd;
                p.out[(static_cast<int64_t>(b) * C + c) * (p.L * p.dd) + l * p.dd + x] = state_at(c, g_hi[i]) * state_at(c, g_ti[i]);
            }
        }
        for (int idx = tid + GI * nthreads; idx < C * p.dd; idx += nthreads) {
            const int c = idx / p.dd, x = idx % p.dd;
            const int64_t io = b * p.idx_bs + static_cast<int64_t>(c) * p.dd + x;
            const int hi = static_cast<int>(p.head[io]), ti = static_cast<int>(p.tail[io]);
            p.out[(static_cast<int64_t>(b) * C + c) * (p.L * p.dd) + l * p.dd + x] = state_at(c, hi) * state_at(c, ti);
        }
        // the next hop's first barrier orders these reads before its writes
    }
}

// ------------------------------------------------------------------------------- P2 forward, wave-independent form
// Channels never mix, so ONE WAVE owns 16 channels of one graph for all L hops and needs no workgroup barrier:
// its state H^T [16][S] lives in REGISTERS as MFMA A-fragments (NT float4 per lane), every hop streams the whole
// A_l through B-fragment registers (the 4-5 waves of a graph re-read it from L2), and the hop's outputs go
// through a private LDS scratch [16][S+4] only to be re-laid out as next hop's A-fragments and for the
// head*tail gather.  Used when S <= 144 (NT <= 9 register fragments); larger S falls back to k_propagate_fwd.
template <int NT, bool VEC4>
__global__ void __launch_bounds__(256) k_propagate_fwd_w(const PropK p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NB = (NT >= 3) ? 3 : NT;                          // column tiles in flight (independent accumulators)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int MTn = (p.C + 15) >> 4;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); give every XCD a contiguous
    // chunk of units so that the waves of one graph, which stream the same A_l, hit the same L2 (PMC: 49 % L2 hits before)
    const int unit = xcd_block(blockIdx.x, gridDim.x) * 4 + wave;
    if (unit >= p.B * MTn) return;
    const int b = unit / MTn, m = unit % MTn;
    const int S = p.S, pitch = p.pitch;
    float* scr = lds + static_cast<int64_t>(wave) * 16 * pitch;
    const int li = lane & 15, lq = lane >> 4;
    const int cmine = 16 * m + li;                                  // channel whose row this lane holds as A-fragment
    float af[NT][4];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const int k = 16 * s + 4 * lq;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            af[s][t] = (cmine < p.C && k + t < S) ? p.h0[b * p.h0_bs + static_cast<int64_t>(cmine) * S + k + t] : 0.f;
    }
    for (int l = 0; l < p.L; ++l) {
        const float* A = p.adj[l] + static_cast<int64_t>(b) * S * S;
#pragma unroll 1
        for (int nt0 = 0; nt0 < NT; nt0 += NB) {
            f32x4 acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            float bq[NB][4], bn[NB][4];
#pragma unroll
            for (int j = 0; j < NB; ++j) load_a_row4<VEC4>(bq[j], A, S, 16 * (nt0 + j) + li, 4 * lq, nt0 + j < NT && 16 * (nt0 + j) + li < S);
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                if (s + 1 < NT) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        load_a_row4<VEC4>(bn[j], A, S, 16 * (nt0 + j) + li, 16 * (s + 1) + 4 * lq, nt0 + j < NT && 16 * (nt0 + j) + li < S);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][t], bq[j][t], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) bq[j][t] = bn[j][t];
            }
            // C layout: col (s) = lane&15, row (channel) = (lane>>4)*4 + r
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (nt0 + j < NT)
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[(4 * lq + r) * pitch + 16 * (nt0 + j) + li] = act_fwd(acc[j][r], p.act);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): this wave's scratch writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(scr + li * pitch + 16 * s + 4 * lq);
            af[s][0] = v.x; af[s][1] = v.y; af[s][2] = v.z; af[s][3] = v.w;
        }
        // relation_l = gather(h, heads) * gather(h, tails)   (models/models.py:270-273)
        for (int idx = lane; idx < 16 * p.dd; idx += 64) {
            const int cl = idx / p.dd, x = idx % p.dd;
            const int c = 16 * m + cl;
            if (c < p.C) {
                const int64_t io = b * p.idx_bs + static_cast<int64_t>(c) * p.dd + x;
                const int hi = static_cast<int>(p.head[io]), ti = static_cast<int>(p.tail[io]);
                p.out[(static_cast<int64_t>(b) * p.C + c) * (p.L * p.dd) + l * p.dd + x] = scr[cl * pitch + hi] * scr[cl * pitch + ti];
            }
        }
        if (p.hsave) {
            float* hs = p.hsave + ((static_cast<int64_t>(l) * p.B + b) * p.C) * S;
            for (int idx = lane; idx < 16 * S; idx += 64) {
                const int cl = idx / S, sidx = idx % S;
                if (16 * m + cl < p.C) hs[static_cast<int64_t>(16 * m + cl) * S + sidx] = scr[cl * pitch + sidx];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // scratch reads done before the next hop overwrites it
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------- P2 forward, staged form
// One workgroup = ONE graph, one wave per 16 channels (state in registers as MFMA A-fragments, as in the wave form), but
// A_l is no longer fetched fragment by fragment (16 rows x 64 bytes per instruction, once per wave: PMC showed the texture
// path 59 % busy and 49 % L2 hits).  It is streamed ONCE per graph through a ring of three LDS slabs of 16 rows (= one
// column tile of the output) by the LDS-DMA path in full 1 KiB pieces; all waves read their B fragments from the slab.
// Raw s_barrier + counted vmcnt keep two slabs in flight across each barrier.  Slab rows are 16 bytes longer than S
// floats, which puts the 16 rows of a ds_read_b128 on distinct bank groups and keeps the image lane-linear (37 slots of
// 16 bytes per row at S = 144; the extra slot is filled with a don't-care load).  Needs S % 16 == 0.
template <int NT>
__global__ void __launch_bounds__(1024) k_propagate_fwd_s(const PropK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = p.S, pitch = p.pitch;                              // pitch = S + 4 floats
    const int rowb = (S + 4) * 4, slots_per_row = (S + 4) / 4;       // slab row in bytes / in 16-byte slots
    const int slab_bytes = 16 * rowb, slab_slots = 16 * slots_per_row;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;                                  // = channel groups of the graph
    const int b = blockIdx.x, m = wave;
    unsigned char* ring = smem;                                      // [3][16][rowb]
    float* scr = reinterpret_cast<float*>(smem + 3 * slab_bytes) + static_cast<int64_t>(wave) * 16 * pitch;
    const int li = lane & 15, lq = lane >> 4;
    const int cmine = 16 * m + li;
    const int ndma = (slab_slots + 63) / 64;                         // DMA instructions per slab (10 at S = 144)
    const int my_dma = (ndma - wave + nw - 1) / nw;                  // this wave's share: instructions wave, wave + nw, ...
    const int steps = p.L * NT;

    float af[NT][4];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const int k = 16 * s + 4 * lq;
#pragma unroll
        for (int t = 0; t < 4; ++t) af[s][t] = (cmine < p.C) ? p.h0[b * p.h0_bs + static_cast<int64_t>(cmine) * S + k + t] : 0.f;
    }
    // slab g = rows 16 (g % NT) .. +15 of A_{g / NT}
    auto dma_slab = [&](int g) {
        const float* A = p.adj[g / NT] + static_cast<int64_t>(b) * S * S + static_cast<int64_t>(16 * (g % NT)) * S;
        unsigned char* dst = ring + (g % 3) * slab_bytes;
        for (int i = wave; i < ndma; i += nw) {
            const int slot = 64 * i + lane;
            if (slot < slab_slots) {
                const int row = slot / slots_per_row, sl = slot % slots_per_row;
                const float* src = A + static_cast<int64_t>(row) * S + 4 * (sl < S / 4 ? sl : 0);      // last slot of a row: padding
                // issued as inline asm: through the builtin the compiler knows an LDS-DMA is in flight and waits vmcnt(0) before
                // the next ds_read of the ring (it cannot tell the slots apart), which would serialise the whole pipeline.
                // Hidden from its counters the DMA only makes the compiler's own vmcnt waits more conservative (in-order
                // completion); the waits that matter for the ring are the explicit ones below.
                const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(
                    static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)(dst + 1024 * i))));
                // m0 is saved and restored inside the one asm block (listing it as a clobber is undefined behaviour for a reserved register:
                // the compiler keeps its own LDS base there for its own LDS-DMA / GWS uses)
                uint32_t m0_saved;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_saved) : "v"(src), "s"(lds_addr) : "memory");
            }
        }
    };
    // the h0 loads must be complete BEFORE the loop: otherwise their s_waitcnt vmcnt(0) sits at the first MFMA inside the loop
    // body, where it would drain the DMA pipeline in every iteration
    __builtin_amdgcn_s_waitcnt(0x0f70);
    dma_slab(0);
    if (steps > 1) dma_slab(1);
    f32x4 acc0, acc1;
    for (int g = 0; g < steps; ++g) {
        const int nt = g % NT, l = g / NT;
        // this wave's pieces of slab g have landed (the pieces of slab g+1 may still fly), then everybody's have
        if (g + 1 < steps) {
            if (my_dma >= 3) __builtin_amdgcn_s_waitcnt(0x0f70 | 0);            // conservative for unusual shapes: vmcnt(0)
            else if (my_dma == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | 2);       // vmcnt(2)
            else if (my_dma == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | 1);
            else __builtin_amdgcn_s_waitcnt(0x0f70 | 0);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0f70 | 0);
        }
        __builtin_amdgcn_s_barrier();
        if (g + 2 < steps) dma_slab(g + 2);                          // its ring slot was last read in step g-1: everyone is past that
        const unsigned char* slab = ring + (g % 3) * slab_bytes + li * rowb + lq * 16;
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0;
        float4 bq[NT];
#pragma unroll
        for (int s = 0; s < NT; ++s) bq[s] = *reinterpret_cast<const float4*>(slab + 64 * s);
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            f32x4& a = (s & 1) ? acc1 : acc0;
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][0], bq[s].x, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][1], bq[s].y, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][2], bq[s].z, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][3], bq[s].w, a, 0, 0, 0);
        }
        // C layout: col (s) = lane&15, row (channel) = (lane>>4)*4 + r
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[(4 * lq + r) * pitch + 16 * nt + li] = act_fwd(acc0[r] + acc1[r], p.act);
        if (nt == NT - 1) {                                          // hop finished: wave-private epilogue, as in the wave form
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                const float4 v = *reinterpret_cast<const float4*>(scr + li * pitch + 16 * s + 4 * lq);
                af[s][0] = v.x; af[s][1] = v.y; af[s][2] = v.z; af[s][3] = v.w;
            }
            for (int idx = lane; idx < 16 * p.dd; idx += 64) {       // relation_l = gather(h, heads) * gather(h, tails)
                const int cl = idx / p.dd, x = idx % p.dd;
                const int c = 16 * m + cl;
                if (c < p.C) {
                    const int64_t io = b * p.idx_bs + static_cast<int64_t>(c) * p.dd + x;
                    const int hi = static_cast<int>(p.head[io]), ti = static_cast<int>(p.tail[io]);
                    p.out[(static_cast<int64_t>(b) * p.C + c) * (p.L * p.dd) + l * p.dd + x] = scr[cl * pitch + hi] * scr[cl * pitch + ti];
                }
            }
            if (p.hsave) {
                float* hs = p.hsave + ((static_cast<int64_t>(l) * p.B + b) * p.C) * S;
                for (int cl = 0; cl < 16 && 16 * m + cl < p.C; ++cl)
                    for (int sidx = lane; sidx < S; sidx += 64) hs[static_cast<int64_t>(16 * m + cl) * S + sidx] = scr[cl * pitch + sidx];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------- P2 backward (one hop per launch)
template <int MT, bool VEC4>
__global__ void __launch_bounds__(1024) k_propagate_bwd_hop(const PropBwdK p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = blockDim.x >> 6;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int c0 = chunk * p.CC;
    const int pitch = p.pitch, S = p.S, CC = p.CC;
    float* X = lds;                                       // H^l, later H^l-1
    float* Y = lds + static_cast<int64_t>(CC) * pitch;    // grad wrt H^l -> grad wrt pre-activation
    // VEC4 (S % 4 == 0, 16-byte aligned states, <= kStage 16-byte pieces per thread): every thread requests all its pieces of a
    // state at once, branch free (rows past C re-read row C - 1 and are zeroed on the way into LDS), so a phase is ONE round trip.
    // s_memtime stamps at cfg 3b with the row-by-row form below: of a workgroup's 92.6 k cycles 25.4 k went into this phase and
    // 14.2 k into the H^l-1 one — guarded 4-byte loads, one dependent round trip per row and 64-column step.
    constexpr int kStage = 4;
    const int nf4 = S >> 2, npieces = CC * nf4;
    const bool staged = VEC4 && npieces <= kStage * nthreads;
    auto stage = [&](const float* src0, int64_t row_stride, bool zero_all, float* dst) {      // dst[cl][0 .. pitch) = row c0 + cl of src0
        float4 v[kStage];
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int idx = min(tid + i * nthreads, npieces - 1);
            const int cl = idx / nf4, j = idx - cl * nf4;
            v[i] = *reinterpret_cast<const float4*>(src0 + static_cast<int64_t>(min(c0 + cl, p.C - 1)) * row_stride + 4 * j);
        }
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < npieces) {
                const int cl = idx / nf4, j = idx - cl * nf4;
                const bool ok = c0 + cl < p.C && !zero_all;
                *reinterpret_cast<float4*>(dst + cl * pitch + 4 * j) = ok ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int idx = tid; idx < CC * (pitch - S); idx += nthreads) {    // the pad columns S .. pitch
            const int cl = idx / (pitch - S), s = S + idx % (pitch - S);
            dst[cl * pitch + s] = 0.f;
        }
    };
    if (staged) {
        stage(p.Hl + static_cast<int64_t>(b) * p.C * S, S, false, X);
        stage(p.gH + static_cast<int64_t>(b) * p.C * S, S, p.first != 0, Y);
    } else {
        // one wave per channel row, lanes along s: coalesced, no per-element division
        for (int cl = wave; cl < CC; cl += nwaves) {
            const int c = c0 + cl;
            const int64_t g0 = (static_cast<int64_t>(b) * p.C + c) * S;
            for (int s = lane; s < pitch; s += 64) {
                const bool ok = c < p.C && s < S;
                X[cl * pitch + s] = ok ? p.Hl[g0 + s] : 0.f;
                Y[cl * pitch + s] = (ok && !p.first) ? p.gH[g0 + s] : 0.f;
            }
        }
    }
    __syncthreads();
    // relation gradient: d(h[head]*h[tail])
    for (int idx = tid; idx < CC * p.dd; idx += nthreads) {
        const int cl = idx / p.dd, x = idx % p.dd;
        const int c = c0 + cl;
        if (c < p.C) {
            const int64_t io = b * p.idx_bs + static_cast<int64_t>(c) * p.dd + x;
            const int hi = static_cast<int>(p.head[io]), ti = static_cast<int>(p.tail[io]);
            const float g = p.gout[(static_cast<int64_t>(b) * p.C + c) * (p.L * p.dd) + p.hop * p.dd + x];
            atomicAdd(&Y[cl * pitch + hi], g * X[cl * pitch + ti]);
            atomicAdd(&Y[cl * pitch + ti], g * X[cl * pitch + hi]);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < CC * pitch; idx += nthreads) Y[idx] *= act_bwd(X[idx], p.act);
    __syncthreads();
    if (staged) {
        stage(p.Hprev + b * p.hprev_bs, S, false, X);
    } else {
        for (int cl = wave; cl < CC; cl += nwaves) {
            const int c = c0 + cl;
            const float* src = p.Hprev + b * p.hprev_bs + static_cast<int64_t>(c) * S;
            for (int s = lane; s < pitch; s += 64) X[cl * pitch + s] = (c < p.C && s < S) ? src[s] : 0.f;
        }
    }
    __syncthreads();
    const int li = lane & 15, lq = lane >> 4;
    const int NT = p.Sp >> 4;
    // (i) gA[s][t] = sum_c Y[c][s] * X[c][t]       (M = s, N = t, K = channel)
    // Work split.  (ii) gives column tile nt to wave nt (NT slabs x 4 x MT MFMAs); a tile of (i) costs CC/4 MFMAs, so NT
    // tiles of (i) weigh exactly one column of (ii): the waves beyond NT take NT tiles of (i) each while the first NT
    // waves run (ii), and whatever is left of (i) is dealt round-robin to everybody afterwards.  No barrier in between:
    // both products only read X, Y and A.
    const int wii = NT < nwaves ? NT : nwaves;                      // waves busy with (ii)
    const int nfree = nwaves - wii;
    const int tiles_a = p.gA ? min(NT * NT, nfree * NT) : 0;        // pass A: free waves only
    auto ga_tile = [&](int tile) {
        float* gA = p.gA + static_cast<int64_t>(b) * S * S;
        {
            const int ms = tile / NT, nt = tile % NT;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kc = 0; kc < CC; kc += 16) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int k = kc + 4 * lq + st;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Y[k * pitch + 16 * ms + li], X[k * pitch + 16 * nt + li], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 16 * ms + 4 * lq + r, t = 16 * nt + li;
                if (s < S && t < S) {
                    if (p.chunks == 1) gA[static_cast<int64_t>(s) * S + t] = acc[r];
                    else atomicAdd(&gA[static_cast<int64_t>(s) * S + t], acc[r]);
                }
            }
        }
    };
    if (p.gA && wave >= wii)
        for (int tile = wave - wii; tile < tiles_a; tile += nfree) ga_tile(tile);
    // (ii) gHprev[c][t] = sum_s Y[c][s] * A[s][t]    (M = channel, N = t, K = s)
    {
        const float* A = p.A + static_cast<int64_t>(b) * S * S;
        for (int nt = wave; nt < NT; nt += nwaves) {
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int col = 16 * nt + li;
            float bq[4], bn[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) bq[st] = (4 * lq + st < S && col < S) ? A[static_cast<int64_t>(4 * lq + st) * S + col] : 0.f;
            for (int slab = 0; slab < NT; ++slab) {
                const int kc = 16 * slab + 4 * lq;
#pragma unroll
                for (int st = 0; st < 4; ++st)          // rows of the next slab are in flight during this slab's MFMAs
                    bn[st] = (slab + 1 < NT && kc + 16 + st < S && col < S) ? A[static_cast<int64_t>(kc + 16 + st) * S + col] : 0.f;
                float4 aq[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) aq[m] = *reinterpret_cast<const float4*>(Y + (16 * m + li) * pitch + kc);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].x, bq[0], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].y, bq[1], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].z, bq[2], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].w, bq[3], acc[m], 0, 0, 0);
#pragma unroll
                for (int st = 0; st < 4; ++st) bq[st] = bn[st];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + 16 * m + 4 * lq + r;
                    if (c < p.C && col < S) p.gH[(static_cast<int64_t>(b) * p.C + c) * S + col] = acc[m][r];
                }
        }
    }
    if (p.gA)
        for (int tile = tiles_a + wave; tile < NT * NT; tile += nwaves) ga_tile(tile);      // pass B: the rest of (i), all waves
}

struct PropGeom { int CC, Sp, pitch, chunks, MT, fwd_waves; size_t lds, fwd_lds; };
bool prop_geometry(int C, int S, PropGeom* g) {
    g->Sp = (S + 15) / 16 * 16;
    g->pitch = g->Sp + 4;
    const int Cp = (C + 15) / 16 * 16;
    int cc = Cp < 80 ? Cp : 80;                                     // <= 5 MFMA row tiles per block
    const size_t budget = static_cast<size_t>(cfg_int(CFG_PROP_LDS_KB, 150)) * 1024;      // S = 512: 32-channel chunks instead of 16 (half the re-reads of A_l)
    while (cc > 16 && 2ull * cc * g->pitch * sizeof(float) > budget) cc -= 16;
    if (2ull * cc * g->pitch * sizeof(float) > 160 * 1024) return false;
    g->CC = cc;
    g->MT = cc / 16;
    g->chunks = (C + cc - 1) / cc;
    g->lds = 2ull * cc * g->pitch * sizeof(float);                  // backward: two state buffers
    const int NT = g->Sp / 16;
    g->fwd_waves = NT < 16 ? NT : 16;                               // forward: one wave per column tile
    g->fwd_lds = (NT > 16 ? 2ull : 1ull) * cc * g->pitch * sizeof(float);
    return true;
}

int check_prop(const recon_prop_args* a) {
    if (!a) return RECON_ERR_INVALID;
    if (a->B < 0 || a->C <= 0 || a->S <= 0 || a->L <= 0 || a->dd <= 0) return RECON_ERR_INVALID;
    if (a->L > kMaxHops) return RECON_ERR_UNSUPPORTED;
    if (a->B > 65535) return RECON_ERR_UNSUPPORTED;
    if (a->act < 0 || a->act > 2) return RECON_ERR_INVALID;
    if (!a->h0 || !a->head_idx || !a->tail_idx || !a->out) return RECON_ERR_INVALID;
    if (a->trans) {                                                 // block mode: dd == 16, S = 16 n, C = n (n - 1)
        if (!a->identity) return RECON_ERR_INVALID;
        for (int l = 0; l < a->L; ++l) if (!a->trans[l]) return RECON_ERR_INVALID;
        const int n = a->S / 16;
        if (a->dd != 16 || a->S != 16 * n || a->C != n * (n - 1)) return RECON_ERR_UNSUPPORTED;
        return RECON_OK;
    }
    if (!a->adj) return RECON_ERR_INVALID;
    for (int l = 0; l < a->L; ++l) if (!a->adj[l]) return RECON_ERR_INVALID;
    return RECON_OK;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace recon

using namespace recon;

extern "C" int recon_block_adjacency_fwd(const float* T, const float* identity, int32_t B, int32_t n, int32_t dd, float* A,
                                         recon_stream_t stream) {
    if (B < 0 || n < 1 || dd < 1 || !identity || !A || (n > 1 && B > 0 && !T)) return RECON_ERR_INVALID;
    const int64_t total = 1LL * B * n * dd * n * dd;
    if (total == 0) return RECON_OK;
    const bool v4 = dd % 4 == 0 && B <= 65535 && !((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(identity) | reinterpret_cast<uintptr_t>(A)) & 15);
    if (v4) hipLaunchKernelGGL(k_block_adj_fwd4, dim3(static_cast<unsigned>(ceil_div64(1LL * n * dd * n * dd / 4, 256)), static_cast<unsigned>(B)), dim3(256), 0,
                               as_stream(stream), T, identity, n, dd, A);
    else hipLaunchKernelGGL(k_block_adj_fwd, dim3(static_cast<unsigned>(ceil_div64(total, 256))), dim3(256), 0, as_stream(stream), T,
                            identity, B, n, dd, A);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" size_t recon_block_adjacency_bwd_workspace_floats(int32_t B, int32_t n, int32_t dd) {
    (void)B; (void)n;
    return static_cast<size_t>(kIdentSlices) * (dd > 0 ? dd : 1) * (dd > 0 ? dd : 1);
}

extern "C" int recon_block_adjacency_bwd(const float* gA, int32_t B, int32_t n, int32_t dd, float* gT, float* g_identity,
                                         float* workspace, recon_stream_t stream) {
    if (B < 0 || n < 1 || dd < 1 || !gA) return RECON_ERR_INVALID;
    if (g_identity && !workspace) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    const int64_t total = 1LL * B * n * (n - 1) * dd * dd;
    const bool v4 = dd % 4 == 0 && B <= 65535 && !((reinterpret_cast<uintptr_t>(gA) | reinterpret_cast<uintptr_t>(gT)) & 15);
    if (gT && total > 0) {
        if (v4) hipLaunchKernelGGL(k_block_adj_bwd_T4, dim3(static_cast<unsigned>(ceil_div64(1LL * n * (n - 1) * dd * dd / 4, 256)), static_cast<unsigned>(B)),
                                   dim3(256), 0, st, gA, n, dd, gT);
        else hipLaunchKernelGGL(k_block_adj_bwd_T, dim3(static_cast<unsigned>(ceil_div64(total, 256))), dim3(256), 0, st, gA, B, n, dd, gT);
    }
    if (g_identity) {
        const int64_t pairs = 1LL * B * n;
        const int slices = static_cast<int>(pairs < kIdentSlices ? (pairs > 0 ? pairs : 1) : kIdentSlices);
        hipLaunchKernelGGL(k_block_adj_bwd_I, dim3(slices), dim3(256), 0, st, gA, B, n, dd, workspace);
        hipLaunchKernelGGL(k_sum_rows, dim3(static_cast<unsigned>(ceil_div64(dd * dd, 16))), dim3(1024), 0, st, workspace, slices, dd * dd,
                           g_identity);
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_start_entity_embeddings(const float* ent, const int64_t* pos, const float* templ, int32_t B, int32_t n,
                                             int32_t d, float* out, recon_stream_t stream) {
    if (B < 0 || n < 2 || d < 1 || !ent || !pos || !templ || !out) return RECON_ERR_INVALID;
    const int64_t total = 1LL * B * n * (n - 1) * 2 * d * n;
    if (total == 0) return RECON_OK;
    hipLaunchKernelGGL(k_start_entity, dim3(static_cast<unsigned>(ceil_div64(total, 256))), dim3(256), 0, as_stream(stream), ent, pos,
                       templ, B, n, d, out);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_start_entity_embeddings_bwd(const float* grad_out, const int64_t* pos, const float* templ, int32_t B, int32_t n, int32_t d,
                                                 float* rows, int64_t* key, recon_stream_t stream) {
    if (B < 0 || n < 2 || d < 1 || !grad_out || !pos || !templ || !rows) return RECON_ERR_INVALID;
    const int64_t total = 2LL * B * n * (n - 1) * d;
    if (total == 0) return RECON_OK;
    hipLaunchKernelGGL(k_start_entity_bwd, dim3(static_cast<unsigned>(ceil_div64(total, 256))), dim3(256), 0, as_stream(stream), grad_out, pos, templ, B,
                       n, d, rows, key);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

#define RECON_DISPATCH_MT(MTV, V4, KERNEL, ...)                                                                        \
    do {                                                                                                                \
        switch (MTV) {                                                                                                  \
            case 1: if (V4) hipLaunchKernelGGL((KERNEL<1, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<1, false>), __VA_ARGS__); break; \
            case 2: if (V4) hipLaunchKernelGGL((KERNEL<2, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<2, false>), __VA_ARGS__); break; \
            case 3: if (V4) hipLaunchKernelGGL((KERNEL<3, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<3, false>), __VA_ARGS__); break; \
            case 4: if (V4) hipLaunchKernelGGL((KERNEL<4, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<4, false>), __VA_ARGS__); break; \
            default: if (V4) hipLaunchKernelGGL((KERNEL<5, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<5, false>), __VA_ARGS__); break; \
        }                                                                                                               \
    } while (0)

extern "C" int recon_propagate_fwd(const recon_prop_args* a, recon_stream_t stream) {
    int rc = check_prop(a);
    if (rc != RECON_OK) return rc;
    if (a->B == 0) return RECON_OK;
    PropGeom g;
    if (!prop_geometry(a->C, a->S, &g)) return RECON_ERR_UNSUPPORTED;
    PropK p;
    bool v4 = (a->S % 4) == 0;
    const bool blk = a->trans != nullptr;
    for (int l = 0; l < kMaxHops; ++l) {
        p.adj[l] = (l < a->L && !blk) ? a->adj[l] : nullptr;
        p.trans[l] = (l < a->L && blk) ? a->trans[l] : nullptr;
        if (l < a->L && !al16(blk ? a->trans[l] : a->adj[l])) v4 = false;
    }
    p.identity = blk ? a->identity : nullptr;
    p.h0 = a->h0; p.h0_bs = a->h0_batch_stride; p.head = a->head_idx; p.tail = a->tail_idx; p.idx_bs = a->idx_batch_stride;
    p.out = a->out; p.hsave = a->h_saved;
    p.B = a->B; p.C = a->C; p.S = a->S; p.L = a->L; p.dd = a->dd; p.act = a->act;
    p.CC = g.CC; p.Sp = g.Sp; p.pitch = g.pitch;
    p.stats = a->h_saved ? a->stats : nullptr;
    p.ws = a->split_ws; p.ws_bytes = a->split_ws_bytes;
    hipStream_t st = as_stream(stream);
    const int NTn = g.Sp / 16;
    const int mtn_s = (a->C + 15) / 16;
    {
        // default: two-term half operands on the f16 matrix cores (prop_h.hip) wherever that form exists; RECON_PROP_FWD = w | b | s | x
        // selects one of the forms below (fp32 MFMA per wave / per workgroup / staged, bf16 x 3), h forces the default
        const char* form = cfg(CFG_PROP_FWD);
        if ((!form || form[0] == 'h' || form[0] == '\0') && prop_fwd_h_supported(p)) return prop_fwd_h(p, st);
        if ((!form || form[0] == 'h' || form[0] == '\0') && prop_fwd_hl_supported(p)) return prop_fwd_hl(p, st);     // wide states, given a workspace
        if (blk) return RECON_ERR_UNSUPPORTED;                          // the other forms need a materialised adjacency
        const int ntc = (a->C + 15) / 16, ks = (a->S + 31) / 32, mw = (a->S + 15) / 16;
        if (form && form[0] == 'x' && ntc <= 8 && ks <= 4 && mw <= 16) {     // S <= 128: wider states spill in this form             // bf16 x 3 on the bf16 matrix cores: opt-in (see the kernel's header)
            const size_t xlds = 3ull * ks * ntc * 16 * 64;
            bool launched = true;
#define CALL_X(N_, K_) do { if (xlds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_fwd_x<N_, K_>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(xlds)); \
                            hipLaunchKernelGGL((k_propagate_fwd_x<N_, K_>), dim3(static_cast<unsigned>(a->B)), dim3(64 * mw), xlds, st, p); } while (0)
#define CALL_XK(N_) switch (ks) { case 1: CALL_X(N_, 1); break; case 2: CALL_X(N_, 2); break; case 3: CALL_X(N_, 3); break; default: CALL_X(N_, 4); break; }
            switch (ntc) { case 1: CALL_XK(1); break; case 2: CALL_XK(2); break; case 3: CALL_XK(3); break; case 4: CALL_XK(4); break; case 5: CALL_XK(5); break;
                           case 6: CALL_XK(6); break; case 7: CALL_XK(7); break; case 8: CALL_XK(8); break; default: launched = false; break; }
#undef CALL_XK
#undef CALL_X
            if (launched) { RECON_CHECK_LAUNCH(); return RECON_OK; }
        }
    }
    if (NTn <= 9 && (a->S % 16) == 0 && v4 && mtn_s <= 16 && g.pitch == a->S + 4 && cfg_char(CFG_PROP_FWD) == 's') {   // staged form
        const size_t slds = 3ull * 16 * (a->S + 4) * sizeof(float) + static_cast<size_t>(mtn_s) * 16 * g.pitch * sizeof(float);
#define CALL_S(N_) do { if (slds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_fwd_s<N_>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(slds)); \
                        hipLaunchKernelGGL((k_propagate_fwd_s<N_>), dim3(static_cast<unsigned>(a->B)), dim3(64 * mtn_s), slds, st, p); } while (0)
        switch (NTn) { case 1: CALL_S(1); break; case 2: CALL_S(2); break; case 3: CALL_S(3); break; case 4: CALL_S(4); break;
                       case 5: CALL_S(5); break; case 6: CALL_S(6); break; case 7: CALL_S(7); break; case 8: CALL_S(8); break;
                       default: CALL_S(9); break; }
#undef CALL_S
        RECON_CHECK_LAUNCH();
        return RECON_OK;
    }
    if (NTn <= 9 && cfg_char(CFG_PROP_FWD) != 'b') {      // wave-independent form (RECON_PROP_FWD=w, or shapes the bf16 form does not take)
        const int64_t units = 1LL * a->B * ((a->C + 15) / 16);
        dim3 wgrid(static_cast<unsigned>(ceil_div64(units, 4)));
        const size_t wlds = 4ull * 16 * g.pitch * sizeof(float);
#define CALL_W(N_) do { if (v4) hipLaunchKernelGGL((k_propagate_fwd_w<N_, true>), wgrid, dim3(256), wlds, st, p); \
                        else hipLaunchKernelGGL((k_propagate_fwd_w<N_, false>), wgrid, dim3(256), wlds, st, p); } while (0)
        switch (NTn) { case 1: CALL_W(1); break; case 2: CALL_W(2); break; case 3: CALL_W(3); break; case 4: CALL_W(4); break;
                       case 5: CALL_W(5); break; case 6: CALL_W(6); break; case 7: CALL_W(7); break; case 8: CALL_W(8); break;
                       default: CALL_W(9); break; }
#undef CALL_W
        RECON_CHECK_LAUNCH();
        return RECON_OK;
    }
    dim3 grid(static_cast<unsigned>(g.chunks), static_cast<unsigned>(a->B));
    if (g.fwd_lds > 64 * 1024) {
#define SET_ATTR(MTV, V)                                                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_fwd<MTV, V>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              static_cast<int>(g.fwd_lds))
        switch (g.MT) { case 1: SET_ATTR(1, true); SET_ATTR(1, false); break; case 2: SET_ATTR(2, true); SET_ATTR(2, false); break;
                        case 3: SET_ATTR(3, true); SET_ATTR(3, false); break; case 4: SET_ATTR(4, true); SET_ATTR(4, false); break;
                        default: SET_ATTR(5, true); SET_ATTR(5, false); break; }
#undef SET_ATTR
    }
    RECON_DISPATCH_MT(g.MT, v4, k_propagate_fwd, grid, dim3(64 * g.fwd_waves), g.fwd_lds, st, p);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

static bool prop_h_form_env() {
    const char* form = cfg(CFG_PROP_FWD);
    return !form || form[0] == 'h' || form[0] == '\0';
}

extern "C" size_t recon_propagate_ws_bytes(const recon_prop_args* a) {
    if (!a || !prop_h_form_env() || (a->S % 4) != 0 || a->dd < 1) return 0;
    if (a->trans && (a->dd != 16 || a->S % 16 != 0 || a->C != (a->S / 16) * (a->S / 16 - 1))) return 0;
    return prop_hl_ws_bytes(a->B, a->S, a->L);
}

extern "C" size_t recon_propagate_identity_ws_floats(int32_t dd) {
    return static_cast<size_t>(prop_h_grid(1 << 30)) * (dd > 0 ? dd : 1) * (dd > 0 ? dd : 1);
}

extern "C" int recon_propagate_form(const recon_prop_args* a) {
    if (!a || a->B < 0 || a->C <= 0 || a->S <= 0 || a->L <= 0 || a->L > kMaxHops || a->dd <= 0 || !a->h0) return 0;
    PropGeom g;
    if (!prop_geometry(a->C, a->S, &g)) return 0;
    PropK p{};
    for (int l = 0; l < kMaxHops; ++l) p.adj[l] = nullptr;              // the adjacency pointers' alignment is checked at the call
    p.h0 = a->h0; p.h0_bs = a->h0_batch_stride; p.hsave = nullptr;
    p.B = a->B; p.C = a->C; p.S = a->S; p.L = 0; p.dd = a->dd; p.pitch = g.pitch;
    if (!(prop_h_form_env() && prop_fwd_h_supported(p))) return 0;
    return prop_bwd_h_shape_ok(a->C, a->S) ? 3 : 1;                     // bit 1: the backward's two-term form exists for this shape too
}

// ------------------------------------------------------------------------------- P2 backward for wide states (S > 160)
// At S = 512, C = 992 neither product of a hop fits a workgroup's LDS with all channels, and the channel-chunked kernel above pays for
// it (d A_l needs every channel: chunks meet in atomics; A_l is re-read per chunk).  Written as what they are, both products are plain
// batched GEMMs over the graphs, on the library's own fp32 matrix-core GEMM (gemm_f32.hip):
//     (d)  G_l[b]   = Y_l[b] [C x S] . A_l[b] [S x S]              -> d loss / d H^l-1 before the relation term
//     (c)  dA_l[b]  = Y_l[b]^T [S x C] . H^l-1[b] [C x S]
// with Y_l = (G_l+1 + relation gradient of hop l) . act'(H^l) produced in place by k_prop_bwd_post (one wave per (graph, channel) row:
// the row in LDS, the 2 dd scatter terms as LDS float atomics, act', one pass).  Needs one [B, C, S] workspace besides g_h.
namespace {
__global__ void __launch_bounds__(256) k_prop_bwd_post(const float* __restrict__ G, const float* __restrict__ Hl, const int64_t* __restrict__ head,
                                                        const int64_t* __restrict__ tail, int64_t idx_bs, const float* __restrict__ gout, float* __restrict__ Y,
                                                        int64_t rows, int32_t C, int32_t S, int32_t L, int32_t dd, int32_t hop, int32_t act) {
    extern __shared__ float rowbuf[];                                   // [4][S]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + w;
    if (row >= rows) return;
    const int64_t b = row / C;
    const int c = static_cast<int>(row - b * C);
    float* buf = rowbuf + w * S;
    const float* g = G ? G + row * S : nullptr;
    const float* h = Hl + row * S;
    for (int t = lane; t < S; t += 64) buf[t] = g ? g[t] : 0.f;
    __builtin_amdgcn_s_waitcnt(0xc07f);                                 // this wave's LDS writes have landed (one wave per row: no barrier)
    const int64_t io = b * idx_bs + static_cast<int64_t>(c) * dd;
    const float* go = gout + row * (static_cast<int64_t>(L) * dd) + static_cast<int64_t>(hop) * dd;
    for (int x = lane; x < dd; x += 64) {                               // out = h[head] * h[tail]  (models/models.py:270-273)
        const int hi = static_cast<int>(head[io + x]), ti = static_cast<int>(tail[io + x]);
        const float gv = go[x];
        atomicAdd(buf + hi, gv * h[ti]);
        atomicAdd(buf + ti, gv * h[hi]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float* y = Y + row * S;
    for (int t = lane; t < S; t += 64) y[t] = buf[t] * act_bwd(h[t], act);
}

int prop_bwd_wide(const recon_prop_args* a, const recon_prop_bwd_args* ba, hipStream_t st) {
    const int32_t B = a->B, C = a->C, S = a->S, L = a->L;
    const int64_t CS = 1LL * C * S, BCS = CS * B, rows = 1LL * B * C;
    // the last product must land in g_h: the L products alternate between the two buffers
    float* bufY = (L & 1) ? ba->wide_ws : ba->g_h;                      // holds Y_L first
    float* bufG = (L & 1) ? ba->g_h : ba->wide_ws;
    const dim3 pgrid(static_cast<unsigned>(ceil_div64(rows, 4)));
    const size_t plds = 4ull * S * sizeof(float);
    hipLaunchKernelGGL(k_prop_bwd_post, pgrid, dim3(256), plds, st, nullptr, a->h_saved + static_cast<int64_t>(L - 1) * BCS, a->head_idx, a->tail_idx,
                       a->idx_batch_stride, ba->grad_out, bufY, rows, C, S, L, a->dd, L - 1, a->act);
    for (int l = L; l >= 1; --l) {
        const float* Hprev = l == 1 ? a->h0 : a->h_saved + static_cast<int64_t>(l - 2) * BCS;
        const int64_t hprev_bs = l == 1 ? a->h0_batch_stride : CS;
        GemmBatch bt;
        bt.batch = B; bt.epilogue = 0;
        if (ba->g_adj && ba->g_adj[l - 1]) {                            // (c): A = Y_l as [K = c][M = s], B = H^l-1 as [K = c][N = t]
            bt.a_bs = CS; bt.b_bs = hprev_bs; bt.c_bs = 1LL * S * S;
            const int rc = gemm_f32_batched(S, S, C, plain_operand(bufY, S), false, plain_operand(Hprev, S), false, plain_output(ba->g_adj[l - 1], S), bt, 1,
                                            nullptr, st);
            if (rc != RECON_OK) return rc;
        }
        {                                                               // (d): A = Y_l [M = c][K = s], B = A_l as [K = s][N = t]
            bt.a_bs = CS; bt.b_bs = 1LL * S * S; bt.c_bs = CS;
            const int rc = gemm_f32_batched(C, S, S, plain_operand(bufY, S), true, plain_operand(a->adj[l - 1], S), false, plain_output(bufG, S), bt, 1,
                                            nullptr, st);
            if (rc != RECON_OK) return rc;
        }
        if (l > 1)
            hipLaunchKernelGGL(k_prop_bwd_post, pgrid, dim3(256), plds, st, bufG, a->h_saved + static_cast<int64_t>(l - 2) * BCS, a->head_idx, a->tail_idx,
                               a->idx_batch_stride, ba->grad_out, bufG, rows, C, S, L, a->dd, l - 2, a->act);
        float* t = bufY; bufY = bufG; bufG = t;
    }
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// The same backward with the chain G_{l-1} = Y_l A_l, Y_{l-1} = (G_{l-1} + relation gradient) . act'(H^{l-1}) on the forward's two-term f16 kernel
// (prop_hl.hip: k_propagate_fwd_hl<.., true> over the TRANSPOSED adjacencies, all L steps of a slice of graphs in one launch — no G round
// trip, no row kernel between the hops); the Y_l leave as fp32 and feed the d A_l products.  Needs the gather indices as blocks of 16 columns
// (GP-GNN's: utils/embedding_utils.py:184-202), the forward's split workspace and L slices of [G, C, S] floats.
int prop_bwd_wide_chain(const recon_prop_args* a, const recon_prop_bwd_args* ba, hipStream_t st) {
    const int32_t B = a->B, C = a->C, S = a->S, L = a->L;
    const int64_t CS = 1LL * C * S, BCS = CS * B;
    const bool blk = a->trans != nullptr;                               // block mode: transition tensors read / d T written in place, no adjacency
    const int nn = S / 16;
    const int64_t G = prop_bwd_hl_slice(C, S, L, a->split_ws_bytes, B);
    if (G <= 0) return RECON_ERR_UNSUPPORTED;
    float* diag_ws = blk ? ba->chain_ws + prop_bwd_hl_ws_floats(C, S, L, G) : nullptr;      // [L][B][n][256]
    const bool want_ident = blk && ba->g_identity;
    for (int64_t g0 = 0; g0 < B; g0 += G) {
        const int32_t Gs = static_cast<int32_t>(B - g0 < G ? B - g0 : G);
        PropBwdHL c{};
        c.G = Gs; c.C = C; c.S = S; c.L = L; c.dd = a->dd; c.act = a->act; c.ws = a->split_ws; c.ws_bytes = a->split_ws_bytes;
        c.gout = ba->grad_out + g0 * C * L * a->dd; c.hblk = ba->head_blk; c.tblk = ba->tail_blk;
        c.identity = blk ? a->identity : nullptr;
        float* y_in; unsigned char* planes; float* isg; size_t pset, iset;
        prop_bwd_hl_ws_layout(C, S, L, G, ba->chain_ws, &y_in, &planes, &isg, &pset, &iset);
        c.yplanes = planes; c.yisg = isg;
        c.y_in = nullptr;
        c.hlast = a->h_saved + static_cast<int64_t>(L - 1) * BCS + g0 * CS;  // Y_L = relation gradient of the last hop . act'(H^L), formed by the chain kernel
        for (int k = 0; k < L; ++k) {
            const int l = L - k;
            c.adj_step[k] = blk ? a->trans[l - 1] + g0 * C * 256 : a->adj[l - 1] + g0 * S * S;
            c.hmask[k] = l >= 2 ? a->h_saved + static_cast<int64_t>(l - 2) * BCS + g0 * CS : nullptr;
            c.ysave[k] = l >= 2 ? nullptr : ba->g_h + g0 * CS;           // the Y_l leave as fragment planes
            c.gout_off[k] = l >= 2 ? (l - 2) * a->dd : 0;
        }
        int rc = prop_bwd_hl_chain(c, st);
        if (rc != RECON_OK) return rc;
        for (int k = 0; k < L; ++k) {                                   // d A_l[b] = Y_l[b]^T [S x C] . H^l-1[b] [C x S]
            const int l = L - k;
            float* gA = (!blk && ba->g_adj) ? ba->g_adj[l - 1] : nullptr;
            float* gT = (blk && ba->g_trans) ? ba->g_trans[l - 1] : nullptr;
            if (!gA && !gT && !want_ident) continue;
            const float* Hprev = l == 1 ? a->h0 + g0 * a->h0_batch_stride : a->h_saved + static_cast<int64_t>(l - 2) * BCS + g0 * CS;
            // NOTE: plane sets / scale sets are laid out for slices of G graphs; a short last slice uses the front of each set
            rc = prop_bwd_hl_gadj(planes + static_cast<size_t>(k) * (pset / G) * Gs, isg + static_cast<size_t>(k) * (iset / G) * Gs, Hprev,
                                  l == 1 ? a->h0_batch_stride : CS, gA ? gA + g0 * S * S : nullptr, gT ? gT + g0 * C * 256 : nullptr,
                                  blk ? diag_ws + (static_cast<int64_t>(l - 1) * B + g0) * nn * 256 : nullptr, Gs, C, S, st);
            if (rc != RECON_OK) return rc;
        }
    }
    if (want_ident)                                                     // d identity = the diagonal blocks of every hop, graph and node, fixed order
        hipLaunchKernelGGL(k_sum_rows, dim3(16), dim3(1024), 0, st, diag_ws, static_cast<int32_t>(1LL * L * B * nn), 256, ba->g_identity);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}
}  // namespace

extern "C" size_t recon_propagate_bwd_chain_ws_floats(const recon_prop_args* a) {
    if (!a || a->B <= 0 || a->dd != 16 || a->idx_batch_stride != 0 || !a->split_ws) return 0;
    if (a->trans && (!a->identity || a->S != 16 * (a->S / 16) || a->C != (a->S / 16) * (a->S / 16 - 1))) return 0;
    if (1LL * a->L * a->B * (a->S / 16) >= (1LL << 31)) return 0;
    const int64_t G = prop_bwd_hl_slice(a->C, a->S, a->L, a->split_ws_bytes, a->B);
    const bool off = cfg_char(CFG_PROP_BWD_CHAIN) == '0';
    if (G <= 0 || off) return 0;
    return prop_bwd_hl_ws_floats(a->C, a->S, a->L, G) + (a->trans ? static_cast<size_t>(a->L) * a->B * (a->S / 16) * 256 : 0);
}

extern "C" size_t recon_propagate_bwd_ws_floats(const recon_prop_args* a) {
    if (!a || a->trans || a->S <= 160 || a->B <= 0) return 0;
    return static_cast<size_t>(a->B) * a->C * a->S;
}

extern "C" int recon_propagate_bwd(const recon_prop_bwd_args* ba, recon_stream_t stream) {
    if (!ba) return RECON_ERR_INVALID;
    const recon_prop_args* a = &ba->fwd;
    int rc = check_prop(a);
    if (rc != RECON_OK) return rc;
    if (!a->h_saved || !ba->grad_out || !ba->g_h) return RECON_ERR_INVALID;
    if (a->B == 0) return RECON_OK;
    if (a->S > 160 && ba->chain_ws && ba->head_blk && ba->tail_blk && recon_propagate_bwd_chain_ws_floats(a) > 0 &&
        cfg_char(CFG_PROP_BWD_WIDE) != '0') {
        return prop_bwd_wide_chain(a, ba, as_stream(stream));          // wide states, structured indices: chain + d A on the two-term f16 kernels (block mode too)
    }
    if (a->stats && prop_h_form_env() && cfg_char(CFG_PROP_BWD) != 'f') {      // two-term f16 form (RECON_PROP_BWD=f: fp32 MFMA form)
        PropBwdH q{};
        const bool blk = a->trans != nullptr;
        for (int l = 0; l < kMaxHops; ++l) {
            q.adj[l] = (l < a->L && !blk) ? a->adj[l] : nullptr; q.gadj[l] = (l < a->L && !blk && ba->g_adj) ? ba->g_adj[l] : nullptr;
            q.trans[l] = (l < a->L && blk) ? a->trans[l] : nullptr; q.gtrans[l] = (l < a->L && blk && ba->g_trans) ? ba->g_trans[l] : nullptr;
        }
        q.identity = blk ? a->identity : nullptr;
        q.gident_ws = (blk && ba->g_identity) ? ba->identity_ws : nullptr;
        if (blk && ba->g_identity && !ba->identity_ws) return RECON_ERR_INVALID;
        q.h0 = a->h0; q.h0_bs = a->h0_batch_stride; q.hsave = a->h_saved; q.head = a->head_idx; q.tail = a->tail_idx; q.idx_bs = a->idx_batch_stride;
        q.gout = ba->grad_out; q.gH = ba->g_h; q.stats = a->stats;
        q.B = a->B; q.C = a->C; q.S = a->S; q.L = a->L; q.dd = a->dd; q.act = a->act;
        if (prop_bwd_h_supported(q)) {
            rc = prop_bwd_h(q, as_stream(stream));
            if (rc == RECON_OK && q.gident_ws)                          // per-workgroup partial sums -> g_identity, fixed order
                hipLaunchKernelGGL(k_sum_rows, dim3(16), dim3(1024), 0, as_stream(stream), q.gident_ws, prop_h_grid(a->B), 256, ba->g_identity);
            return rc;
        }
        if (blk) return RECON_ERR_UNSUPPORTED;
    } else if (a->trans) return RECON_ERR_UNSUPPORTED;
    {
        const bool wide_off = cfg_char(CFG_PROP_BWD_WIDE) == '0';
        if (!wide_off && ba->wide_ws && a->S > 160 && 4ull * a->S * sizeof(float) <= 64 * 1024)     // wide states: both products as batched GEMMs
            return prop_bwd_wide(a, ba, as_stream(stream));
    }
    PropGeom g;
    if (!prop_geometry(a->C, a->S, &g)) return RECON_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t BCS = 1LL * a->B * a->C * a->S;
    dim3 grid(static_cast<unsigned>(g.chunks), static_cast<unsigned>(a->B));
    if (g.lds > 64 * 1024) {
#define SET_ATTR(MTV, V)                                                                                                 \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_bwd_hop<MTV, V>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              static_cast<int>(g.lds))
        switch (g.MT) { case 1: SET_ATTR(1, true); SET_ATTR(1, false); break; case 2: SET_ATTR(2, true); SET_ATTR(2, false); break;
                        case 3: SET_ATTR(3, true); SET_ATTR(3, false); break; case 4: SET_ATTR(4, true); SET_ATTR(4, false); break;
                        default: SET_ATTR(5, true); SET_ATTR(5, false); break; }
#undef SET_ATTR
    }
    for (int l = a->L; l >= 1; --l) {
        PropBwdK p;
        p.A = a->adj[l - 1];
        p.Hl = a->h_saved + static_cast<int64_t>(l - 1) * BCS;
        if (l == 1) { p.Hprev = a->h0; p.hprev_bs = a->h0_batch_stride; }
        else { p.Hprev = a->h_saved + static_cast<int64_t>(l - 2) * BCS; p.hprev_bs = 1LL * a->C * a->S; }
        p.head = a->head_idx; p.tail = a->tail_idx; p.idx_bs = a->idx_batch_stride;
        p.gout = ba->grad_out; p.gH = ba->g_h;
        p.gA = ba->g_adj ? ba->g_adj[l - 1] : nullptr;
        p.B = a->B; p.C = a->C; p.S = a->S; p.L = a->L; p.dd = a->dd; p.act = a->act;
        p.CC = g.CC; p.Sp = g.Sp; p.pitch = g.pitch; p.hop = l - 1; p.first = (l == a->L) ? 1 : 0; p.chunks = g.chunks;
        if (p.gA && g.chunks > 1 && hipMemsetAsync(p.gA, 0, sizeof(float) * a->B * a->S * a->S, st) != hipSuccess) return RECON_ERR_LAUNCH;
        const bool v4b = (a->S % 4) == 0 && al16(p.Hl) && al16(p.Hprev) && al16(p.gH) && (p.hprev_bs % 4) == 0;
        RECON_DISPATCH_MT(g.MT, v4b, k_propagate_bwd_hop, grid, dim3(1024), g.lds, st, p);
        RECON_CHECK_LAUNCH();
    }
    return RECON_OK;
}

// ------------------------------------------------------------------------------- P5 / K6 GraphConvolution
namespace {

// Y[b][i][o] = epilogue( sum_j M[i][j] * Xin[b][j][o] ),  M = adj[b] (TRANS = false) or adj[b]^T (TRANS = true), on the
// matrix cores (v_mfma_f32_16x16x4_f32).  Block = (64-column tile, graph, 32-row tile); wave w owns 16 columns and both
// 16-row tiles; the contraction index is walked in chunks of 32 through LDS, so any n is accepted (n <= 32, the
// reference's regime, is one chunk and one row tile).
// MASK: Xin = gout * (fwd_out > 0)   (ReLU backward folded into the load)
// EPI : + bias, ReLU
// LDS images are k-major with pitches = 16 (mod 32) floats, so the two k groups a 32-lane read touches fall on disjoint
// banks.  (A VALU form with two LDS reads per four FMAs was LDS-issue bound: 37 us at cfg 3a for 80 MB of traffic.)
template <bool TRANS, bool MASK, bool EPI>
__global__ void __launch_bounds__(256) k_gcn_aggregate_mfma(const float* __restrict__ adj, const float* __restrict__ Xin,
                                                            const float* __restrict__ fwd_out, const float* __restrict__ bias,
                                                            int32_t n, int32_t O, float* __restrict__ Y) {
    constexpr int PM = 48, PX = 80;
    __shared__ float Mk[32 * PM];           // Mk[k][i] = M[i0 + i][k0 + k]
    __shared__ float Xs[32 * PX];           // Xs[k][o]
    const int b = blockIdx.y, o0 = blockIdx.x * 64, i0 = blockIdx.z * 32;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* A = adj + static_cast<int64_t>(b) * n * n;
    const int li = lane & 15, lq = lane >> 4;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int k0 = 0; k0 < n; k0 += 32) {
        if (k0) __syncthreads();
        for (int idx = t; idx < 32 * 32; idx += 256) {
            const int k = idx >> 5, i = idx & 31;
            float v = 0.f;
            if (k0 + k < n && i0 + i < n)
                v = TRANS ? A[static_cast<int64_t>(k0 + k) * n + i0 + i] : A[static_cast<int64_t>(i0 + i) * n + k0 + k];
            Mk[k * PM + i] = v;
        }
        for (int idx = t; idx < 32 * 64; idx += 256) {
            const int k = idx >> 6, o = o0 + (idx & 63);
            float v = 0.f;
            if (k0 + k < n && o < O) {
                const int64_t g = (static_cast<int64_t>(b) * n + k0 + k) * O + o;
                v = Xin[g];
                if constexpr (MASK) v = fwd_out[g] > 0.f ? v : 0.f;
            }
            Xs[k * PX + (idx & 63)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = 4 * s + lq;
            const float bx = Xs[k * PX + 16 * w + li];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Mk[k * PM + li], bx, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Mk[k * PM + 16 + li], bx, acc[1], 0, 0, 0);
        }
    }
    const int o = o0 + 16 * w + li;                                 // C layout: col = lane & 15, row = (lane >> 4) * 4 + r
    if (o < O) {
        const float bv = (EPI && bias) ? bias[o] : 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 16 * tt + 4 * lq + r;
                if (i < n) {
                    float v = acc[tt][r];
                    if constexpr (EPI) { v += bv; v = v > 0.f ? v : 0.f; }
                    Y[(static_cast<int64_t>(b) * n + i) * O + o] = v;
                }
            }
    }
}

// g_adj[b][i][j] = sum_o gpre[b][i][o] * support[b][j][o]
__global__ void __launch_bounds__(256) k_gcn_grad_adj(const float* __restrict__ gout, const float* __restrict__ fwd_out,
                                                      const float* __restrict__ sup, int32_t n, int32_t O,
                                                      float* __restrict__ gadj) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx % n;
    const int64_t ri = (static_cast<int64_t>(b) * n + i) * O, rj = (static_cast<int64_t>(b) * n + j) * O;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(fwd_out[ri + o] > 0.f ? gout[ri + o] : 0.f, sup[rj + o], s);
    gadj[static_cast<int64_t>(b) * n * n + idx] = s;
}

// g_bias[o] = sum_rows gpre[row][o]: per-block partial rows, then a fixed-order second pass
__global__ void __launch_bounds__(256) k_gcn_bias_partial(const float* __restrict__ gout, const float* __restrict__ fwd_out,
                                                          int64_t rows, int32_t O, int32_t rows_per_block,
                                                          float* __restrict__ partial) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                     // four independent chains: the loop is latency bound
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += fwd_out[r * O + o] > 0.f ? gout[r * O + o] : 0.f;
        s1 += fwd_out[(r + 1) * O + o] > 0.f ? gout[(r + 1) * O + o] : 0.f;
        s2 += fwd_out[(r + 2) * O + o] > 0.f ? gout[(r + 2) * O + o] : 0.f;
        s3 += fwd_out[(r + 3) * O + o] > 0.f ? gout[(r + 3) * O + o] : 0.f;
    }
    for (; r < r1; ++r) s0 += fwd_out[r * O + o] > 0.f ? gout[r * O + o] : 0.f;
    partial[static_cast<int64_t>(blockIdx.y) * O + o] = (s0 + s1) + (s2 + s3);
}
int check_gcn(const recon_gcn_args* a) {
    if (!a || a->B < 0 || a->n <= 0 || a->in_features <= 0 || a->out_features <= 0) return RECON_ERR_INVALID;
    if (!a->x || !a->adj || !a->weight || !a->support || !a->out) return RECON_ERR_INVALID;
    if (a->B > 65535 || a->n > 65535 * 32) return RECON_ERR_UNSUPPORTED;      // grid.y = graphs, grid.z = 32-row tiles
    return RECON_OK;
}
constexpr int kBiasBlocks = 1024;

}  // namespace

static size_t gcn_split_part(int32_t rows, int32_t K) { return align_up(static_cast<size_t>(3) * rows * bx3_kp(K) * 2, 256); }
extern "C" size_t recon_gcn_split_bytes(int32_t in_features, int32_t out_features) {
    if (in_features <= 0 || out_features <= 0) return 256;
    return gcn_split_part(out_features, in_features) + gcn_split_part(in_features, out_features);
}

extern "C" int recon_gcn_fwd(const recon_gcn_args* a, recon_stream_t stream) {
    int rc = check_gcn(a);
    if (rc != RECON_OK) return rc;
    if (a->B == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    const int32_t rows = a->B * a->n, I = a->in_features, O = a->out_features;
    // support = x @ W      (models/layers.py:58)
    GemmBatch one;
    one.batch = 1; one.a_bs = one.b_bs = one.c_bs = 0; one.epilogue = 0;
    const OperandDesc X = plain_operand(a->x, I);
    if (a->w_split && !(reinterpret_cast<uintptr_t>(a->w_split) & 15) && bx3_supported(X, I, one)) {
        // bf16 term planes of W^T [3][O][kp(I)] (this product) and of W [3][I][kp(O)] (g_x in the backward)
        char* ws = static_cast<char*>(a->w_split);
        rc = bx3_split_planes(a->weight, O, 0, true, O, I, 1, ws, st);
        if (rc == RECON_OK) rc = bx3_split_planes(a->weight, O, 0, false, I, O, 1, ws + gcn_split_part(O, I), st);
        if (rc == RECON_OK) rc = gemm_bx3_batched(rows, O, I, X, ws, plain_output(a->support, O), one, st);
    } else {
        rc = gemm_f32(rows, O, I, X, true, plain_operand(a->weight, O), false, plain_output(a->support, O), 1, nullptr, st);
    }
    if (rc != RECON_OK) return rc;
    // out = relu(adj @ support + bias)     (models/layers.py:59-63)
    dim3 grid(static_cast<unsigned>(ceil_div64(O, 64)), static_cast<unsigned>(a->B), static_cast<unsigned>(ceil_div64(a->n, 32)));
    hipLaunchKernelGGL((k_gcn_aggregate_mfma<false, false, true>), grid, dim3(256), 0, st, a->adj, a->support, nullptr, a->bias, a->n, O,
                       a->out);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" size_t recon_gcn_bwd_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t out_features) {
    size_t need = static_cast<size_t>(kBiasBlocks) * out_features;
    int sk = gemm_pick_split_k(in_features, out_features, B * n);
    const int sk3 = bx3_kmajor_splits(B * n, bx3_kmajor_split_k(in_features, out_features, B * n, 1));      // split-precision form
    if (sk3 > sk) sk = sk3;
    const size_t g = static_cast<size_t>(sk) * in_features * out_features;
    return g > need ? g : need;
}

extern "C" size_t recon_gcn_bwd_split_bytes(int32_t B, int32_t n, int32_t out_features) {
    if (B <= 0 || n <= 0 || out_features <= 0) return 256;
    return align_up(static_cast<size_t>(3) * B * n * bx3_kp(out_features) * 2, 256);
}

extern "C" int recon_gcn_bwd(const recon_gcn_bwd_args* b, recon_stream_t stream) {
    if (!b) return RECON_ERR_INVALID;
    const recon_gcn_args* a = &b->fwd;
    int rc = check_gcn(a);
    if (rc != RECON_OK) return rc;
    if (!b->grad_out || !b->g_support || !b->partial) return RECON_ERR_INVALID;
    if (a->B == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    const int32_t rows = a->B * a->n, I = a->in_features, O = a->out_features, n = a->n;
    dim3 grid(static_cast<unsigned>(ceil_div64(O, 64)), static_cast<unsigned>(a->B), static_cast<unsigned>(ceil_div64(n, 32)));
    // g_support = adj^T @ (grad_out * (out > 0))
    hipLaunchKernelGGL((k_gcn_aggregate_mfma<true, true, false>), grid, dim3(256), 0, st, a->adj, b->grad_out, a->out, nullptr, n, O,
                       b->g_support);
    if (b->g_adj)
        hipLaunchKernelGGL(k_gcn_grad_adj, dim3(static_cast<unsigned>(ceil_div64(n * n, 256)), static_cast<unsigned>(a->B)), dim3(256), 0,
                           st, b->grad_out, a->out, a->support, n, O, b->g_adj);
    if (b->g_bias) {
        const int rpb = static_cast<int>(ceil_div64(rows, kBiasBlocks));
        const int nb = static_cast<int>(ceil_div64(rows, rpb));
        hipLaunchKernelGGL(k_gcn_bias_partial, dim3(static_cast<unsigned>(ceil_div64(O, 256)), static_cast<unsigned>(nb)), dim3(256), 0, st,
                           b->grad_out, a->out, static_cast<int64_t>(rows), O, rpb, b->partial);
        hipLaunchKernelGGL(k_sum_rows, dim3(static_cast<unsigned>(ceil_div64(O, 16))), dim3(1024), 0, st, b->partial, nb, O, b->g_bias);
    }
    RECON_CHECK_LAUNCH();
    // g_x = g_support @ W^T
    if (b->g_x) {
        GemmBatch one;
        one.batch = 1; one.a_bs = one.b_bs = one.c_bs = 0; one.epilogue = 0;
        const OperandDesc G = plain_operand(b->g_support, O);
        if (a->w_split && !(reinterpret_cast<uintptr_t>(a->w_split) & 15) && bx3_supported(plain_operand(a->x, I), I, one) &&
            bx3_supported(G, O, one))                                  // same condition as the forward, which filled the planes
            rc = gemm_bx3_batched(rows, I, O, G, static_cast<const char*>(a->w_split) + gcn_split_part(O, I), plain_output(b->g_x, I), one, st);
        else
            rc = gemm_f32(rows, I, O, G, true, plain_operand(a->weight, O), true, plain_output(b->g_x, I), 1, nullptr, st);
        if (rc != RECON_OK) return rc;
    }
    // g_W = x^T @ g_support   (split-K over the B*n rows, deterministic second pass)
    if (b->g_weight) {
        const int64_t ldp = bx3_kp(O);
        if (b->gs_split && a->w_split && !(reinterpret_cast<uintptr_t>(b->gs_split) & 15) && bx3_kmajor_supported(a->x, I, 0, ldp, 0, I, O)) {
            // split-precision, both operands k-major: x split on the fly, g_support from its bf16 term planes
            rc = bx3_split_planes(b->g_support, O, 0, false, rows, O, 1, b->gs_split, st);
            const int sk = bx3_kmajor_splits(rows, bx3_kmajor_split_k(I, O, rows, 1));
            if (rc == RECON_OK) rc = gemm_bx3_kmajor_batched(I, O, rows, a->x, I, 0, b->gs_split, ldp, static_cast<int64_t>(rows) * ldp, 0, 1, sk, b->partial, st);
            if (rc == RECON_OK) rc = splitk_reduce(b->partial, sk, I, O, plain_output(b->g_weight, O), 0, 1, 0, false, st);
        } else {
            const int sk = gemm_pick_split_k(I, O, rows);
            rc = gemm_f32(I, O, rows, plain_operand(a->x, I), false, plain_operand(b->g_support, O), false, plain_output(b->g_weight, O), sk,
                          b->partial, st);
        }
        if (rc != RECON_OK) return rc;
    }
    return RECON_OK;
}
