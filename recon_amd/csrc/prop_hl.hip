// K5''' — GP-GNN propagation (models/models.py:260-274) for WIDE states (160 < S <= 512: up to 32 nodes at 2d = 16, BASELINE.json
// configs[2] "n=32") on the f16 matrix cores with two-term operands, the arithmetic of prop_h.hip (fp32-class accuracy at 3 x
// v_mfma_f32_16x16x32_f16 per 16x16x32 block).  A graph's state no longer fits one CU's LDS, but channels never mix
// (h^l[c] = act(A_l h^l-1[c])), so a workgroup owns (graph, 64 channels) for all L hops and the 16 workgroups of a graph share
// A_l through their XCD's L2:
//   * k_prop_split_adj (once per slice of graphs): every row of every A_l gets its own power-of-two scale (s.amax in [2^14, 2^15)) and
//     is written as two half terms IN MFMA FRAGMENT ORDER — one KiB per (16 rows, K step, term), lane after lane — so that the
//     propagation kernel's operand loads are plain coalesced 16-byte loads with no arithmetic behind them (each A_l is consumed by
//     C / 64 workgroups: converting it once instead of C / 64 times is what this pass buys).  In block mode it reads the transition
//     tensors / the identity in place (models/models.py:240-259): the S x S adjacency never exists in fp32.
//   * k_propagate_fwd_hl: 8 waves; wave w owns rows 16 RT w .. of A_l (RT row tiles at once: a B fragment read from LDS feeds RT MFMAs
//     per term), walks the K steps with the next step's A fragments in flight (across hop boundaries), keeps its RT x 4 output tiles in
//     accumulators over the barrier that ends the hop's reads and writes the new state IN PLACE: 64 channels x 512 columns x two
//     half planes = 128 KiB of LDS, under per-channel power-of-two scales (max magnitude gathered with LDS atomics, as in prop_h.hip);
//     head (.) tail gather and the saved state are reconstructed from hi + lo.
// Workgroup -> (graph, chunk) puts the chunks of one graph on ONE XCD (blockIdx.x & 7 is the XCD under round-robin dispatch).
#include <stdlib.h>
#include "prop_common.h"
#include "prop_h_util.h"

namespace recon {
namespace {

constexpr int kCH = 64;                 // channels per workgroup
constexpr int kSTEP = kCH * 64;         // bytes of one K step (32 columns) of one plane: 64 bytes per channel
constexpr int kHLWaves = 8;
// buffer-load aux bit 31: a volatile access — the compiler neither moves nor merges it (see k_prop_gadj_hl: as plain read-only loads the A-fragment
// requests of a loop without stores were sunk next to their uses, i.e. the two-K-step prefetch was compiled away)
#ifdef RECON_HL_PLAIN_LOADS
constexpr int kVolatileLoad = 0;
#else
constexpr int kVolatileLoad = static_cast<int>(0x80000000u);
#endif

struct PropHL {
    PropK p;
    unsigned char* split;               // [L][G][RTT = 8 RT][NKS][2 terms][64 lanes x 16 B]
    float* alpha;                       // [L][G][RP = 128 RT] inverse row scales
    int32_t g0, G, nchunks, NKS, RT;    // slice = graphs g0 .. g0 + G - 1; NKS even (K padded with zeros)
    // ---- backward chain (k_propagate_fwd_hl<.., true>): step k multiplies by A_{L-k}^T (p.adj[k], split TRANSPOSED) and forms
    //      Y_{l-1} = (G + relation gradient of hop l-1) . act'(H^{l-1}) in its epilogue, l = L - k
    const float* hmask[kMaxHops];       // H^{l-1} of step k ([G][C][S], slice-local) or null (l = 1: the product itself is d loss / d h^0)
    float* ysave[kMaxHops];             // where step k's new state goes ([G][C][S]): Y_{l-1}, the last one d loss / d h^0
    const float* gout;                  // [G][C][gout_ld]
    const int32_t* hblk; const int32_t* tblk;   // [C]: head / tail indices are blocks of 16 columns starting here (multiples of 16)
    int32_t gout_ld, gout_off[kMaxHops];        // L dd; column of step k's relation gradient ((l - 2) dd)
    // the states Y_L, Y_{L-1}, .., Y_1 leave as the d A products' streamed operand: two half planes in MFMA fragment order, rows = state
    // column s, K = channel, as they sit in LDS (i.e. times the channel's power-of-two scale, whose inverse goes to yisg)
    const float* hlast;                 // H^L [G][C][S]: the first state Y_L = relation gradient of hop L . act'(H^L) is formed in the prologue (p.h0 unused)
    unsigned char* yplanes;             // [L emissions][G][RTT][NKC][2][1 KiB] or null
    float* yisg;                        // [L emissions][G][32 NKC]
    int32_t NKC;                        // K steps of the d A products: channels / 32, rounded up to a multiple of 2
};

// ---------------------------------------------------------------------------------------------------------------- split pass
// One wave per (hop, graph, 16 rows): the rows' fragments for all K steps in registers, row maximum across the four lanes of a row,
// scale, two half terms, 16-byte stores in fragment order.  K step ks, lane (li, lq): t = 32 ks + 4 lq .. + 3 and 32 ks + 16 + 4 lq .. + 3
// (the k permutation of prop_h.hip: both pieces are contiguous 16 bytes of the fp32 row).
template <bool BLK, bool TRANS = false>
__global__ void __launch_bounds__(256) k_prop_split_adj(const PropHL q) {
    const PropK& p = q.p;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int RTT = 8 * q.RT, RP = 128 * q.RT, NKS = q.NKS;
    int64_t unit = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const int rt = static_cast<int>(unit % RTT);
    unit /= RTT;
    const int g = static_cast<int>(unit % q.G), l = static_cast<int>(unit / q.G);
    if (l >= p.L) return;
    const int b = q.g0 + g, S = p.S;
    constexpr int MAXK = 16;
    u32x4 raw[MAXK][2];
    if constexpr (TRANS && !BLK) {
        // rows of A_l^T: element (row, t) = A_l[t][row]; the 16 lanes li of a group read 64 contiguous bytes of row t of A_l
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.adj[l] + static_cast<int64_t>(b) * S * S), 0, S * S * 4, 0x00020000);
        const int row = 16 * rt + li;
#pragma unroll
        for (int ks = 0; ks < MAXK; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t0 = 32 * ks + 16 * h + 4 * lq;
                uint32_t e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    e[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, (row < S && t0 + i < S) ? static_cast<uint32_t>((t0 + i) * S + row) * 4u : kOOB, 0, 0);
                raw[ks][h] = u32x4{e[0], e[1], e[2], e[3]};
            }
    } else if constexpr (!BLK) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.adj[l] + static_cast<int64_t>(b) * S * S), 0, S * S * 4, 0x00020000);
        const int row = 16 * rt + li;
        const uint32_t base = row < S ? static_cast<uint32_t>(row * S + 4 * lq) * 4u : kOOB;
#pragma unroll
        for (int ks = 0; ks < MAXK; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t0 = 32 * ks + 16 * h + 4 * lq;
                raw[ks][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, (row < S && t0 < S) ? base + static_cast<uint32_t>(32 * ks + 16 * h) * 4u : kOOB, 0, 0);
            }
    } else if constexpr (BLK && TRANS) {
        // rows of node i = rt of A_l^T: block (i, j) = trans[l][b, e(j, i)]^T — element (li, c) = T[e(j, i)][c][li] — identity^T on the diagonal
        const int nn = S >> 4, C = p.C;
        const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.trans[l] + static_cast<int64_t>(b) * C * 256), 0, C * 1024, 0x00020000);
        const auto rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.identity), 0, 1024, 0x00020000);
#pragma unroll
        for (int ks = 0; ks < MAXK; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * ks + h;
                const bool ok = rt < nn && j < nn, diag = j == rt;
                const uint32_t e = static_cast<uint32_t>(j * (nn - 1) + (rt < j ? rt : rt - 1));
                const auto rs = diag ? rs_i : rs_t;
                uint32_t v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? static_cast<uint32_t>((4 * lq + i) * 16 + li) * 4u : kOOB, diag ? 0 : static_cast<int>(e * 1024u), 0);
                raw[ks][h] = u32x4{v[0], v[1], v[2], v[3]};
            }
    } else {
        // rows of node i = rt: block (i, j) = trans[l][b, e(i, j)] (row li, columns 4 lq ..), the identity on the diagonal
        const int nn = S >> 4, C = p.C;
        const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.trans[l] + static_cast<int64_t>(b) * C * 256), 0, C * 1024, 0x00020000);
        const auto rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.identity), 0, 1024, 0x00020000);
        const uint32_t vb = static_cast<uint32_t>(li * 16 + 4 * lq) * 4u;
#pragma unroll
        for (int ks = 0; ks < MAXK; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * ks + h;
                const bool ok = rt < nn && j < nn, diag = j == rt;
                const uint32_t e = static_cast<uint32_t>(rt * (nn - 1) + (j < rt ? j : j - 1));
                const auto rs = diag ? rs_i : rs_t;                     // wave-uniform: a scalar select
                raw[ks][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? vb : kOOB, diag ? 0 : static_cast<int>(e * 1024u), 0);
            }
    }
    float m = 0.f;
#pragma unroll
    for (int ks = 0; ks < MAXK; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32x4 v = raw[ks][h];
            m = fmaxf(fmaxf(fabsf(as_f(v.x)), fabsf(as_f(v.y))), m);
            m = fmaxf(fmaxf(fabsf(as_f(v.z)), fabsf(as_f(v.w))), m);
        }
    m = rows_max(m);
    const float alpha = hx2_scale_of(m);
    if (lq == 0) q.alpha[(static_cast<int64_t>(l) * q.G + g) * RP + 16 * rt + li] = hx2_inv(alpha);
    unsigned char* dst = q.split + (((static_cast<int64_t>(l) * q.G + g) * RTT + rt) * NKS) * 2048 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < MAXK; ++ks) {
        if (ks < NKS) {
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 v = raw[ks][h];
                hx2_split2(as_f(v.x) * alpha, as_f(v.y) * alpha, hi[2 * h], lo[2 * h]);
                hx2_split2(as_f(v.z) * alpha, as_f(v.w) * alpha, hi[2 * h + 1], lo[2 * h + 1]);
            }
            *reinterpret_cast<u32x4*>(dst + ks * 2048) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4*>(dst + ks * 2048 + 1024) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- propagation
// State image (as in prop_h.hip, with 64 channels per K step): element (channel c, column t) of plane q lives at byte
//     q PLANE + (t >> 5) kSTEP + 64 c + 16 (((t >> 2) & 3) ^ ((c >> 1) & 3)) + 8 ((t >> 4) & 1) + 2 (t & 3).
__device__ __forceinline__ int hl_pos4(int c, int t0) {                 // t0 % 4 == 0: the 8 bytes holding columns t0 .. t0 + 3
    return (t0 >> 5) * kSTEP + 64 * c + ((((t0 >> 2) & 3) ^ ((c >> 1) & 3)) << 4) + (((t0 >> 4) & 1) << 3);
}

template <int RT, int NKS, bool BWD = false>
__global__ void __launch_bounds__(64 * kHLWaves) k_propagate_fwd_hl(const PropHL q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const PropK& p = q.p;
    constexpr int PLANE = NKS * kSTEP, KP = NKS * 32;
    const int S = p.S, C = p.C, L = p.L;
    unsigned char* Hs = sm;
    uint32_t* chmax = reinterpret_cast<uint32_t*>(sm + 2 * PLANE);      // [2][kCH]
    float* isg = reinterpret_cast<float*>(chmax + 2 * kCH);             // [kCH] inverse channel scales
    float* atab = isg + kCH;                                            // [L][128 RT] inverse row scales of this graph's A_l
    uint32_t* gtab = reinterpret_cast<uint32_t*>(atab + p.L * 128 * RT);   // [8][512] gather items of each thread (kept out of the registers)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int wg = blockIdx.x, xcd = wg & 7, rr = wg >> 3;
    const int chunk = rr % q.nchunks, gs = (rr / q.nchunks) * 8 + xcd;
    if (gs >= q.G) return;
    const int b = q.g0 + gs, c0 = chunk * kCH;
    const int nch = C - c0 < kCH ? C - c0 : kCH;                        // valid channels of this chunk
    const int RTT = 8 * RT, RP = 128 * RT;
    const uint32_t hop_bytes = static_cast<uint32_t>(RTT) * NKS * 2048u;

    // ---- A fragments: two slots of one K step each, refilled row tile by row tile two K steps ahead
    u32x4 ring[2][RT][2];
    uint32_t voff_rt[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) voff_rt[r] = static_cast<uint32_t>((wave * RT + r) * NKS) * 2048u + lane * 16u;
    auto rsrc_hop = [&](int l) {
        return __builtin_amdgcn_make_buffer_rsrc(q.split + (static_cast<int64_t>(l) * q.G + gs) * hop_bytes, 0, static_cast<int>(hop_bytes), 0x00020000);
    };
    auto rs_cur = rsrc_hop(0);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)                                       // K steps 0 and 1 of hop 0 (NKS >= 2)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            ring[sl][r][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, voff_rt[r], sl * 2048, 0);
            ring[sl][r][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, voff_rt[r] + 1024u, sl * 2048, 0);
        }

    auto store_state4 = [&](int off, float v0, float v1, float v2, float v3) {
        uint32_t h0, l0, h1, l1;
        hx2_split2(v0, v1, h0, l0);
        hx2_split2(v2, v3, h1, l1);
        *reinterpret_cast<uint2*>(Hs + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(Hs + PLANE + off) = make_uint2(l0, l1);
    };

    // BWD: the state image -> fragment planes of Y^T for the d A product (lane (li, lq) of row tile T, K step ks: s = 16 T + li, channels
    // 32 ks + 4 lq .. + 3 and + 16): two transposing reads per plane — lane ip of a 16-lane group addresses the 8 bytes (4 columns) of
    // channel base + (ip >> 2), columns 16 T + 4 (ip & 3) .., and receives column 16 T + ip of the 4 channels
    [[maybe_unused]] auto emit_planes = [&](int e) {
        if constexpr (BWD) {
            if (!q.yplanes) return;
            typedef short i16x4_t __attribute__((ext_vector_type(4)));
            unsigned char* dst = q.yplanes + ((static_cast<int64_t>(e) * q.G + gs) * RTT) * q.NKC * 2048 + lane * 16;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int T = wave * RT + r, t4 = 16 * T + 4 * (li & 3);
                if (16 * T >= KP) continue;                             // wave-uniform: row tiles past the image (their rows of d A are not stored)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const int ca = 32 * kk + 4 * lq + (li >> 2);
                        const unsigned char* a0 = Hs + pl * PLANE + hl_pos4(ca, t4);
                        const unsigned char* a1 = Hs + pl * PLANE + hl_pos4(ca + 16, t4);
                        const i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)(a0));
                        const i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)(a1));
                        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        *reinterpret_cast<u32x4*>(dst + ((static_cast<int64_t>(T) * q.NKC + 2 * chunk + kk) * 2 + pl) * 1024) = v;
                    }
            }
            if (tid < kCH) q.yisg[(static_cast<int64_t>(e) * q.G + gs) * 32 * q.NKC + c0 + tid] = isg[tid];
        }
    };

    // ---- h^0 of this chunk: wave w stages channels w, w + 8, ... (a whole channel per wave: its max magnitude is a wave reduction), four
    // channels per batch; columns past S and channels past C come back as zeros and are written as zeros
    {
        if (tid < 2 * kCH) chmax[tid] = 0u;
        for (int i = tid; i < L * RP; i += 64 * kHLWaves) {             // all hops' row scales: no global load inside the hop loop but the A fragments
            const int l = i / RP;
            atab[i] = q.alpha[(static_cast<int64_t>(l) * q.G + gs) * RP + (i - l * RP)];
        }
        if constexpr (BWD) {
            {
                // (compile-time branch: a run-time one in front of the hop loop left the request counters of the two paths different at the
                // loop header, and the compiler answered with s_waitcnt vmcnt(0) at every K step — 1.77 -> 2.8 ms per slice)
                // Y_L[c][s] = grad_out[c][L-1][x] H^L[c][partner block + x] act'(H^L[c][s]) on the 16 + 16 columns of channel c's head and tail
                // blocks, zero everywhere else: 48 values per channel come from memory instead of a [C, S] tensor written by a kernel of its own
                const int64_t row0 = static_cast<int64_t>(b) * C + c0;
                const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.hlast + row0 * S), 0, nch * S * 4, 0x00020000);
                const int goff = (L - 1) * p.dd;
                const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.gout + row0 * q.gout_ld + goff), 0, (nch * q.gout_ld - goff) * 4, 0x00020000);
                constexpr int NC = kCH / kHLWaves;                      // channels per wave: every request of all of them is issued before the first use
                int mine[NC], other[NC];
                const int x = lane & 15;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int cc = min(c0 + wave + kHLWaves * i, C - 1);
                    const int hbc = q.hblk[cc], tbc = q.tblk[cc];
                    mine[i] = (lane < 16 ? hbc : tbc) + x; other[i] = (lane < 16 ? tbc : hbc) + x;
                }
                float gv[NC], ho[NC], hm_[NC];
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int cl = wave + kHLWaves * i;
                    const bool ok = cl < nch && lane < 32;
                    gv[i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rs_g, ok ? static_cast<uint32_t>(cl * q.gout_ld + x) * 4u : kOOB, 0, 0));
                    ho[i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rs_h, ok ? static_cast<uint32_t>(cl * S + other[i]) * 4u : kOOB, 0, 0));
                    hm_[i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rs_h, ok ? static_cast<uint32_t>(cl * S + mine[i]) * 4u : kOOB, 0, 0));
                }
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int cl = wave + kHLWaves * i;
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int t0 = 4 * lane + 256 * ps;
                        if (t0 < KP) store_state4(hl_pos4(cl, t0), 0.f, 0.f, 0.f, 0.f);
                    }
                }
                lds_wait();                                             // this wave's zeros have landed: the 32 values per channel go on top
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int cl = wave + kHLWaves * i;
                    const float v = gv[i] * ho[i] * act_bwd(hm_[i], p.act);
                    const float sg = hx2_scale_of(wave_max(fabsf(v)));
                    if (cl < nch && lane < 32) {
                        uint32_t h, l_;
                        hx2_split2(v * sg, 0.f, h, l_);
                        const int pos = hl_pos4(cl, mine[i] & ~3) + 2 * (mine[i] & 3);
                        *reinterpret_cast<uint16_t*>(Hs + pos) = static_cast<uint16_t>(h & 0xffffu);
                        *reinterpret_cast<uint16_t*>(Hs + PLANE + pos) = static_cast<uint16_t>(l_ & 0xffffu);
                    }
                    if (lane == 0) isg[cl] = hx2_inv(sg);
                }
            }
        }
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h0 + b * p.h0_bs + static_cast<int64_t>(c0) * S), 0, nch * S * 4, 0x00020000);
#pragma unroll 1
        for (int batch = 0; batch < (BWD ? 0 : 2); ++batch) {
            u32x4 hv[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cl = wave + 8 * (4 * batch + i);
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int t0 = 4 * lane + 256 * ps;
                    hv[i][ps] = __builtin_amdgcn_raw_buffer_load_b128(rs, (cl < nch && t0 < S) ? static_cast<uint32_t>(cl * S + t0) * 4u : kOOB, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cl = wave + 8 * (4 * batch + i);
                float mx = 0.f;
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const u32x4 v = hv[i][ps];
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(as_f(v.x)), fabsf(as_f(v.y))), fmaxf(fabsf(as_f(v.z)), fabsf(as_f(v.w)))));
                }
                mx = wave_max(mx);
                const float sg = hx2_scale_of(mx);
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int t0 = 4 * lane + 256 * ps;
                    const u32x4 v = hv[i][ps];
                    if (t0 < KP) store_state4(hl_pos4(cl, t0), as_f(v.x) * sg, as_f(v.y) * sg, as_f(v.z) * sg, as_f(v.w) * sg);
                }
                if (lane == 0) isg[cl] = hx2_inv(sg);
            }
        }
    }
    // ---- gather items of this thread (the same in every hop): (channel cl, x) -> byte positions of head / tail in plane 0
    const int nitems = nch * p.dd, Ldd = L * p.dd;
    if constexpr (!BWD) {
        uint32_t g_hi[2], g_ti[2], g_o[2], g_c[2];
        const int64_t* hd = p.head + b * p.idx_bs + static_cast<int64_t>(c0) * p.dd;
        const int64_t* tl = p.tail + b * p.idx_bs + static_cast<int64_t>(c0) * p.dd;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = min(tid + i * 64 * kHLWaves, nitems - 1);
            const int cl = idx / p.dd;
            const int th = static_cast<int>(hd[idx]), tt = static_cast<int>(tl[idx]);
            g_hi[i] = hl_pos4(cl, th & ~3) + 2 * (th & 3);
            g_ti[i] = hl_pos4(cl, tt & ~3) + 2 * (tt & 3);
            g_o[i] = 4u * static_cast<uint32_t>(idx + cl * (Ldd - p.dd));                  // cl L dd + x
            g_c[i] = 4u * cl;
            gtab[(4 * i + 0) * 512 + tid] = g_hi[i]; gtab[(4 * i + 1) * 512 + tid] = g_ti[i];
            gtab[(4 * i + 2) * 512 + tid] = g_o[i]; gtab[(4 * i + 3) * 512 + tid] = g_c[i];
        }
    }
    lds_barrier();
    emit_planes(0);                                                     // Y_L
    float inv_sig[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inv_sig[j] = isg[16 * j + li];

    const int swz = ((li >> 1) & 3) << 4;
    const int b_rd = li * 64 + ((lq << 4) ^ swz);                       // B fragment: + 1024 j + kSTEP ks (+ PLANE for the low terms)
    // backward chain: everything a step applies to its product is linear in it (a mask, 1 - h^2, an added term): the channel scale commutes
    const bool homog = BWD || p.act != RECON_ACT_TANH, relu = !BWD && p.act == RECON_ACT_RELU;
    // BWD: first columns of the channels' head / tail blocks live in the (otherwise unused) gather table: [0 .. 63] head, [64 .. 127] tail
    if constexpr (BWD) {
        if (tid < 2 * kCH) {
            const int cc = min(c0 + (tid & (kCH - 1)), C - 1);
            gtab[tid] = static_cast<uint32_t>(tid < kCH ? q.hblk[cc] : q.tblk[cc]);
        }
        lds_barrier();
    }

#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const auto rs_next = rsrc_hop(l + 1 < L ? l + 1 : l);
        f32x4 acc[RT][4];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // one K step out of ring slot ks & 1.  Row tile after row tile: as soon as a tile's 12 MFMAs are issued its two fragment registers
        // are free and take the SAME tile of K step ks + 2 (next hop past the end; out of range past the last hop: zeros, no branch) — two K
        // steps of A in flight in two slots' worth of registers.  The K loop is unrolled: straight-line code lets the compiler count the
        // outstanding requests (s_waitcnt vmcnt(N) instead of draining them at every step).
        const uint32_t next_ok = l + 1 < L ? 0u : kOOB;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int BUF = ks & 1;
            const bool wrap = ks + 2 >= NKS;
            const auto rs = wrap ? rs_next : rs_cur;
            const uint32_t so = static_cast<uint32_t>(wrap ? ks + 2 - NKS : ks + 2) * 2048u;
            const uint32_t oob = wrap ? next_ok : 0u;
            const unsigned char* hb = Hs + ks * kSTEP + b_rd;
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(hb + 1024 * j);
                bl[j] = *reinterpret_cast<const f16x8*>(hb + PLANE + 1024 * j);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f16x8 ah = __builtin_bit_cast(f16x8, ring[BUF][r][0]), al = __builtin_bit_cast(f16x8, ring[BUF][r][1]);
#ifndef RECON_HL_NOMFMA
                // the wave that issues matrix-core work goes first (the other wave of the SIMD is typically waiting for its fragments): -3 % on the forward;
                // the chain form (heavier epilogue, spills) loses 10 % with it
                if constexpr (!BWD) { if (r == 0) __builtin_amdgcn_s_setprio(2); }
                // small terms first; four independent accumulator chains
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], acc[r][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], acc[r][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc[r][j], 0, 0, 0);
#else
                acc[r][0][0] += static_cast<float>(ah[0]) + static_cast<float>(bl[r & 3][1]); acc[r][1][1] += static_cast<float>(al[7]) + static_cast<float>(bh[r & 3][2]);
#endif
#ifndef RECON_HL_NOLOAD
                // (plain loads here: the compiler keeps them 8 .. 9 requests ahead of their uses; as volatile ones — 14 .. 15 ahead — the forward is
                // unchanged at 1.24 ms per slice and the chain form spills 57 registers: 1.69 -> 1.82 ms)
                ring[BUF][r][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_rt[r] | oob, so, 0);
                ring[BUF][r][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (voff_rt[r] + 1024u) | oob, so, 0);
#endif
                __builtin_amdgcn_sched_barrier(0x078f);                 // everything but VMEM may move across: the requests stay where they are written
            }
            if constexpr (!BWD) __builtin_amdgcn_s_setprio(0);
        }
        rs_cur = rs_next;

        // ---- epilogue 1: row scales off, activation, channel maxima.  C layout: column (lane & 15) = channel 16 j + li, rows 4 lq + r of row tile r.
        uint32_t* cm = chmax + (l & 1) * kCH;
        float4 ia[RT];                                                  // inverse row scales of this wave's rows (C layout: rows 4 lq + r of each row tile)
#pragma unroll
        for (int r = 0; r < RT; ++r) if constexpr (!BWD) ia[r] = *reinterpret_cast<const float4*>(atab + l * RP + 16 * (wave * RT + r) + 4 * lq);
        const float* ia_lds = atab + l * RP + 16 * wave * RT + 4 * lq;   // BWD re-reads them where they are used (registers)
        // BWD: H^{l-1} of this chunk (activation derivative, relation gradient) and grad_out as range-checked buffers: rows past S and
        // channels past C read zeros
        const float* hm = nullptr;
        __amdgpu_buffer_rsrc_t rs_hm, rs_go;
        if constexpr (BWD) {
            hm = q.hmask[l];
            const int64_t row0 = static_cast<int64_t>(b) * C + c0;
            rs_hm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((hm ? hm : p.h0) + row0 * S), 0, nch * S * 4, 0x00020000);
            rs_go = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.gout + row0 * q.gout_ld + q.gout_off[l]), 0,
                                                      (nch * q.gout_ld - q.gout_off[l]) * 4, 0x00020000);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mm = 0.f;
            // BWD: everything this column of tiles needs from memory is requested at once, branch-free (what does not exist reads zeros):
            // H^{l-1}[c][rows of this lane] for the activation derivative, and the relation gradient of hop l-1 (models/models.py:270-273
            // backwards: out = h[head] * h[tail] puts grad_out * h[tail block] on the head block's rows and grad_out * h[head block] on the
            // tail block's) — the head / tail block of channel c is ONE row tile each, owned by one wave
            [[maybe_unused]] u32x4 dm[RT], g4, hx, hy;
            [[maybe_unused]] float sig = 0.f;
            [[maybe_unused]] int rh = -1, rt_ = -1;
            if constexpr (BWD) {
                if (hm) {
                    const uint32_t cb = static_cast<uint32_t>((16 * j + li) * S) * 4u;
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        const int t0 = 16 * (wave * RT + r) + 4 * lq;
                        dm[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_hm, t0 < S ? cb + static_cast<uint32_t>(t0) * 4u : kOOB, 0, 0);
                    }
                    const int hbj = static_cast<int>(gtab[16 * j + li]), tbj = static_cast<int>(gtab[kCH + 16 * j + li]);
                    const bool okc = 16 * j + li < nch;
                    rh = (hbj >> 4) - wave * RT; rt_ = (tbj >> 4) - wave * RT;
                    const bool has_h = okc && rh >= 0 && rh < RT, has_t = okc && rt_ >= 0 && rt_ < RT;
                    g4 = __builtin_amdgcn_raw_buffer_load_b128(rs_go, (has_h || has_t) ? static_cast<uint32_t>((16 * j + li) * q.gout_ld + 4 * lq) * 4u : kOOB, 0, 0);
                    hx = __builtin_amdgcn_raw_buffer_load_b128(rs_hm, has_h ? cb + static_cast<uint32_t>(tbj + 4 * lq) * 4u : kOOB, 0, 0);
                    hy = __builtin_amdgcn_raw_buffer_load_b128(rs_hm, has_t ? cb + static_cast<uint32_t>(hbj + 4 * lq) * 4u : kOOB, 0, 0);
                    sig = hx2_inv(inv_sig[j]);                          // the product is in units of the old channel scale: so is what is added to it
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float u = homog ? 1.f : inv_sig[j];
                if constexpr (BWD) {
                    const f32x4 t = *reinterpret_cast<const volatile f32x4*>(ia_lds + 16 * r);
                    ia[r] = make_float4(t[0], t[1], t[2], t[3]);
                }
                float v0 = acc[r][j][0] * (ia[r].x * u), v1 = acc[r][j][1] * (ia[r].y * u), v2 = acc[r][j][2] * (ia[r].z * u), v3 = acc[r][j][3] * (ia[r].w * u);
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                if (!homog) { v0 = tanh_fast(v0); v1 = tanh_fast(v1); v2 = tanh_fast(v2); v3 = tanh_fast(v3); }
                if constexpr (BWD) {
                    if (hm) {
                        const float sh = r == rh ? sig : 0.f, st_ = r == rt_ ? sig : 0.f;
                        v0 += as_f(g4.x) * (as_f(hx.x) * sh + as_f(hy.x) * st_); v1 += as_f(g4.y) * (as_f(hx.y) * sh + as_f(hy.y) * st_);
                        v2 += as_f(g4.z) * (as_f(hx.z) * sh + as_f(hy.z) * st_); v3 += as_f(g4.w) * (as_f(hx.w) * sh + as_f(hy.w) * st_);
                        const u32x4 d = dm[r];
                        v0 *= act_bwd(as_f(d.x), p.act); v1 *= act_bwd(as_f(d.y), p.act); v2 *= act_bwd(as_f(d.z), p.act); v3 *= act_bwd(as_f(d.w), p.act);
                    }
                }
                acc[r][j] = f32x4{v0, v1, v2, v3};
                mm = fmaxf(mm, fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))));
            }
            atomicMax(cm + 16 * j + li, __builtin_bit_cast(uint32_t, mm));
        }
        lds_barrier();                                                  // everybody has read H^l-1; maxima complete
        // ---- epilogue 2: H^l under its new channel scales, in place
        if (tid < kCH) chmax[((l + 1) & 1) * kCH + tid] = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float inv_u = homog ? inv_sig[j] : 1.f;               // the unit acc[.][j] is in now
            const float sg = hx2_scale_of(__builtin_bit_cast(float, cm[16 * j + li]) * inv_u);
            const float f = sg * inv_u;
            inv_sig[j] = hx2_inv(sg);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int t0 = 16 * (wave * RT + r) + 4 * lq;
                if (t0 < KP) store_state4(hl_pos4(16 * j + li, t0), acc[r][j][0] * f, acc[r][j][1] * f, acc[r][j][2] * f, acc[r][j][3] * f);
            }
            if (tid < 16) isg[16 * j + li] = inv_sig[j];
        }
        lds_barrier();                                                  // H^l complete
        if constexpr (BWD) { if (l + 1 < L) emit_planes(l + 1); }
        // ---- relation_l = gather(h, heads) * gather(h, tails)   (models/models.py:270-273), saved state; values = (hi + lo) / scale
        auto state_at = [&](uint32_t pos) {
            return static_cast<float>(*reinterpret_cast<const _Float16*>(Hs + pos)) + static_cast<float>(*reinterpret_cast<const _Float16*>(Hs + PLANE + pos));
        };
        char* out = BWD ? nullptr : reinterpret_cast<char*>(p.out + ((static_cast<int64_t>(b) * C + c0) * L + l) * p.dd);
        if constexpr (!BWD) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + i * 64 * kHLWaves < nitems) {
                const uint32_t ghi = gtab[(4 * i + 0) * 512 + tid], gti = gtab[(4 * i + 1) * 512 + tid], go = gtab[(4 * i + 2) * 512 + tid], gc = gtab[(4 * i + 3) * 512 + tid];
                const float k = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(isg) + gc);
                *reinterpret_cast<float*>(out + go) = (state_at(ghi) * k) * (state_at(gti) * k);
            }
        for (int idx = tid + 2 * 64 * kHLWaves; idx < nitems; idx += 64 * kHLWaves) {      // dd > 16
            const int cl = idx / p.dd, x = idx - cl * p.dd;
            const int64_t io = b * p.idx_bs + static_cast<int64_t>(c0) * p.dd + idx;
            const int th = static_cast<int>(p.head[io]), tt = static_cast<int>(p.tail[io]);
            const float k = isg[cl];
            reinterpret_cast<float*>(out)[cl * Ldd + x] = (state_at(hl_pos4(cl, th & ~3) + 2 * (th & 3)) * k) * (state_at(hl_pos4(cl, tt & ~3) + 2 * (tt & 3)) * k);
        }
        }
        if (BWD ? q.ysave[l] != nullptr : p.hsave != nullptr) {                                           // wave w: channels w, w + 8, ...; lane = (K step, half, slot), two passes of 256 columns
            char* hs = BWD ? reinterpret_cast<char*>(q.ysave[l] + (static_cast<int64_t>(b) * C + c0) * S)
                           : reinterpret_cast<char*>(p.hsave + ((static_cast<int64_t>(l) * p.B + b) * C + c0) * S);
            for (int cl = wave; cl < nch; cl += kHLWaves) {
                const float k = isg[cl];
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int ts = 256 * ps + 32 * (lane >> 3) + 16 * ((lane >> 2) & 1) + 4 * (lane & 3);
                    if (ts < S) {
                        const int off = hl_pos4(cl, ts);
                        const uint2 hi = *reinterpret_cast<const uint2*>(Hs + off), lo = *reinterpret_cast<const uint2*>(Hs + PLANE + off);
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        const h2 a0 = __builtin_bit_cast(h2, hi.x), a1 = __builtin_bit_cast(h2, hi.y), b0 = __builtin_bit_cast(h2, lo.x), b1 = __builtin_bit_cast(h2, lo.y);
                        float4 v;
                        v.x = (static_cast<float>(a0[0]) + static_cast<float>(b0[0])) * k; v.y = (static_cast<float>(a0[1]) + static_cast<float>(b0[1])) * k;
                        v.z = (static_cast<float>(a1[0]) + static_cast<float>(b1[0])) * k; v.w = (static_cast<float>(a1[1]) + static_cast<float>(b1[1])) * k;
                        *reinterpret_cast<float4*>(hs + static_cast<uint32_t>(cl * S + ts) * 4u) = v;
                    }
                }
            }
        }
        // the next hop's post-compute barrier orders these reads before its writes
    }
}

// ---------------------------------------------------------------------------------------------------------------- d A_l products
// d A_l[s][t] = sum_c Y_l[c][s] H^{l-1}[c][t] per graph, K = channels.  The chain kernel left Y_l^T as fragment planes UNDER ITS CHANNEL SCALES
// (Yhat[c][s] = sigma_c Y[c][s], sigma_c a power of two) — a scale on the contraction index, which the other operand absorbs exactly:
//     d A_l[s][t] = sum_c Yhat[c][s] (H[c][t] / sigma_c).
// Workgroup = (graph, 64 columns t): Htilde = H / sigma as a two-term image [64 t][K = up to 512 channels] in LDS under per-column scales
// (max over ALL channels, so the passes over K share one unit and the accumulators run through), Yhat^T fragments straight from global memory
// into registers two K steps ahead — the forward's loop with the roles of state and adjacency taken by H and Y.  K > 512 (n = 32: 992
// channels): the image is restaged between the passes.
struct PropGadj {
    const unsigned char* yplanes;       // [G][RTT][NKC][2][1 KiB]
    const float* yisg;                  // [G][32 NKC] inverse channel scales of Y_l
    const float* Hprev; int64_t h_bs;   // H^{l-1} [G][C][S] (h_bs = 0: one h^0 for all graphs)
    float* out;                         // d A_l [G][S][S]
    float* gtrans;                      // BLOCK MODE (gdiag != null): d T_l [G][C][256] in the transition tensor's layout, or null
    float* gdiag;                       // block mode: the diagonal blocks of d A_l, [G][n][256] (summed into d identity by the caller)
    int32_t G, C, S, NKC, npass, nchunks;
};

template <int RT, int NKS>
__global__ void __launch_bounds__(64 * kHLWaves) k_prop_gadj_hl(const PropGadj q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int PLANE = NKS * kSTEP, KP = NKS * 32, RTT = 8 * RT;
    unsigned char* Hs = sm;
    uint32_t* chmax = reinterpret_cast<uint32_t*>(sm + 2 * PLANE);      // [kCH]
    float* isg_t = reinterpret_cast<float*>(chmax + kCH);               // [kCH] inverse column scales
    float* fy = isg_t + kCH;                                            // [KP] inverse channel scales of Y for the current pass
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int wg = blockIdx.x, xcd = wg & 7, rr = wg >> 3;
    const int chunk = rr % q.nchunks, gs = (rr / q.nchunks) * 8 + xcd;
    if (gs >= q.G) return;
    const int S = q.S, C = q.C, NKC = q.NKC, t0 = chunk * kCH;

    // ---- Y fragments: ring of two K steps, as in the forward
    u32x4 ring[2][RT][2];
    uint32_t voff_rt[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) voff_rt[r] = static_cast<uint32_t>((wave * RT + r) * NKC) * 2048u + lane * 16u;
    const uint32_t ybytes = static_cast<uint32_t>(RTT) * NKC * 2048u;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(q.yplanes) + static_cast<int64_t>(gs) * ybytes, 0, static_cast<int>(ybytes), 0x00020000);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            ring[sl][r][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, voff_rt[r], sl * 2048, 0);
            ring[sl][r][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, voff_rt[r] + 1024u, sl * 2048, 0);
        }

    auto store_state4 = [&](int off, float v0, float v1, float v2, float v3) {
        uint32_t h0, l0, h1, l1;
        hx2_split2(v0, v1, h0, l0);
        hx2_split2(v2, v3, h1, l1);
        *reinterpret_cast<uint2*>(Hs + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(Hs + PLANE + off) = make_uint2(l0, l1);
    };

    // ---- H^{l-1}[c][t0 .. t0 + 63] of this graph: lane (li, lq) takes columns 16 jj + li and channels 4 lq .. + 3 of a 16-channel group
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.Hprev + gs * q.h_bs), 0, C * S * 4, 0x00020000);
    const float* yi = q.yisg + static_cast<int64_t>(gs) * 32 * NKC;
    if (tid < kCH) chmax[tid] = 0u;
    lds_barrier();
    {   // column maxima of Htilde over all channels, 16-byte loads: lane (tq, cq) takes columns t0 + 4 tq .. + 3 of channel 4 (wave + 8 it) + cq,
        // eight channels in flight per lane
        float m4[4] = {0.f, 0.f, 0.f, 0.f};
        const int tq = lane & 15, cq = lane >> 4, tt = t0 + 4 * tq;
        constexpr int U = 8;
        for (int c4 = 4 * wave; c4 < C; c4 += 4 * kHLWaves * U) {
            u32x4 v[U];
            float f[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c4 + 4 * kHLWaves * u + cq;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_h, (c < C && tt < S) ? static_cast<uint32_t>(c * S + tt) * 4u : kOOB, 0, 0);
                f[u] = c < C ? yi[c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                m4[0] = fmaxf(m4[0], fabsf(as_f(v[u].x) * f[u])); m4[1] = fmaxf(m4[1], fabsf(as_f(v[u].y) * f[u]));
                m4[2] = fmaxf(m4[2], fabsf(as_f(v[u].z) * f[u])); m4[3] = fmaxf(m4[3], fabsf(as_f(v[u].w) * f[u]));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mm = rows_max(m4[i]);
            if (cq == 0) atomicMax(chmax + 4 * tq + i, __builtin_bit_cast(uint32_t, mm));
        }
    }
    lds_barrier();
    float sg_t[4];                                                      // scales of this lane's four columns
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) sg_t[jj] = hx2_scale_of(__builtin_bit_cast(float, chmax[16 * jj + li]));
    if (tid < kCH) isg_t[tid] = hx2_inv(hx2_scale_of(__builtin_bit_cast(float, chmax[tid])));

    f32x4 acc[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int swz = ((li >> 1) & 3) << 4;
    const int b_rd = li * 64 + ((lq << 4) ^ swz);

#pragma unroll 1
    for (int pass = 0; pass < q.npass; ++pass) {
        const int c_pass = pass * KP;
        lds_barrier();                                                  // the previous pass's reads of the image are done (pass 0: isg_t visible)
        static_assert(KP <= 64 * kHLWaves && (2 * NKS) % kHLWaves == 0, "one table element per thread; whole rounds of 16-channel groups");
        if (tid < KP) fy[tid] = c_pass + tid < C ? yi[c_pass + tid] : 0.f;
        lds_barrier();
        // stage: wave w converts the 16-channel groups w, w + 8, ... of this pass.  Straight-line (compile-time trip count): with a run-time loop
        // between the ring's requests and the K loop the compiler cannot count what is outstanding and drains (s_waitcnt vmcnt(0)) at every K step
#pragma unroll
        for (int it = 0; it < 2 * NKS / kHLWaves; ++it) {
            const int cg = wave + kHLWaves * it;
            const int cl = 16 * cg + 4 * lq, cb = c_pass + cl;
            const float4 f4 = *reinterpret_cast<const float4*>(fy + cl);
            const float f[4] = {f4.x, f4.y, f4.z, f4.w};
            float v[4][4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = t0 + 16 * jj + li;
                    v[jj][i] = as_f(__builtin_amdgcn_raw_buffer_load_b32(rs_h, (cb + i < C && t < S) ? static_cast<uint32_t>((cb + i) * S + t) * 4u : kOOB, 0, 0));
                }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                store_state4(hl_pos4(16 * jj + li, cl), v[jj][0] * (f[0] * sg_t[jj]), v[jj][1] * (f[1] * sg_t[jj]), v[jj][2] * (f[2] * sg_t[jj]),
                             v[jj][3] * (f[3] * sg_t[jj]));
        }
        lds_barrier();
        const int kg0 = pass * NKS;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int BUF = ks & 1;
            const int kg = kg0 + ks + 2;                                // the K step requested now (zeros past the last one)
            const uint32_t so = static_cast<uint32_t>(kg) * 2048u;
            const uint32_t oob = kg < NKC ? 0u : kOOB;
            const unsigned char* hb = Hs + ks * kSTEP + b_rd;
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(hb + 1024 * j);
                bl[j] = *reinterpret_cast<const f16x8*>(hb + PLANE + 1024 * j);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f16x8 ah = __builtin_bit_cast(f16x8, ring[BUF][r][0]), al = __builtin_bit_cast(f16x8, ring[BUF][r][1]);
                if (r == 0) __builtin_amdgcn_s_setprio(2);              // (as in the forward: -2 %)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], acc[r][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], acc[r][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc[r][j], 0, 0, 0);
                // aux bit 31 = volatile: these requests stay WHERE THEY ARE WRITTEN.  As plain read-only loads the compiler sank them one K step down,
                // next to their uses (nothing in this loop writes memory, so nothing stopped it): every fragment was then awaited right behind its
                // request (s_waitcnt vmcnt(1) / vmcnt(0) throughout, RT = 2 and 4; the RT = 3 instances kept their place by luck of a heuristic)
                ring[BUF][r][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, voff_rt[r] | oob, so, kVolatileLoad);
                ring[BUF][r][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (voff_rt[r] + 1024u) | oob, so, kVolatileLoad);
                __builtin_amdgcn_sched_barrier(0x0787);                 // MFMA and VMEM stay on their side: at 255 registers the scheduler otherwise runs each row tile's
                                                                        // accumulator chain depth-first to free the ring early — every fragment is then awaited right behind its request
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // ---- d A_l[s][t]: C layout — column (lane & 15) = t0 + 16 j + li, rows s = 16 (wave RT + r) + 4 lq + i; 64-byte runs per row
    if (q.gdiag) {
        // block mode (S = 16 n): row tile = node i, column tile = node j; block (i, j) goes to T's layout [e(i, j)][row][column], the diagonal
        // blocks to their own buffer
        const int nn = S >> 4;
        float* gt = q.gtrans ? q.gtrans + static_cast<int64_t>(gs) * C * 256 : nullptr;
        float* gd = q.gdiag + static_cast<int64_t>(gs) * nn * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float k = isg_t[16 * j + li];
            const int jn = (t0 >> 4) + j;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int in_ = wave * RT + r;
                if (in_ >= nn || jn >= nn) continue;                    // wave-uniform
                float* dst = in_ == jn ? gd + in_ * 256 : (gt ? gt + static_cast<int64_t>(in_ * (nn - 1) + (jn < in_ ? jn : jn - 1)) * 256 : nullptr);
                if (!dst) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[(4 * lq + i) * 16 + li] = acc[r][j][i] * k;
            }
        }
        return;
    }
    float* out = q.out + static_cast<int64_t>(gs) * S * S;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, S * S * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float k = isg_t[16 * j + li];
        const int t = t0 + 16 * j + li;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s_ = 16 * (wave * RT + r) + 4 * lq + i;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, acc[r][j][i] * k), rs_o,
                                                      (s_ < S && t < S) ? static_cast<uint32_t>(s_ * S + t) * 4u : kOOB, 0, 0);
            }
    }
}

struct HLGeom { int NKS, RT; size_t per_graph_split, per_graph_alpha, lds; };
HLGeom hl_geom(int S, int L) {
    HLGeom g;
    g.NKS = ((S + 31) / 32 + 1) & ~1;                                   // even: the ring alternates between two slots
    g.RT = (S + 127) / 128;
    g.per_graph_split = static_cast<size_t>(L) * 8 * g.RT * g.NKS * 2048;
    g.per_graph_alpha = static_cast<size_t>(L) * 128 * g.RT * sizeof(float);
    g.lds = 2ull * g.NKS * kSTEP + 3ull * kCH * sizeof(uint32_t) + static_cast<size_t>(L) * 128 * g.RT * sizeof(float) + 8ull * 512 * sizeof(uint32_t);
    return g;
}
constexpr int kHLSlice = 256;           // graphs per split pass (bounds the workspace: 0.8 GB at S = 512, L = 3)

}  // namespace

size_t prop_hl_ws_bytes(int B, int S, int L) {
    if (S <= 160 || S > 512 || B <= 0 || L <= 0) return 0;
    const HLGeom g = hl_geom(S, L);
    if (g.lds > 160 * 1024) return 0;
    const int G = B < kHLSlice ? B : kHLSlice;
    return (g.per_graph_split + g.per_graph_alpha) * static_cast<size_t>(G) + 256;
}

bool prop_fwd_hl_supported(const PropK& p) {
    if (p.S <= 160 || p.S > 512 || (p.S % 4) != 0 || p.dd < 1 || p.L < 1 || !p.ws) return false;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (p.identity && (p.dd != 16 || (p.S % 16) != 0 || p.C != (p.S / 16) * (p.S / 16 - 1) || !al16(p.identity))) return false;
    for (int l = 0; l < p.L; ++l) if (!al16(p.identity ? p.trans[l] : p.adj[l])) return false;
    if (!al16(p.h0) || (p.h0_bs % 4) != 0 || (p.hsave && !al16(p.hsave)) || !al16(p.ws)) return false;
    if (static_cast<int64_t>(p.C) * p.S * 4 >= (1LL << 31)) return false;
    const HLGeom g = hl_geom(p.S, p.L);
    if (g.lds > 160 * 1024) return false;                               // 8 hops at S = 512: the row-scale table no longer fits beside the state
    return p.ws_bytes >= static_cast<int64_t>(g.per_graph_split + g.per_graph_alpha + 256);
}

int prop_fwd_hl(const PropK& p, hipStream_t st) {
    if (!prop_fwd_hl_supported(p)) return RECON_ERR_UNSUPPORTED;
    const HLGeom g = hl_geom(p.S, p.L);
    int64_t G = (p.ws_bytes - 256) / static_cast<int64_t>(g.per_graph_split + g.per_graph_alpha);
    if (G > p.B) G = p.B;
    if (G > kHLSlice) G = kHLSlice;
    PropHL q;
    q.p = p;
    q.split = static_cast<unsigned char*>(p.ws);
    q.alpha = reinterpret_cast<float*>(q.split + ((g.per_graph_split * static_cast<size_t>(G) + 255) & ~static_cast<size_t>(255)));
    q.NKS = g.NKS; q.RT = g.RT; q.nchunks = (p.C + kCH - 1) / kCH;
    for (int64_t g0 = 0; g0 < p.B; g0 += G) {
        q.g0 = static_cast<int32_t>(g0);
        q.G = static_cast<int32_t>(p.B - g0 < G ? p.B - g0 : G);
        const int64_t units = static_cast<int64_t>(p.L) * q.G * 8 * g.RT;
        const dim3 sgrid(static_cast<unsigned>((units + 3) / 4));
        if (p.identity) hipLaunchKernelGGL(k_prop_split_adj<true>, sgrid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL(k_prop_split_adj<false>, sgrid, dim3(256), 0, st, q);
        const dim3 grid(static_cast<unsigned>(((q.G + 7) / 8) * 8 * q.nchunks));
#define CALL_HL(R_, K_)                                                                                                                 \
    do {                                                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_fwd_hl<R_, K_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  static_cast<int>(g.lds));                                                                             \
        hipLaunchKernelGGL((k_propagate_fwd_hl<R_, K_>), grid, dim3(64 * kHLWaves), g.lds, st, q);                                      \
    } while (0)
        switch (g.NKS) {                                                // RT = rows / 128 follows from the K steps
            case 6: CALL_HL(2, 6); break; case 8: CALL_HL(2, 8); break; case 10: CALL_HL(3, 10); break; case 12: CALL_HL(3, 12); break;
            case 14: CALL_HL(4, 14); break; default: CALL_HL(4, 16); break;
        }
#undef CALL_HL
    }
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// ---------------------------------------------------------------------------------------------------------------- backward chain
// d loss / d H^{l-1} = A_l^T Y_l is a propagation with the transposed adjacencies, Y_{l-1} = (that + relation gradient) . act'(H^{l-1}) its
// "activation": the forward's kernel with another epilogue (k_propagate_fwd_hl<.., true>) runs all L steps of a slice of graphs in one
// launch, the Y_l (operands of the d A_l products) leaving as fp32 like the forward's saved states.
int64_t prop_bwd_hl_slice(int C, int S, int L, int64_t ws_bytes, int B) {
    if (S <= 160 || S > 512 || (S % 16) != 0 || L < 1 || L > kMaxHops || B <= 0) return 0;
    const HLGeom g = hl_geom(S, L);
    if (g.lds > 160 * 1024 || static_cast<int64_t>(C) * S * 4 >= (1LL << 31)) return 0;
    int64_t G = (ws_bytes - 256) / static_cast<int64_t>(g.per_graph_split + g.per_graph_alpha);
    if (G > B) G = B;
    if (G > kHLSlice) G = kHLSlice;
    return G < 0 ? 0 : G;
}

// chain workspace of a slice of G graphs, in floats: Y_L as fp32 [G][C][S] | L plane sets [G][RTT][NKC][2][256 floats] | L x [G][32 NKC] scales
static int hl_nkc(int C) { return 2 * ((C + kCH - 1) / kCH); }
size_t prop_bwd_hl_ws_floats(int C, int S, int L, int64_t G) {
    const HLGeom g = hl_geom(S, L);
    const size_t planes = static_cast<size_t>(8) * g.RT * hl_nkc(C) * 512;
    return static_cast<size_t>(G) * (static_cast<size_t>(C) * S + static_cast<size_t>(L) * (planes + 32ull * hl_nkc(C)));
}
void prop_bwd_hl_ws_layout(int C, int S, int L, int64_t G, float* ws, float** y_in, unsigned char** planes, float** isg, size_t* plane_set_bytes, size_t* isg_set_floats) {
    const HLGeom g = hl_geom(S, L);
    *plane_set_bytes = static_cast<size_t>(G) * 8 * g.RT * hl_nkc(C) * 2048;
    *isg_set_floats = static_cast<size_t>(G) * 32 * hl_nkc(C);
    *y_in = ws;
    *planes = reinterpret_cast<unsigned char*>(ws + static_cast<size_t>(G) * C * S);
    *isg = reinterpret_cast<float*>(*planes + static_cast<size_t>(L) * *plane_set_bytes);
}

int prop_bwd_hl_gadj(const unsigned char* yplanes, const float* yisg, const float* Hprev, int64_t h_bs, float* out, float* gtrans, float* gdiag, int G,
                     int C, int S, hipStream_t st) {
    const HLGeom g = hl_geom(S, 1);
    PropGadj q{};
    q.yplanes = yplanes; q.yisg = yisg; q.Hprev = Hprev; q.h_bs = h_bs; q.out = out; q.gtrans = gtrans; q.gdiag = gdiag;
    q.G = G; q.C = C; q.S = S; q.NKC = hl_nkc(C); q.nchunks = (S + kCH - 1) / kCH;
    const int NKS = q.NKC <= 8 ? 8 : 16;
    q.npass = (q.NKC + NKS - 1) / NKS;
    if (static_cast<int64_t>(C) * S * 4 >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;
    const size_t lds = 2ull * NKS * kSTEP + 2ull * kCH * sizeof(uint32_t) + static_cast<size_t>(NKS) * 32 * sizeof(float);
    const dim3 grid(static_cast<unsigned>(((G + 7) / 8) * 8 * q.nchunks));
#define CALL_GA(R_, K_)                                                                                                             \
    do {                                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prop_gadj_hl<R_, K_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  static_cast<int>(lds));                                                                           \
        hipLaunchKernelGGL((k_prop_gadj_hl<R_, K_>), grid, dim3(64 * kHLWaves), lds, st, q);                                        \
    } while (0)
    if (NKS == 8) { if (g.RT == 2) CALL_GA(2, 8); else if (g.RT == 3) CALL_GA(3, 8); else CALL_GA(4, 8); }
    else { if (g.RT == 2) CALL_GA(2, 16); else if (g.RT == 3) CALL_GA(3, 16); else CALL_GA(4, 16); }
#undef CALL_GA
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

int prop_bwd_hl_chain(const PropBwdHL& a, hipStream_t st) {
    const int64_t G = prop_bwd_hl_slice(a.C, a.S, a.L, a.ws_bytes, a.G);
    if (G < a.G || a.dd != 16 || !a.hblk || !a.tblk) return RECON_ERR_UNSUPPORTED;
    const HLGeom g = hl_geom(a.S, a.L);
    PropHL q{};
    q.p.B = a.G; q.p.C = a.C; q.p.S = a.S; q.p.L = a.L; q.p.dd = a.dd; q.p.act = a.act;
    q.p.h0 = a.y_in; q.p.h0_bs = static_cast<int64_t>(a.C) * a.S;
    for (int k = 0; k < a.L; ++k) { q.p.adj[k] = a.adj_step[k]; q.hmask[k] = a.hmask[k]; q.ysave[k] = a.ysave[k]; q.gout_off[k] = a.gout_off[k]; }
    q.gout = a.gout; q.gout_ld = a.L * a.dd; q.hblk = a.hblk; q.tblk = a.tblk;
    q.yplanes = a.yplanes; q.yisg = a.yisg; q.NKC = hl_nkc(a.C); q.hlast = a.hlast;
    if (a.hlast) { q.p.h0 = a.hlast; q.p.h0_bs = 0; }                   // (unused: the prologue forms Y_L itself)
    q.split = static_cast<unsigned char*>(a.ws);
    q.alpha = reinterpret_cast<float*>(q.split + ((g.per_graph_split * static_cast<size_t>(a.G) + 255) & ~static_cast<size_t>(255)));
    q.NKS = g.NKS; q.RT = g.RT; q.nchunks = (a.C + kCH - 1) / kCH; q.g0 = 0; q.G = a.G;
    const int64_t units = static_cast<int64_t>(a.L) * q.G * 8 * g.RT;
    if (a.identity) {
        for (int k = 0; k < a.L; ++k) q.p.trans[k] = a.adj_step[k];    // block mode: the steps' transition tensors [G][C][256]
        q.p.identity = a.identity;
        hipLaunchKernelGGL((k_prop_split_adj<true, true>), dim3(static_cast<unsigned>((units + 3) / 4)), dim3(256), 0, st, q);
    } else {
        hipLaunchKernelGGL((k_prop_split_adj<false, true>), dim3(static_cast<unsigned>((units + 3) / 4)), dim3(256), 0, st, q);
    }
    const dim3 grid(static_cast<unsigned>(((q.G + 7) / 8) * 8 * q.nchunks));
#define CALL_HLB(R_, K_)                                                                                                                      \
    do {                                                                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_fwd_hl<R_, K_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  static_cast<int>(g.lds));                                                                                   \
        hipLaunchKernelGGL((k_propagate_fwd_hl<R_, K_, true>), grid, dim3(64 * kHLWaves), g.lds, st, q);                                      \
    } while (0)
    switch (g.NKS) {
        case 6: CALL_HLB(2, 6); break; case 8: CALL_HLB(2, 8); break; case 10: CALL_HLB(3, 10); break; case 12: CALL_HLB(3, 12); break;
        case 14: CALL_HLB(4, 14); break; default: CALL_HLB(4, 16); break;
    }
#undef CALL_HLB
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace recon
