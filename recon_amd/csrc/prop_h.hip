// K5'' — GP-GNN propagation (models/models.py:260-274 and its copies :470-485, :680-694, :918-932) on the f16 matrix cores with
// TWO-TERM operands: fp32-class accuracy at 3 x v_mfma_f32_16x16x32_f16 per 16x16x32 block (the scheme of gemm_hx2.hip), which
// takes the matrix-pipe time of cfg 3b from 58 us (fp32 MFMA) to ~11 us and leaves the kernel bound by the ONE pass over the
// adjacency stack it has to make (L * B * S^2 * 4 bytes).
//
// Per graph b and hop l:  Hnew^T [S x C] = A_l [S x S] . H^T [S x C]      (M = s, N = channel, K = t)
//   * one PERSISTENT workgroup per CU walks graphs b = blockIdx.x, + gridDim.x, ...; wave w owns the 16 rows s = 16 w .. 16 w + 15
//     of every A_l;
//   * A_l is read exactly once from HBM: every lane fetches its own MFMA A-fragments for the WHOLE hop (row s = lane & 15,
//     8 NKS columns) as 16-byte loads into registers — one hop ahead of their use, ACROSS hop and graph boundaries, so a CU
//     always has one full A_l (83 KB at cfg 3b) in flight.  Holding a row's whole hop lets the wave find the row's max magnitude
//     in registers: every row of A gets its own power-of-two scale (s.amax in [2^14, 2^15), half's 5 exponent bits), the scaled
//     row is written as x0 + x1 (11 + 11 significant bits);
//   * the channel states live in LDS as two half planes [K step][channel][32 t] under a per-channel power-of-two scale (the
//     channel's max magnitude is gathered across the waves by LDS atomics in the hop's epilogue), plus an fp32 copy for the
//     head (.) tail gather and the saved states;  a0 b1 + a1 b0 + a0 b0 accumulate in fp32;
//   * the contraction index is permuted inside a K step (lane group q holds t = 4 q .. 4 q + 3 and 16 + 4 q .. + 3), identically in
//     both operands, so that one load instruction covers 64 contiguous bytes per row;
//   * workgroup barriers are raw s_barrier + lgkmcnt(0): __syncthreads() would also drain the prefetch (vmcnt(0)).
// No load other than the prefetch is issued inside the hop loop (the gather indices are fetched once per graph): a load's use
// would wait for everything issued before it, i.e. for the prefetch.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "prop_common.h"
#include "prop_h_util.h"

namespace recon {
namespace {

// s_memtime stamps for tools/probe/prop_h_stamps.hip (compiled out of the library)
#ifdef RECON_PROP_STAMPS
__device__ unsigned long long* g_stamps;
#define STAMP(slot)                                                                                                   \
    do {                                                                                                              \
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 77) && (wave == 0 || wave == NW - 1) && gi < 4) {          \
            g_stamps[((((blockIdx.x ? 1 : 0) * 2 + (wave ? 1 : 0)) * 4 + gi) * 8 + hop_i) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
        }                                                                                                             \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

constexpr int kGatherRegs = 2;                    // gather items per thread whose indices stay in registers (cfg 3b: exactly 2)

// NKS = K steps of 32 (S = 32 NKS or 32 NKS - 16), NTC = channel tiles of 16 (C <= 16 NTC); blockDim.x = 4 S (one wave per 16 rows of A).
// Dynamic LDS: half planes [2][NKS][16 NTC][64 B] | channel maxima [2][16 NTC] | inverse channel scales [16 NTC] | inverse row scales [waves][16]
//
// State image.  Element (channel c, column t) of plane q lives at byte
//     q PLANE + (t >> 5) STEP + 64 c + 16 (((t >> 2) & 3) ^ ((c >> 1) & 3)) + 8 ((t >> 4) & 1) + 2 (t & 3):
// one 64-byte row per (K step, channel); its four 16-byte slots are the B fragments of the four lane groups (slot q holds t = 4 q .. 4 q + 3
// and 16 + 4 q .. + 3 of the K step), XOR-rotated by the channel so that the ds_read_b128 of a fragment is conflict free and the 8-byte
// writes of the epilogue (16 channels x 4 columns per lane group) meet two to a bank pair instead of four.
template <int NKS, int NTC, bool BLK>
__global__ void __launch_bounds__(128 * NKS) k_propagate_fwd_h(const PropK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int CH = NTC * 16;
    constexpr int STEP = CH * 64;                 // bytes of one K step of one plane
    constexpr int PLANE = NKS * STEP;
    constexpr int KP = NKS * 32;
    constexpr int NCW = (CH + 2 * NKS - 2) / (2 * NKS - 1);      // channels a wave stages (at least 2 NKS - 1 waves)
    const int S = p.S, C = p.C;
    unsigned char* Hs = sm;
    uint32_t* chmax = reinterpret_cast<uint32_t*>(sm + 2 * PLANE);
    float* isg = reinterpret_cast<float*>(chmax + 2 * CH);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = nthreads >> 6;
    float* atab = isg + CH + 16 * wave;           // this wave's 16 inverse row scales
    uint32_t* gstat = reinterpret_cast<uint32_t*>(isg + CH + 16 * NW);      // [2 kMaxHops + 1] max magnitudes of this graph (for the backward)
    const int li = lane & 15, lq = lane >> 4;
    const uint32_t SSb = static_cast<uint32_t>(S) * S * 4;                      // bytes of one A_l of one graph
    // A fragments: k slots 0..3 = t 32 ks + 4 lq .., slots 4..7 = t 32 ks + 16 + 4 lq ..  One buffer descriptor per (graph, hop), one
    // lane offset for all ten loads (the rest is the instruction's immediate); the half K step past S is requested out of range.
    const uint32_t voff_a = (static_cast<uint32_t>(16 * wave + li) * S + 4 * lq) * 4;
    const uint32_t voff_tail = (S & 16) ? kOOB : voff_a + (KP - 16) * 4;
    auto rsrc_a = [&](int l, int bb) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.adj[l]) + static_cast<int64_t>(bb) * SSb), 0,
                                                 static_cast<int>(SSb), 0x00020000);
    };
    // BLOCK MODE (p.identity, dd == 16: this wave's 16 rows are node i = wave, a K step covers nodes j = 2 ks and 2 ks + 1): the fragment of
    // block (i, j) is rows r = li, columns 4 lq .. of trans[l][b, e(i, j)] — one contiguous KiB per load instruction — or of `identity` when
    // j == i; e(i, j) = i (n - 1) + (j < i ? j : j - 1).  Everything but the lane offset is wave-uniform: descriptor, soffset and the
    // choice of the identity are scalar selects.
    constexpr bool blocks = BLK;
    const int nn = S >> 4;                                               // nodes (block mode)
    const uint32_t voff_blk = static_cast<uint32_t>(li * 16 + 4 * lq) * 4u;
    const auto rs_ident = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blocks ? p.identity : p.h0), 0, 1024, 0x00020000);
    auto rsrc_t = [&](int l, int bb) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.trans[l] + static_cast<int64_t>(bb) * C * 256), 0, C * 1024, 0x00020000);
    };
    auto load_rows = [&](u32x4 (&raw)[NKS][2], int ks, int l, int bb, decltype(rs_ident) rs_adj) {      // the two loads of K step ks
        if constexpr (!blocks) {
            raw[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, voff_a + 128 * ks, 0, 0);
            raw[ks][1] = ks == NKS - 1 ? __builtin_amdgcn_raw_buffer_load_b128(rs_adj, voff_tail, 0, 0)
                                       : __builtin_amdgcn_raw_buffer_load_b128(rs_adj, voff_a + 128 * ks + 64, 0, 0);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * ks + h;
                const bool diag = j == wave;
                const int e = wave * (nn - 1) + (j < wave ? j : j - 1);
                const auto rs = diag ? rs_ident : rs_adj;
                raw[ks][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, j < nn ? voff_blk : kOOB, diag ? 0 : e * 1024, 0);
            }
        }
    };
    // two half terms of four consecutive (scaled) columns -> 8 bytes of each plane
    auto store_state4 = [&](int off, float v0, float v1, float v2, float v3) {
        uint32_t h0, l0, h1, l1;
        hx2_split2(v0, v1, h0, l0);
        hx2_split2(v2, v3, h1, l1);
        *reinterpret_cast<uint2*>(Hs + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(Hs + PLANE + off) = make_uint2(l0, l1);
    };
    auto col_off = [](int t0) { return (t0 >> 5) * STEP + (((t0 >> 2) & 3) << 4) + (((t0 >> 4) & 1) << 3); };      // XOR ((c >> 1) & 3) << 4, + 64 c
    const int swz = ((li >> 1) & 3) << 4;                               // channel 16 j + li: the same for every j
    const int b_rd = li * 64 + ((lq << 4) ^ swz);                       // B fragment: + 1024 j + STEP ks (+ PLANE for the low terms)
    const int t0w = 16 * wave + 4 * lq;                                 // this lane's four state columns in the C layout
    const int so_w = (col_off(t0w) ^ swz) + 64 * li;                    // + 1024 j
    const int nitems = C * p.dd, Ldd = p.L * p.dd;
    const bool homog = p.act != RECON_ACT_TANH;                         // act(k v) = k act(v) for k > 0: the old channel scale stays on
    const bool relu = p.act == RECON_ACT_RELU;

    // h^0 of graph bb: this wave's NCW channels, one 16-byte piece per lane
    const uint32_t vo_h = 4 * lane < S ? 16u * lane : kOOB;
    u32x4 raw[NKS][2];
    int b = blockIdx.x;
    if (b < p.B) {
        const auto rs = blocks ? rsrc_t(0, b) : rsrc_a(0, b);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) load_rows(raw, ks, 0, b, rs);
    }
    int gi = -1, hop_i = 0;
    (void)gi; (void)hop_i;
#pragma unroll 1
    for (; b < p.B; b += gridDim.x) {
        ++gi; hop_i = 7;
        STAMP(0);
        // ---- h^0: wave w stages channels w, w + NW, ... (a whole channel per wave: its max magnitude is a wave reduction); columns past S
        // and channels past C come back as zeros (out-of-range offsets).  (Requesting the next graph's h^0 during the last hop was
        // measured: 89 -> 95 us — it competes with the adjacency prefetch for the CU's ~12 B/clk share of HBM.)
        float mh0 = 0.f;                                                           // max |h^0| over this wave's channels
        {
            const int t0 = 4 * lane;
            const int so = col_off(t0);
            if (wave == 0) {
#pragma unroll
                for (int k = 0; k < (2 * CH + 63) / 64; ++k)
                    if (lane + 64 * k < 2 * CH) chmax[lane + 64 * k] = 0u;
                if (lane < 2 * kMaxHops + 1) gstat[lane] = 0u;
            }
            // in batches of at most 12 channels per wave (cfg 3b: 9, one batch): narrow states with many channels (S <= 32: one or two waves for up
            // to 96 channels) would otherwise hold every channel's piece at once — 384 registers
            constexpr int NB = NCW < 12 ? NCW : 12;
#pragma unroll 1
            for (int i0 = 0; i0 < NCW; i0 += NB) {
                u32x4 hv[NB];
                {
                    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h0 + b * p.h0_bs), 0, C * S * 4, 0x00020000);
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const int c = wave + (i0 + i) * NW;
                        hv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (c < C && i0 + i < NCW) ? vo_h + static_cast<uint32_t>(c * S * 4) : kOOB, 0, 0);
                    }
                }
                float mx[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i)                                        // independent reduction chains
                    mx[i] = wave_max(fmaxf(fmaxf(fabsf(as_f(hv[i].x)), fabsf(as_f(hv[i].y))), fmaxf(fabsf(as_f(hv[i].z)), fabsf(as_f(hv[i].w)))));
#pragma unroll
                for (int i = 0; i < NB; ++i) mh0 = fmaxf(mh0, mx[i]);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int c = wave + (i0 + i) * NW;                             // wave-uniform
                    if (c < CH && i0 + i < NCW) {
                        const float sg = hx2_scale_of(mx[i]);
                        if (t0 < KP) store_state4((so ^ (((c >> 1) & 3) << 4)) + 64 * c, as_f(hv[i].x) * sg, as_f(hv[i].y) * sg, as_f(hv[i].z) * sg, as_f(hv[i].w) * sg);
                        if (lane == 0) isg[c] = hx2_inv(sg);
                    }
                }
            }
        }
        // ---- gather items of this thread (the same in every hop): byte positions of head / tail in plane 0, of the channel's inverse
        // scale and of the result in `out`
        uint32_t g_hi[kGatherRegs], g_ti[kGatherRegs], g_o[kGatherRegs], g_c[kGatherRegs];
        {
            const int64_t* hd = p.head + b * p.idx_bs;
            const int64_t* tl = p.tail + b * p.idx_bs;
            auto pos = [&](uint32_t c, uint32_t t) {
                return (t >> 5) * STEP + 64 * c + (((((t >> 2) & 3) ^ ((c >> 1) & 3))) << 4) + (((t >> 4) & 1) << 3) + 2 * (t & 3);
            };
#pragma unroll
            for (int i = 0; i < kGatherRegs; ++i) {
                const uint32_t idx = min(tid + i * nthreads, nitems - 1);
                const uint32_t c = idx / static_cast<uint32_t>(p.dd);
                g_hi[i] = pos(c, static_cast<uint32_t>(hd[idx])); g_ti[i] = pos(c, static_cast<uint32_t>(tl[idx]));
                g_o[i] = 4u * (idx + c * (Ldd - p.dd));                         // c L dd + x
                g_c[i] = 4u * c;
            }
        }
        STAMP(1);
        lds_barrier();
        STAMP(2);
        if (p.stats && lane == 0) atomicMax(gstat, __builtin_bit_cast(uint32_t, mh0));       // gstat was cleared before the barrier
        float inv_sig[NTC];
#pragma unroll
        for (int j = 0; j < NTC; ++j) inv_sig[j] = isg[16 * j + li];

#pragma unroll 1
        for (int l = 0; l < p.L; ++l) {
            hop_i = l;
            STAMP(0);
            // ---- this hop's rows: per-row scale, two half terms per element
            float m = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 q = raw[ks][h];
                    m = fmaxf(fmaxf(fabsf(as_f(q.x)), fabsf(as_f(q.y))), m);
                    m = fmaxf(fmaxf(fabsf(as_f(q.z)), fabsf(as_f(q.w))), m);
                }
            m = rows_max(m);                                                        // lanes li, li + 16, li + 32, li + 48 hold one row
            if (p.stats) {                                                          // max |A_l| of the graph, for the backward's scale
                float mw = dpp_max<0xB1>(m);
                mw = dpp_max<0x4E>(mw); mw = dpp_max<0x141>(mw); mw = dpp_max<0x140>(mw);
                if (lane == 0) atomicMax(gstat + p.L + 1 + l, __builtin_bit_cast(uint32_t, mw));
            }
            const float alpha = hx2_scale_of(m);
            if (lq == 0) atab[li] = hx2_inv(alpha);
            f16x8 a_hi[NKS], a_lo[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 q = raw[ks][h];
                    hx2_split2(as_f(q.x) * alpha, as_f(q.y) * alpha, hi[2 * h], lo[2 * h]);
                    hx2_split2(as_f(q.z) * alpha, as_f(q.w) * alpha, hi[2 * h + 1], lo[2 * h + 1]);
                }
                a_hi[ks] = __builtin_bit_cast(f16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
                a_lo[ks] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
            }
            STAMP(1);
            // ---- products, with the rows of the NEXT step (next hop, or hop 0 of this workgroup's next graph) requested between the K
            // steps: a CU takes ~40 cycles per 1 KiB load instruction, so ten of them in a row block the wave for as long as the
            // matrix pipe needs for the hop (s_memtime stamps: 3-4 k cycles) — interleaved they ride under the MFMAs
            const bool more_hops = l + 1 < p.L;
            const int nb = more_hops ? b : b + static_cast<int>(gridDim.x);
#ifdef RECON_PROP_NOPREFETCH
            const bool pre = false;
#else
            const bool pre = nb < p.B;
#endif
            const auto rs_n = blocks ? rsrc_t(more_hops ? l + 1 : 0, pre ? nb : b) : rsrc_a(more_hops ? l + 1 : 0, pre ? nb : b);
            f32x4 acc[NTC];
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            STAMP(2);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int j = 0; j < NTC; j += 2) {
                    f16x8 bh[2], bl[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        if (j + jj < NTC) {
                            bh[jj] = *reinterpret_cast<const f16x8*>(Hs + ks * STEP + 1024 * (j + jj) + b_rd);
                            bl[jj] = *reinterpret_cast<const f16x8*>(Hs + PLANE + ks * STEP + 1024 * (j + jj) + b_rd);
                        }
                    // small terms first; two independent accumulator chains
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) if (j + jj < NTC) acc[j + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[ks], bl[jj], acc[j + jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) if (j + jj < NTC) acc[j + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[ks], bh[jj], acc[j + jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) if (j + jj < NTC) acc[j + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[ks], bh[jj], acc[j + jj], 0, 0, 0);
                }
                if (pre) load_rows(raw, ks, 0, 0, rs_n);
                __builtin_amdgcn_sched_barrier(0x078f);                            // everything but VMEM may move across: the requests stay where they are written
            }
            STAMP(3);
            // ---- epilogue 1: row scales off, activation, channel maxima (every lane sends its own ds_max: reducing the four lanes of a channel
            // on the VALU first measured no faster).  C layout: column (lane & 15) = channel 16 j + li, rows 4 lq + r.  Values stay under the OLD channel scale
            // where the activation commutes with it (relu, linear).
            lds_wait();
            const float4 ia = *reinterpret_cast<const float4*>(atab + 4 * lq);
            uint32_t* cm = chmax + (l & 1) * CH;
            if (homog) {
#pragma unroll
                for (int j = 0; j < NTC; ++j) {
                    acc[j][0] *= ia.x; acc[j][1] *= ia.y; acc[j][2] *= ia.z; acc[j][3] *= ia.w;
                    if (relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[j][r] = fmaxf(acc[j][r], 0.f);
                    }
                    const float mm = fmaxf(fmaxf(fabsf(acc[j][0]), fabsf(acc[j][1])), fmaxf(fabsf(acc[j][2]), fabsf(acc[j][3])));
                    atomicMax(cm + 16 * j + li, __builtin_bit_cast(uint32_t, mm));
                }
            } else {
#pragma unroll
                for (int j = 0; j < NTC; ++j) {
                    acc[j][0] = tanh_fast(acc[j][0] * (ia.x * inv_sig[j])); acc[j][1] = tanh_fast(acc[j][1] * (ia.y * inv_sig[j]));
                    acc[j][2] = tanh_fast(acc[j][2] * (ia.z * inv_sig[j])); acc[j][3] = tanh_fast(acc[j][3] * (ia.w * inv_sig[j]));
                    const float mm = fmaxf(fmaxf(fabsf(acc[j][0]), fabsf(acc[j][1])), fmaxf(fabsf(acc[j][2]), fabsf(acc[j][3])));
                    atomicMax(cm + 16 * j + li, __builtin_bit_cast(uint32_t, mm));
                }
            }
                          // unconditional: a conditional definition would keep the old value live through every hop                                 // the next graph's h^0 travels under the rest of this hop
            STAMP(4);
            lds_barrier();                                                          // everybody has read H^l-1; maxima complete
            STAMP(5);
            // ---- epilogue 2: H^l under its new channel scales into the planes
            if (wave == 0) {
#pragma unroll
                for (int k = 0; k < (CH + 63) / 64; ++k)
                    if (lane + 64 * k < CH) chmax[((l + 1) & 1) * CH + lane + 64 * k] = 0u;
                if (p.stats) {                                                      // max |H^l| of the graph
                    float mh = 0.f;
#pragma unroll
                    for (int j = 0; j < NTC; ++j) mh = fmaxf(mh, __builtin_bit_cast(float, cm[16 * j + li]) * (homog ? inv_sig[j] : 1.f));
                    mh = dpp_max<0xB1>(mh); mh = dpp_max<0x4E>(mh); mh = dpp_max<0x141>(mh); mh = dpp_max<0x140>(mh);
                    if (lane == 0) gstat[l + 1] = __builtin_bit_cast(uint32_t, mh);
                }
            }
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                const float inv_u = homog ? inv_sig[j] : 1.f;                       // the unit acc[j] is in now
                const float sg = hx2_scale_of(__builtin_bit_cast(float, cm[16 * j + li]) * inv_u);
                const float f = sg * inv_u;
                inv_sig[j] = hx2_inv(sg);
                store_state4(so_w + 1024 * j, acc[j][0] * f, acc[j][1] * f, acc[j][2] * f, acc[j][3] * f);
                if (tid < 16) isg[16 * j + li] = inv_sig[j];
            }
            STAMP(6);
            lds_barrier();                                                          // H^l complete
            STAMP(7);
            // ---- relation_l = gather(h, heads) * gather(h, tails)   (models/models.py:270-273), saved state; values = (hi + lo) / scale
            auto state_at = [&](uint32_t pos) {
                return static_cast<float>(*reinterpret_cast<const _Float16*>(Hs + pos)) + static_cast<float>(*reinterpret_cast<const _Float16*>(Hs + PLANE + pos));
            };
            char* out = reinterpret_cast<char*>(p.out + (static_cast<int64_t>(b) * C * p.L + l) * p.dd);
#pragma unroll
            for (int i = 0; i < kGatherRegs; ++i)
                if (tid + i * nthreads < nitems) {
                    const float k = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(isg) + g_c[i]);
                    *reinterpret_cast<float*>(out + g_o[i]) = (state_at(g_hi[i]) * k) * (state_at(g_ti[i]) * k);
                }
            if (nitems > kGatherRegs * nthreads) {                                   // more items per thread: their index loads wait for the prefetch
                for (int idx = tid + kGatherRegs * nthreads; idx < nitems; idx += nthreads) {
                    const uint32_t c = idx / p.dd, x = idx - c * p.dd;
                    const int64_t io = b * p.idx_bs + idx;
                    const uint32_t th = static_cast<uint32_t>(p.head[io]), tt = static_cast<uint32_t>(p.tail[io]);
                    const uint32_t base = 64 * c, sw = (c >> 1) & 3;
                    const uint32_t ph = (th >> 5) * STEP + base + ((((th >> 2) & 3) ^ sw) << 4) + (((th >> 4) & 1) << 3) + 2 * (th & 3);
                    const uint32_t pt = (tt >> 5) * STEP + base + ((((tt >> 2) & 3) ^ sw) << 4) + (((tt >> 4) & 1) << 3) + 2 * (tt & 3);
                    const float k = isg[c];
                    reinterpret_cast<float*>(out)[c * Ldd + x] = (state_at(ph) * k) * (state_at(pt) * k);
                }
            }
            if (p.hsave) {                                                          // wave w: rows w, w + NW, ...; lane = (K step, half, slot)
                char* hs = reinterpret_cast<char*>(p.hsave + ((static_cast<int64_t>(l) * p.B + b) * C) * S);
                const int ts = 32 * (lane >> 3) + 16 * ((lane >> 2) & 1) + 4 * (lane & 3);       // the four columns behind this lane's 8 bytes
                const int lo_off = (lane >> 3) * STEP + (((lane >> 2) & 1) << 3);
                if (ts < S)
                    for (int c = wave; c < C; c += NW) {
                        const float k = isg[c];
                        const int off = lo_off + 64 * c + (((lane & 3) ^ ((c >> 1) & 3)) << 4);
                        const uint2 hi = *reinterpret_cast<const uint2*>(Hs + off), lo = *reinterpret_cast<const uint2*>(Hs + PLANE + off);
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        const h2 a0 = __builtin_bit_cast(h2, hi.x), a1 = __builtin_bit_cast(h2, hi.y), b0 = __builtin_bit_cast(h2, lo.x), b1 = __builtin_bit_cast(h2, lo.y);
                        float4 v;
                        v.x = (static_cast<float>(a0[0]) + static_cast<float>(b0[0])) * k; v.y = (static_cast<float>(a0[1]) + static_cast<float>(b0[1])) * k;
                        v.z = (static_cast<float>(a1[0]) + static_cast<float>(b1[0])) * k; v.w = (static_cast<float>(a1[1]) + static_cast<float>(b1[1])) * k;
                        *reinterpret_cast<float4*>(hs + static_cast<uint32_t>(c * S + ts) * 4u) = v;
                    }
            }
            STAMP(8);
        }
        hop_i = 6; STAMP(0);
        lds_barrier();                                                              // the gathers are done before the next graph's h^0 lands
        if (p.stats && tid < 2 * p.L + 1) p.stats[static_cast<int64_t>(b) * (2 * p.L + 1) + tid] = __builtin_bit_cast(float, gstat[tid]);
    }
}

// ================================================================================================ backward, two-term f16 form
// All L hops of a graph in one persistent workgroup (one per CU, graphs b = blockIdx.x, + gridDim.x, ...); wave w owns the 16 columns
// t = 16 w .. 16 w + 15 of both products of a hop (M = t):
//   (c)  gA_l^T [t][s]     = sum_c  H^l-1[c][t] . Y_l[c][s]          (K = channels)   -> g_adj, 16-byte stores
//   (d)  gH^l-1^T [t][c]   = sum_s  A_l[s][t]   . Y_l[c][s]          (K = s)          -> next hop's Y after (+ relation gradient) . act'
// with Y_l = d loss / d (pre-activation of hop l), [C][S], resident in LDS as two half planes under ONE power-of-two scale per graph and
// hop (the contraction runs over channels in (c) and over columns in (d): only a scalar scale commutes with both).  The other
// operands stream through a ring of two LDS slabs of 32 rows (A_l: 32 rows s; H^l-1: 32 channels), fp32 -> two half terms on the way in,
// under per-graph scales known from the forward (`stats`: max |A_l|, max |H^l|); their fragments — 8 consecutive k for one t, where k is
// the slab's ROW index — come out through the transposing read ds_read_b64_tr_b16, as do Y's in (c).  The relation gradient
// (d out / d h[head], h[tail]) is scattered into an fp32 image R one hop ahead, with LDS float atomics (indices are arbitrary).
// One workgroup barrier per slab step and one for the max magnitude of the new Y.
template <int NKS, int NTC, bool BLK>
__global__ void __launch_bounds__(128 * NKS) k_propagate_bwd_h(const PropBwdH p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int CH = NTC * 16;
    constexpr int NKC = (CH + 31) / 32;           // K steps of (c)
    constexpr int YCH = NKC * 32;                 // channel rows of the Y image: rows CH .. YCH stay zero ((c) contracts over them)
    constexpr int STEP = YCH * 64;                // bytes of one K step (32 columns) of one plane of Y
    constexpr int PLANE = NKS * STEP;
    const int S = p.S, C = p.C, L = p.L;
    const int RS = 2 * S + 16;                    // slab row: S halves + 16 bytes (rows 8 apart land on different banks)
    const int SLAB = 32 * RS;                     // one plane of one slab
    const int pitch = S + 4;                      // fp32 image of the relation gradient
    unsigned char* Ys = sm;                                          // [2][PLANE]
    unsigned char* ring = sm + 2 * PLANE;                            // [2 slots][2 planes][SLAB]
    unsigned char* Rb = ring + 4 * SLAB;                             // fp32 [CH][pitch]
    uint32_t* ymax = reinterpret_cast<uint32_t*>(Rb + CH * pitch * 4);      // [2]
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = nthreads >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const uint32_t SSb = static_cast<uint32_t>(S) * S * 4;
    const int nstat = 2 * L + 1;

    // ---- slab staging: 32 rows x S floats = 8 S float4 = two per thread (nthreads = 4 S); unit u = tid + i nthreads -> (row, 4 columns)
    const int nf4 = S >> 2;
    int s_row[2], s_off[2], s_j[2];
    uint32_t s_goff[2], s_boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + i * nthreads;
        const int r = u / nf4, c4 = u - r * nf4;
        s_row[i] = r;
        s_off[i] = r * RS + 8 * c4;
        s_goff[i] = static_cast<uint32_t>(r * S + 4 * c4) * 4u;
        s_j[i] = c4 >> 2;                                               // block mode (dd == 16): column block of this unit ...
        s_boff[i] = static_cast<uint32_t>((r & 15) * 16 + 4 * (c4 & 3)) * 4u;      // ... and its place inside a 16 x 16 block
    }
    // BLOCK MODE (p.identity): A_l[i dd + r][j dd + c] = trans[l-1][b, e(i, j), r, c] (identity on the diagonal), read in place; the gradient
    // of product (c) goes straight into the transition tensors' layout and the diagonal blocks into a per-workgroup sum
    constexpr bool blocks = BLK;
    const int nn = S >> 4;
    const auto rs_ident = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blocks ? p.identity : p.h0), 0, 1024, 0x00020000);
    const uint32_t voff_blk = static_cast<uint32_t>(li * 16 + 4 * lq) * 4u;
    f32x4 gI_acc = f32x4{0.f, 0.f, 0.f, 0.f};                          // d loss / d identity [r = li][c = 4 lq ..], this wave's node, all graphs and hops
    auto want_c = [&](int l) { return blocks ? (p.gtrans[l - 1] != nullptr || p.gident_ws != nullptr) : p.gadj[l - 1] != nullptr; };
    // ---- slab requests.  The slabs of one graph come in PHASES (hop l = L .. 1; per hop first the NKC slabs of H^l-1 — kind 0, product (c),
    // only when g_adj[l-1] is wanted — then the NKS slabs of A_l — kind 1, product (d)).  A phase has ONE buffer descriptor (the whole
    // [C][S] state / [S][S] adjacency of the graph: rows past it come back as zeros) and ONE scale; a request inside a phase costs two loads
    // and an add, the scalar work (64-bit addresses, the statistics) is paid once per phase.
    struct Phase { int bb, l, kind, k; };
    auto phase_len = [&](int kind) { return kind == 0 ? NKC : NKS; };
    auto first_kind = [&](int l) { return want_c(l) ? 0 : 1; };
    Phase ph{static_cast<int>(blockIdx.x), L, first_kind(L), 0};
    auto phase_rsrc = [&](const Phase& q) {
        if (q.kind == 0) {
            const float* P = q.l >= 2 ? p.hsave + ((static_cast<int64_t>(q.l) - 2) * p.B + q.bb) * C * S : p.h0 + q.bb * p.h0_bs;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, C * S * 4, 0x00020000);
        }
        if (blocks) return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.trans[q.l - 1] + static_cast<int64_t>(q.bb) * C * 256), 0, C * 1024, 0x00020000);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.adj[q.l - 1]) + static_cast<int64_t>(q.bb) * SSb), 0,
                                                 static_cast<int>(SSb), 0x00020000);
    };
    auto phase_scale = [&](const Phase& q) {
        const float* sb = p.stats + static_cast<int64_t>(q.bb) * nstat;
        return hx2_scale_of(q.kind == 0 ? sb[q.l - 1] : sb[L + q.l]);
    };
    bool ph_live = ph.bb < p.B;
    auto ph_rs = phase_rsrc(ph_live ? ph : Phase{0, L, 1, 0});
    float ph_sc = ph_live ? phase_scale(ph) : 1.f;
    float stg_sc[2] = {1.f, 1.f};                                      // scale of the slab in each register set
    auto request = [&](u32x4 (&dst)[2], float& sc) {
        sc = ph_sc;
        if (ph_live && blocks && ph.kind == 1) {                        // rows of nodes 2 k, 2 k + 1 out of the transition blocks / the identity
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ni = 2 * ph.k + (s_row[i] >> 4), j = s_j[i];
                const bool valid = ni < nn, diag = ni == j;
                const uint32_t e = static_cast<uint32_t>(ni * (nn - 1) + (j < ni ? j : j - 1));
                const u32x4 vt = __builtin_amdgcn_raw_buffer_load_b128(ph_rs, (valid && !diag) ? e * 1024u + s_boff[i] : kOOB, 0, 0);
                const u32x4 vi = __builtin_amdgcn_raw_buffer_load_b128(rs_ident, (valid && diag) ? s_boff[i] : kOOB, 0, 0);
                dst[i] = vt | vi;                                       // one of the two is zeros (out of range)
            }
        } else if (ph_live) {
            const uint32_t koff = static_cast<uint32_t>(ph.k) * 128u * S;          // 32 rows
#pragma unroll
            for (int i = 0; i < 2; ++i) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(ph_rs, s_goff[i] + koff, 0, 0);
        } else {
            dst[0] = u32x4{0u, 0u, 0u, 0u}; dst[1] = dst[0];
        }
        if (++ph.k == phase_len(ph.kind)) {                            // next phase
            ph.k = 0;
            if (ph.kind == 0) ph.kind = 1;
            else {
                if (--ph.l == 0) { ph.l = L; ph.bb += static_cast<int>(gridDim.x); }
                ph.kind = first_kind(ph.l);
            }
            ph_live = ph.bb < p.B;
            if (ph_live) { ph_rs = phase_rsrc(ph); ph_sc = phase_scale(ph); }
        }
    };
    auto store_slab = [&](const u32x4 (&src)[2], int slot, float scale) {
        unsigned char* d = ring + slot * 2 * SLAB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t h0, l0, h1, l1;
            hx2_split2(as_f(src[i].x) * scale, as_f(src[i].y) * scale, h0, l0);
            hx2_split2(as_f(src[i].z) * scale, as_f(src[i].w) * scale, h1, l1);
            *reinterpret_cast<uint2*>(d + s_off[i]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + SLAB + s_off[i]) = make_uint2(l0, l1);
        }
    };
    // ---- fragment addresses.  Transposing read: the 16 lanes of group lq address a 4 (k) x 16 (m) block of halves, lane ip at row ip >> 2,
    // columns 4 (ip & 3) ..; lane ip receives column ip of the four rows.  Two reads (k rows 8 lq .. + 3 and + 4 .. + 7) make a fragment.
    const int tr_slab = (8 * lq + (li >> 2)) * RS + (16 * wave + 4 * (li & 3)) * 2;        // + 4 RS for the second read; + slot, plane
    auto tr_frag = [](const unsigned char* lo_p, const unsigned char* hi_p) {
        typedef short i16x4 __attribute__((ext_vector_type(4)));
        const i16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lo_p));
        const i16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(hi_p));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // Y image (the forward's state image with NATURAL column order inside a K step): element (c, s) of plane q at
    //   q PLANE + (s >> 5) STEP + 64 c + 16 (((s >> 3) & 3) ^ ((c >> 1) & 3)) + 2 (s & 7)
    // (c): Y^T fragment of column tile n, K step kc — rows (k) = channels c0 = 32 kc + 8 lq + (li >> 2) (+ 4 for the second read), columns
    // (m) = s0 = 16 n + 4 (li & 3) ..: with (s0 >> 3) & 3 = 2 (n & 1) | ((li & 3) >> 1) and (c0 >> 1) & 3 = (li >> 3) & 1 (| 2 for the second read)
    // the address separates into a per-lane base and constants of (n, kc):
    //   first read : yc_lane + 2048 kc + (n >> 1) STEP + 32 (n & 1);   second: the same + 256 with the 32 (n & 1) term flipped
    const int yc_lane = 64 * (8 * lq + (li >> 2)) + (((((li & 3) >> 1) ^ ((li >> 3) & 1))) << 4) + 8 * (li & 1);
    const int swz = ((li >> 1) & 3) << 4;
    const int yb_rd = li * 64 + ((lq << 4) ^ swz);                     // (d): B fragment of channel 16 j + li, columns 32 ks + 8 lq ..: + 1024 j + STEP ks
    const int t0w = 16 * wave + 4 * lq;                                 // C layout of (d): channel 16 j + li, columns t0w .. + 3
    const int yw = (t0w >> 5) * STEP + 64 * li + (((((t0w >> 3) & 3) << 4)) ^ swz) + 2 * (t0w & 7);       // + 1024 j
    const int rw = (li * pitch + t0w) * 4;                              // R image: + 64 j pitch
    const int nitems = C * p.dd, Ldd = L * p.dd;

    // ---- zero state of the workgroup: R, the maxima, and the Y image (its columns S .. 32 NKS are never written and meet zero rows of A)
    for (int i = tid; i < CH * pitch; i += nthreads) reinterpret_cast<float*>(Rb)[i] = 0.f;
    for (int i = tid; i < 2 * PLANE / 16; i += nthreads) reinterpret_cast<uint4*>(Ys)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 2) ymax[tid] = 0u;
    // ---- slab pipeline.  Step n consumes slab n from ring slot n & 1.  At the start of step n: slot n & 1 holds slab n (written during step
    // n - 1), register set (n + 1) & 1 holds slab n + 1 and set n & 1 slab n + 2 (both in flight).  Step n: barrier; set (n + 1) & 1 -> slot
    // (n + 1) & 1; request slab n + 3 into that set; products of slab n.
    u32x4 stg[2][2];
    request(stg[0], stg_sc[0]);                                         // slab 0
    request(stg[1], stg_sc[1]);                                         // slab 1
    store_slab(stg[0], 0, stg_sc[0]);                                   // slab 0 -> slot 0
    request(stg[0], stg_sc[0]);                                         // slab 2
    int stepno = 0;
    int gi = -1, hop_i = 0;
    (void)gi; (void)hop_i;
    auto step_in = [&](bool stamp = false) {
        if (stamp) STAMP(13);
        lds_barrier();                                                  // slab `stepno` (and a freshly written Y image) visible; the other slot is free
        if (stamp) STAMP(14);
        // (the register set is selected by a uniform BRANCH, not by an index: indexed, the compiler addressed the sets through s_set_gpr_idx moves,
        // and a move of a register a request is still filling waits for the request — the two-slab prefetch was awaited right behind its issue)
        if ((stepno + 1) & 1) store_slab(stg[1], 1, stg_sc[1]);
        else store_slab(stg[0], 0, stg_sc[0]);
        if (stamp) STAMP(15);
    };
    auto step_out = [&]() {                                              // behind the step's products: the request (its issue waits on the CU's memory queue)
        if ((stepno + 1) & 1) request(stg[1], stg_sc[1]);
        else request(stg[0], stg_sc[0]);
        ++stepno;
    };

#pragma unroll 1
    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
        ++gi; hop_i = 7;
        STAMP(0);
        const float* st_b = p.stats + static_cast<int64_t>(b) * nstat;
        // relation-gradient items of this thread (the same positions in every hop)
        uint32_t it_h[kGatherRegs], it_t[kGatherRegs], it_c[kGatherRegs], it_x[kGatherRegs];
        {
            const int64_t* hd = p.head + b * p.idx_bs;
            const int64_t* tl = p.tail + b * p.idx_bs;
#pragma unroll
            for (int i = 0; i < kGatherRegs; ++i) {
                const uint32_t idx = min(tid + i * nthreads, nitems - 1);
                const uint32_t c = idx / static_cast<uint32_t>(p.dd);
                it_c[i] = c; it_x[i] = idx - c * p.dd;
                it_h[i] = static_cast<uint32_t>(hd[idx]); it_t[i] = static_cast<uint32_t>(tl[idx]);
            }
        }
        // d out_l / d h^l into R: R[c][head] += g X[c][tail], R[c][tail] += g X[c][head]   (X = h^l, g = grad_out[b, c, (l-1) dd + x]);
        // the loads are issued by scatter_load and committed (LDS float atomics) later, so that their latency is not waited for
        float sc_g[kGatherRegs], sc_h[kGatherRegs], sc_t[kGatherRegs];
        auto scatter_load = [&](int l) {
            const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.hsave + ((static_cast<int64_t>(l) - 1) * p.B + b) * C * S), 0, C * S * 4, 0x00020000);
            const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gout + static_cast<int64_t>(b) * C * Ldd), 0, C * Ldd * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < kGatherRegs; ++i) {
                sc_g[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, 4u * (it_c[i] * Ldd + (l - 1) * p.dd + it_x[i]), 0, 0));
                sc_h[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 4u * (it_c[i] * S + it_h[i]), 0, 0));
                sc_t[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 4u * (it_c[i] * S + it_t[i]), 0, 0));
            }
        };
        auto scatter_commit = [&](int l) {
            float* R = reinterpret_cast<float*>(Rb);
#pragma unroll
            for (int i = 0; i < kGatherRegs; ++i)
                if (tid + i * nthreads < nitems) {
                    atomicAdd(R + it_c[i] * pitch + it_h[i], sc_g[i] * sc_t[i]);
                    atomicAdd(R + it_c[i] * pitch + it_t[i], sc_g[i] * sc_h[i]);
                }
            if (nitems > kGatherRegs * nthreads) {
                const float* X = p.hsave + ((static_cast<int64_t>(l) - 1) * p.B + b) * C * S;
                const float* go = p.gout + static_cast<int64_t>(b) * C * Ldd + (l - 1) * p.dd;
                for (int idx = tid + kGatherRegs * nthreads; idx < nitems; idx += nthreads) {
                    const int c = idx / p.dd, x = idx - c * p.dd;
                    const int64_t io = b * p.idx_bs + idx;
                    const int hi = static_cast<int>(p.head[io]), ti = static_cast<int>(p.tail[io]);
                    const float g = go[c * Ldd + x];
                    atomicAdd(R + c * pitch + hi, g * X[c * S + ti]);
                    atomicAdd(R + c * pitch + ti, g * X[c * S + hi]);
                }
            }
        };
        // h^lx at this lane's positions of the C layout (for act'): requested early, used by make_y
        u32x4 xq[NTC];
        auto x_load = [&](int lx) {
            const float* X = p.hsave + ((static_cast<int64_t>(lx) - 1) * p.B + b) * C * S;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, C * S * 4, 0x00020000);
#pragma unroll
            for (int j = 0; j < NTC; ++j)
                xq[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, static_cast<uint32_t>((16 * j + li) * S + t0w) * 4u, 0, 0);      // channels past C: zeros
        };
        // new Y = (acc / unit + R) . act'(X) from (d)'s result (zero for the first Y of a graph); writes the image under its own scale,
        // leaves R zeroed, returns 1 / scale
        auto make_y = [&](f32x4 (&acc)[NTC], float inv_unit, int par, int l_next_scatter) {
            float mm = 0.f;
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                const float4 r = *reinterpret_cast<const float4*>(Rb + rw + 64 * j * pitch);
                *reinterpret_cast<float4*>(Rb + rw + 64 * j * pitch) = make_float4(0.f, 0.f, 0.f, 0.f);
                const float xv[4] = {as_f(xq[j].x), as_f(xq[j].y), as_f(xq[j].z), as_f(xq[j].w)};
                const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = fmaf(acc[j][q], inv_unit, rv[q]);
                    const float dact = p.act == RECON_ACT_RELU ? (xv[q] > 0.f ? 1.f : 0.f) : (p.act == RECON_ACT_TANH ? 1.f - xv[q] * xv[q] : 1.f);
                    acc[j][q] = (16 * j + li < C) ? v * dact : 0.f;
                    mm = fmaxf(mm, fabsf(acc[j][q]));
                }
            }
            mm = wave_max(mm);
            if (lane == 0) atomicMax(ymax + par, __builtin_bit_cast(uint32_t, mm));
            lds_barrier();                                              // every wave is through with the old image; the maximum is complete
            const float sg = hx2_scale_of(__builtin_bit_cast(float, ymax[par]));
            if (tid == 0) ymax[par ^ 1] = 0u;
            // R is zero again and nobody reads it before the next make_y (one hop and several barriers away): the relation gradient of the
            // hop below goes in now — its operands were requested with this hop's X and arrived under the same wait
            if (l_next_scatter >= 1) scatter_commit(l_next_scatter);
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                uint32_t h0, l0, h1, l1;
                hx2_split2(acc[j][0] * sg, acc[j][1] * sg, h0, l0);
                hx2_split2(acc[j][2] * sg, acc[j][3] * sg, h1, l1);
                *reinterpret_cast<uint2*>(Ys + yw + 1024 * j) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(Ys + PLANE + yw + 1024 * j) = make_uint2(l0, l1);
            }
            return hx2_inv(sg);                                         // the next slab step's barrier publishes the image
        };

        int par = 0;
        scatter_load(L);
        x_load(L);
        lds_barrier();                                                  // R zeroed (kernel start / previous graph's last make_y) before anyone adds to it
        scatter_commit(L);
        lds_barrier();
        f32x4 accd[NTC];
#pragma unroll
        for (int j = 0; j < NTC; ++j) accd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (L >= 2) scatter_load(L - 1);
        float inv_sy = make_y(accd, 0.f, par, L - 1);
        par ^= 1;
        STAMP(1);

#pragma unroll 1
        for (int l = L; l >= 1; --l) {
            const float sA = hx2_scale_of(st_b[L + l]), sP = hx2_scale_of(st_b[l - 1]);
            const bool want_gA = want_c(l);
            hop_i = L - l;
            STAMP(0);

            // ---------------- (c): g_adj[l-1]^T tile  [t in wave's tile][all s] = sum_c P^T[t][c] Y^T[s][c]
            if (want_gA) {
                constexpr int NTS = NKS * 2;                            // column tiles of s (S = 16 NW <= 16 NTS)
                f32x4 accc[NTS];
#pragma unroll
                for (int n = 0; n < NTS; ++n) accc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int kc = 0; kc < NKC; ++kc) {
                    step_in();
                    STAMP(5 + kc);
                    const unsigned char* sl = ring + (stepno & 1) * 2 * SLAB;
                    const f16x8 a_hi = tr_frag(sl + tr_slab, sl + tr_slab + 4 * RS), a_lo = tr_frag(sl + SLAB + tr_slab, sl + SLAB + tr_slab + 4 * RS);
                    const unsigned char* yk = Ys + yc_lane + 2048 * kc;
                    // every column tile of the image, unconditionally (tiles past S hold zero columns and are not stored): a uniform `if (n < NW)`
                    // per tile made ten basic blocks of read -> wait -> three MFMAs; straight-line, with the next tile's fragments requested
                    // before this tile's MFMAs, the LDS latency rides under the matrix pipe
                    auto yfrag = [&](int n, f16x8& b_hi, f16x8& b_lo) {
                        const unsigned char* q0 = yk + (n >> 1) * STEP + 32 * (n & 1);
                        const unsigned char* q1 = yk + 256 + (n >> 1) * STEP + 32 * ((n & 1) ^ 1);
                        b_hi = tr_frag(q0, q1); b_lo = tr_frag(q0 + PLANE, q1 + PLANE);
                    };
                    f16x8 bh[2], bl[2];
                    yfrag(0, bh[0], bl[0]);
#pragma unroll
                    for (int n = 0; n < NTS; ++n) {
                        if (n + 1 < NTS) yfrag(n + 1, bh[(n + 1) & 1], bl[(n + 1) & 1]);
                        accc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bl[n & 1], accc[n], 0, 0, 0);
                        accc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bh[n & 1], accc[n], 0, 0, 0);
                        accc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bh[n & 1], accc[n], 0, 0, 0);
                    }
                    step_out();
                }
                STAMP(1);
                // C layout: column (lane & 15) = s = 16 n + li, rows t = 16 w + 4 lq + r: g_adj[s][t .. t + 3] as one 16-byte store
                const float k = hx2_inv(sP) * inv_sy;
                if (blocks) {
                    // tile (s-tile n, this wave's t-tile) = block (i = n, j = wave) of g_A: rows r = li, columns 4 lq .. of g_trans[l-1][b, e(n, wave)] — one
                    // contiguous KiB per store instruction; the diagonal block adds to the identity's gradient
                    const bool st = p.gtrans[l - 1] != nullptr;
                    const auto rgt = __builtin_amdgcn_make_buffer_rsrc(st ? p.gtrans[l - 1] + static_cast<int64_t>(b) * C * 256 : const_cast<float*>(p.h0), 0, st ? C * 1024 : 0, 0x00020000);
#pragma unroll
                    for (int n = 0; n < NTS; ++n)
                        if (n < NW) {
                            const f32x4 v = accc[n] * k;
                            if (n == wave) gI_acc += v;
                            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rgt, voff_blk, (n * (nn - 1) + (wave < n ? wave : wave - 1)) * 1024, 0);
                        }
                } else {
                const auto rga = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.gadj[l - 1]) + static_cast<int64_t>(b) * SSb, 0, static_cast<int>(SSb), 0x00020000);
                const uint32_t go_lane = static_cast<uint32_t>(li * S + t0w) * 4u;
#pragma unroll
                for (int n = 0; n < NTS; ++n)
                    if (n < NW) {
                        const f32x4 v = accc[n] * k;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rga, go_lane + static_cast<uint32_t>(n) * 64u * S, 0, 0);
                    }
                }
            }
            STAMP(2);
            // ---------------- (d): gH^l-1^T tile [t in wave's tile][all c] = sum_s A_l^T[t][s] Y[c][s]
            if (l > 1) x_load(l - 1);                                   // h^l-1 at this lane's positions, for act' behind (d)
            if (l > 2) scatter_load(l - 2);                             // operands of the relation gradient one hop further down
#pragma unroll
            for (int j = 0; j < NTC; ++j) accd[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ks = 0; ks < NKS; ++ks) {
                step_in(ks == 2);
                STAMP(8 + ks);

                const unsigned char* sl = ring + (stepno & 1) * 2 * SLAB;
                const f16x8 a_hi = tr_frag(sl + tr_slab, sl + tr_slab + 4 * RS), a_lo = tr_frag(sl + SLAB + tr_slab, sl + SLAB + tr_slab + 4 * RS);
                f16x8 dh[2], dl[2];                                     // the next channel tile's fragments are requested before this tile's MFMAs
                dh[0] = *reinterpret_cast<const f16x8*>(Ys + ks * STEP + yb_rd);
                dl[0] = *reinterpret_cast<const f16x8*>(Ys + PLANE + ks * STEP + yb_rd);
#pragma unroll
                for (int j = 0; j < NTC; ++j) {
                    if (j + 1 < NTC) {
                        dh[(j + 1) & 1] = *reinterpret_cast<const f16x8*>(Ys + ks * STEP + 1024 * (j + 1) + yb_rd);
                        dl[(j + 1) & 1] = *reinterpret_cast<const f16x8*>(Ys + PLANE + ks * STEP + 1024 * (j + 1) + yb_rd);
                    }
                    accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, dl[j & 1], accd[j], 0, 0, 0);
                    accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, dh[j & 1], accd[j], 0, 0, 0);
                    accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, dh[j & 1], accd[j], 0, 0, 0);
                }
                step_out();
            }
            STAMP(3);
            const float inv_unit = hx2_inv(sA) * inv_sy;
            if (l > 1) {
                inv_sy = make_y(accd, inv_unit, par, l - 2);
                par ^= 1;
                STAMP(4);
            } else {                                                    // d loss / d h^0: column (lane & 15) = channel, rows t
                const auto rgh = __builtin_amdgcn_make_buffer_rsrc(p.gH + static_cast<int64_t>(b) * C * S, 0, C * S * 4, 0x00020000);      // channels past C: dropped (out of range)
                const uint32_t gh_lane = static_cast<uint32_t>(li * S + t0w) * 4u;
#pragma unroll
                for (int j = 0; j < NTC; ++j) {
                    const f32x4 v = accd[j] * inv_unit;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rgh, gh_lane + static_cast<uint32_t>(j) * 64u * S, 0, 0);
                }
            }
        }
    }
    if (blocks && p.gident_ws) {                                        // this workgroup's share of d loss / d identity: the waves' sums added in wave order
        lds_barrier();
        float* red = reinterpret_cast<float*>(ring);                    // [NW][16][16]
        *reinterpret_cast<f32x4*>(red + wave * 256 + li * 16 + 4 * lq) = gI_acc;
        lds_barrier();
        if (tid < 256) {
            float sum = 0.f;
            for (int w = 0; w < NW; ++w) sum += red[w * 256 + tid];
            p.gident_ws[static_cast<int64_t>(blockIdx.x) * 256 + tid] = sum;
        }
    }
}

size_t fwd_h_lds(int nks, int ntc, int S) {
    const size_t ch = 16ull * ntc;
    return 2ull * nks * ch * 64 + 3ull * ch * sizeof(uint32_t) + static_cast<size_t>(S / 16) * 16 * sizeof(float) + (2 * kMaxHops + 1) * sizeof(uint32_t);
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

size_t bwd_h_lds(int nks, int ntc, int S) {
    const size_t ch = 16ull * ntc;
    const size_t ych = (ch + 31) / 32 * 32;
    return 2ull * nks * ych * 64 + 4ull * 32 * (2 * S + 16) + ch * (S + 4) * sizeof(float) + 64;
}

}  // namespace

int prop_h_grid(int B) { return B < num_cus() ? B : num_cus(); }
bool prop_bwd_h_shape_ok(int C, int S) {
    return S % 16 == 0 && S >= 16 && S <= 160 && C <= 96 && bwd_h_lds((S + 31) / 32, (C + 15) / 16, S) <= 160 * 1024;
}

bool prop_bwd_h_supported(const PropBwdH& p) {
    if (p.identity && (p.dd != 16 || p.C != (p.S / 16) * (p.S / 16 - 1))) return false;
    if (p.S % 16 != 0 || p.S > 160 || p.C > 96 || p.S < 16 || p.dd < 1 || !p.stats || !p.hsave || !p.gH || !p.gout) return false;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    for (int l = 0; l < p.L; ++l)
        if (p.identity ? (!al16(p.trans[l]) || !al16(p.gtrans[l])) : (!al16(p.adj[l]) || !al16(p.gadj[l]))) return false;
    if (p.identity && !al16(p.identity)) return false;
    if (!al16(p.h0) || (p.h0_bs % 4) != 0 || !al16(p.hsave) || !al16(p.gH)) return false;
    const int nks = (p.S + 31) / 32, ntc = (p.C + 15) / 16;
    return bwd_h_lds(nks, ntc, p.S) <= 160 * 1024;
}

int prop_bwd_h(const PropBwdH& p, hipStream_t st) {
    if (!prop_bwd_h_supported(p)) return RECON_ERR_UNSUPPORTED;
    const int nks = (p.S + 31) / 32, ntc = (p.C + 15) / 16, nw = p.S / 16;
    const size_t lds = bwd_h_lds(nks, ntc, p.S);
    const int grid = p.B < num_cus() ? p.B : num_cus();
#define CALL_B(K_, N_, X_)                                                                                                              \
    do {                                                                                                                                \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_bwd_h<K_, N_, X_>),                 \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));             \
        hipLaunchKernelGGL((k_propagate_bwd_h<K_, N_, X_>), dim3(static_cast<unsigned>(grid)), dim3(64 * nw), lds, st, p);              \
    } while (0)
    if (p.identity) {                                                   // block mode: S = 16 n, C = n (n - 1): one (NKS, NTC) per n
        switch (nw) { case 2: CALL_B(1, 1, true); break; case 3: case 4: CALL_B(2, 1, true); break; case 5: case 6: CALL_B(3, 2, true); break;
                      case 7: CALL_B(4, 3, true); break; case 8: CALL_B(4, 4, true); break; case 9: CALL_B(5, 5, true); break; default: return RECON_ERR_UNSUPPORTED; }
    } else {
#define CALL_BN(K_)                                                                                                                     \
    switch (ntc) { case 1: CALL_B(K_, 1, false); break; case 2: CALL_B(K_, 2, false); break; case 3: CALL_B(K_, 3, false); break; case 4: CALL_B(K_, 4, false); break; \
                   case 5: CALL_B(K_, 5, false); break; default: CALL_B(K_, 6, false); break; }
    switch (nks) { case 1: CALL_BN(1); break; case 2: CALL_BN(2); break; case 3: CALL_BN(3); break; case 4: CALL_BN(4); break; default: CALL_BN(5); break; }
#undef CALL_BN
    }
#undef CALL_B
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

bool prop_fwd_h_supported(const PropK& p) {
    if (p.identity && (p.dd != 16 || p.C != (p.S / 16) * (p.S / 16 - 1) || (reinterpret_cast<uintptr_t>(p.identity) & 15) != 0)) return false;
    if (p.S % 16 != 0 || p.S > 160 || p.C > 96 || p.S < 16 || p.dd < 1) return false;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    for (int l = 0; l < p.L; ++l) if (!al16(p.identity ? p.trans[l] : p.adj[l])) return false;
    if (!al16(p.h0) || (p.h0_bs % 4) != 0 || (p.hsave && !al16(p.hsave))) return false;
    const int nks = (p.S + 31) / 32, ntc = (p.C + 15) / 16;
    return fwd_h_lds(nks, ntc, p.S) <= 160 * 1024;
}

int prop_fwd_h(const PropK& p, hipStream_t st) {
    if (!prop_fwd_h_supported(p)) return RECON_ERR_UNSUPPORTED;
    const int nks = (p.S + 31) / 32, ntc = (p.C + 15) / 16, nw = p.S / 16;
    const size_t lds = fwd_h_lds(nks, ntc, p.S);
    const int grid = p.B < num_cus() ? p.B : num_cus();
#define CALL_H(K_, N_, X_)                                                                                                              \
    do {                                                                                                                                \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_fwd_h<K_, N_, X_>),                 \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));             \
        hipLaunchKernelGGL((k_propagate_fwd_h<K_, N_, X_>), dim3(static_cast<unsigned>(grid)), dim3(64 * nw), lds, st, p);              \
    } while (0)
    if (p.identity) {                                                   // block mode: S = 16 n, C = n (n - 1): one (NKS, NTC) per n
        switch (nw) { case 2: CALL_H(1, 1, true); break; case 3: case 4: CALL_H(2, 1, true); break; case 5: case 6: CALL_H(3, 2, true); break;
                      case 7: CALL_H(4, 3, true); break; case 8: CALL_H(4, 4, true); break; case 9: CALL_H(5, 5, true); break; case 10: CALL_H(5, 6, true); break;
                      default: return RECON_ERR_UNSUPPORTED; }
    } else {
#define CALL_HN(K_)                                                                                                                     \
    switch (ntc) { case 1: CALL_H(K_, 1, false); break; case 2: CALL_H(K_, 2, false); break; case 3: CALL_H(K_, 3, false); break; case 4: CALL_H(K_, 4, false); break; \
                   case 5: CALL_H(K_, 5, false); break; default: CALL_H(K_, 6, false); break; }
    switch (nks) { case 1: CALL_HN(1); break; case 2: CALL_HN(2); break; case 3: CALL_HN(3); break; case 4: CALL_HN(4); break; default: CALL_HN(5); break; }
#undef CALL_HN
    }
#undef CALL_H
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace recon
