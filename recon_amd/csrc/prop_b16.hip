// K5 in bfloat16 — GP-GNN propagation (models/models.py:260-274 and its copies :470-485, :680-694, :918-932) on bf16 tensors:
// bf16 storage, fp32 accumulation on v_mfma_f32_16x16x32_bf16, every state rounded to bf16 once per hop — what the reference's
// torch.matmul / relu / gather chain computes when its tensors are bfloat16 (BASELINE.json configs[2], configs[4]).
// ONE MFMA per 16x16x32 block, no operand splitting, no scale search: the small form below is bound by its single pass over
// the adjacency stack (L B S^2 2 bytes), the wide form by the matrix pipe.
//
//   k_prop_b16_fwd<NKS,NTC,BLK>     S <= 160, C <= 96 (cfg 3b at n = 9): all L hops of a graph in one persistent workgroup, the
//                                   state resident in LDS (two images, one barrier per hop), A_l fetched once as MFMA A fragments
//                                   one hop ahead; block mode reads the transition tensors in place.
//   k_bgemm_b16<PK,QK,...>          batched GEMM over the graphs (any S % 8 == 0): the forward's hops for wide states
//                                   (H^l = act(H^l-1 A_l^T), n = 32: S = 512, C = 992) and both products of a backward hop
//                                   (G_l = Y_l A_l,  dA_l = Y_l^T H^l-1).  Operands by LDS-DMA into double-buffered images;
//                                   k-major operands come out of them through ds_read_b64_tr_b16.
//   k_prop_b16_ypost                Y_l = (G_l+1 + relation gradient of hop l) . act'(H^l), one wave per (graph, channel) row
//   k_prop_b16_gather               relation_l = h[head] * h[tail] from the saved states (wide form)
//   k_block_adj_b16_*               P1 in bf16 (models/models.py:240-259)
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "prop_common.h"
#include "prop_h_util.h"

namespace recon {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

// buffer-load aux bit 31 = a volatile access: the request stays where it is written.  As plain read-only loads the compiler sinks the A-fragment
// requests of a K loop that writes no memory towards their uses (found in csrc/prop_hl.hip's k_prop_gadj_hl: the prefetch distance was compiled away)
#ifdef RECON_B16_PLAIN_LOADS
constexpr int kPinnedLoad = 0;
#else
constexpr int kPinnedLoad = static_cast<int>(0x80000000u);
#endif

// timing experiments are compile-time builds (make EXTRA="-DRECON_BGEMM_ABLATE=2"): a run-time switch in front of a copy or a request costs the
// product kernels their counted waits.  k_bgemm_b16: 1 copies read the zero page, 2 no copies, 4 no stores; k_prop_b16_fwd_wide: 1 = the A
// fragments are not fetched
#ifndef RECON_BGEMM_ABLATE
#define RECON_BGEMM_ABLATE 0
#endif
#ifndef RECON_PROP_B16_ABLATE
#define RECON_PROP_B16_ABLATE 0
#endif
constexpr int kBGemmAblate = RECON_BGEMM_ABLATE, kWideAblate = RECON_PROP_B16_ABLATE;

struct PropB16K {
    const uint16_t* adj[kMaxHops];
    const uint16_t* trans[kMaxHops];
    const uint16_t* identity;
    const uint16_t* h0; int64_t h0_bs;
    const int64_t* head; const int64_t* tail; int64_t idx_bs;
    uint16_t* out; uint16_t* hsave;
    int32_t B, C, S, L, dd, act;
};

// by-value helper: __builtin_bit_cast applied directly to a vector ELEMENT yields element 0 (prop_h_util.h)
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float bf2f(uint32_t bits16) { return __builtin_bit_cast(float, bits16 << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {      // round to nearest even, a in the low half
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == RECON_ACT_RELU) return fmaxf(v, 0.f);
    if (act == RECON_ACT_TANH) return tanh_fast(v);
    return v;
}

// COMPILER HAZARD (hipcc, ROCm 7.2, gfx950): the wait states between an MFMA and the first read of its result are inserted by the compiler
// and counted along the FALL-THROUGH path only: with the accumulators in AGPRs (one accumulator tile per wave) and a uniform branch between
// the last MFMA and v_accvgpr_read — taken in inference, where the saved-state address arithmetic is skipped — the read came two
// instructions behind the MFMA and returned rows 2, 3 of the tile before they were written (block mode, n = 4, tanh: wrong values in
// exactly the columns t % 4 >= 2).  Every accumulator read-out below is preceded by an unconditional gap longer than the MFMA.
__device__ __forceinline__ void mfma_drain() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ================================================================================================ small states: fused forward
// NKS = K steps of 32 (S <= 32 NKS), NTC = channel tiles of 16 (C <= 16 NTC); blockDim.x = 4 S (one wave per 16 rows of A; S % 16 == 0).
// Dynamic LDS: two state images [NKS][16 NTC][64 B].  Element (channel c, column t) of an image lives at byte
//     (t >> 5) STEP + 64 c + 16 (((t >> 3) & 3) ^ ((c >> 1) & 3)) + 2 (t & 7):
// one 64-byte row per (K step, channel); its four 16-byte slots are the B fragments of the four lane groups (natural k order: lane group q
// holds t = 32 ks + 8 q .. + 7), XOR-rotated by the channel so that the ds_read_b128 of a fragment is conflict free.
// Hop l reads image `cur` and writes image `cur ^ 1`: one barrier per hop (the prop_h.hip form needs the channel maxima between them).
template <int NKS, int NTC, bool BLK>
__global__ void __launch_bounds__(128 * NKS) k_prop_b16_fwd(const PropB16K p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int CH = NTC * 16;
    constexpr int STEP = CH * 64;
    constexpr int PLANE = NKS * STEP;
    constexpr int NG = 2;                                               // gather pairs per thread whose positions stay in registers
    const int S = p.S, C = p.C, L = p.L;
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const uint32_t SSb = static_cast<uint32_t>(S) * S * 2;

    // A fragment of K step ks: row s = 16 wave + li, columns 32 ks + 8 lq .. + 7 (16 bytes); the part of the last K step past S is requested out
    // of range (zeros).  BLOCK MODE (dd == 16: this wave's rows are node i = wave, a K step covers nodes 2 ks, 2 ks + 1): columns 8 (lq & 1) ..
    // of row li of trans[l][b, e(i, j)], j = 2 ks + (lq >> 1); the diagonal block comes from `identity` (fetched once per kernel).
    const uint32_t voff_row = (static_cast<uint32_t>(16 * wave + li) * S + 8 * lq) * 2u;
    const uint32_t voff_last = (32 * (NKS - 1) + 8 * lq < S) ? voff_row + 64u * (NKS - 1) : kOOB;
    const int nn = S >> 4;
    const uint32_t voff_blk = static_cast<uint32_t>(li * 32 + (lq & 1) * 16);
    auto rsrc_a = [&](int l, int bb) {
        if constexpr (BLK)
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.trans[l] + static_cast<int64_t>(bb) * C * 256), 0, C * 512, 0x00020000);
        else
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.adj[l]) + static_cast<int64_t>(bb) * SSb), 0,
                                                     static_cast<int>(SSb), 0x00020000);
    };
    auto a_off = [&](int ks) -> uint32_t {
        if constexpr (BLK) {
            const int j = 2 * ks + (lq >> 1);
            const int e = wave * (nn - 1) + (j < wave ? j : j - 1);
            return (j < nn && j != wave) ? static_cast<uint32_t>(e) * 512u + voff_blk : kOOB;
        } else {
            return ks == NKS - 1 ? voff_last : voff_row + 64u * ks;
        }
    };
    u32x4 diag = u32x4{0u, 0u, 0u, 0u};
    if constexpr (BLK) {
        const auto rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.identity), 0, 512, 0x00020000);
        diag = __builtin_amdgcn_raw_buffer_load_b128(rs_i, (lq >> 1) == (wave & 1) ? voff_blk : kOOB, 0, 0);
    }
    const int swz = ((li >> 1) & 3) << 4;
    const int b_rd = li * 64 + ((lq << 4) ^ swz);                       // B fragment of channel 16 j + li: + 1024 j + STEP ks
    // C layout of the hop's result: channel 16 j + li, columns t = 16 wave + 4 lq .. + 3 -> 8 bytes of the next image
    const int t0w = 16 * wave + 4 * lq;
    const int so_w = (wave >> 1) * STEP + 64 * li + (((2 * (wave & 1) + (lq >> 1)) << 4) ^ swz) + 8 * (lq & 1);      // + 1024 j
    const int nitems = C * p.dd, Ldd = L * p.dd, npairs = nitems >> 1;
    auto pos = [&](uint32_t c, uint32_t t) {
        return (t >> 5) * STEP + 64u * c + (((((t >> 3) & 3u) ^ ((c >> 1) & 3u))) << 4) + 2u * (t & 7u);
    };

    for (int i = tid; i < 2 * PLANE / 16; i += nthreads) reinterpret_cast<uint4*>(sm)[i] = make_uint4(0u, 0u, 0u, 0u);
    lds_barrier();                                                      // pad columns (t >= S) stay zero for the whole kernel: nobody writes them

    // gather items (pairs of neighbouring x: one 4-byte store): byte positions of head / tail in an image, element offset in `out`
    uint32_t g_h0[NG], g_h1[NG], g_t0[NG], g_t1[NG], g_o[NG];
    auto gather_setup = [&](int bb) {
        const int64_t* hd = p.head + bb * p.idx_bs;
        const int64_t* tl = p.tail + bb * p.idx_bs;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const uint32_t it = 2u * min(tid + i * nthreads, npairs - 1);
            const uint32_t c = it / static_cast<uint32_t>(p.dd), x = it - c * p.dd;
            g_h0[i] = pos(c, static_cast<uint32_t>(hd[it])); g_h1[i] = pos(c, static_cast<uint32_t>(hd[it + 1]));
            g_t0[i] = pos(c, static_cast<uint32_t>(tl[it])); g_t1[i] = pos(c, static_cast<uint32_t>(tl[it + 1]));
            g_o[i] = c * Ldd + x;
        }
    };
    if (p.idx_bs == 0) gather_setup(0);

    u32x4 raw[NKS];
    int b = blockIdx.x;
    {
        const auto rs = rsrc_a(0, b < p.B ? b : 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) raw[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, b < p.B ? a_off(ks) : kOOB, 0, 0);
    }
    int cur = 0;
    const int s8 = S >> 3, npieces = C * s8;
#pragma unroll 1
    for (; b < p.B; b += gridDim.x) {
        // ---- h^0 of graph b into image `cur`: 16-byte pieces (8 columns of one channel), consecutive threads = consecutive pieces
        {
            const uint16_t* h0b = p.h0 + b * p.h0_bs;
            unsigned char* Hc = sm + cur * PLANE;
            for (int q = tid; q < npieces; q += nthreads) {
                const int c = q / s8, t = 8 * (q - c * s8);
                const u32x4 v = *reinterpret_cast<const u32x4*>(h0b + static_cast<int64_t>(c) * S + t);
                *reinterpret_cast<u32x4*>(Hc + (t >> 5) * STEP + 64 * c + ((((t >> 3) & 3) ^ ((c >> 1) & 3)) << 4)) = v;
            }
        }
        if (p.idx_bs != 0) gather_setup(b);
        lds_barrier();

#pragma unroll 1
        for (int l = 0; l < L; ++l) {
            const unsigned char* Hc = sm + cur * PLANE;
            unsigned char* Hn = sm + (cur ^ 1) * PLANE;
            // the rows of the NEXT step (next hop, or hop 0 of this workgroup's next graph) are requested between the K steps, into the
            // registers whose fragment the matrix pipe has just been handed
            const bool more_hops = l + 1 < L;
            const int nb = more_hops ? b : b + static_cast<int>(gridDim.x);
            const bool pre = nb < p.B;
            const auto rs_n = rsrc_a(more_hops ? l + 1 : 0, pre ? nb : b);
            const auto rs_hs = __builtin_amdgcn_make_buffer_rsrc(
                p.hsave ? p.hsave + ((static_cast<int64_t>(l) * p.B + b) * C) * S : p.out, 0, p.hsave ? C * S * 2 : 0, 0x00020000);
            f32x4 acc[NTC];
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                u32x4 av = raw[ks];
                if constexpr (BLK) {
                    const bool dk = ks == (wave >> 1);                  // wave-uniform
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] |= dk ? diag[e] : 0u;
                }
                const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                for (int j = 0; j < NTC; ++j) {
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Hc + ks * STEP + 1024 * j + b_rd);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf, acc[j], 0, 0, 0);
                }
                raw[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_n, pre ? a_off(ks) : kOOB, 0, 0);
                __builtin_amdgcn_sched_barrier(0x078f);                // everything but VMEM may move across: the requests stay where they are written
            }
            mfma_drain();
            // ---- epilogue: activation, bf16, into the other image and (training) the saved states.  One straight-line copy per activation
            // (a run-time `act` inside the element loop became four basic blocks per value)
            auto epilogue = [&](auto act_c) {
                constexpr int ACT = decltype(act_c)::value;
#pragma unroll
                for (int j = 0; j < NTC; ++j) {
                    float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ACT == RECON_ACT_RELU ? fmaxf(v[r], 0.f) : (ACT == RECON_ACT_TANH ? tanh_fast(v[r]) : v[r]);
                    const uint32_t w0 = pack_bf2(v[0], v[1]), w1 = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(Hn + so_w + 1024 * j) = make_uint2(w0, w1);
                    const int c = 16 * j + li;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{w0, w1}, rs_hs, (p.hsave && c < C) ? static_cast<uint32_t>(c * S + t0w) * 2u : kOOB, 0, 0);
                }
            };
            if (p.act == RECON_ACT_RELU) epilogue(std::integral_constant<int, RECON_ACT_RELU>{});
            else if (p.act == RECON_ACT_TANH) epilogue(std::integral_constant<int, RECON_ACT_TANH>{});
            else epilogue(std::integral_constant<int, RECON_ACT_LINEAR>{});
            lds_barrier();                                              // H^l complete; everybody has read H^l-1
            // ---- relation_l = gather(h, heads) * gather(h, tails)   (models/models.py:270-273)
            uint16_t* outb = p.out + (static_cast<int64_t>(b) * C * L + l) * p.dd;
            auto hv = [&](uint32_t q) { return bf2f(*reinterpret_cast<const uint16_t*>(Hn + q)); };
#pragma unroll
            for (int i = 0; i < NG; ++i)
                if (tid + i * nthreads < npairs)
                    *reinterpret_cast<uint32_t*>(outb + g_o[i]) = pack_bf2(hv(g_h0[i]) * hv(g_t0[i]), hv(g_h1[i]) * hv(g_t1[i]));
            if (npairs > NG * nthreads) {                               // more items per thread: their index loads wait for the prefetch
                for (int pi = tid + NG * nthreads; pi < npairs; pi += nthreads) {
                    const uint32_t it = 2u * pi, c = it / static_cast<uint32_t>(p.dd), x = it - c * p.dd;
                    const int64_t io = b * p.idx_bs + it;
                    const float v0 = hv(pos(c, static_cast<uint32_t>(p.head[io]))) * hv(pos(c, static_cast<uint32_t>(p.tail[io])));
                    const float v1 = hv(pos(c, static_cast<uint32_t>(p.head[io + 1]))) * hv(pos(c, static_cast<uint32_t>(p.tail[io + 1])));
                    *reinterpret_cast<uint32_t*>(outb + c * Ldd + x) = pack_bf2(v0, v1);
                }
            }
            cur ^= 1;
        }
        cur ^= 1;                                                       // the next graph's h^0 goes where nobody gathers from
    }
}

// ================================================================================================ wide states: fused forward
// 160 < S <= 512 (S % 32 == 0; n = 32: S = 512, C = 992).  The state of a graph no longer fits a CU, but channels never mix: a workgroup
// owns (graph, 128 channels) for ALL hops, the chunk's state [NKS][128][64 B] (128 KiB at S = 512) resident in LDS and updated IN PLACE,
// so that no state travels through HBM between hops (the batched-GEMM form writes and re-reads 1 GB per hop at B = 1024: it is bound by
// that traffic, 2.5 GB per hop at ~5 TB/s, not by its 213 us of MFMA time).  Wave w of 8 owns RT row tiles of every A_l and fetches their
// A fragments straight into registers two K steps ahead (the 8 chunks of a graph run on one XCD: its L2 serves 7 of the 8 reads of A_l);
// one B fragment from LDS feeds RT MFMAs.  Per hop: K loop (fully unrolled: the compiler counts the outstanding requests only in
// straight-line code), barrier, epilogue (activation, bf16, 16-byte LDS / saved-state writes after a lane-row exchange between
// neighbouring row tiles), barrier, gather.  Block mode reads the transition tensors in place.
template <int NKS, int RT, bool BLK>
__global__ void __launch_bounds__(512, 2) k_prop_b16_fwd_wide(const PropB16K p, const int nchunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int NTC = 8, CH = 128, STEP = CH * 64, PD = 2, NG = 2;
    static_assert(NKS % 2 == 0 && NKS >= PD, "the request ring is two steps deep");
    const int S = p.S, C = p.C, L = p.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int gslot = within / nchunk, chunk = within - gslot * nchunk;
    const int b = gslot * 8 + xcd;
    if (b >= p.B) return;
    const int c0 = chunk * CH, cn = min(CH, C - c0);
    const uint32_t SSb = static_cast<uint32_t>(S) * S * 2;
    const int nn = S >> 4;

    auto rsrc_a = [&](int l) {
        if constexpr (BLK)
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.trans[l] + static_cast<int64_t>(b) * C * 256), 0, C * 512, 0x00020000);
        else
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.adj[l]) + static_cast<int64_t>(b) * SSb), 0,
                                                     static_cast<int>(SSb), 0x00020000);
    };
    // A fragment of (row tile R = wave RT + r, K step ks): one lane offset per row tile, the K step in the instruction's scalar offset (the
    // offsets of all 16 K steps, hoisted out of the hop loop, had cost 64 registers and spilled).  BLOCK MODE: columns 8 (lq & 1) .. of row li
    // of trans[l][b, e(R, j)], j = 2 ks + (lq >> 1), e = R (nn - 1) + j - [j > R]; the diagonal block comes from `identity`.
    const uint32_t voff_blk = static_cast<uint32_t>(li * 32 + (lq & 1) * 16);
    uint32_t voff_r[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int R = wave * RT + r;
        if constexpr (BLK) voff_r[r] = voff_blk + 512u * static_cast<uint32_t>(R * (nn - 1) + (lq >> 1));
        else voff_r[r] = R < nn ? (static_cast<uint32_t>(16 * R + li) * S + 8 * lq) * 2u : kOOB;
    }
    auto load_a = [&](decltype(rsrc_a(0)) rs, int r, int ks, int lq_hi, bool live) -> u32x4 {
        if constexpr (BLK) {
            const int R = wave * RT + r, j = 2 * ks + lq_hi;
            const bool valid = live && R < nn && j < nn && j != R;
            // (the whole offset in the lane part: the range check looks at the lane offset alone, and R = 0 would make it negative)
            return __builtin_amdgcn_raw_buffer_load_b128(rs, valid ? voff_r[r] + 1024u * ks - (j > R ? 512u : 0u) : kOOB, 0, kPinnedLoad);
        } else {
            return __builtin_amdgcn_raw_buffer_load_b128(rs, (live && !(kWideAblate & 1)) ? voff_r[r] : kOOB, 64 * ks, kPinnedLoad);
        }
    };
    u32x4 dpar[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};   // identity rows for row tiles of even / odd node index
    if constexpr (BLK) {
        const auto rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.identity), 0, 512, 0x00020000);
        const u32x4 idv = __builtin_amdgcn_raw_buffer_load_b128(rs_i, voff_blk, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { dpar[0][e] = (lq >> 1) == 0 ? idv[e] : 0u; dpar[1][e] = (lq >> 1) == 1 ? idv[e] : 0u; }
    }
    // ---- gather items of this chunk (pairs of neighbouring x), positions in the image
    auto pos = [&](uint32_t c, uint32_t t) {
        return (t >> 5) * STEP + 64u * c + (((((t >> 3) & 3u) ^ ((c >> 1) & 3u))) << 4) + 2u * (t & 7u);
    };
    const int Ldd = L * p.dd, npairs = (cn * p.dd) >> 1;
    uint32_t* gtab = reinterpret_cast<uint32_t*>(sm + NKS * STEP);     // [NG][5][512]: the positions live in LDS, not in registers (the K loop needs them all)
    {
        const int64_t* hd = p.head + b * p.idx_bs + static_cast<int64_t>(c0) * p.dd;
        const int64_t* tl = p.tail + b * p.idx_bs + static_cast<int64_t>(c0) * p.dd;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const uint32_t it = 2u * min(tid + i * 512, npairs - 1);
            const uint32_t c = it / static_cast<uint32_t>(p.dd), x = it - c * p.dd;
            uint32_t* g = gtab + i * 5 * 512 + tid;
            g[0] = pos(c, static_cast<uint32_t>(hd[it])); g[512] = pos(c, static_cast<uint32_t>(hd[it + 1]));
            g[1024] = pos(c, static_cast<uint32_t>(tl[it])); g[1536] = pos(c, static_cast<uint32_t>(tl[it + 1]));
            g[2048] = c * Ldd + x;
        }
    }
    // ---- h^0 of the chunk into the image
    {
        const uint16_t* h0b = p.h0 + b * p.h0_bs + static_cast<int64_t>(c0) * S;
        const int s8 = S >> 3, npieces = cn * s8;
        for (int q = tid; q < npieces; q += 512) {
            const int c = q / s8, t = 8 * (q - c * s8);
            const u32x4 v = *reinterpret_cast<const u32x4*>(h0b + static_cast<int64_t>(c) * S + t);
            *reinterpret_cast<u32x4*>(sm + (t >> 5) * STEP + 64 * c + ((((t >> 3) & 3) ^ ((c >> 1) & 3)) << 4)) = v;
        }
        const int zp = (CH - cn) * s8;                                  // channels past the chunk's last: zeros (their results are never stored)
        for (int q = tid; q < zp; q += 512) {
            const int c = cn + q / s8, t = 8 * (q % s8);
            *reinterpret_cast<u32x4*>(sm + (t >> 5) * STEP + 64 * c + ((((t >> 3) & 3) ^ ((c >> 1) & 3)) << 4)) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    // ---- A fragments of the first PD steps of the first hop
    u32x4 ar[PD][RT];
    {
        const auto rs = rsrc_a(0);
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int r = 0; r < RT; ++r) ar[d][r] = load_a(rs, r, d, lq >> 1, true);
    }
    const int swz = ((li >> 1) & 3) << 4;
    const int b_rd = li * 64 + ((lq << 4) ^ swz);
    lds_barrier();

#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const auto rs_cur = rsrc_a(l);
        const auto rs_nxt = rsrc_a(l + 1 < L ? l + 1 : l);
        const bool more = l + 1 < L;
        int lq_hi = lq >> 1;
        asm volatile("" : "+v"(lq_hi));                                 // keeps the block-mode offset arithmetic inside the hop (not hoisted: registers)
        if constexpr (BLK) {                                            // likewise the masked identity rows of all (row tile, K step) pairs
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t d0 = dpar[0][e], d1 = dpar[1][e];
                asm volatile("" : "+v"(d0), "+v"(d1));
                dpar[0][e] = d0; dpar[1][e] = d1;
            }
        }
        f32x4 acc[RT][NTC];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            bf16x8 a[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                u32x4 av = ar[ks % PD][r];
                if constexpr (BLK) {
                    const int R = wave * RT + r;
                    const bool dk = ks == (R >> 1);                     // wave-uniform
                    const bool odd = R & 1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] |= dk ? (odd ? dpar[1][e] : dpar[0][e]) : 0u;
                }
                a[r] = __builtin_bit_cast(bf16x8, av);
            }
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(sm + ks * STEP + 1024 * j + b_rd);
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], bf, acc[r][j], 0, 0, 0);
            }
            // the fragments of step ks + PD (of the next hop behind this hop's last steps) into the registers just consumed.  Measured and not
            // kept: the state's fragments one step ahead in a second register set (pinned with sched_barrier: the scheduler otherwise sinks
            // every LDS read to its use) — no change; requests in pairs of K steps four steps ahead (both halves of a 128-byte line back to
            // back) — no change either.  Without the requests the kernel takes 1.39 ms at n = 32, with them 2.15.
            if (ks + PD < NKS) {
#pragma unroll
                for (int r = 0; r < RT; ++r) ar[ks % PD][r] = load_a(rs_cur, r, ks + PD, lq_hi, true);
            } else {
#pragma unroll
                for (int r = 0; r < RT; ++r) ar[ks % PD][r] = load_a(rs_nxt, r, ks + PD - NKS, lq_hi, more);
            }
            __builtin_amdgcn_sched_barrier(0x078f);                    // everything but VMEM may move across
        }
        mfma_drain();
        lds_barrier();                                                  // everybody has read H^l-1
        // ---- epilogue: activation, bf16; row tiles r, r + 1 exchange halves between lane rows: a lane then holds 8 consecutive columns
        // of one channel — 16-byte writes into the image and into the saved states
        const auto rs_hs = __builtin_amdgcn_make_buffer_rsrc(
            p.hsave ? p.hsave + ((static_cast<int64_t>(l) * p.B + b) * C + c0) * S : p.out, 0, p.hsave ? cn * S * 2 : 0, 0x00020000);
        int li_e = li, lq_e = lq;                                       // laundered: the 32 write positions below are loop invariant, and hoisted
        asm volatile("" : "+v"(li_e), "+v"(lq_e));                      // out of the hop loop they stay live through the K loop (spills)
        auto epilogue = [&](auto act_c) {
            constexpr int A = decltype(act_c)::value;
            auto fin = [&](const f32x4& v4, uint32_t& w0, uint32_t& w1) {
                float v[4] = {v4[0], v4[1], v4[2], v4[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = A == RECON_ACT_RELU ? fmaxf(v[e], 0.f) : (A == RECON_ACT_TANH ? tanh_fast(v[e]) : v[e]);
                w0 = pack_bf2(v[0], v[1]); w1 = pack_bf2(v[2], v[3]);
            };
            const int csw = ((li_e >> 1) & 3);
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                const int c = 16 * j + li_e;
#pragma unroll
                for (int r = 0; r + 1 < RT; r += 2) {
                    uint32_t a0, a1, b0, b1;
                    fin(acc[r][j], a0, a1);
                    fin(acc[r + 1][j], b0, b1);
                    const auto x0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto x1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    const int t = 16 * (wave * RT + r + (lq_e & 1)) + 8 * (lq_e >> 1);
                    const u32x4 v = u32x4{x0[0], x1[0], x0[1], x1[1]};
                    if (t < S) *reinterpret_cast<u32x4*>(sm + (t >> 5) * STEP + 64 * c + ((((t >> 3) & 3) ^ csw) << 4)) = v;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_hs, (p.hsave && c < cn && t < S) ? static_cast<uint32_t>(c * S + t) * 2u : kOOB, 0, 0);
                }
                if constexpr (RT & 1) {
                    uint32_t a0, a1;
                    fin(acc[RT - 1][j], a0, a1);
                    const int t = 16 * (wave * RT + RT - 1) + 4 * lq_e;
                    if (t < S) *reinterpret_cast<uint2*>(sm + (t >> 5) * STEP + 64 * c + ((((t >> 3) & 3) ^ csw) << 4) + 2 * (t & 7)) = make_uint2(a0, a1);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{a0, a1}, rs_hs, (p.hsave && c < cn && t < S) ? static_cast<uint32_t>(c * S + t) * 2u : kOOB, 0, 0);
                }
            }
        };
        if (p.act == RECON_ACT_RELU) epilogue(std::integral_constant<int, RECON_ACT_RELU>{});
        else if (p.act == RECON_ACT_TANH) epilogue(std::integral_constant<int, RECON_ACT_TANH>{});
        else epilogue(std::integral_constant<int, RECON_ACT_LINEAR>{});
        lds_barrier();                                                  // H^l complete
        // ---- relation_l = gather(h, heads) * gather(h, tails)   (models/models.py:270-273)
        uint16_t* outb = p.out + ((static_cast<int64_t>(b) * C + c0) * L + l) * p.dd;
        auto hv = [&](uint32_t q) { return bf2f(*reinterpret_cast<const uint16_t*>(sm + q)); };
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (tid + i * 512 < npairs) {
                const uint32_t* g = gtab + i * 5 * 512 + tid;
                *reinterpret_cast<uint32_t*>(outb + g[2048]) = pack_bf2(hv(g[0]) * hv(g[1024]), hv(g[512]) * hv(g[1536]));
            }
        if (npairs > NG * 512) {
            for (int pi = tid + NG * 512; pi < npairs; pi += 512) {
                const uint32_t it = 2u * pi, c = it / static_cast<uint32_t>(p.dd), x = it - c * p.dd;
                const int64_t io = b * p.idx_bs + static_cast<int64_t>(c0) * p.dd + it;
                const float v0 = hv(pos(c, static_cast<uint32_t>(p.head[io]))) * hv(pos(c, static_cast<uint32_t>(p.tail[io])));
                const float v1 = hv(pos(c, static_cast<uint32_t>(p.head[io + 1]))) * hv(pos(c, static_cast<uint32_t>(p.tail[io + 1])));
                *reinterpret_cast<uint32_t*>(outb + c * Ldd + x) = pack_bf2(v0, v1);
            }
        }
    }
}

// ================================================================================================ batched GEMM over the graphs
// C[b][m][n] = sum_k Q[b](m, k) . P[b](n, k), bf16 operands, fp32 accumulation, bf16 result with n contiguous.  P is the operand
// indexed by the output's contiguous index: its fragments are the MFMA's A operand, so that a lane holds four consecutive n of
// one m (8-byte stores).  Either operand is k-contiguous ([row][k], XK = false) or k-major ([k][row], XK = true):
//   forward hop        H^l[c][s]   = sum_t H^l-1[c][t] A_l[s][t]          P = A_l   (k-contiguous), Q = H^l-1 (k-contiguous), act
//   backward (d)       G[c][t]     = sum_s Y[c][s]     A_l[s][t]          P = A_l   (k-major),      Q = Y     (k-contiguous)
//   backward (c)       dA[s][t]    = sum_c Y[c][s]     H^l-1[c][t]        P = H^l-1 (k-major),      Q = Y     (k-major)
// Workgroup = WM x WN waves, each MT x NT tiles of 16 x 16; a stage = KSUB K steps of 32, double buffered, one barrier per stage;
// every operand piece (1 KiB = one wave instruction) is a global -> LDS copy without staging registers.  k-contiguous images:
// [row][32 k] with the 16-byte slots rotated by 2 (row >> 3) (conflict-free ds_read_b128); k-major images: [32 k][row], 32-byte
// column slots XORed with (k & 3) | ((k >> 3) & 1) << 2, fragments through the transposing read.  K tails read a page of zeros.
// Workgroup id -> (graph, tile) keeps the tiles of one graph on one XCD (its L2 serves the re-reads of the operands).
struct BGemmB16 {
    const uint16_t* P; int64_t p_bs; int32_t ldp;
    const uint16_t* Q; int64_t q_bs; int32_t ldq;
    uint16_t* C; int64_t c_bs; int32_t ldc;
    const uint16_t* zeros;
    int32_t M, N, K, batch, tiles_m, tiles_n, act;
    // PBLK: the P operand is a block adjacency read IN PLACE — P = transition tensors [batch][nodes (nodes - 1)][256], p_ident = identity
    // [256]: element (s, t) = P[e(s >> 4, t >> 4)][s & 15][t & 15], the identity on the diagonal blocks (models/models.py:240-259)
    int32_t p_nodes; const uint16_t* p_ident;
    // EPI_YPOST: the result G becomes Y = (G + relation gradient) . act'(H) before it is stored (structured gather indices: channel c reads
    // the dd consecutive columns from yp_head[c] and from yp_tail[c]): yp_h = H [batch][M][N], yp_g = grad_out + hop offset [batch][M][yp_ldg]
    const uint16_t* yp_h; const uint16_t* yp_g; int64_t yp_g_bs; int32_t yp_ldg, yp_dd; const int32_t* yp_head; const int32_t* yp_tail;
    // EPI_BLOCKS: the result (a gradient of a block adjacency) leaves in the transition tensors' layout: C = g_T [batch][nodes (nodes - 1)][256]
    // (or null), the diagonal blocks into blk_diag [batch][nodes][256] (or null)
    uint16_t* blk_diag;
};
enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_YPOST = 2, EPI_BLOCKS = 3 };
__device__ __forceinline__ int kc_off(int row, int kq) { return row * 64 + (((kq + 2 * (row >> 3)) & 3) << 4); }
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lo_p, const unsigned char* hi_p) {
    const i16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lo_p));
    const i16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(hi_p));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
constexpr int km_slots(int rows) { int s = 8; while (s * 16 < rows) s *= 2; return s; }      // 32-byte column slots per k row (power of two, >= 8)

// Persistent form.  A workgroup keeps ONE tile position (tm, tn) and walks graphs g = g0, g0 + gstride, ...: the copy plan (row / column
// offsets inside a graph) is computed once, and the copy pipeline runs across tile boundaries — with K = 512 a tile is only 16 K steps,
// and a pipeline that restarts per tile spends as long waiting for its first operands and draining its stores as it computes (first form
// of this kernel: 128 x 128 tiles, one workgroup per tile, vmcnt(0) + __syncthreads per stage: 0.20 of the dense peak at n = 32).
// NBUF LDS stages of one K step (32) each; at step s the wave waits until ITS copies of step s have landed (counted vmcnt: the copies of
// the NBUF - 2 following steps stay in flight), passes a raw s_barrier (everybody's copies of step s are there, everybody is done reading
// stage (s - 1) % NBUF), requests step s + NBUF - 1 into that stage, reads its fragments and issues the MFMAs.  Every wave issues exactly
// ND copy instructions per step (missing pieces and steps past the end copy a page of zeros to a scratch KiB): s_waitcnt takes an
// immediate.  After a tile's stores the next wait is vmcnt(0): loads and stores share the counter and complete out of order with respect
// to each other.
template <bool PK, bool QK, int WM, int WN, int MT, int NT, int NBUF, int EPI, bool PBLK>
__global__ void __launch_bounds__(64 * WM * WN, (MT * NT >= 32 ? 2 : 3) * WM * WN / 4 > 0 ? (MT * NT >= 32 ? 2 : 3) : 1) k_bgemm_b16(const BGemmB16 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int NWV = WM * WN, BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int PS = km_slots(BN), QS = km_slots(BM);
    constexpr int P_BYTES = PK ? 32 * PS * 32 : BN * 64, Q_BYTES = QK ? 32 * QS * 32 : BM * 64;
    constexpr int STAGE = P_BYTES + Q_BYTES;
    constexpr int NPP = P_BYTES / 1024, NPQ = Q_BYTES / 1024, NP = NPP + NPQ, ND = (NP + NWV - 1) / NWV;
    constexpr int SCRATCH = NBUF * STAGE;                               // 1 KiB nobody reads
    static_assert((NBUF - 2) * ND <= 63, "vmcnt field");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    // ---- tile position and graph sequence of this workgroup: XCD x = id & 7 owns graphs x, x + 8, ...; its workgroups are `groups` sets
    // of T tile positions, set q walking graphs x + 8 q, x + 8 (q + groups), ...: the tiles of one graph run side by side on one XCD
    const int T = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int grp = within / T, tile = within - grp * T;
    const int groups = (gridDim.x >> 3) / T;
    const int g0 = xcd + 8 * grp, gstride = 8 * groups;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int KT = (p.K + 31) >> 5;

    // ---- copy plan of this wave: piece pc = i NWV + wid of a stage; element offset d_a + (k0 + d_k) * stride inside the graph's operand
    int d_a[ND], d_k[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int pc = min(i * NWV + wid, NP - 1);
        const bool isp = pc < NPP;
        const int pi = isp ? pc : pc - NPP;
        const int s = 64 * pi + lane;
        const bool kmaj = isp ? PK : QK;
        const int row0 = isp ? n0 : m0, rows = isp ? p.N : p.M, ld = isp ? p.ldp : p.ldq;
        if (kmaj) {
            const int ns2 = 2 * (isp ? PS : QS);                        // 16-byte units per k row
            const int k = s / ns2, phys = s - k * ns2;
            const int t = (phys >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2));
            d_k[i] = k;
            d_a[i] = min(row0 + 16 * t + 8 * (phys & 1), ((rows + 7) & ~7) - 8);
        } else {
            const int rowL = s >> 2, kq = ((s & 3) - 2 * (rowL >> 3)) & 3;
            d_k[i] = 8 * kq;
            d_a[i] = min(row0 + rowL, rows - 1) * ((PBLK && isp) ? 1 : ld);      // block mode: the row index itself
        }
    }
    const uint16_t* zlane = p.zeros + 8 * lane;
    // issue side of the pipeline: step (graph gi, K step ki) goes to stage ib
    int gi = g0, ki = 0, ib = 0;
    const uint16_t* Pi = p.P + g0 * p.p_bs;
    const uint16_t* Qi = p.Q + g0 * p.q_bs;
    auto issue_piece = [&](int i) {                                     // copy instruction i of this wave for step (gi, ki)
        const bool live = gi < p.batch;
        const int pc = i * NWV + wid;                                   // wave-uniform
        const bool real = live && pc < NP;
        const bool isp = pc < NPP;
        const bool kmaj = isp ? PK : QK;
        const int k = 32 * ki + d_k[i];
        const uint16_t* base = isp ? Pi : Qi;
        int64_t off = kmaj ? static_cast<int64_t>(k) * (isp ? p.ldp : p.ldq) + d_a[i] : static_cast<int64_t>(d_a[i]) + k;
        if constexpr (PBLK) {
            if (isp) {                                                  // (row s, eight columns t ..) of the block adjacency: one 16-byte piece of a block row
                const int sr = PK ? k : d_a[i], tc = PK ? d_a[i] : k;
                const int bi = sr >> 4, bj = tc >> 4, e = bi * (p.p_nodes - 1) + (bj < bi ? bj : bj - 1);
                const int in_blk = (sr & 15) * 16 + (tc & 15);
                base = bi == bj ? p.p_ident : Pi;
                off = bi == bj ? in_blk : e * 256 + in_blk;
            }
        }
        const uint16_t* src = (real && k < p.K && !(kBGemmAblate & 1)) ? base + off : zlane;
        unsigned char* dst = sm + (real ? ib * STAGE + 1024 * pc : SCRATCH);
        // hidden from the compiler's LDS bookkeeping (recon_common.h): through the builtin every copy below was followed by a vmcnt(0) in front
        // of the next fragment read, and the loop-head barrier by another — the three-stage ring ran as a serial loop
        if constexpr (!(kBGemmAblate & 2)) dma16_to_lds(src, dst);
    };
    auto issue_advance = [&]() {
        if (++ki == KT) { ki = 0; gi += gstride; Pi += gstride * p.p_bs; Qi += gstride * p.q_bs; }
        ib = ib + 1 == NBUF ? 0 : ib + 1;
    };
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) issue_piece(i);
        issue_advance();
    };
    // ---- fragment addresses inside a stage
    int p_rd[NT], q_rd[MT];
    const int kk = 8 * lq + (li >> 2);                                  // k row of the first transposing read (+ 4 for the second)
    const int kx = (li >> 2) | ((lq & 1) << 2);                          // slot XOR of that row (the same for k + 4)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = wn * NT + j;
        p_rd[j] = PK ? kk * (PS * 32) + ((t ^ kx) << 5) + ((li & 3) << 3) : kc_off(16 * t + li, lq);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int t = wm * MT + i;
        q_rd[i] = P_BYTES + (QK ? kk * (QS * 32) + ((t ^ kx) << 5) + ((li & 3) << 3) : kc_off(16 * t + li, lq));
    }
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) issue();
    int cb = 0;
    // EPI_YPOST: the head / tail block starts of this lane's MT rows — the tile position is fixed, they are the same for every graph
    int yp_hb[MT], yp_tb[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + 16 * (wm * MT + i) + li;
        yp_hb[i] = (EPI == EPI_YPOST && m < p.M) ? p.yp_head[m] : 0;
        yp_tb[i] = (EPI == EPI_YPOST && m < p.M) ? p.yp_tail[m] : 0;
    }
#pragma unroll 1
    for (int g = g0; g < p.batch; g += gstride) {
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < KT; ++ks) {
            // the one wait of a step.  It also holds behind a tile's stores: requests of one kind complete in order and the counter counts both,
            // so "at most (NBUF - 2) ND requests outstanding" leaves at most that many COPIES outstanding — the youngest ones
            dma_wait<(NBUF - 2) * ND>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // fragments in the order of their use, the step's copy instructions (for step s + NBUF - 1: into the stage everybody has just
            // left) spread behind the first rows of MFMAs: a wave that issued 12 reads + 4 copies and only then its 32 MFMAs left the
            // matrix pipe idle for a third of the step (PMC: MFMA busy 33 %, waves parked 33 % — all waves of a workgroup are in the
            // same phase at the same time)
            const unsigned char* im = sm + cb * STAGE;
            bf16x8 pf[NT], qf[MT];
            auto read_q = [&](int i) {
                if constexpr (QK) qf[i] = tr_frag(im + q_rd[i], im + q_rd[i] + 4 * QS * 32);
                else qf[i] = *reinterpret_cast<const bf16x8*>(im + q_rd[i]);
            };
            read_q(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (PK) pf[j] = tr_frag(im + p_rd[j], im + p_rd[j] + 4 * PS * 32);
                else pf[j] = *reinterpret_cast<const bf16x8*>(im + p_rd[j]);
            }
            if constexpr (MT > 1) read_q(1);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i + 2 < MT) read_q(i + 2);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[i][j], 0, 0, 0);
                if (i < ND) issue_piece(i);
            }
#pragma unroll
            for (int i = MT; i < ND; ++i) issue_piece(i);
            issue_advance();
            // this wave's fragment reads are complete before it reaches the next barrier (the MFMAs have consumed them)
            cb = cb + 1 == NBUF ? 0 : cb + 1;
        }
        mfma_drain();
        // ---- MFMA result: rows (4 lq + r) = n, column li = m.
        // Stores are ISSUE bound (one store instruction occupies the CU's store path for ~70 cycles whatever it carries): neighbouring
        // column tiles exchange halves between lane rows (v_permlane16_swap) so that a lane holds EIGHT consecutive columns of one row —
        // half as many instructions, 16 bytes each, 64-byte runs per row.  EPI_YPOST works on the eight fp32 values before they are rounded.
        uint16_t* Cb = p.C + g * p.c_bs;
        auto put8 = [&](int m, int n, const float (&v)[8]) {            // eight consecutive columns n .. n + 7 of row m
            if (m >= p.M || n >= p.N) return;                           // N % 8 == 0
            const u32x4 w = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
            if constexpr (EPI == EPI_BLOCKS) {
                const int bi = m >> 4, bj = n >> 4, in_blk = (m & 15) * 16 + (n & 15);
                if (bi != bj) {
                    if (p.C) *reinterpret_cast<u32x4*>(Cb + (bi * (p.p_nodes - 1) + (bj < bi ? bj : bj - 1)) * 256 + in_blk) = w;
                } else if (p.blk_diag) {
                    *reinterpret_cast<u32x4*>(p.blk_diag + (static_cast<int64_t>(g) * p.p_nodes + bi) * 256 + in_blk) = w;
                }
            } else {
                *reinterpret_cast<u32x4*>(Cb + static_cast<int64_t>(m) * p.ldc + n) = w;
            }
        };
        auto store_tile = [&](auto act_c) {
            constexpr int A = decltype(act_c)::value;
            auto rsrc_of = [&](const uint16_t* q, int64_t elems) {
                return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(q), 0, static_cast<int>(elems * 2), 0x00020000);
            };
            const auto rs_h = rsrc_of(EPI == EPI_YPOST ? p.yp_h + g * p.c_bs : p.zeros, EPI == EPI_YPOST ? static_cast<int64_t>(p.M) * p.N : 0);
            const auto rs_g = rsrc_of(EPI == EPI_YPOST ? p.yp_g + g * p.yp_g_bs : p.zeros, EPI == EPI_YPOST ? static_cast<int64_t>(p.M) * p.yp_ldg : 0);
            // Y = (G + R) . act'(H): H's eight values; the relation term where the columns lie in the channel's head block (partner: the same
            // offset in its tail block) or tail block (partner in the head block) — out-of-range offsets read zeros.  Requests and arithmetic are
            // separate steps: the epilogue asks for the operands of a whole batch of column groups before it touches the first (issued group by
            // group in front of each group's stores, the compiler kept them there — a store may alias a later load — and a tile's epilogue was
            // sixteen dependent round trips: 1.21 ms for the product that takes 0.71 without this epilogue)
            struct YpOps { u32x4 h8, g8, p8; };
            auto yp_load = [&](int m, int hb, int tb, int n) {
                const bool ok = m < p.M && n < p.N;
                const bool in_h = ok && n >= hb && n < hb + p.yp_dd, in_t = ok && !in_h && n >= tb && n < tb + p.yp_dd;
                const int x = n - (in_h ? hb : tb);
                YpOps o;
                o.h8 = __builtin_amdgcn_raw_buffer_load_b128(rs_h, ok ? static_cast<uint32_t>(m * p.N + n) * 2u : kOOB, 0, 0);
                o.g8 = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (in_h || in_t) ? static_cast<uint32_t>(m * p.yp_ldg + x) * 2u : kOOB, 0, 0);
                o.p8 = __builtin_amdgcn_raw_buffer_load_b128(rs_h, (in_h || in_t) ? static_cast<uint32_t>(m * p.N + (in_h ? tb : hb) + x) * 2u : kOOB, 0, 0);
                return o;
            };
            auto yp_apply = [&](float (&v)[8], const YpOps& o) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t hw = o.h8[e], gw = o.g8[e], pw = o.p8[e];
                    const float h0 = bf2f(hw & 0xffffu), h1 = bf2f(hw >> 16);
                    const float y0 = v[2 * e] + bf2f(gw & 0xffffu) * bf2f(pw & 0xffffu), y1 = v[2 * e + 1] + bf2f(gw >> 16) * bf2f(pw >> 16);
                    v[2 * e] = y0 * (A == RECON_ACT_RELU ? (h0 > 0.f ? 1.f : 0.f) : (A == RECON_ACT_TANH ? 1.f - h0 * h0 : 1.f));
                    v[2 * e + 1] = y1 * (A == RECON_ACT_RELU ? (h1 > 0.f ? 1.f : 0.f) : (A == RECON_ACT_TANH ? 1.f - h1 * h1 : 1.f));
                }
            };
            auto finish8 = [&](int m, int n, float (&v)[8]) {
                if constexpr (EPI == EPI_ACT) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = A == RECON_ACT_RELU ? fmaxf(v[r], 0.f) : (A == RECON_ACT_TANH ? tanh_fast(v[r]) : v[r]);
                } else if constexpr (EPI == EPI_YPOST) {
                    const bool ok = m < p.M && n < p.N;
                    yp_apply(v, yp_load(m, ok ? p.yp_head[m] : 0, ok ? p.yp_tail[m] : 0, n));
                }
            };
            constexpr int NG = NT / 2;                                  // paired column groups per row of tiles
            constexpr int RB = EPI == EPI_YPOST ? (MT % 2 == 0 ? 2 : 1) : 1;                          // rows of tiles per batch of requests (4: the 256 x 256 tile spills)
#pragma unroll
            for (int i0 = 0; i0 < MT; i0 += RB) {
                YpOps ops[RB][NG > 0 ? NG : 1];
                if constexpr (EPI == EPI_YPOST) {
#pragma unroll
                    for (int ii = 0; ii < RB; ++ii) {
                        const int m = m0 + 16 * (wm * MT + i0 + ii) + li;
#pragma unroll
                        for (int jj = 0; jj < NG; ++jj)
                            ops[ii][jj] = yp_load(m, yp_hb[i0 + ii], yp_tb[i0 + ii], n0 + 16 * (wn * NT + 2 * jj + (lq & 1)) + 8 * (lq >> 1));
                    }
                }
#pragma unroll
            for (int i = i0; i < i0 + RB; ++i) {
                const int m = m0 + 16 * (wm * MT + i) + li;
#pragma unroll
                for (int j = 0; j + 1 < NT; j += 2) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                       // rows of lanes exchange: even rows keep tile j, odd rows tile j + 1
                        const auto x = __builtin_amdgcn_permlane16_swap(as_u(acc[i][j][r]), as_u(acc[i][j + 1][r]), false, false);
                        v[r] = as_f(x[0]); v[4 + r] = as_f(x[1]);
                    }
                    const int n = n0 + 16 * (wn * NT + j + (lq & 1)) + 8 * (lq >> 1);
                    if constexpr (EPI == EPI_YPOST) yp_apply(v, ops[i - i0][j / 2]);
                    else finish8(m, n, v);
                    put8(m, n, v);
                }
                if constexpr (NT & 1) {                                 // the unpaired tile: the same through a half-filled group (four columns)
                    const int n = n0 + 16 * (wn * NT + NT - 1) + 4 * lq;
                    if constexpr (EPI == EPI_PLAIN || EPI == EPI_ACT) {
                        float v[4] = {acc[i][NT - 1][0], acc[i][NT - 1][1], acc[i][NT - 1][2], acc[i][NT - 1][3]};
                        if constexpr (EPI == EPI_ACT) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = A == RECON_ACT_RELU ? fmaxf(v[r], 0.f) : (A == RECON_ACT_TANH ? tanh_fast(v[r]) : v[r]);
                        }
                        if (m < p.M && n < p.N) *reinterpret_cast<uint2*>(Cb + static_cast<int64_t>(m) * p.ldc + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    } else {
                        // lane rows exchange inside the tile: rows lq = 0 / 1 and 2 / 3 pair up, the even row takes both halves of eight columns
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t own = as_u(acc[i][NT - 1][r]);
                            const auto x = __builtin_amdgcn_permlane16_swap(own, own, false, false);      // x[1] on an even row = the odd partner's value
                            v[r] = as_f(own); v[4 + r] = as_f(x[1]);
                        }
                        const int n8 = n0 + 16 * (wn * NT + NT - 1) + 8 * (lq >> 1);
                        const bool even = (lq & 1) == 0;
                        finish8(even ? m : p.M, n8, v);                 // odd rows: masked out (their values travel with the even partner)
                        put8(even ? m : p.M, n8, v);
                    }
                }
            }
            }
        };
        if constexpr (kBGemmAblate & 4) { asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[MT - 1][NT - 1][3])); }
        else if ((EPI == EPI_ACT || EPI == EPI_YPOST) && p.act == RECON_ACT_RELU) store_tile(std::integral_constant<int, RECON_ACT_RELU>{});
        else if ((EPI == EPI_ACT || EPI == EPI_YPOST) && p.act == RECON_ACT_TANH) store_tile(std::integral_constant<int, RECON_ACT_TANH>{});
        else store_tile(std::integral_constant<int, RECON_ACT_LINEAR>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the dummy copies of the last steps target this workgroup's LDS
}

// ================================================================================================ Y_l of the backward
// Y[row] = (G[row] + relation gradient of hop `hop`) . act'(H[row]), row = (graph, channel): one wave per row, the row as fp32 in LDS,
// the 2 dd scatter terms as LDS float atomics (indices are arbitrary).  G == nullptr: the last hop (nothing arrives from above).
// The FIRST Y of a backward, Y_L = (relation gradient of the last hop) . act'(H^L), for block-structured gather indices: a row is zero except for the dd
// columns of its head block and of its tail block, so it is written in one pass of 16-byte pieces — zeros, or for the (at most 2 dd / 8) pieces inside a block
// grad_out[x] h[partner block + x] act'(h[this column]) from three 16-byte loads — instead of being assembled through LDS atomics (1.0 -> 0.3 ms at n = 32)
// A wave writes kYLastRows consecutive rows: with one row per wave (1 M rows at n = 32 = 254 k workgroups of a few hundred cycles each) the launch ran at the
// dispatcher's pace, 2.3 TB/s of zeros
constexpr int kYLastRows = 8;
__global__ void __launch_bounds__(256) k_prop_b16_y_last(const uint16_t* __restrict__ Hl, const int32_t* __restrict__ hblk, const int32_t* __restrict__ tblk,
                                                          const uint16_t* __restrict__ gout, uint16_t* __restrict__ Y, int64_t rows, int32_t C, int32_t S,
                                                          int32_t L, int32_t dd, int32_t hop, int32_t act) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * 4 + w) * kYLastRows;
    if (row0 >= rows) return;
    // the block starts of the wave's rows with ONE request (lane r: row r), then every row's operands before the first store: as a loop over rows
    // each row cost two dependent round trips (block starts -> operands -> store)
    const int64_t myrow = row0 + (lane & (kYLastRows - 1));
    const int myc = static_cast<int>((myrow < rows ? myrow : row0) % C);
    const int hb_l = hblk[myc], tb_l = tblk[myc];
    // broadcast HERE, with every lane active: read inside the column loop (where only S / 8 lanes are), the compiler sinks the two loads into
    // that region too and the lanes that hold rows 4 .. 7 never execute them (S = 32: wrong block starts, caught by the n = 2 oracle test)
    int hbs[kYLastRows], tbs[kYLastRows];
#pragma unroll
    for (int rr = 0; rr < kYLastRows; ++rr) { hbs[rr] = __builtin_amdgcn_readlane(hb_l, rr); tbs[rr] = __builtin_amdgcn_readlane(tb_l, rr); }
    for (int t8 = lane; t8 < (S >> 3); t8 += 64) {
        const int col = 8 * t8;
        u32x4 o[kYLastRows];
#pragma unroll
        for (int rr = 0; rr < kYLastRows; ++rr) {
            const int64_t row = row0 + rr;
            const int hb = hbs[rr], tb = tbs[rr];
            const bool in_h = col >= hb && col < hb + dd, in_t = col >= tb && col < tb + dd;
            o[rr] = u32x4{0u, 0u, 0u, 0u};
            if ((in_h || in_t) && row < rows) {
                const uint16_t* h = Hl + row * S;
                const uint16_t* go = gout + row * (static_cast<int64_t>(L) * dd) + static_cast<int64_t>(hop) * dd;
                const int x = col - (in_h ? hb : tb);
                const u32x4 g8 = *reinterpret_cast<const u32x4*>(go + x);
                const u32x4 p8 = *reinterpret_cast<const u32x4*>(h + (in_h ? tb : hb) + x);
                const u32x4 m8 = *reinterpret_cast<const u32x4*>(h + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t gw = g8[e], pw = p8[e], mw = m8[e];
                    o[rr][e] = pack_bf2((bf2f(gw & 0xffffu) * bf2f(pw & 0xffffu)) * act_bwd(bf2f(mw & 0xffffu), act),
                                        (bf2f(gw >> 16) * bf2f(pw >> 16)) * act_bwd(bf2f(mw >> 16), act));
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < kYLastRows; ++rr)
            if (row0 + rr < rows) *reinterpret_cast<u32x4*>(Y + (row0 + rr) * S + col) = o[rr];
    }
}

__global__ void __launch_bounds__(256) k_prop_b16_ypost(const uint16_t* __restrict__ G, const uint16_t* __restrict__ Hl, const int64_t* __restrict__ head,
                                                         const int64_t* __restrict__ tail, int64_t idx_bs, const uint16_t* __restrict__ gout,
                                                         uint16_t* __restrict__ Y, int64_t rows, int32_t C, int32_t S, int32_t L, int32_t dd, int32_t hop,
                                                         int32_t act) {
    extern __shared__ float rowbuf[];                                   // [4][2][S]: gradient row, state row
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + w;
    if (row >= rows) return;
    const int64_t b = row / C;
    const int c = static_cast<int>(row - b * C);
    float* buf = rowbuf + w * 2 * S;
    float* hb = buf + S;
    const uint16_t* g = G ? G + row * S : nullptr;
    const uint16_t* h = Hl + row * S;
    for (int t8 = lane; t8 < (S >> 3); t8 += 64) {
        const u32x4 hv = *reinterpret_cast<const u32x4*>(h + 8 * t8);
        u32x4 gv = u32x4{0u, 0u, 0u, 0u};
        if (g) gv = *reinterpret_cast<const u32x4*>(g + 8 * t8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t hw = hv[e], gw = gv[e];
            hb[8 * t8 + 2 * e] = bf2f(hw & 0xffffu); hb[8 * t8 + 2 * e + 1] = bf2f(hw >> 16);
            buf[8 * t8 + 2 * e] = bf2f(gw & 0xffffu); buf[8 * t8 + 2 * e + 1] = bf2f(gw >> 16);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                                 // this wave's LDS writes have landed (one wave per row: no barrier)
    const int64_t io = b * idx_bs + static_cast<int64_t>(c) * dd;
    const uint16_t* go = gout + row * (static_cast<int64_t>(L) * dd) + static_cast<int64_t>(hop) * dd;
    for (int x = lane; x < dd; x += 64) {                               // out = h[head] * h[tail]  (models/models.py:270-273)
        const int hi = static_cast<int>(head[io + x]), ti = static_cast<int>(tail[io + x]);
        const float gv = bf2f(go[x]);
        atomicAdd(buf + hi, gv * hb[ti]);
        atomicAdd(buf + ti, gv * hb[hi]);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    uint16_t* y = Y + row * S;
    for (int t8 = lane; t8 < (S >> 3); t8 += 64) {
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = 8 * t8 + 2 * e;
            o[e] = pack_bf2(buf[t] * act_bwd(hb[t], act), buf[t + 1] * act_bwd(hb[t + 1], act));
        }
        *reinterpret_cast<u32x4*>(y + 8 * t8) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// relation_l for every hop out of the saved states [L][B][C][S]: thread = one element of out [B][C][L][dd]
__global__ void __launch_bounds__(256) k_prop_b16_gather(const uint16_t* __restrict__ hs, const int64_t* __restrict__ head, const int64_t* __restrict__ tail,
                                                          int64_t idx_bs, uint16_t* __restrict__ out, int64_t total, int32_t B, int32_t C, int32_t S, int32_t L,
                                                          int32_t dd) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = static_cast<int>(idx % dd);
    const int l = static_cast<int>((idx / dd) % L);
    const int64_t bc = idx / (static_cast<int64_t>(dd) * L);
    const int64_t b = bc / C;
    const int c = static_cast<int>(bc - b * C);
    const int64_t io = b * idx_bs + static_cast<int64_t>(c) * dd + x;
    const uint16_t* h = hs + ((static_cast<int64_t>(l) * B + b) * C + c) * S;
    const float v = bf2f(h[head[io]]) * bf2f(h[tail[io]]);
    out[idx] = static_cast<uint16_t>(pack_bf2(v, 0.f) & 0xffffu);
}

// ================================================================================================ P1 in bf16
// A[b, i dd + r, j dd + c] = T[b, e(i, j), r dd + c] (identity on the diagonal blocks); VEC = 8 columns per thread where dd % 8 == 0
template <int VEC>
__global__ void __launch_bounds__(256) k_block_adj_b16_fwd(const uint16_t* __restrict__ T, const uint16_t* __restrict__ I, int32_t n, int32_t dd,
                                                            uint16_t* __restrict__ A) {
    const int S = n * dd, SV = S / VEC;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= S * SV) return;
    const int b = blockIdx.y;
    const int row = t / SV, col = (t - row * SV) * VEC;
    const int i = row / dd, r = row - i * dd, j = col / dd, c = col - j * dd;
    const uint16_t* src = i == j ? I + r * dd + c
                                 : T + (static_cast<int64_t>(b) * n * (n - 1) + i * (n - 1) + (j < i ? j : j - 1)) * dd * dd + r * dd + c;
    uint16_t* dst = A + static_cast<int64_t>(b) * S * S + static_cast<int64_t>(row) * S + col;
    if constexpr (VEC == 8) *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
    else *dst = *src;
}
template <int VEC>
__global__ void __launch_bounds__(256) k_block_adj_b16_bwd_T(const uint16_t* __restrict__ gA, int32_t n, int32_t dd, uint16_t* __restrict__ gT) {
    const int S = n * dd, d2 = dd * dd, Cn = n * (n - 1);
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Cn * (d2 / VEC)) return;
    const int b = blockIdx.y;
    const int e = t / (d2 / VEC), rc = (t - e * (d2 / VEC)) * VEC;
    const int r = rc / dd, c = rc - r * dd;
    const int i = e / (n - 1);
    int j = e - i * (n - 1);
    if (j >= i) ++j;
    const uint16_t* src = gA + static_cast<int64_t>(b) * S * S + static_cast<int64_t>(i * dd + r) * S + j * dd + c;
    uint16_t* dst = gT + (static_cast<int64_t>(b) * Cn + e) * d2 + rc;
    if constexpr (VEC == 8) *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
    else *dst = *src;
}
// d loss / d identity: slice `blk` of the (graph, node) pairs, thread e = (r, c), fp32 partial sums [slice][dd dd]; k_sum_rows_b16 adds the
// slices in a fixed order and rounds once
constexpr int kIdentSlicesB16 = 256;
__global__ void __launch_bounds__(256) k_block_adj_b16_bwd_I(const uint16_t* __restrict__ gA, int32_t B, int32_t n, int32_t dd, float* __restrict__ partial) {
    const int64_t S = 1LL * n * dd, pairs = 1LL * B * n;
    const int64_t per = (pairs + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = blockIdx.x * per, t1 = min(pairs, t0 + per);
    for (int e = threadIdx.x; e < dd * dd; e += 256) {
        const int r = e / dd, c = e % dd;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        auto at = [&](int64_t t) { const int64_t b = t / n; const int i = static_cast<int>(t % n); return bf2f(gA[(b * S + i * dd + r) * S + i * dd + c]); };
        int64_t t = t0;
        for (; t + 4 <= t1; t += 4) { s0 += at(t); s1 += at(t + 1); s2 += at(t + 2); s3 += at(t + 3); }
        for (; t < t1; ++t) s0 += at(t);
        partial[static_cast<int64_t>(blockIdx.x) * dd * dd + e] = (s0 + s1) + (s2 + s3);
    }
}
__global__ void __launch_bounds__(256) k_sum_rows_b16(const float* __restrict__ partial, int32_t nrows, int32_t O, uint16_t* __restrict__ out) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              // eight loads in flight: one chain of 256 dependent round trips took 59 us
    int r = 0;
    for (; r + 8 <= nrows; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += partial[static_cast<int64_t>(r + u) * O + o];
    }
    for (; r < nrows; ++r) a[0] += partial[static_cast<int64_t>(r) * O + o];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    out[o] = static_cast<uint16_t>(pack_bf2(s, 0.f) & 0xffffu);
}

// ================================================================================================ host side
bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

int num_cus_b16() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

int check_b16(const recon_prop_b16_args* a) {
    if (!a) return RECON_ERR_INVALID;
    if (a->B < 0 || a->C <= 0 || a->S <= 0 || a->L <= 0 || a->dd <= 0) return RECON_ERR_INVALID;
    if (a->L > kMaxHops || a->B > 65535) return RECON_ERR_UNSUPPORTED;
    if (a->act < 0 || a->act > 2) return RECON_ERR_INVALID;
    if (!a->h0 || !a->head_idx || !a->tail_idx || !a->out) return RECON_ERR_INVALID;
    if (a->trans) {
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

size_t fused_lds(int nks, int ntc) { return 2ull * nks * ntc * 16 * 64; }

// 1: the fused small-state kernel, 2: the batched-GEMM form (materialised adjacency, saved states, zeros page), 0: neither
int form_b16(const recon_prop_b16_args* a, bool check_ptrs) {
    if (a->S % 8 != 0 || a->S < 8) return 0;
    const bool blk = a->trans != nullptr;
    if (check_ptrs) {
        if (!al16(a->h0) || (a->h0_batch_stride % 8) != 0 || (a->h_saved && !al16(a->h_saved)) || (reinterpret_cast<uintptr_t>(a->out) & 3)) return 0;
        for (int l = 0; l < a->L; ++l) if (!al16(blk ? a->trans[l] : a->adj[l])) return 0;
        if (blk && !al16(a->identity)) return 0;
    }
    const char* env = cfg(CFG_PROP_B16);                                // "g": the GEMM form everywhere (tests, A/B)
    const bool force_gemm = env && env[0] == 'g';
    if (!force_gemm && a->S % 16 == 0 && a->S <= 160 && a->C <= 96 && (a->dd % 2) == 0 && ((a->C * a->dd) & 1) == 0) return 1;
    // wide states in LDS: S = 32 NKS, NKS even in 6 .. 16 (S % 64 == 0, 192 <= S <= 512), even dd; inference or training alike
    if (!force_gemm && !(env && env[0] == 'n') && a->S % 64 == 0 && a->S >= 192 && a->S <= 512 && (a->dd % 2) == 0) return 3;
    if (static_cast<int64_t>(a->C) * a->S >= (1LL << 31) || static_cast<int64_t>(a->S) * a->S >= (1LL << 31)) return 0;
    return 2;                                                           // block mode too: the GEMMs read the transition tensors in place
}

template <bool PK, bool QK, int EPI, bool PBLK, int WM, int WN, int MT, int NT, int NBUF>
int launch_bgemm_cfg(BGemmB16 g, hipStream_t st) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr size_t lds = static_cast<size_t>(NBUF) * ((PK ? 32 * km_slots(BN) * 32 : BN * 64) + (QK ? 32 * km_slots(BM) * 32 : BM * 64)) + 1024;
    g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = (g.N + BN - 1) / BN;
    const int64_t T = static_cast<int64_t>(g.tiles_m) * g.tiles_n;
    auto kern = k_bgemm_b16<PK, QK, WM, WN, MT, NT, NBUF, EPI, PBLK>;
    static int occ_dev[16] = {};                                        // per instantiation and device (the attribute is per device too)
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& occ = occ_dev[dev & 15];
    if (occ == 0) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WM * WN, lds) != hipSuccess || occ < 1) occ = 1;
    }
    // resident capacity per XCD / tile positions = sets of tile positions per XCD; no more sets than graphs per XCD
    const int64_t cap_xcd = static_cast<int64_t>(occ) * num_cus_b16() / 8;
    int64_t groups = cap_xcd / T;
    if (groups < 1) groups = 1;
    const int64_t per_xcd = ceil_div64(g.batch, 8);
    if (groups > per_xcd) groups = per_xcd;
    const int64_t rounds = ceil_div64(per_xcd, groups);
    groups = ceil_div64(per_xcd, rounds);                               // the same number of rounds with fewer idle sets
    const int64_t nblk = 8 * groups * T;
    if (nblk >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nblk)), dim3(64 * WM * WN), lds, st, g);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// Tile shapes: a whole graph of <= 9 nodes per workgroup (144 x 144, 9 waves); 256 x 256 with 8 waves of 128 x 64 otherwise (the copies come
// from L2 at ~30 TB/s chip-wide: bytes per flop halve against 128 x 128, measured 657 -> 495 TF/s the other way); RECON_BGEMM_CFG=a: 128 x 128.
template <bool PK, bool QK, int EPI, bool PBLK>
int launch_bgemm(const BGemmB16& g, hipStream_t st) {
    if (g.M <= 144 && g.N <= 144) return launch_bgemm_cfg<PK, QK, EPI, PBLK, 3, 3, 3, 3, 3>(g, st);
    const char tile = cfg_char(CFG_BGEMM_CFG);
    if (tile == 'a') return launch_bgemm_cfg<PK, QK, EPI, PBLK, 2, 2, 4, 4, 3>(g, st);
    if (tile == 'h') return launch_bgemm_cfg<PK, QK, EPI, PBLK, 1, 4, 8, 4, 3>(g, st);
    return launch_bgemm_cfg<PK, QK, EPI, PBLK, 2, 4, 8, 4, 3>(g, st);
}

int fwd_fused(const recon_prop_b16_args* a, hipStream_t st) {
    PropB16K p{};
    const bool blk = a->trans != nullptr;
    for (int l = 0; l < kMaxHops; ++l) {
        p.adj[l] = (l < a->L && !blk) ? static_cast<const uint16_t*>(a->adj[l]) : nullptr;
        p.trans[l] = (l < a->L && blk) ? static_cast<const uint16_t*>(a->trans[l]) : nullptr;
    }
    p.identity = blk ? static_cast<const uint16_t*>(a->identity) : nullptr;
    p.h0 = static_cast<const uint16_t*>(a->h0); p.h0_bs = a->h0_batch_stride;
    p.head = a->head_idx; p.tail = a->tail_idx; p.idx_bs = a->idx_batch_stride;
    p.out = static_cast<uint16_t*>(a->out); p.hsave = static_cast<uint16_t*>(a->h_saved);
    p.B = a->B; p.C = a->C; p.S = a->S; p.L = a->L; p.dd = a->dd; p.act = a->act;
    const int nks = (a->S + 31) / 32, ntc = (a->C + 15) / 16, nw = a->S / 16;
    const size_t lds = fused_lds(nks, ntc);
#define CALL_F(K_, N_, X_)                                                                                                              \
    do {                                                                                                                                \
        auto kern = k_prop_b16_fwd<K_, N_, X_>;                                                                                         \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                       static_cast<int>(lds));                                                         \
        int occ = 0;                                                                                                                    \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * nw, lds) != hipSuccess || occ < 1) occ = 1;                   \
        const int64_t cap = static_cast<int64_t>(occ) * num_cus_b16();                                                                  \
        const int64_t rounds = ceil_div64(a->B, cap);                                                                                   \
        const int grid = static_cast<int>(ceil_div64(a->B, rounds));      /* every workgroup walks `rounds` graphs (the last ones one less) */ \
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(64 * nw), lds, st, p);                                         \
    } while (0)
    if (blk) {                                                          // block mode: S = 16 n, C = n (n - 1): one (NKS, NTC) per n
        switch (nw) { case 2: CALL_F(1, 1, true); break; case 3: case 4: CALL_F(2, 1, true); break; case 5: case 6: CALL_F(3, 2, true); break;
                      case 7: CALL_F(4, 3, true); break; case 8: CALL_F(4, 4, true); break; case 9: CALL_F(5, 5, true); break; case 10: CALL_F(5, 6, true); break;
                      default: return RECON_ERR_UNSUPPORTED; }
    } else {
#define CALL_FN(K_)                                                                                                                     \
    switch (ntc) { case 1: CALL_F(K_, 1, false); break; case 2: CALL_F(K_, 2, false); break; case 3: CALL_F(K_, 3, false); break; case 4: CALL_F(K_, 4, false); break; \
                   case 5: CALL_F(K_, 5, false); break; default: CALL_F(K_, 6, false); break; }
        switch (nks) { case 1: CALL_FN(1); break; case 2: CALL_FN(2); break; case 3: CALL_FN(3); break; case 4: CALL_FN(4); break; default: CALL_FN(5); break; }
#undef CALL_FN
    }
#undef CALL_F
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

int fwd_wide(const recon_prop_b16_args* a, hipStream_t st) {
    PropB16K p{};
    const bool blk = a->trans != nullptr;
    for (int l = 0; l < kMaxHops; ++l) {
        p.adj[l] = (l < a->L && !blk) ? static_cast<const uint16_t*>(a->adj[l]) : nullptr;
        p.trans[l] = (l < a->L && blk) ? static_cast<const uint16_t*>(a->trans[l]) : nullptr;
    }
    p.identity = blk ? static_cast<const uint16_t*>(a->identity) : nullptr;
    p.h0 = static_cast<const uint16_t*>(a->h0); p.h0_bs = a->h0_batch_stride;
    p.head = a->head_idx; p.tail = a->tail_idx; p.idx_bs = a->idx_batch_stride;
    p.out = static_cast<uint16_t*>(a->out); p.hsave = static_cast<uint16_t*>(a->h_saved);
    p.B = a->B; p.C = a->C; p.S = a->S; p.L = a->L; p.dd = a->dd; p.act = a->act;
    const int nks = a->S / 32, nchunk = (a->C + 127) / 128;
    const size_t lds = static_cast<size_t>(nks) * 128 * 64 + 2 * 5 * 512 * sizeof(uint32_t);
    const int64_t nblk = ceil_div64(a->B, 8) * 8 * nchunk;
    if (nblk >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;
#define CALL_W(K_, R_, X_)                                                                                                              \
    do {                                                                                                                                \
        auto kern = k_prop_b16_fwd_wide<K_, R_, X_>;                                                                                    \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                       static_cast<int>(lds));                                                         \
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nblk)), dim3(512), lds, st, p, nchunk);                                     \
    } while (0)
#define CALL_WK(X_)                                                                                                                     \
    switch (nks) { case 6: CALL_W(6, 2, X_); break; case 8: CALL_W(8, 2, X_); break; case 10: CALL_W(10, 3, X_); break; case 12: CALL_W(12, 3, X_); break; \
                   case 14: CALL_W(14, 4, X_); break; case 16: CALL_W(16, 4, X_); break; default: return RECON_ERR_UNSUPPORTED; }
    if (blk) { CALL_WK(true); } else { CALL_WK(false); }
#undef CALL_WK
#undef CALL_W
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

int fwd_gemm(const recon_prop_b16_args* a, hipStream_t st) {
    if (!a->h_saved || !a->zeros || !al16(a->zeros)) return RECON_ERR_INVALID;
    const int32_t B = a->B, C = a->C, S = a->S, L = a->L;
    const int64_t CS = 1LL * C * S, BCS = CS * B;
    uint16_t* hs = static_cast<uint16_t*>(a->h_saved);
    for (int l = 1; l <= L; ++l) {
        BGemmB16 g{};
        const bool blk = a->trans != nullptr;
        g.P = static_cast<const uint16_t*>(blk ? a->trans[l - 1] : a->adj[l - 1]); g.p_bs = blk ? 1LL * C * 256 : 1LL * S * S; g.ldp = S;
        g.p_nodes = blk ? S / 16 : 0; g.p_ident = static_cast<const uint16_t*>(a->identity);
        g.Q = l == 1 ? static_cast<const uint16_t*>(a->h0) : hs + (l - 2) * BCS; g.q_bs = l == 1 ? a->h0_batch_stride : CS; g.ldq = S;
        g.C = hs + (l - 1) * BCS; g.c_bs = CS; g.ldc = S;
        g.zeros = static_cast<const uint16_t*>(a->zeros);
        g.M = C; g.N = S; g.K = S; g.batch = B; g.act = a->act;
        const int rc = blk ? launch_bgemm<false, false, EPI_ACT, true>(g, st) : launch_bgemm<false, false, EPI_ACT, false>(g, st);
        if (rc != RECON_OK) return rc;
    }
    const int64_t total = 1LL * B * C * L * a->dd;
    hipLaunchKernelGGL(k_prop_b16_gather, dim3(static_cast<unsigned>(ceil_div64(total, 256))), dim3(256), 0, st, hs, a->head_idx, a->tail_idx,
                       a->idx_batch_stride, static_cast<uint16_t*>(a->out), total, B, C, S, L, a->dd);
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace
}  // namespace recon

using namespace recon;

extern "C" int recon_propagate_b16_form(const recon_prop_b16_args* a) {
    if (!a || a->B < 0 || a->C <= 0 || a->S <= 0 || a->L <= 0 || a->L > kMaxHops || a->dd <= 0) return 0;
    if (a->trans && (a->dd != 16 || a->S % 16 != 0 || a->C != (a->S / 16) * (a->S / 16 - 1))) return 0;
    return form_b16(a, false);
}

extern "C" int recon_propagate_b16_fwd(const recon_prop_b16_args* a, recon_stream_t stream) {
    const int rc = check_b16(a);
    if (rc != RECON_OK) return rc;
    if (a->B == 0) return RECON_OK;
    const int form = form_b16(a, true);
    if (form == 1) return fwd_fused(a, as_stream(stream));
    if (form == 2) return fwd_gemm(a, as_stream(stream));
    if (form == 3) return fwd_wide(a, as_stream(stream));
    return RECON_ERR_UNSUPPORTED;
}

// Both products of every hop as batched GEMMs over the graphs.  Y_l = (G_l+1 + relation gradient) . act'(H^l) is formed in the epilogue of the
// product that delivers G_l+1 where the gather indices are blocks of dd consecutive columns (head_blk / tail_blk: what the GP-GNN models
// use), else in place by k_prop_b16_ypost.  The L products of kind (d) alternate between `ws` and g_h so that the last one (d loss / d h^0
// per graph) lands in g_h.  BLOCK MODE (fwd.trans): A_l is read out of the transition tensors in place, d A_l leaves in their layout
// (g_trans) and its diagonal blocks are summed into g_identity (fp32, fixed order, rounded once) — no adjacency is ever materialised.
namespace {
__global__ void __launch_bounds__(256) k_diag_partial_b16(const uint16_t* __restrict__ diag, int64_t nblk, float* __restrict__ partial) {
    const int64_t per = (nblk + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = blockIdx.x * per, t1 = min(nblk, t0 + per);
    const int e = threadIdx.x;                                          // 256 elements of a 16 x 16 block
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t t = t0;
    for (; t + 4 <= t1; t += 4) { s0 += bf2f(diag[t * 256 + e]); s1 += bf2f(diag[(t + 1) * 256 + e]); s2 += bf2f(diag[(t + 2) * 256 + e]); s3 += bf2f(diag[(t + 3) * 256 + e]); }
    for (; t < t1; ++t) s0 += bf2f(diag[t * 256 + e]);
    partial[static_cast<int64_t>(blockIdx.x) * 256 + e] = (s0 + s1) + (s2 + s3);
}
}  // namespace

extern "C" size_t recon_propagate_b16_bwd_diag_elems(const recon_prop_b16_args* a) {
    if (!a || !a->trans || a->S % 16 != 0 || a->B <= 0 || a->L <= 0) return 0;
    return static_cast<size_t>(a->L) * a->B * (a->S / 16) * 256;
}

extern "C" int recon_propagate_b16_bwd(const recon_prop_b16_bwd_args* ba, recon_stream_t stream) {
    if (!ba) return RECON_ERR_INVALID;
    const recon_prop_b16_args* a = &ba->fwd;
    const int rc0 = check_b16(a);
    if (rc0 != RECON_OK) return rc0;
    if (!a->h_saved || !ba->grad_out || !ba->g_h || !ba->ws || !a->zeros) return RECON_ERR_INVALID;
    if (a->B == 0) return RECON_OK;
    const int32_t B = a->B, C = a->C, S = a->S, L = a->L;
    const bool blk = a->trans != nullptr;
    if (blk && ba->g_identity && (!ba->diag_ws || !ba->ident_ws)) return RECON_ERR_INVALID;
    if (S % 8 != 0 || !al16(a->h0) || (a->h0_batch_stride % 8) != 0 || !al16(a->h_saved) || !al16(ba->g_h) || !al16(ba->ws) || !al16(a->zeros) ||
        static_cast<int64_t>(C) * S >= (1LL << 31) || static_cast<int64_t>(S) * S >= (1LL << 31) || 8ull * S * sizeof(float) > 64 * 1024)
        return RECON_ERR_UNSUPPORTED;
    for (int l = 0; l < L; ++l) {
        if (!al16(blk ? a->trans[l] : a->adj[l])) return RECON_ERR_UNSUPPORTED;
        if (blk ? (ba->g_trans && ba->g_trans[l] && !al16(ba->g_trans[l])) : (ba->g_adj && ba->g_adj[l] && !al16(ba->g_adj[l]))) return RECON_ERR_UNSUPPORTED;
    }
    if (blk && (!al16(a->identity) || (ba->diag_ws && !al16(ba->diag_ws)))) return RECON_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int64_t CS = 1LL * C * S, BCS = CS * B, rows = 1LL * B * C;
    const int nn = S / 16;
    const uint16_t* hs = static_cast<const uint16_t*>(a->h_saved);
    const uint16_t* gout = static_cast<const uint16_t*>(ba->grad_out);
    uint16_t* bufY = static_cast<uint16_t*>((L & 1) ? ba->ws : ba->g_h);
    uint16_t* bufG = static_cast<uint16_t*>((L & 1) ? ba->g_h : ba->ws);
    const dim3 pgrid(static_cast<unsigned>(ceil_div64(rows, 4)));
    const dim3 ylgrid(static_cast<unsigned>(ceil_div64(rows, 4 * kYLastRows)));
    const size_t plds = 8ull * S * sizeof(float);
    // the fused Y: structured gather indices shared by the batch, dd and L dd multiples of 8 (16-byte pieces of grad_out rows)
    const bool fuse_y = ba->head_blk && ba->tail_blk && a->idx_batch_stride == 0 && (a->dd % 8) == 0 && cfg_char(CFG_PROP_B16_YPOST) != 'k';
    if (fuse_y)
        hipLaunchKernelGGL(k_prop_b16_y_last, ylgrid, dim3(256), 0, st, hs + (L - 1) * BCS, ba->head_blk, ba->tail_blk, gout, bufY, rows, C, S, L, a->dd, L - 1,
                           a->act);
    else
        hipLaunchKernelGGL(k_prop_b16_ypost, pgrid, dim3(256), plds, st, nullptr, hs + (L - 1) * BCS, a->head_idx, a->tail_idx, a->idx_batch_stride, gout, bufY,
                           rows, C, S, L, a->dd, L - 1, a->act);
    for (int l = L; l >= 1; --l) {
        const uint16_t* Hprev = l == 1 ? static_cast<const uint16_t*>(a->h0) : hs + (l - 2) * BCS;
        const int64_t hprev_bs = l == 1 ? a->h0_batch_stride : CS;
        void* gdst = blk ? (ba->g_trans ? ba->g_trans[l - 1] : nullptr) : (ba->g_adj ? ba->g_adj[l - 1] : nullptr);
        uint16_t* diag = (blk && ba->g_identity) ? static_cast<uint16_t*>(ba->diag_ws) + static_cast<int64_t>(l - 1) * B * nn * 256 : nullptr;
        if (gdst || diag) {                                             // (c): dA[s][t] = sum_c Y[c][s] H^l-1[c][t]
            BGemmB16 g{};
            g.P = Hprev; g.p_bs = hprev_bs; g.ldp = S;
            g.Q = bufY; g.q_bs = CS; g.ldq = S;
            g.C = static_cast<uint16_t*>(gdst); g.c_bs = blk ? 1LL * C * 256 : 1LL * S * S; g.ldc = S;
            g.zeros = static_cast<const uint16_t*>(a->zeros);
            g.M = S; g.N = S; g.K = C; g.batch = B; g.act = 0;
            g.p_nodes = nn; g.blk_diag = diag;
            const int rc = blk ? launch_bgemm<true, true, EPI_BLOCKS, false>(g, st) : launch_bgemm<true, true, EPI_PLAIN, false>(g, st);
            if (rc != RECON_OK) return rc;
        }
        {                                                               // (d): G[c][t] = sum_s Y[c][s] A_l[s][t]   (+ the next Y in its epilogue)
            BGemmB16 g{};
            g.P = static_cast<const uint16_t*>(blk ? a->trans[l - 1] : a->adj[l - 1]); g.p_bs = blk ? 1LL * C * 256 : 1LL * S * S; g.ldp = S;
            g.p_nodes = blk ? nn : 0; g.p_ident = static_cast<const uint16_t*>(a->identity);
            g.Q = bufY; g.q_bs = CS; g.ldq = S;
            g.C = bufG; g.c_bs = CS; g.ldc = S;
            g.zeros = static_cast<const uint16_t*>(a->zeros);
            g.M = C; g.N = S; g.K = S; g.batch = B; g.act = a->act;
            const bool yp = fuse_y && l > 1;
            if (yp) {
                g.yp_h = hs + (l - 2) * BCS; g.yp_g = gout + static_cast<int64_t>(l - 2) * a->dd; g.yp_g_bs = 1LL * C * L * a->dd; g.yp_ldg = L * a->dd;
                g.yp_dd = a->dd; g.yp_head = ba->head_blk; g.yp_tail = ba->tail_blk;
            }
            const int rc = blk ? (yp ? launch_bgemm<true, false, EPI_YPOST, true>(g, st) : launch_bgemm<true, false, EPI_PLAIN, true>(g, st))
                               : (yp ? launch_bgemm<true, false, EPI_YPOST, false>(g, st) : launch_bgemm<true, false, EPI_PLAIN, false>(g, st));
            if (rc != RECON_OK) return rc;
        }
        if (l > 1 && !fuse_y)
            hipLaunchKernelGGL(k_prop_b16_ypost, pgrid, dim3(256), plds, st, bufG, hs + (l - 2) * BCS, a->head_idx, a->tail_idx, a->idx_batch_stride, gout,
                               bufG, rows, C, S, L, a->dd, l - 2, a->act);
        uint16_t* t = bufY; bufY = bufG; bufG = t;
    }
    if (blk && ba->g_identity) {                                        // every hop's diagonal blocks -> one [16][16] sum
        const int64_t nb = static_cast<int64_t>(L) * B * nn;
        const int slices = static_cast<int>(nb < kIdentSlicesB16 ? nb : kIdentSlicesB16);
        hipLaunchKernelGGL(k_diag_partial_b16, dim3(slices), dim3(256), 0, st, static_cast<const uint16_t*>(ba->diag_ws), nb, ba->ident_ws);
        hipLaunchKernelGGL(k_sum_rows_b16, dim3(1), dim3(256), 0, st, ba->ident_ws, slices, 256, static_cast<uint16_t*>(ba->g_identity));
    }
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

extern "C" int recon_block_adjacency_b16_fwd(const void* T, const void* identity, int32_t B, int32_t n, int32_t dd, void* A, recon_stream_t stream) {
    if (B < 0 || n < 1 || dd < 1 || !identity || !A || (n > 1 && B > 0 && !T)) return RECON_ERR_INVALID;
    if (B > 65535) return RECON_ERR_UNSUPPORTED;
    if (B == 0) return RECON_OK;
    const int S = n * dd;
    const bool v8 = dd % 8 == 0 && !((reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(identity) | reinterpret_cast<uintptr_t>(A)) & 15);
    const uint16_t* Tp = static_cast<const uint16_t*>(T);
    const uint16_t* Ip = static_cast<const uint16_t*>(identity);
    if (v8) hipLaunchKernelGGL(k_block_adj_b16_fwd<8>, dim3(static_cast<unsigned>(ceil_div64(1LL * S * S / 8, 256)), static_cast<unsigned>(B)), dim3(256), 0,
                               as_stream(stream), Tp, Ip, n, dd, static_cast<uint16_t*>(A));
    else hipLaunchKernelGGL(k_block_adj_b16_fwd<1>, dim3(static_cast<unsigned>(ceil_div64(1LL * S * S, 256)), static_cast<unsigned>(B)), dim3(256), 0,
                            as_stream(stream), Tp, Ip, n, dd, static_cast<uint16_t*>(A));
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" size_t recon_block_adjacency_b16_bwd_workspace_floats(int32_t dd) {
    return static_cast<size_t>(kIdentSlicesB16) * (dd > 0 ? dd : 1) * (dd > 0 ? dd : 1);
}

extern "C" int recon_block_adjacency_b16_bwd(const void* gA, int32_t B, int32_t n, int32_t dd, void* gT, void* g_identity, float* workspace,
                                             recon_stream_t stream) {
    if (B < 0 || n < 1 || dd < 1 || !gA) return RECON_ERR_INVALID;
    if (g_identity && !workspace) return RECON_ERR_INVALID;
    if (B > 65535) return RECON_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const uint16_t* g = static_cast<const uint16_t*>(gA);
    const int64_t per_graph = 1LL * n * (n - 1) * dd * dd;
    if (gT && per_graph > 0 && B > 0) {
        const bool v8 = dd % 8 == 0 && !((reinterpret_cast<uintptr_t>(gA) | reinterpret_cast<uintptr_t>(gT)) & 15);
        if (v8) hipLaunchKernelGGL(k_block_adj_b16_bwd_T<8>, dim3(static_cast<unsigned>(ceil_div64(per_graph / 8, 256)), static_cast<unsigned>(B)), dim3(256), 0,
                                   st, g, n, dd, static_cast<uint16_t*>(gT));
        else hipLaunchKernelGGL(k_block_adj_b16_bwd_T<1>, dim3(static_cast<unsigned>(ceil_div64(per_graph, 256)), static_cast<unsigned>(B)), dim3(256), 0, st, g,
                                n, dd, static_cast<uint16_t*>(gT));
    }
    if (g_identity) {
        const int64_t pairs = 1LL * B * n;
        const int slices = static_cast<int>(pairs < kIdentSlicesB16 ? (pairs > 0 ? pairs : 1) : kIdentSlicesB16);
        hipLaunchKernelGGL(k_block_adj_b16_bwd_I, dim3(slices), dim3(256), 0, st, g, B, n, dd, workspace);
        hipLaunchKernelGGL(k_sum_rows_b16, dim3(static_cast<unsigned>(ceil_div64(dd * dd, 256))), dim3(256), 0, st, workspace, slices, dd * dd,
                           static_cast<uint16_t*>(g_identity));
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
