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

namespace recon {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// COMPILER HAZARD (hipcc, ROCm 7.2): __builtin_bit_cast applied directly to a vector ELEMENT (q.w, a[1]) yields element 0 — e.g.
// fmaxf(bit_cast(a[0]), bit_cast(a[1])) folds to a[0].  Every element goes through this by-value helper (tools/probe/h_probe.hip).
__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }

__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS traffic is done; vmcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_wait() {      // this wave's LDS writes have landed (same-wave hand-over, no barrier)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("" ::: "memory");
}
// max over lanes l, l ^ 16, l ^ 32, l ^ 48 (the four 16-lane rows) on the VALU: v_permlane{16,32}_swap exchange rows between two
// registers, no LDS round trip as __shfl_xor (ds_bpermute) would take
__device__ __forceinline__ float rows_max(float m) {
    const uint32_t u = __builtin_bit_cast(uint32_t, m);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    m = fmaxf(as_f(a[0]), as_f(a[1]));
    const uint32_t w = __builtin_bit_cast(uint32_t, m);
    auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(as_f(c[0]), as_f(c[1]));
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) { return fmaxf(v, dpp_mov<CTRL>(v)); }
// max over all 64 lanes: DPP inside the rows, row exchange across them
__device__ __forceinline__ float wave_max(float m) {
    m = dpp_max<0xB1>(m);                         // quad_perm [1,0,3,2]
    m = dpp_max<0x4E>(m);                         // quad_perm [2,3,0,1]
    m = dpp_max<0x141>(m);                        // row_half_mirror
    m = dpp_max<0x140>(m);                        // row_mirror
    return rows_max(m);
}

// tanh to ~3e-7 relative without ocml's branchy tanhf (20 activations per lane and hop): odd polynomial below 0.1 (truncation
// error 2e-11 there), (1 - t) / (1 + t) with t = exp(-2|x|) above (1 - t >= 0.18: no cancellation)
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float poly = x * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    const float t = __expf(-2.f * ax);
    const float big = copysignf((1.f - t) * __frcp_rn(1.f + t), x);
    return ax < 0.1f ? poly : big;
}

constexpr int kGatherRegs = 2;                    // gather items per thread whose indices stay in registers (cfg 3b: exactly 2)
constexpr uint32_t kOOB = 0xfffffff0u;            // a buffer offset past every num_records: the load returns zeros

// NKS = K steps of 32 (S = 32 NKS or 32 NKS - 16), NTC = channel tiles of 16 (C <= 16 NTC); blockDim.x = 4 S (one wave per 16 rows of A).
// Dynamic LDS: half planes [2][NKS][16 NTC][64 B] | fp32 state (under the channel scales) [16 NTC][S + 4] | channel maxima [2][16 NTC] |
// inverse channel scales [16 NTC] | inverse row scales [waves][16]
template <int NKS, int NTC>
__global__ void __launch_bounds__(128 * NKS) k_propagate_fwd_h(const PropK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int CH = NTC * 16;
    constexpr int STEP = CH * 64;                 // bytes of one K step of one plane
    constexpr int PLANE = NKS * STEP;
    constexpr int KP = NKS * 32;
    constexpr int NCW = (CH + 2 * NKS - 2) / (2 * NKS - 1);      // channels a wave stages (at least 2 NKS - 1 waves)
    const int S = p.S, C = p.C, pitch = S + 4;
    unsigned char* Hs = sm;
    unsigned char* Hfb = sm + 2 * PLANE;          // fp32 state, addressed in bytes
    uint32_t* chmax = reinterpret_cast<uint32_t*>(Hfb + CH * pitch * 4);
    float* isg = reinterpret_cast<float*>(chmax + 2 * CH);
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = nthreads >> 6;
    float* atab = isg + CH + 16 * wave;           // this wave's 16 inverse row scales
    const int li = lane & 15, lq = lane >> 4;
    const uint32_t SSb = static_cast<uint32_t>(S) * S * 4;                      // bytes of one A_l of one graph
    // A fragments: k slots 0..3 = t 32 ks + 4 lq .., slots 4..7 = t 32 ks + 16 + 4 lq ..  One buffer descriptor per (graph, hop), one
    // lane offset for all ten loads (the rest is the instruction's immediate); the half K step past S is requested out of range.
    const uint32_t voff_a = (static_cast<uint32_t>(16 * wave + li) * S + 4 * lq) * 4;
    const uint32_t voff_tail = (S & 16) ? kOOB : voff_a + (KP - 16) * 4;
    auto load_a = [&](u32x4 (&raw)[NKS][2], int l, int bb) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.adj[l]) + static_cast<int64_t>(bb) * SSb), 0,
                                                          static_cast<int>(SSb), 0x00020000);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                raw[ks][h] = (ks == NKS - 1 && h == 1) ? __builtin_amdgcn_raw_buffer_load_b128(rs, voff_tail, 0, 0)
                                                       : __builtin_amdgcn_raw_buffer_load_b128(rs, voff_a + (32 * ks + 16 * h) * 4, 0, 0);
    };
    // four consecutive t of one channel, scaled, into the two planes at byte offset `off` of plane 0.
    // off(c, t0) = (t0 >> 5) STEP + 64 c + 16 (((t0 >> 2) & 3) ^ (2 ((c >> 3) & 1))) + 8 ((t0 >> 4) & 1): adding 2 (c >> 3) mod 4 is an XOR
    // with bit 1, so the channel part and the column part separate (one of them is wave-uniform wherever this is called)
    auto store_state4 = [&](int off, float v0, float v1, float v2, float v3) {
        uint32_t h0, l0, h1, l1;
        hx2_split2(v0, v1, h0, l0);
        hx2_split2(v2, v3, h1, l1);
        *reinterpret_cast<uint2*>(Hs + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(Hs + PLANE + off) = make_uint2(l0, l1);
    };
    auto col_off = [](int t0) { return (t0 >> 5) * STEP + (((t0 >> 2) & 3) << 4) + (((t0 >> 4) & 1) << 3); };
    const int b_rd = li * 64 + (((lq + 2 * (li >> 3)) & 3) << 4);      // B fragment: + 1024 j + STEP ks (+ PLANE for the low terms)
    const int t0w = 16 * wave + 4 * lq;                                 // this lane's four state columns in the C layout
    const int so_w = (col_off(t0w) + 64 * li) ^ (((li >> 3) & 1) << 5);  // + 1024 j: (16 j + li) >> 3 has li's parity bit
    const int hf_w = (li * pitch + t0w) * 4;                            // + 64 j pitch
    const int nitems = C * p.dd, Ldd = p.L * p.dd;
    const bool homog = p.act != RECON_ACT_TANH;                         // act(k v) = k act(v) for k > 0: the old channel scale stays on

    u32x4 raw[NKS][2];
    int b = blockIdx.x;
    if (b < p.B) load_a(raw, 0, b);
#pragma unroll 1
    for (; b < p.B; b += gridDim.x) {
        // ---- gather items of this thread (the same in every hop): byte positions of head / tail in the fp32 state, of the channel's
        // inverse scale and of the result in `out`
        uint32_t g_hi[kGatherRegs], g_ti[kGatherRegs], g_o[kGatherRegs], g_c[kGatherRegs];
        {
            const int64_t* hd = p.head + b * p.idx_bs;
            const int64_t* tl = p.tail + b * p.idx_bs;
#pragma unroll
            for (int i = 0; i < kGatherRegs; ++i) {
                const uint32_t idx = min(tid + i * nthreads, nitems - 1);
                const uint32_t c = idx / static_cast<uint32_t>(p.dd);
                g_hi[i] = 4u * (c * pitch + static_cast<uint32_t>(hd[idx])); g_ti[i] = 4u * (c * pitch + static_cast<uint32_t>(tl[idx]));
                g_o[i] = 4u * (idx + c * (Ldd - p.dd));                         // c L dd + x
                g_c[i] = 4u * c;
            }
        }
        // ---- h^0: wave w stages channels w, w + NW, ... (a whole channel per wave: its max magnitude is a wave reduction); columns
        // past S and channels past C come back as zeros (out-of-range offsets)
        {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.h0 + b * p.h0_bs), 0, C * S * 4, 0x00020000);
            const int t0 = 4 * lane;
            const uint32_t vo = t0 < S ? 16u * lane : kOOB;
            const int so = col_off(t0);
            u32x4 hv[NCW];
#pragma unroll
            for (int i = 0; i < NCW; ++i) {
                const int c = wave + i * NW;
                hv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, c < C ? vo + static_cast<uint32_t>(c * S * 4) : kOOB, 0, 0);
            }
            for (int i = tid; i < 2 * CH; i += nthreads) chmax[i] = 0u;
#pragma unroll
            for (int i = 0; i < NCW; ++i) {
                const int c = wave + i * NW;                                        // wave-uniform
                if (c < CH) {
                    const float v0 = as_f(hv[i].x), v1 = as_f(hv[i].y);
                    const float v2 = as_f(hv[i].z), v3 = as_f(hv[i].w);
                    const float m = wave_max(fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))));
                    const float sg = hx2_scale_of(m);
                    if (t0 < KP) store_state4((so + 64 * c) ^ (((c >> 3) & 1) << 5), v0 * sg, v1 * sg, v2 * sg, v3 * sg);
                    if (t0 < S) *reinterpret_cast<float4*>(Hfb + (c * pitch + t0) * 4) = make_float4(v0 * sg, v1 * sg, v2 * sg, v3 * sg);
                    if (lane == 0) isg[c] = hx2_inv(sg);
                }
            }
        }
        lds_barrier();
        float inv_sig[NTC];
#pragma unroll
        for (int j = 0; j < NTC; ++j) inv_sig[j] = isg[16 * j + li];

#pragma unroll 1
        for (int l = 0; l < p.L; ++l) {
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
            // ---- the rows of the NEXT step (next hop, or hop 0 of this workgroup's next graph) start their trip now
            {
                const bool more_hops = l + 1 < p.L;
                const int nb = more_hops ? b : b + static_cast<int>(gridDim.x);
                if (nb < p.B) load_a(raw, more_hops ? l + 1 : 0, nb);
            }
            // ---- products
            f32x4 acc[NTC];
#pragma unroll
            for (int j = 0; j < NTC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
            }
            // ---- epilogue 1: row scales off, activation, channel maxima (every lane sends its own: the LDS unit orders the four lanes of
            // a channel).  C layout: column (lane & 15) = channel 16 j + li, rows 4 lq + r.  Values stay under the OLD channel scale
            // where the activation commutes with it (relu, linear).
            lds_wait();
            const float4 ia = *reinterpret_cast<const float4*>(atab + 4 * lq);
            uint32_t* cm = chmax + (l & 1) * CH;
            if (homog) {
                const bool relu = p.act == RECON_ACT_RELU;
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
            lds_barrier();                                                          // everybody has read H^l-1; maxima complete
            // ---- epilogue 2: H^l under its new channel scales into the planes and the fp32 copy
            for (int i = tid; i < CH; i += nthreads) chmax[((l + 1) & 1) * CH + i] = 0u;
#pragma unroll
            for (int j = 0; j < NTC; ++j) {
                const float inv_u = homog ? inv_sig[j] : 1.f;                       // the unit acc[j] is in now
                const float sg = hx2_scale_of(__builtin_bit_cast(float, cm[16 * j + li]) * inv_u);
                const float f = sg * inv_u;
                inv_sig[j] = hx2_inv(sg);
                const float w0 = acc[j][0] * f, w1 = acc[j][1] * f, w2 = acc[j][2] * f, w3 = acc[j][3] * f;
                store_state4(so_w + 1024 * j, w0, w1, w2, w3);
                *reinterpret_cast<float4*>(Hfb + hf_w + 64 * j * pitch) = make_float4(w0, w1, w2, w3);
                if (tid < 16) isg[16 * j + li] = inv_sig[j];
            }
            lds_barrier();                                                          // H^l complete
            // ---- relation_l = gather(h, heads) * gather(h, tails)   (models/models.py:270-273), saved state
            char* out = reinterpret_cast<char*>(p.out + (static_cast<int64_t>(b) * C * p.L + l) * p.dd);
#pragma unroll
            for (int i = 0; i < kGatherRegs; ++i)
                if (tid + i * nthreads < nitems) {
                    const float k = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(isg) + g_c[i]);
                    *reinterpret_cast<float*>(out + g_o[i]) = (*reinterpret_cast<const float*>(Hfb + g_hi[i]) * k) * (*reinterpret_cast<const float*>(Hfb + g_ti[i]) * k);
                }
            if (nitems > kGatherRegs * nthreads) {                                   // more items per thread: their index loads wait for the prefetch
                for (int idx = tid + kGatherRegs * nthreads; idx < nitems; idx += nthreads) {
                    const int c = idx / p.dd, x = idx - c * p.dd;
                    const int64_t io = b * p.idx_bs + idx;
                    const float* hf = reinterpret_cast<const float*>(Hfb) + c * pitch;
                    const float k = isg[c];
                    reinterpret_cast<float*>(out)[c * Ldd + x] = (hf[static_cast<int>(p.head[io])] * k) * (hf[static_cast<int>(p.tail[io])] * k);
                }
            }
            if (p.hsave) {                                                          // wave w: rows w, w + NW, ...; lanes: 16-byte pieces
                char* hs = reinterpret_cast<char*>(p.hsave + ((static_cast<int64_t>(l) * p.B + b) * C) * S);
                if (4 * lane < S)
                    for (int c = wave; c < C; c += NW) {
                        const float k = isg[c];
                        float4 v = *reinterpret_cast<const float4*>(Hfb + (c * pitch + 4 * lane) * 4);
                        v.x *= k; v.y *= k; v.z *= k; v.w *= k;
                        *reinterpret_cast<float4*>(hs + static_cast<uint32_t>(c * S + 4 * lane) * 4u) = v;
                    }
            }
        }
        lds_barrier();                                                              // the gathers are done before the next graph's h^0 lands
    }
}

size_t fwd_h_lds(int nks, int ntc, int S) {
    const size_t ch = 16ull * ntc;
    return 2ull * nks * ch * 64 + ch * (S + 4) * sizeof(float) + 3ull * ch * sizeof(uint32_t) + static_cast<size_t>(S / 16) * 16 * sizeof(float);
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

}  // namespace

bool prop_fwd_h_supported(const PropK& p) {
    if (p.S % 16 != 0 || p.S > 160 || p.C > 96 || p.S < 16 || p.dd < 1) return false;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    for (int l = 0; l < p.L; ++l) if (!al16(p.adj[l])) return false;
    if (!al16(p.h0) || (p.h0_bs % 4) != 0 || (p.hsave && !al16(p.hsave))) return false;
    const int nks = (p.S + 31) / 32, ntc = (p.C + 15) / 16;
    return fwd_h_lds(nks, ntc, p.S) <= 160 * 1024;
}

int prop_fwd_h(const PropK& p, hipStream_t st) {
    if (!prop_fwd_h_supported(p)) return RECON_ERR_UNSUPPORTED;
    const int nks = (p.S + 31) / 32, ntc = (p.C + 15) / 16, nw = p.S / 16;
    const size_t lds = fwd_h_lds(nks, ntc, p.S);
    const int grid = p.B < num_cus() ? p.B : num_cus();
#define CALL_H(K_, N_)                                                                                                                  \
    do {                                                                                                                                \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_propagate_fwd_h<K_, N_>),                     \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));             \
        hipLaunchKernelGGL((k_propagate_fwd_h<K_, N_>), dim3(static_cast<unsigned>(grid)), dim3(64 * nw), lds, st, p);                  \
    } while (0)
#define CALL_HN(K_)                                                                                                                     \
    switch (ntc) { case 1: CALL_H(K_, 1); break; case 2: CALL_H(K_, 2); break; case 3: CALL_H(K_, 3); break; case 4: CALL_H(K_, 4); break; \
                   case 5: CALL_H(K_, 5); break; default: CALL_H(K_, 6); break; }
    switch (nks) { case 1: CALL_HN(1); break; case 2: CALL_HN(2); break; case 3: CALL_HN(3); break; case 4: CALL_HN(4); break; default: CALL_HN(5); break; }
#undef CALL_HN
#undef CALL_H
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace recon
