// K1 / K2 — edge-wise attention + segment aggregation kernels of the KB-GAT layer, and the host
// orchestration of the full layer forward / backward (projection GEMMs in gemm_f32.hip).
//
// Replaces GAT/layers.py:129-178 (forward) and the autograd graph derived from it together with
// SpecialSpmmFunctionFinal.backward (GAT/layers.py:67-79).  All H heads that share inputs
// (GAT/models.py:71-72) are processed by one launch: blockIdx.y = head.
//
// Data layout in HBM (fp32):
//   P  [2][H][N][D]  node projections (dst half / src half), head-major so that one (head, node)
//                    row is D contiguous floats;
//   Q  [H][E][D]     edge projections in CSR-slot order (edges sorted by destination), so the
//                    kernel streams Q strictly sequentially per head;
//   out[N][H*D]      the reference's concatenated layout (torch.cat(dim=1), GAT/models.py:71).
// Work mapping: a group of G lanes (G = 8..64, power of two) owns one (node, head) pair; each
// lane holds KR vectors of VEC floats of the D-wide row.  The segment (CSR row) is walked in
// fixed order with register accumulators: no atomics, run-to-run deterministic.
#include <math.h>
#include "recon_common.h"

namespace recon {
namespace {

constexpr int kBlock = 256;

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, each
// with a private L2); neighbouring node blocks share P_src rows (same graph) and cache lines, so give
// every XCD one CONTIGUOUS chunk of node blocks.  Bijective for any grid size; speed only.
__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
template <int KR> constexpr int unroll_for() { return KR >= 8 ? 1 : (KR >= 4 ? 2 : 4); }

struct EdgeFwdArgs {
    const int32_t* rowptr; const int32_t* src;
    const float* P; const float* Q; const float* a2; const float* keep;
    float* out; float* sigma; float* Z;
    int32_t N, E, D, H, ld_out, concat;
    float alpha;
    int32_t* nan_flag;          // optional (config.hip: recon_set_nan_flag)
};

template <int VEC, int G, int KR, bool TRAIN>
__global__ void __launch_bounds__(kBlock) k_gat_edge_fwd(const EdgeFwdArgs p) {
    constexpr int GPB = kBlock / G;
    constexpr int kUnroll = unroll_for<KR>();
    const int lig = threadIdx.x % G;
    const int node = xcd_block(blockIdx.x, gridDim.x) * GPB + threadIdx.x / G;
    const int h = blockIdx.y;
    if (node >= p.N) return;            // whole groups exit together; group_sum only crosses a group
    const int D = p.D;
    int c[KR]; bool act[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { c[r] = (r * G + lig) * VEC; act[r] = c[r] < D; }

    const float* Pd = p.P + (static_cast<int64_t>(h) * p.N + node) * D;
    const float* Ps = p.P + static_cast<int64_t>(p.H + h) * p.N * D;
    const float* Qh = p.Q + static_cast<int64_t>(h) * p.E * D;
    float a2[KR][VEC], pd[KR][VEC], U[KR][VEC];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) { a2[r][v] = 0.f; pd[r][v] = 0.f; U[r][v] = 0.f; }
        if (act[r]) { load_vec<VEC>(a2[r], p.a2 + h * D + c[r]); load_vec<VEC>(pd[r], Pd + c[r]); }
    }
    const int beg = p.rowptr[node], end = p.rowptr[node + 1];
    float Zs = 0.f;
    for (int k0 = beg; k0 < end; k0 += kUnroll) {
        float m[kUnroll][KR][VEC];
        float part[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int k = k0 + u;
            part[u] = 0.f;
            if (k < end) {
                const int s = p.src[k];
                const float* q = Qh + static_cast<int64_t>(k) * D;
                const float* ps = Ps + static_cast<int64_t>(s) * D;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    if (act[r]) {
                        float qv[VEC], sv[VEC];
                        load_vec<VEC>(qv, q + c[r]);
                        load_vec<VEC>(sv, ps + c[r]);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            m[u][r][v] = (pd[r][v] + sv[v]) + qv[v];
                            part[u] = fmaf(a2[r][v], m[u][r][v], part[u]);
                        }
                    } else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) m[u][r][v] = 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int k = k0 + u;
            const float sg = group_sum<G>(part[u]);     // executed by every lane of the wave
            if (k < end) {
                const float w = expf(-(sg > 0.f ? sg : p.alpha * sg));
                Zs += w;
                float wk = w;
                if constexpr (TRAIN) {
                    if (p.keep) wk = w * p.keep[static_cast<int64_t>(h) * p.E + k];
                    if (lig == 0) p.sigma[static_cast<int64_t>(h) * p.E + k] = sg;
                }
#pragma unroll
                for (int r = 0; r < KR; ++r)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) U[r][v] = fmaf(wk, m[u][r][v], U[r][v]);
            }
        }
    }
    if (Zs == 0.f) Zs = 1e-12f;                          // GAT/layers.py:152
    if (p.nan_flag && !(fabsf(Zs) <= 3.0e38f)) *p.nan_flag = 1;     // the reference's asserts (:147, :167, :172)
    if constexpr (TRAIN) { if (lig == 0) p.Z[static_cast<int64_t>(h) * p.N + node] = Zs; }
    float* o = p.out + static_cast<int64_t>(node) * p.ld_out + h * D;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        if (act[r]) {
            float y[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float hv = U[r][v] / Zs;
                y[v] = (p.concat && hv <= 0.f) ? expm1f(hv) : hv;   // F.elu, GAT/layers.py:173-175
            }
            store_vec<VEC>(o + c[r], y);
        }
    }
}

struct EdgeBwdArgs {
    const int32_t* rowptr; const int32_t* src;
    const float* P; const float* Q; const float* a2; const float* keep;
    const float* out; const float* gout; const float* sigma; const float* Z;
    float* Gm; float* gPdst; float* a2_partial;       // a2_partial: [gridDim.x][H][D]
    int32_t N, E, D, H, ld_out, ld_gout, concat;
    float alpha;
};

// one group walks nodes node0, node0 + stride, ... so that the a_2 gradient can be reduced in a
// fixed order: lane registers -> LDS across the block's groups -> one partial row per block.
template <int VEC, int G, int KR>
__global__ void __launch_bounds__(kBlock) k_gat_edge_bwd(const EdgeBwdArgs p) {
    constexpr int GPB = kBlock / G;
    constexpr int kUnroll = unroll_for<KR>();
    extern __shared__ __attribute__((aligned(16))) float red[];      // [GPB][D]
    const int lig = threadIdx.x % G, grp = threadIdx.x / G;
    const int h = blockIdx.y;
    const int D = p.D;
    int c[KR]; bool act[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { c[r] = (r * G + lig) * VEC; act[r] = c[r] < D; }
    const float* Ps = p.P + static_cast<int64_t>(p.H + h) * p.N * D;
    const float* Qh = p.Q + static_cast<int64_t>(h) * p.E * D;
    float* Gmh = p.Gm + static_cast<int64_t>(h) * p.E * D;
    float a2[KR][VEC], ga2[KR][VEC];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) { a2[r][v] = 0.f; ga2[r][v] = 0.f; }
        if (act[r]) load_vec<VEC>(a2[r], p.a2 + h * D + c[r]);
    }
    const int n_iter = (p.N + gridDim.x * GPB - 1) / (gridDim.x * GPB);
    for (int it = 0; it < n_iter; ++it) {
        const int node = (it * gridDim.x + xcd_block(blockIdx.x, gridDim.x)) * GPB + grp;
        const bool nv = node < p.N;                      // keep every lane in the loop: group_sum is wave-wide code
        float pd[KR][VEC], gU[KR][VEC], gd[KR][VEC];
        float dot = 0.f, Zi = 1.f;
        int beg = 0, end = 0;
        if (nv) {
            Zi = p.Z[static_cast<int64_t>(h) * p.N + node];
            beg = p.rowptr[node]; end = p.rowptr[node + 1];
        }
#pragma unroll
        for (int r = 0; r < KR; ++r) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { pd[r][v] = 0.f; gU[r][v] = 0.f; gd[r][v] = 0.f; }
            if (nv && act[r]) {
                float y[VEC], gy[VEC];
                load_vec<VEC>(pd[r], p.P + (static_cast<int64_t>(h) * p.N + node) * D + c[r]);
                load_vec<VEC>(y, p.out + static_cast<int64_t>(node) * p.ld_out + h * D + c[r]);
                load_vec<VEC>(gy, p.gout + static_cast<int64_t>(node) * p.ld_gout + h * D + c[r]);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float gh = gy[v], hv = y[v];
                    if (p.concat && y[v] <= 0.f) {          // elu'(h) = exp(h) = y + 1 ; h = log1p(y)
                        const float e = y[v] + 1.f;
                        gh = gy[v] * e;
                        hv = e > 0.f ? log1pf(y[v]) : 0.f;  // exp(h)*h -> 0 when exp(h) underflowed
                    }
                    gU[r][v] = gh / Zi;
                    dot = fmaf(gh, hv, dot);
                }
            }
        }
        const float gZ = -group_sum<G>(dot) / Zi;
        for (int k0 = beg; __any(k0 < end); k0 += kUnroll) {
            float m[kUnroll][KR][VEC];
            float part[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int k = k0 + u;
                part[u] = 0.f;
                if (k < end) {
                    const int s = p.src[k];
                    const float* q = Qh + static_cast<int64_t>(k) * D;
                    const float* ps = Ps + static_cast<int64_t>(s) * D;
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        if (act[r]) {
                            float qv[VEC], sv[VEC];
                            load_vec<VEC>(qv, q + c[r]);
                            load_vec<VEC>(sv, ps + c[r]);
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                m[u][r][v] = (pd[r][v] + sv[v]) + qv[v];
                                part[u] = fmaf(gU[r][v], m[u][r][v], part[u]);
                            }
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) m[u][r][v] = 0.f;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int k = k0 + u;
                const float tt = group_sum<G>(part[u]);
                if (k < end) {
                    const float sg = p.sigma[static_cast<int64_t>(h) * p.E + k];
                    const float w = expf(-(sg > 0.f ? sg : p.alpha * sg));
                    const float kf = p.keep ? p.keep[static_cast<int64_t>(h) * p.E + k] : 1.f;
                    const float gw = fmaf(kf, tt, gZ);
                    const float gs = -gw * w * (sg > 0.f ? 1.f : p.alpha);
                    const float kw = kf * w;
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        if (act[r]) {
                            float gm[VEC];
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                gm[v] = fmaf(kw, gU[r][v], gs * a2[r][v]);
                                gd[r][v] += gm[v];
                                ga2[r][v] = fmaf(gs, m[u][r][v], ga2[r][v]);
                            }
                            store_vec<VEC>(Gmh + static_cast<int64_t>(k) * D + c[r], gm);
                        }
                    }
                }
            }
        }
        if (nv) {
#pragma unroll
            for (int r = 0; r < KR; ++r)
                if (act[r]) store_vec<VEC>(p.gPdst + (static_cast<int64_t>(h) * p.N + node) * D + c[r], gd[r]);
        }
    }
    // fixed-order reduction of the a_2 gradient over the block's groups
#pragma unroll
    for (int r = 0; r < KR; ++r)
        if (act[r]) store_vec<VEC>(red + grp * D + c[r], ga2[r]);
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += kBlock) {
        float s = 0.f;
        for (int g = 0; g < GPB; ++g) s += red[g * D + i];
        p.a2_partial[(static_cast<int64_t>(blockIdx.x) * p.H + h) * D + i] = s;
    }
}

// K2b: gP_src[h][j][:] = sum over edges whose source is j of Gm[h][slot][:]  (CSC walk, fixed order)
struct SrcGatherArgs {
    const int32_t* rowptr_src; const int32_t* slot_by_src; const float* Gm; float* gPsrc;
    int32_t N, E, D, H;
};
template <int VEC, int G, int KR>
__global__ void __launch_bounds__(kBlock) k_gat_src_gather(const SrcGatherArgs p) {
    constexpr int GPB = kBlock / G;
    constexpr int kUnroll = 4;
    const int lig = threadIdx.x % G;
    const int node = xcd_block(blockIdx.x, gridDim.x) * GPB + threadIdx.x / G;
    const int h = blockIdx.y;
    if (node >= p.N) return;
    const int D = p.D;
    const float* Gmh = p.Gm + static_cast<int64_t>(h) * p.E * D;
    float acc[KR][VEC];
#pragma unroll
    for (int r = 0; r < KR; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[r][v] = 0.f;
    const int beg = p.rowptr_src[node], end = p.rowptr_src[node + 1];
    for (int k0 = beg; k0 < end; k0 += kUnroll) {
        float t[kUnroll][KR][VEC];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int k = k0 + u;
#pragma unroll
            for (int r = 0; r < KR; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v) t[u][r][v] = 0.f;
            if (k < end) {
                const int slot = p.slot_by_src[k];
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const int cc = (r * G + lig) * VEC;
                    if (cc < D) load_vec<VEC>(t[u][r], Gmh + static_cast<int64_t>(slot) * D + cc);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int r = 0; r < KR; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[r][v] += t[u][r][v];
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        const int cc = (r * G + lig) * VEC;
        if (cc < D) store_vec<VEC>(p.gPsrc + (static_cast<int64_t>(h) * p.N + node) * D + cc, acc[r]);
    }
}

// out[i] = sum_b partial[b][i], fixed order: 64 columns x 16 row groups per block, LDS combine
__global__ void __launch_bounds__(1024) k_reduce_partials(const float* __restrict__ partial, int32_t nblk, int32_t HD,
                                                          float* __restrict__ out) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int per = (nblk + 15) / 16;
    float s = 0.f;
    if (col < HD)
        for (int b = grp * per; b < min(nblk, (grp + 1) * per); ++b) s += partial[static_cast<int64_t>(b) * HD + col];
    red[grp][c] = s;
    __syncthreads();
    if (grp == 0 && col < HD) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][c];
        out[col] = t;
    }
}

// ---- dispatch over (VEC, G, KR) --------------------------------------------------------------
struct Shape { int vec, g, kr; };
bool pick_shape(int D, bool aligned4, bool aligned2, Shape* s) {
    int vec = (aligned4 && D % 4 == 0) ? 4 : ((aligned2 && D % 2 == 0) ? 2 : 1);
    int lanes = (D + vec - 1) / vec;
    int g = 8;
    while (g < 64 && g < lanes) g <<= 1;
    int kr = (lanes + g - 1) / g;
    if (kr > 2 && kr <= 4) kr = 4;
    else if (kr > 4 && kr <= 8) kr = 8;
    else if (kr > 8) return false;
    s->vec = vec; s->g = g; s->kr = kr;
    return true;
}

// F is a generic lambda taking std::integral_constant-like tags; expanded over the instantiated set
#define RECON_DISPATCH_SHAPE(S, CALL)                                                         \
    do {                                                                                      \
        bool ok_ = true;                                                                      \
        if ((S).kr == 1) {                                                                    \
            if ((S).vec == 4) { if ((S).g == 8) { CALL(4, 8, 1); } else if ((S).g == 16) { CALL(4, 16, 1); } else if ((S).g == 32) { CALL(4, 32, 1); } else { CALL(4, 64, 1); } } \
            else if ((S).vec == 2) { if ((S).g == 8) { CALL(2, 8, 1); } else if ((S).g == 16) { CALL(2, 16, 1); } else if ((S).g == 32) { CALL(2, 32, 1); } else { CALL(2, 64, 1); } } \
            else { if ((S).g == 8) { CALL(1, 8, 1); } else if ((S).g == 16) { CALL(1, 16, 1); } else if ((S).g == 32) { CALL(1, 32, 1); } else { CALL(1, 64, 1); } } \
        } else if ((S).kr == 2) {                                                             \
            if ((S).vec == 4) { CALL(4, 64, 2); } else if ((S).vec == 2) { CALL(2, 64, 2); } else { CALL(1, 64, 2); } \
        } else if ((S).kr == 4) {                                                             \
            if ((S).vec == 4) { CALL(4, 64, 4); } else if ((S).vec == 2) { CALL(2, 64, 4); } else { CALL(1, 64, 4); } \
        } else if ((S).kr == 8) {                                                             \
            if ((S).vec == 4) { CALL(4, 64, 8); } else if ((S).vec == 2) { CALL(2, 64, 8); } else { CALL(1, 64, 8); } \
        } else ok_ = false;                                                                   \
        if (!ok_) return RECON_ERR_UNSUPPORTED;                                               \
    } while (0)

bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

int check_fwd(const recon_graph* g, const recon_gat_fwd_args* a) {
    if (!g || !a) return RECON_ERR_INVALID;
    if (a->N != g->N || a->E != g->E) return RECON_ERR_INVALID;
    if (a->N < 0 || a->E < 0 || a->F <= 0 || a->R <= 0 || a->D <= 0 || a->H <= 0) return RECON_ERR_INVALID;
    if (a->ld_out < a->H * a->D) return RECON_ERR_INVALID;
    if (!a->x || !a->a || !a->a_2 || !a->P || !a->out) return RECON_ERR_INVALID;
    if (a->E > 0 && (!a->edge_embed || !a->Q)) return RECON_ERR_INVALID;
    if (a->H > 65535) return RECON_ERR_UNSUPPORTED;
    return RECON_OK;
}

int bwd_blocks(int N, int gpb) {
    int64_t nb = ceil_div64(N, gpb);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    return static_cast<int>(nb);
}

}  // namespace
}  // namespace recon

using namespace recon;

extern "C" int recon_gat_project(const recon_graph* g, const recon_gat_fwd_args* a, recon_stream_t stream) {
    int rc = check_fwd(g, a);
    if (rc != RECON_OK) return rc;
    hipStream_t st = as_stream(stream);
    const int32_t N = a->N, E = a->E, F = a->F, R = a->R, D = a->D, H = a->H;
    const int64_t W = 2LL * F + R, HD = 1LL * H * D;
    if (2 * HD > 0x7fffffffLL) return RECON_ERR_UNSUPPORTED;
    // P[s][h][n][d] = sum_f x[n][f] * a[h][d][s*F + f]
    {
        OperandDesc A = plain_operand(a->x, F);
        OperandDesc B = plain_operand(a->a, W);
        B.P = static_cast<int32_t>(HD); B.S2 = F;              // column (s,h,d): row (h,d) of `a`, shifted by s*F
        OutputDesc C = plain_output(a->P, D);
        C.Dseg = D; C.Sseg = static_cast<int64_t>(N) * D;      // column (s*H+h, d) -> head-major block
        rc = gemm_f32(N, static_cast<int32_t>(2 * HD), F, A, true, B, true, C, 1, nullptr, st);
        if (rc != RECON_OK) return rc;
    }
    // Q[h][k][d] = sum_r edge_embed[eid[k]][r] * a[h][d][2F + r]
    if (E > 0) {
        OperandDesc A = plain_operand(a->edge_embed, R);
        A.gather = g->eid;
        OperandDesc B = plain_operand(a->a + 2 * F, W);
        OutputDesc C = plain_output(a->Q, D);
        C.Dseg = D; C.Sseg = static_cast<int64_t>(E) * D;
        rc = gemm_f32(E, static_cast<int32_t>(HD), R, A, true, B, true, C, 1, nullptr, st);
        if (rc != RECON_OK) return rc;
    }
    return RECON_OK;
}

extern "C" int recon_gat_edge_fwd(const recon_graph* g, const recon_gat_fwd_args* a, recon_stream_t stream) {
    int rc = check_fwd(g, a);
    if (rc != RECON_OK) return rc;
    if (a->N == 0) return RECON_OK;
    const bool train = a->Z != nullptr;                   // training call: save sigma / Z for the backward
    if (train && a->E > 0 && !a->sigma) return RECON_ERR_INVALID;
    if (a->keep && !train) return RECON_ERR_INVALID;
    EdgeFwdArgs p;
    p.rowptr = g->rowptr_dst; p.src = g->src; p.P = a->P; p.Q = a->Q; p.a2 = a->a_2; p.keep = a->keep;
    p.out = a->out; p.sigma = a->sigma; p.Z = a->Z;
    p.N = a->N; p.E = a->E; p.D = a->D; p.H = a->H; p.ld_out = a->ld_out; p.concat = a->concat; p.alpha = a->alpha;
    p.nan_flag = recon::nan_flag();
    const bool a4 = al(a->P, 16) && al(a->Q, 16) && al(a->a_2, 16) && al(a->out, 16) && a->ld_out % 4 == 0;
    const bool a2ok = al(a->P, 8) && al(a->Q, 8) && al(a->a_2, 8) && al(a->out, 8) && a->ld_out % 2 == 0;
    Shape s;
    if (!pick_shape(a->D, a4, a2ok, &s)) return RECON_ERR_UNSUPPORTED;
    const int gpb = kBlock / s.g;
    dim3 grid(static_cast<unsigned>(ceil_div64(a->N, gpb)), static_cast<unsigned>(a->H));
    hipStream_t st = as_stream(stream);
#define CALL_FWD(V, G_, K_)                                                                              \
    do {                                                                                                 \
        if (train) hipLaunchKernelGGL((k_gat_edge_fwd<V, G_, K_, true>), grid, dim3(kBlock), 0, st, p);  \
        else hipLaunchKernelGGL((k_gat_edge_fwd<V, G_, K_, false>), grid, dim3(kBlock), 0, st, p);       \
    } while (0)
    RECON_DISPATCH_SHAPE(s, CALL_FWD);
#undef CALL_FWD
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_gat_fwd(const recon_graph* g, const recon_gat_fwd_args* a, recon_stream_t stream) {
    int rc = recon_gat_project(g, a, stream);
    if (rc != RECON_OK) return rc;
    return recon_gat_edge_fwd(g, a, stream);
}

extern "C" size_t recon_gat_bwd_partial_floats(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H) {
    const int64_t HD = 1LL * H * D;
    size_t need = static_cast<size_t>(512) * HD;                                     // a_2 partial rows (<= 512 blocks)
    const int s1 = gemm_pick_split_k(static_cast<int32_t>(2 * HD), F, N);
    const int s2 = gemm_pick_split_k(static_cast<int32_t>(HD), R, E);
    const size_t g1 = static_cast<size_t>(s1 > 1 ? s1 : 0) * 2 * HD * F;
    const size_t g2 = static_cast<size_t>(s2 > 1 ? s2 : 0) * HD * R;
    if (g1 > need) need = g1;
    if (g2 > need) need = g2;
    return need;
}

extern "C" int recon_gat_bwd(const recon_graph* g, const recon_gat_bwd_args* b, recon_stream_t stream) {
    if (!b) return RECON_ERR_INVALID;
    const recon_gat_fwd_args* a = &b->fwd;
    int rc = check_fwd(g, a);
    if (rc != RECON_OK) return rc;
    if (!a->Z || !b->grad_out || !b->gP || !b->partial) return RECON_ERR_INVALID;
    if (a->E > 0 && (!a->sigma || !b->Gm)) return RECON_ERR_INVALID;
    if (b->ld_gout < a->H * a->D) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    const int32_t N = a->N, E = a->E, F = a->F, R = a->R, D = a->D, H = a->H;
    const int64_t W = 2LL * F + R, HD = 1LL * H * D;
    if (N == 0) return RECON_OK;

    const bool a4 = al(a->P, 16) && al(a->Q, 16) && al(a->a_2, 16) && al(a->out, 16) && a->ld_out % 4 == 0 &&
                    al(b->grad_out, 16) && b->ld_gout % 4 == 0 && al(b->Gm, 16) && al(b->gP, 16) && al(b->partial, 16);
    const bool a2ok = al(a->P, 8) && al(a->Q, 8) && al(a->a_2, 8) && al(a->out, 8) && a->ld_out % 2 == 0 &&
                      al(b->grad_out, 8) && b->ld_gout % 2 == 0 && al(b->Gm, 8) && al(b->gP, 8) && al(b->partial, 8);
    Shape s;
    if (!pick_shape(D, a4, a2ok, &s)) return RECON_ERR_UNSUPPORTED;
    const int gpb = kBlock / s.g;
    const int nblk = bwd_blocks(N, gpb);

    // (1) edge pass over the destination CSR: Gm, gP_dst, a_2 partials
    {
        EdgeBwdArgs p;
        p.rowptr = g->rowptr_dst; p.src = g->src; p.P = a->P; p.Q = a->Q; p.a2 = a->a_2; p.keep = a->keep;
        p.out = a->out; p.gout = b->grad_out; p.sigma = a->sigma; p.Z = a->Z;
        p.Gm = b->Gm; p.gPdst = b->gP; p.a2_partial = b->partial;
        p.N = N; p.E = E; p.D = D; p.H = H; p.ld_out = a->ld_out; p.ld_gout = b->ld_gout; p.concat = a->concat;
        p.alpha = a->alpha;
        dim3 grid(static_cast<unsigned>(nblk), static_cast<unsigned>(H));
        const size_t lds = static_cast<size_t>(gpb) * D * sizeof(float);
#define CALL_BWD(V, G_, K_) hipLaunchKernelGGL((k_gat_edge_bwd<V, G_, K_>), grid, dim3(kBlock), lds, st, p)
        RECON_DISPATCH_SHAPE(s, CALL_BWD);
#undef CALL_BWD
        RECON_CHECK_LAUNCH();
        if (b->g_a_2) {
            hipLaunchKernelGGL(k_reduce_partials, dim3(static_cast<unsigned>(ceil_div64(HD, 64))), dim3(1024), 0, st,
                               b->partial, nblk, static_cast<int32_t>(HD), b->g_a_2);
            RECON_CHECK_LAUNCH();
        }
    }
    // (2) source-side segment sum over the CSC view
    {
        SrcGatherArgs p;
        p.rowptr_src = g->rowptr_src; p.slot_by_src = g->slot_by_src; p.Gm = b->Gm;
        p.gPsrc = b->gP + static_cast<int64_t>(H) * N * D;
        p.N = N; p.E = E; p.D = D; p.H = H;
        dim3 grid(static_cast<unsigned>(ceil_div64(N, gpb)), static_cast<unsigned>(H));
#define CALL_SG(V, G_, K_) hipLaunchKernelGGL((k_gat_src_gather<V, G_, K_>), grid, dim3(kBlock), 0, st, p)
        RECON_DISPATCH_SHAPE(s, CALL_SG);
#undef CALL_SG
        RECON_CHECK_LAUNCH();
    }
    // (3) g_x[n][f] = sum_{s,h,d} gP[s][h][n][d] * a[h][d][s*F + f]
    if (b->g_x) {
        OperandDesc A = plain_operand(b->gP, D);
        A.Dseg = D; A.Sseg = static_cast<int64_t>(N) * D;
        OperandDesc B = plain_operand(a->a, W);
        B.P = static_cast<int32_t>(HD); B.S2 = F;
        rc = gemm_f32(N, F, static_cast<int32_t>(2 * HD), A, true, B, false, plain_output(b->g_x, F), 1, nullptr, st);
        if (rc != RECON_OK) return rc;
    }
    // (4) g_edge_embed[eid[k]][r] = sum_{h,d} Gm[h][k][d] * a[h][d][2F + r]
    if (b->g_edge_embed && E > 0) {
        OperandDesc A = plain_operand(b->Gm, D);
        A.Dseg = D; A.Sseg = static_cast<int64_t>(E) * D;
        OperandDesc B = plain_operand(a->a + 2 * F, W);
        OutputDesc C = plain_output(b->g_edge_embed, R);
        C.scatter = g->eid;
        rc = gemm_f32(E, R, static_cast<int32_t>(HD), A, true, B, false, C, 1, nullptr, st);
        if (rc != RECON_OK) return rc;
    }
    if (b->g_a) {
        // (5) g_a[h][d][s*F + f] = sum_n gP[s][h][n][d] * x[n][f]
        {
            OperandDesc A = plain_operand(b->gP, D);          // major = k (node), minor = m (s,h,d) segmented
            A.Dseg = D; A.Sseg = static_cast<int64_t>(N) * D;
            OperandDesc B = plain_operand(a->x, F);
            OutputDesc C = plain_output(b->g_a, W);
            C.P = static_cast<int32_t>(HD); C.S2 = F;
            const int sk = gemm_pick_split_k(static_cast<int32_t>(2 * HD), F, N);
            rc = gemm_f32(static_cast<int32_t>(2 * HD), F, N, A, false, B, false, C, sk, b->partial, st);
            if (rc != RECON_OK) return rc;
        }
        // (6) g_a[h][d][2F + r] = sum_k Gm[h][k][d] * edge_embed[eid[k]][r]
        if (E > 0) {
            OperandDesc A = plain_operand(b->Gm, D);
            A.Dseg = D; A.Sseg = static_cast<int64_t>(E) * D;
            OperandDesc B = plain_operand(a->edge_embed, R);
            B.gather = g->eid;
            OutputDesc C = plain_output(b->g_a + 2 * F, W);
            const int sk = gemm_pick_split_k(static_cast<int32_t>(HD), R, E);
            rc = gemm_f32(static_cast<int32_t>(HD), R, E, A, false, B, false, C, sk, b->partial, st);
            if (rc != RECON_OK) return rc;
        } else {
            // no edges: the relation block of g_a is zero
            for (int64_t row = 0; row < HD; ++row)
                if (hipMemsetAsync(b->g_a + row * W + 2 * F, 0, sizeof(float) * R, st) != hipSuccess) return RECON_ERR_LAUNCH;
        }
    }
    return RECON_OK;
}


// ---- G1-G3: SpecialSpmmFinal as a stand-alone op -------------------------------------------
namespace {
// Segmented row sum, robust to very long segments (a relation type owning 10^4 edges) and to tiny ones:
// pass 1: one wave walks kSL consecutive CSR slots in order (lanes span the columns), sums runs of equal
//         destination in registers, writes segments that lie entirely inside its slot range straight to `out`
//         and the (at most two) segments that cross its range boundary to carry[wave][0 = continues from the
//         left | 1 = continues to the right];
// pass 2: one wave per destination that spans several ranges adds its carries in slot order.  Fixed order,
//         no atomics; rows without slots get their zeros here (no memset of `out` in front).
// 32 slots per wave, sixteen rows in flight: a wave's walk is a chain of dependent round trips — two of them now (256 slots x 8 rows in
// flight = 32 took 75 - 105 us per call whatever the width and 100 k slots filled only 98 workgroups; 128 x 16 = 8: 27 us at the stage-A
// sizes; 32: the iteration on a re-used batch 1.47 -> 1.39 ms, 16 and 64 slots measured beside it: 1.41 / 1.40-1.5).  The carries are
// 2 C floats per 32 slots: 1/16 of the values read.
constexpr int kSL = 32;
template <int VEC>
__global__ void __launch_bounds__(256) k_rowsum_walk(const int32_t* __restrict__ dst,
                                                     const int32_t* __restrict__ eid, const float* __restrict__ w, int32_t E,
                                                     int32_t C, float* __restrict__ out, float* __restrict__ carry, int32_t row_mod) {
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int k0 = wv * kSL;
    if (k0 >= E) return;
    const int k1 = min(E, k0 + kSL);
    const int c = blockIdx.y * 64 * VEC + lane * VEC;
    const bool ca = c < C;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    int cur = dst[k0];
    // Which run of equal destinations crosses the range's ends follows from the two slots just outside it — requested here, under the first
    // rows' round trip.  (Looking the run's row up in rowptr at every change of destination was a dependent round trip per run: at the
    // stage-A graphs' 2-20 slots per row, most of a wave's time.)
    const bool left_cross = k0 > 0 && dst[k0 - 1] == cur;
    const bool right_cross = k1 < E && dst[k1] == dst[k1 - 1];
    bool first = true;
    auto flush = [&](int node, bool last) {
        float* target;
        if (first && left_cross) target = carry + (static_cast<int64_t>(wv) * 2 + 0) * C;            // continues from the left
        else if (last && right_cross) target = carry + (static_cast<int64_t>(wv) * 2 + 1) * C;       // continues to the right only
        else target = out + static_cast<int64_t>(node) * C;                                          // complete here
        first = false;
        if (ca) store_vec<VEC>(target + c, acc);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    };
    // sixteen rows in flight, branch-free loads (a guarded load gets its own basic block and is then waited for on its own):
    // slots past the range re-read its last slot and are skipped at the accumulate; lanes past C read a clamped column
    const int cc = ca ? c : 0;
    constexpr int U = 16;
    for (int k = k0; k < k1; k += U) {
        float t[U][VEC];
        int d[U], e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = min(k + u, k1 - 1);
            d[u] = dst[kk];
            e[u] = eid[kk];
            if (row_mod) e[u] %= row_mod;                            // several keys share one row of `w` (gather_rows_pair: key k and key k + E)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) load_vec<VEC>(t[u], w + static_cast<int64_t>(e[u]) * C + cc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u < k1) {
                if (d[u] != cur) { flush(cur, false); cur = d[u]; } // wave-uniform
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += t[u][v];
            }
        }
    }
    flush(cur, true);
}
template <int VEC>
__global__ void __launch_bounds__(256) k_rowsum_fix(const int32_t* __restrict__ rowptr, int32_t N, int32_t C,
                                                    const float* __restrict__ carry, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= N) return;
    const int b = rowptr[node], e = rowptr[node + 1];
    if (e <= b) {                                                   // a row without slots: its zeros (the memset launch this replaces cleared all of `out`)
        float z[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] = 0.f;
        for (int c = lane * VEC; c < C; c += 64 * VEC) store_vec<VEC>(out + static_cast<int64_t>(node) * C + c, z);
        return;
    }
    const int wa = b / kSL, wb = (e - 1) / kSL;
    if (wa == wb) return;                                           // lay inside one range: already written
    for (int c = lane * VEC; c < C; c += 64 * VEC) {
        float s[VEC];
        load_vec<VEC>(s, carry + (static_cast<int64_t>(wa) * 2 + 1) * C + c);
        // a relation owning 10^4 edges spans ~80 ranges: eight carries' loads in flight, added in range order
        constexpr int UF = 8;
        int wv = wa + 1;
        for (; wv + UF <= wb + 1; wv += UF) {
            float t[UF][VEC];
#pragma unroll
            for (int u = 0; u < UF; ++u) load_vec<VEC>(t[u], carry + (static_cast<int64_t>(wv + u) * 2 + 0) * C + c);
#pragma unroll
            for (int u = 0; u < UF; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) s[v] += t[u][v];
        }
        for (; wv <= wb; ++wv) {
            float t[VEC];
            load_vec<VEC>(t, carry + (static_cast<int64_t>(wv) * 2 + 0) * C + c);
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[v] += t[v];
        }
        store_vec<VEC>(out + static_cast<int64_t>(node) * C + c, s);
    }
}
__global__ void k_spmm_rowsum_bwd(const int64_t* __restrict__ dst, int64_t E, int32_t C, const float* __restrict__ gout,
                                  float* __restrict__ gw) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= E * C) return;
    gw[idx] = gout[dst[idx / C] * C + idx % C];
}
// table[i0] + table[i1] per output row: one thread per VEC columns of a row (the two index words of a row are one 16-byte line for its threads)
template <int VEC>
__global__ void __launch_bounds__(256) k_gather_rows_pair(const float* __restrict__ table, const int64_t* __restrict__ idx2, int64_t E, int32_t C,
                                                           float* __restrict__ out) {
    const int cpr = C / VEC;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= E * cpr) return;
    const int64_t r = t / cpr;
    const int c = static_cast<int>(t - r * cpr) * VEC;
    const int64_t i0 = idx2[2 * r], i1 = idx2[2 * r + 1];
    float x[VEC], y[VEC];
    load_vec<VEC>(x, table + i0 * C + c);
    load_vec<VEC>(y, table + i1 * C + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) x[v] += y[v];
    store_vec<VEC>(out + r * C + c, x);
}
}  // namespace

extern "C" int recon_gather_rows_pair_fwd(const float* table, const int64_t* idx2, int64_t E, int32_t width, float* out, recon_stream_t stream) {
    if (E < 0 || width <= 0) return RECON_ERR_INVALID;
    if (E == 0) return RECON_OK;
    if (!table || !idx2 || !out) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    const int vec = (width % 4 == 0 && al(table, 16) && al(out, 16)) ? 4 : ((width % 2 == 0 && al(table, 8) && al(out, 8)) ? 2 : 1);
    const int64_t threads = E * (width / vec);
    const int64_t blocks = ceil_div64(threads, 256);
    if (blocks > 0x7fffffffLL) return RECON_ERR_UNSUPPORTED;
    const dim3 grid(static_cast<unsigned>(blocks));
    if (vec == 4) hipLaunchKernelGGL((k_gather_rows_pair<4>), grid, dim3(256), 0, st, table, idx2, E, width, out);
    else if (vec == 2) hipLaunchKernelGGL((k_gather_rows_pair<2>), grid, dim3(256), 0, st, table, idx2, E, width, out);
    else hipLaunchKernelGGL((k_gather_rows_pair<1>), grid, dim3(256), 0, st, table, idx2, E, width, out);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" size_t recon_spmm_rowsum_workspace_floats(int32_t E, int32_t out_features) {
    return static_cast<size_t>(ceil_div64(E > 0 ? E : 1, kSL)) * 2 * static_cast<size_t>(out_features > 0 ? out_features : 1);
}

extern "C" int recon_spmm_rowsum_fwd(const recon_graph* g, const float* edge_w, int32_t out_features, float* out,
                                     float* workspace, recon_stream_t stream) {
    return recon_spmm_rowsum_mod_fwd(g, edge_w, out_features, 0, out, workspace, stream);
}

extern "C" int recon_spmm_rowsum_mod_fwd(const recon_graph* g, const float* edge_w, int32_t out_features, int32_t row_mod, float* out,
                                         float* workspace, recon_stream_t stream) {
    if (!g || !out || out_features <= 0 || row_mod < 0 || (g->E > 0 && (!edge_w || !workspace))) return RECON_ERR_INVALID;
    const int64_t total = static_cast<int64_t>(g->N) * out_features;
    if (total == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    if (g->E == 0) {                                                 // no slots: every row is zero
        if (hipMemsetAsync(out, 0, sizeof(float) * total, st) != hipSuccess) return RECON_ERR_LAUNCH;
        return RECON_OK;
    }
    const int C = out_features;
    const int vec = (C % 4 == 0 && al(edge_w, 16) && al(out, 16) && al(workspace, 16)) ? 4
                  : ((C % 2 == 0 && al(edge_w, 8) && al(out, 8) && al(workspace, 8)) ? 2 : 1);       // C = 50: the reference's relation width
    const int nw = static_cast<int>(ceil_div64(g->E, kSL));
    dim3 grid(static_cast<unsigned>(ceil_div64(nw, 4)), static_cast<unsigned>(ceil_div64(C, 64 * vec)));
    dim3 fgrid(static_cast<unsigned>(ceil_div64(g->N, 4)));
    if (vec == 4) {
        hipLaunchKernelGGL((k_rowsum_walk<4>), grid, dim3(256), 0, st, g->dst, g->eid, edge_w, g->E, C, out, workspace, row_mod);
        hipLaunchKernelGGL((k_rowsum_fix<4>), fgrid, dim3(256), 0, st, g->rowptr_dst, g->N, C, workspace, out);
    } else if (vec == 2) {
        hipLaunchKernelGGL((k_rowsum_walk<2>), grid, dim3(256), 0, st, g->dst, g->eid, edge_w, g->E, C, out, workspace, row_mod);
        hipLaunchKernelGGL((k_rowsum_fix<2>), fgrid, dim3(256), 0, st, g->rowptr_dst, g->N, C, workspace, out);
    } else {
        hipLaunchKernelGGL((k_rowsum_walk<1>), grid, dim3(256), 0, st, g->dst, g->eid, edge_w, g->E, C, out, workspace, row_mod);
        hipLaunchKernelGGL((k_rowsum_fix<1>), fgrid, dim3(256), 0, st, g->rowptr_dst, g->N, C, workspace, out);
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_spmm_rowsum_bwd(const int64_t* edge_dst, int64_t E, const float* grad_out, int32_t out_features,
                                     float* grad_edge_w, recon_stream_t stream) {
    if (E < 0 || out_features <= 0) return RECON_ERR_INVALID;
    if (E == 0) return RECON_OK;
    if (!edge_dst || !grad_out || !grad_edge_w) return RECON_ERR_INVALID;
    hipLaunchKernelGGL(k_spmm_rowsum_bwd, dim3(static_cast<unsigned>(ceil_div64(E * out_features, 256))), dim3(256), 0,
                       as_stream(stream), edge_dst, E, out_features, grad_out, grad_edge_w);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_version(void) { return RECON_ABI_VERSION; }
extern "C" const char* recon_error_string(int code) {
    switch (code) {
        case RECON_OK: return "ok";
        case RECON_ERR_INVALID: return "invalid argument";
        case RECON_ERR_UNSUPPORTED: return "unsupported shape";
        case RECON_ERR_LAUNCH: return "kernel launch failed";
        case RECON_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}
