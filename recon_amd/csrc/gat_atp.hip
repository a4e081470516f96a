// K1' — "aggregate, then project" formulation of the KB-GAT layer (all H heads per wave).
//
// GAT/layers.py:129-178 projects every edge (m_e = a.[x_dst; x_src; r_e], an E x (2F+R) x D GEMM per
// head) and then aggregates.  Both steps are linear, so they commute:
//     s_e   = a_2.m_e = u_dst.x_dst + u_src.x_src + u_rel.r_e            u = a_2^T a   (2F+R per head)
//     h_i   = sum_e k_e w_e m_e / Z_i = a . V_i,
//     V_i   = [ x_i Zk_i/Z_i ;  sum_e k_e w_e x_src(e) / Z_i ;  sum_e k_e w_e r_e / Z_i ]
// i.e. the scores need only three skinny dot products per node / edge, the edge stage aggregates RAW
// feature rows (read once for all heads), and the only large GEMM left is N x (2F+R) x D per head:
// half the MFMA work of project-then-aggregate when E = 4N, and no E x H*D intermediate in HBM.
// Results differ from the reference only in fp32 summation order.
//
// Layouts (fp32): u [H][W], c_node [N][2H] (dst half | src half), c_rel / sigma / keep [E][H] in
// CSR-slot order, Z / Zk [N][H], V [N][H][W], W = 2F+R.
#include <math.h>
#include <stdlib.h>
#include "recon_common.h"

#include <type_traits>

namespace recon {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
// VEC consecutive elements of a feature / embedding row: float32, or (B16) bfloat16 storage widened on the way in — the layer's boundary in
// BASELINE.json configs[4] ("mixed GAT+Propagation stack, bf16"): x and edge_embed are read as they are stored, not from up-cast copies
template <int VEC, bool B16>
__device__ __forceinline__ void load_in(float (&r)[VEC], const float* base, int64_t e) {
    if constexpr (!B16) {
        load_vec<VEC>(r, base + e);
    } else {
        const uint16_t* b = reinterpret_cast<const uint16_t*>(base) + e;
        if constexpr (VEC == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(b);
            r[0] = __builtin_bit_cast(float, t.x << 16); r[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
            r[2] = __builtin_bit_cast(float, t.y << 16); r[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
        } else if constexpr (VEC == 2) {
            const uint32_t t = *reinterpret_cast<const uint32_t*>(b);
            r[0] = __builtin_bit_cast(float, t << 16); r[1] = __builtin_bit_cast(float, t & 0xffff0000u);
        } else {
            r[0] = __builtin_bit_cast(float, static_cast<uint32_t>(*b) << 16);
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }      // v_pk_fma_f32

__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// u[h][w] = sum_d a_2[h][d] * a[h][d][w]      block = (64 columns w) x (16 row groups over d), fixed-order combine.
// 16 groups with 4 independent partial sums each: with 4 groups and one dependent chain the kernel was latency
// bound (15 us for 3.8 MB).
__global__ void __launch_bounds__(1024) k_score_vec(const float* __restrict__ a, const float* __restrict__ a2, int32_t D, int32_t W,
                                                    float* __restrict__ u, uint32_t* __restrict__ aux) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int w = blockIdx.x * 64 + c, h = blockIdx.y;
    const float* ah = a + static_cast<int64_t>(h) * D * W;
    const int per = (D + 15) / 16;
    const int d0 = grp * per, d1 = min(D, (grp + 1) * per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, mx = 0.f;             // mx: max |a| seen by this thread (every element is read once)
    if (w < W) {
        int d = d0;
        for (; d + 4 <= d1; d += 4) {
            const float v0 = ah[static_cast<int64_t>(d) * W + w], v1 = ah[static_cast<int64_t>(d + 1) * W + w];
            const float v2 = ah[static_cast<int64_t>(d + 2) * W + w], v3 = ah[static_cast<int64_t>(d + 3) * W + w];
            s0 = fmaf(a2[h * D + d], v0, s0);
            s1 = fmaf(a2[h * D + d + 1], v1, s1);
            s2 = fmaf(a2[h * D + d + 2], v2, s2);
            s3 = fmaf(a2[h * D + d + 3], v3, s3);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
        }
        for (; d < d1; ++d) {
            const float v0 = ah[static_cast<int64_t>(d) * W + w];
            s0 = fmaf(a2[h * D + d], v0, s0);
            mx = fmaxf(mx, fabsf(v0));
        }
    }
    red[grp][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && w < W) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][c];
        u[static_cast<int64_t>(h) * W + w] = t;
    }
    if (aux) {
        // First kernel of a forward pass in f16 x 2 mode.  (1) Its share of max |a| goes out as ONE plain store per block into
        // a word of quantity 0 that no slot uses (hx2_blkmax_word) — k_hx2_split_both, which needs the scale first, gathers the
        // words and writes the 32 slots the GEMMs read: nothing of quantity 0 has to be zero beforehand.  (2) Quantities 1..3 and
        // the page of zeros, which later kernels fill by atomicMax / read as zeros, are cleared here: the fill launch this
        // replaces cost 4.5 us per step.
        __syncthreads();
        red[grp][c] = mx;
        __syncthreads();
        if (threadIdx.x < 64) {
            float m = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) m = fmaxf(m, red[g][threadIdx.x]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            const int b = blockIdx.y * gridDim.x + blockIdx.x;
            if (threadIdx.x == 0) aux[hx2_blkmax_word(b)] = __builtin_bit_cast(uint32_t, m);
        }
        const int nblk = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        const int total = static_cast<int>(kHx2AuxBytes / 4) - kHx2QuantityWords;      // everything behind quantity 0
        const int per_b = (total + nblk - 1) / nblk;
        for (int i = b * per_b + threadIdx.x; i < min(total, (b + 1) * per_b); i += 1024) aux[kHx2QuantityWords + i] = 0u;
    }
}

// backward of u = a_2^T a:   g_a[h][d][:] += a_2[h][d] * g_u[h][:] ;  g_a_2[h][d] = a[h][d][:] . g_u[h][:]
__global__ void __launch_bounds__(256) k_score_vec_bwd(const float* __restrict__ a, const float* __restrict__ a2,
                                                       const float* __restrict__ gu, int32_t D, int32_t W, float* __restrict__ ga,
                                                       float* __restrict__ ga2, uint32_t* __restrict__ rezero, int32_t rezero_words) {
    __shared__ float red[256];
    const int d = blockIdx.x, h = blockIdx.y;
    // last kernel of a backward pass: hand the max-magnitude slots of grad_out back zeroed, so that a second backward over the
    // same saved state (retain_graph) publishes into clean slots
    if (rezero && d == 0 && h == 0)
        for (int i = threadIdx.x; i < rezero_words; i += 256) rezero[i] = 0u;
    const int64_t row = (static_cast<int64_t>(h) * D + d) * W;
    const float a2v = a2[h * D + d];
    float s = 0.f;
    for (int w = threadIdx.x; w < W; w += 256) {
        const float g = gu[static_cast<int64_t>(h) * W + w];
        s = fmaf(a[row + w], g, s);
        if (ga) ga[row + w] += a2v * g;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && ga2) ga2[h * D + d] = red[0];
}

// Last kernel of a backward pass in f16 x 2 mode: the fixed-order second pass of the split-K weight gradient
// (partial[h][z][w][d] -> g_a[h][d][w], transposed through LDS) with the score path's a_2[h][d] * g_u[h][w] added on the way, and —
// in the blocks behind the tiles — g_a_2[h][d] = a[h][d][:] . g_u[h][:], 8 rows per block.  As k_splitk_reduce_t + k_score_vec_bwd
// this was two launches and a second read-modify-write pass over g_a.  W % 4 == 0.
__global__ void __launch_bounds__(256) k_atp_weights_finish(const float* __restrict__ partial, int32_t splits, int32_t W, int32_t D, int32_t H,
                                                            const float* __restrict__ a, const float* __restrict__ a2,
                                                            const float* __restrict__ gu, float* __restrict__ ga, float* __restrict__ ga2,
                                                            uint32_t* __restrict__ rezero, int32_t rezero_words) {
    __shared__ float tile[32][33];
    if (rezero && blockIdx.x == 0)                                       // see k_score_vec_bwd
        for (int i = threadIdx.x; i < rezero_words; i += 256) rezero[i] = 0u;
    const int tw = (W + 31) / 32, td = (D + 31) / 32, ntile = tw * td * H;
    if (static_cast<int>(blockIdx.x) < ntile) {
        const int b = blockIdx.x, h = b / (td * tw), m0 = ((b / td) % tw) * 32, n0 = (b % td) * 32;      // m = w, n = d
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const int64_t MN = static_cast<int64_t>(W) * D;
        const float* pz = partial + static_cast<int64_t>(h) * splits * MN;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t off[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ty + 8 * i, n = n0 + tx;
            ok[i] = m < W && n < D;
            off[i] = ok[i] ? static_cast<int64_t>(m) * D + n : 0;
        }
        for (int z = 0; z < splits; ++z) {                              // split index outermost: four independent chains, fixed order
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += pz[z * MN + off[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = ok[i] ? acc[i] : 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + ty + 8 * i, m = m0 + tx;
            if (m < W && n < D)
                ga[(static_cast<int64_t>(h) * D + n) * W + m] = fmaf(a2[h * D + n], gu[static_cast<int64_t>(h) * W + m], tile[tx][ty + 8 * i]);
        }
    } else if (ga2) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int r0 = (static_cast<int>(blockIdx.x) - ntile) * 8 + wave * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + i;                                        // r = h * D + d
            if (r >= H * D) break;                                       // wave-uniform
            const int h = r / D;
            float sum = 0.f;
            for (int w = 4 * lane; w < W; w += 256) {
                const float4 av = *reinterpret_cast<const float4*>(a + static_cast<int64_t>(r) * W + w);
                const float4 gv = *reinterpret_cast<const float4*>(gu + static_cast<int64_t>(h) * W + w);
                sum = fmaf(av.x, gv.x, fmaf(av.y, gv.y, fmaf(av.z, gv.z, fmaf(av.w, gv.w, sum))));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if (lane == 0) ga2[r] = sum;
        }
    }
}

// One wave per node: g_h = g_y * elu'(h) (written when concat) and q[node][h] = g_h[h,:] . h[h,:], the
// (g_V . V) term of d loss / d Z  (g_V = g_h a and a V = h, so the 2F+R-wide dot collapses to a D-wide one).
// h is recovered from y = elu(h): y > 0 ? y : log1p(y).
__global__ void __launch_bounds__(256) k_elu_grad_q(const float* __restrict__ gy, int32_t ld_gy, const float* __restrict__ y,
                                                    int32_t ld_y, int32_t N, int32_t H, int32_t D, int32_t concat,
                                                    float* __restrict__ gh, float* __restrict__ q, uint16_t* __restrict__ ghp,
                                                    int64_t ld_p, int64_t plane_p, int32_t f16x2, const Hx2Scale gsc,
                                                    float* __restrict__ row_inv, int64_t row_inv_ld, uint32_t* __restrict__ amax_q,
                                                    const int32_t* __restrict__ row_node) {
    // row_node (row compaction, recon_graph.n_rows; NULL: rows are nodes): N counts ROWS, y and every output go by row, and row r's
    // gradient is row row_node[r] of gy (the caller's grad_out covers all nodes).
    // ghp (optional): term planes of g_h for the split-precision GEMMs — three bfloat16 planes [3][N][ld_p], or (f16x2) the
    // half terms of s * g_h, head-major [H][N][2][D] (ld_p = 2 D, plane_p = D).  The scale s:
    //   row_inv == nullptr   one per tensor, from the published max |grad_out| (|elu'| <= 1, so it bounds |g_h|): a pass over grad_out
    //                        (k_hx2_amax, 52 MB at cfg 2) has to run first;
    //   row_inv              (D <= 256) ONE PER ROW (node, head), a power of two from the row's own maximum, found in the wave that holds
    //                        the row: no pass in front.  1 / s goes to row_inv[h][node] (row stride row_inv_ld, padded with ones to a
    //                        multiple of 8 nodes), the tensor's maximum to amax_q on the way.  The consumers: g_V = g_h a has the rows on its
    //                        M index (the epilogue multiplies row-wise), g_a^T = V^T g_h contracts over them (the product multiplies V's
    //                        fragments by s_tensor / s_row, gemm_hx2.hip).
    constexpr int IPW = 4;                                               // (node, head) rows per wave, loads batched
    const int lane = threadIdx.x & 63;
    const int item0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * IPW;      // N*H < 2^31 (checked by the host)
    const int total = N * H;
    if (item0 >= total) return;
    const bool v4 = ((D | ld_gy | ld_y) & 3) == 0;
    if (v4 && D <= 256) {
        // All 2 IPW row loads and the scale's slot load go out together, branch free (rows past the end re-read the last row,
        // lanes past D column 0; both are masked afterwards): behind a guard each load pair is a basic block of its own and the
        // compiler drains the counter at every join — IPW dependent round trips, plus one for a scale read in front of them.
        const int c = 4 * lane;
        const bool cok = c < D;
        const int cc = cok ? c : 0;
        float4 g4[IPW], y4[IPW];
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int it = min(item0 + j, total - 1);
            const int node = it / H, h = it % H;
            g4[j] = *reinterpret_cast<const float4*>(gy + static_cast<int64_t>(row_node ? row_node[node] : node) * ld_gy + h * D + cc);
            y4[j] = *reinterpret_cast<const float4*>(y + static_cast<int64_t>(node) * ld_y + h * D + cc);
        }
        const float gs_t = (f16x2 && !row_inv) ? hx2_scale_wave(gsc) : 1.f;
        float ov[IPW][4], pv[IPW], rmx[IPW];
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const float gv[4] = {g4[j].x, g4[j].y, g4[j].z, g4[j].w}, yv[4] = {y4[j].x, y4[j].y, y4[j].z, y4[j].w};
            float part = 0.f, mx = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float g = gv[v], hv = yv[v];
                if (concat && yv[v] <= 0.f) { const float e = yv[v] + 1.f; g = gv[v] * e; hv = e > 0.f ? __logf(e) : 0.f; }   // h = log(y+1); its error is multiplied by exp(h) <= 1
                ov[j][v] = g;
                part = fmaf(g, hv, part);
                mx = fmaxf(mx, fabsf(g));
            }
            pv[j] = cok ? part : 0.f;
            rmx[j] = cok ? mx : 0.f;
        }
        float gs_row[IPW], wave_amax = 0.f;
        static_assert(IPW == 4, "row reductions below");
        // the IPW row sums q with 3 permutes + DPP steps (multi_sum: row j's total lands in lanes [16 j, 16 j + 16)) instead of IPW full butterflies
        const float tot_l = multi_sum<4>(pv, lane);
        if (row_inv) {
            const float m = multi_max4(rmx, lane);                      // row j's maximum in lanes [16 j, 16 j + 16)
            float all = 0.f;
#pragma unroll
            for (int j = 0; j < IPW; ++j) {
                const float rm = lane_bcast(m, 16 * j);
                gs_row[j] = hx2_scale_of(rm);
                if (item0 + j < total) all = fmaxf(all, rm);
            }
            wave_amax = all;
        } else {
#pragma unroll
            for (int j = 0; j < IPW; ++j) gs_row[j] = gs_t;
        }
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int it = item0 + j;
            const float gs = gs_row[j];
            const float (&o)[4] = ov[j];
            const float tot = lane_bcast(tot_l, 16 * j);
            if (it < total) {
                if (gh && c < D) *reinterpret_cast<float4*>(gh + static_cast<int64_t>(it) * D + c) = make_float4(o[0], o[1], o[2], o[3]);
                if (ghp && c < D) {
                    const int node = it / H, h = it % H;
                    uint16_t* dst = f16x2 ? ghp + (static_cast<int64_t>(h) * N + node) * ld_p + c : ghp + static_cast<int64_t>(node) * ld_p + h * D + c;
                    float x[4] = {o[0], o[1], o[2], o[3]};
                    if (f16x2) {
                        uint32_t hi[2], lo[2];
                        hx2_split2(x[0] * gs, x[1] * gs, hi[0], lo[0]);
                        hx2_split2(x[2] * gs, x[3] * gs, hi[1], lo[1]);
                        *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0], hi[1]);
                        *reinterpret_cast<uint2*>(dst + plane_p) = make_uint2(lo[0], lo[1]);
                    } else
#pragma unroll
                    for (int pq = 0; pq < 3; ++pq) {
                        uint32_t w[2];
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const __bf16 b0 = static_cast<__bf16>(x[2 * hh]), b1 = static_cast<__bf16>(x[2 * hh + 1]);
                            const uint32_t u0 = __builtin_bit_cast(uint16_t, b0), u1 = __builtin_bit_cast(uint16_t, b1);
                            w[hh] = u0 | (u1 << 16);
                            x[2 * hh] -= __builtin_bit_cast(float, u0 << 16);
                            x[2 * hh + 1] -= __builtin_bit_cast(float, u1 << 16);
                        }
                        *reinterpret_cast<uint2*>(dst + pq * plane_p) = make_uint2(w[0], w[1]);
                    }
                }
                if (lane == 0) {
                    q[it] = tot;
                    if (row_inv) row_inv[static_cast<int64_t>(it % H) * row_inv_ld + it / H] = hx2_inv(gs);
                }
                if (row_inv && it / H == N - 1 && lane >= 1 && lane <= ((8 - (N & 7)) & 7))      // the table's padding rows: scale 1
                    row_inv[static_cast<int64_t>(it % H) * row_inv_ld + N - 1 + lane] = 1.f;
            }
        }
        // the tensor's maximum goes out LAST: the commit reads its slot first (a device-scope round trip), and in front of the stores it held
        // every wave's stores back by that trip
        if (row_inv && amax_q) hx2_amax_commit_uniform(wave_amax, amax_q);
        return;
    }
    const float gs = f16x2 ? hx2_scale_wave(gsc) : 1.f;
    if (v4) {                                                            // wide heads (out_att: D = H*D of the heads): 256 columns per pass
        for (int j = 0; j < IPW; ++j) {
            const int it = item0 + j;
            if (it >= total) break;
            const int node = it / H, h = it % H;
            const float* gr = gy + static_cast<int64_t>(row_node ? row_node[node] : node) * ld_gy + h * D;
            const float* yr = y + static_cast<int64_t>(node) * ld_y + h * D;
            float part = 0.f;
            for (int c = 4 * lane; c < D; c += 256) {
                const float4 g4 = *reinterpret_cast<const float4*>(gr + c), y4 = *reinterpret_cast<const float4*>(yr + c);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
                float x[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float g = gv[v], hv = yv[v];
                    if (concat && yv[v] <= 0.f) { const float e = yv[v] + 1.f; g = gv[v] * e; hv = e > 0.f ? __logf(e) : 0.f; }
                    x[v] = g;
                    part = fmaf(g, hv, part);
                }
                if (gh) *reinterpret_cast<float4*>(gh + static_cast<int64_t>(it) * D + c) = make_float4(x[0], x[1], x[2], x[3]);
                if (ghp) {
                    uint16_t* dst = f16x2 ? ghp + (static_cast<int64_t>(h) * N + node) * ld_p + c : ghp + static_cast<int64_t>(node) * ld_p + h * D + c;
                    if (f16x2) {
                        uint32_t hi[2], lo[2];
                        hx2_split2(x[0] * gs, x[1] * gs, hi[0], lo[0]);
                        hx2_split2(x[2] * gs, x[3] * gs, hi[1], lo[1]);
                        *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0], hi[1]);
                        *reinterpret_cast<uint2*>(dst + plane_p) = make_uint2(lo[0], lo[1]);
                    } else
#pragma unroll
                    for (int pq = 0; pq < 3; ++pq) {
                        uint32_t w[2];
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const __bf16 b0 = static_cast<__bf16>(x[2 * hh]), b1 = static_cast<__bf16>(x[2 * hh + 1]);
                            const uint32_t u0 = __builtin_bit_cast(uint16_t, b0), u1 = __builtin_bit_cast(uint16_t, b1);
                            w[hh] = u0 | (u1 << 16);
                            x[2 * hh] -= __builtin_bit_cast(float, u0 << 16);
                            x[2 * hh + 1] -= __builtin_bit_cast(float, u1 << 16);
                        }
                        *reinterpret_cast<uint2*>(dst + pq * plane_p) = make_uint2(w[0], w[1]);
                    }
                }
            }
            const float tot = group_sum<64>(part);
            if (lane == 0) q[it] = tot;
        }
        return;
    }
    for (int j = 0; j < IPW; ++j) {
        const int it = item0 + j;
        if (it >= total) break;
        const int node = it / H, h = it % H;
        const float* gr = gy + static_cast<int64_t>(row_node ? row_node[node] : node) * ld_gy + h * D;
        const float* yr = y + static_cast<int64_t>(node) * ld_y + h * D;
        float* go = gh ? gh + static_cast<int64_t>(it) * D : nullptr;
        float part = 0.f;
        for (int c = lane; c < D; c += 64) {
            float g = gr[c], hv = yr[c];
            if (concat && hv <= 0.f) { const float e = hv + 1.f; g = g * e; hv = e > 0.f ? __logf(e) : 0.f; }
            if (go) go[c] = g;
            part = fmaf(g, hv, part);
        }
        const float tot = group_sum<64>(part);
        if (lane == 0) q[it] = tot;
    }
}

// out[row][j] = X[row'] . U_j,  U_j = u + (j % H)*W + (j / H)*F + off,  row' = gather ? gather[row] : row.
// One wave per kRowsPerIter consecutive rows (their loads are issued together: the kernel is latency bound, one row
// at a time it reached 1.3 TB/s); U staged in LDS once per block, so blocks are sized for >= 32 rows.
constexpr int kRowsPerIter = 4;
// Two independent jobs (node scores and edge scores) share ONE launch: blocks [0, nb0) run job 0, the rest job 1, so
// the small node job does not leave the chip three quarters empty for its whole latency-bound duration.
struct RowDotsJob { const float* X; const int32_t* gather; int32_t rows, K, F, off, NJ, nb; float* out; uint32_t* amax; };
template <int VEC>
__global__ void __launch_bounds__(kBlock) k_row_dots(const RowDotsJob j0, const RowDotsJob j1, const float* __restrict__ u, int32_t H,
                                                     int32_t W) {
    extern __shared__ __attribute__((aligned(16))) float U[];     // [NJ][K]
    const bool second = static_cast<int>(blockIdx.x) >= j0.nb;
    const RowDotsJob& jb = second ? j1 : j0;
    const float* __restrict__ X = jb.X;
    const int32_t* __restrict__ gather = jb.gather;
    float* __restrict__ out = jb.out;
    const int rows = jb.rows, K = jb.K, F = jb.F, off = jb.off, NJ = jb.NJ;
    const int bid = second ? blockIdx.x - j0.nb : blockIdx.x, nblocks = jb.nb;
    for (int j = 0; j < NJ; ++j) {
        const float* uj = u + static_cast<int64_t>(j % H) * W + (j / H) * F + off;
        for (int k = threadIdx.x; k < K; k += kBlock) U[j * K + k] = uj[k];
    }
    __syncthreads();
    constexpr int RB = kRowsPerIter;
    const int lane = threadIdx.x & 63;
    const int wave = bid * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = nblocks * (kBlock / 64);
    float mx = 0.f;                                                    // max |X| over the rows this wave reads (every row is read by one wave)
    for (int row0 = wave * RB; row0 < rows; row0 += nwaves * RB) {
        const float* xr[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row = min(row0 + b, rows - 1);                      // rows past the end recompute the last row, not stored
            xr[b] = X + static_cast<int64_t>(gather ? gather[row] : row) * K;
        }
        float mine[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) mine[b] = 0.f;
        for (int j0 = 0; j0 < NJ; j0 += 8) {                            // eight dot products share one multi-value reduction
            float part[RB][8];
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) part[b][jj] = 0.f;
            for (int c = lane * VEC; c < K; c += 64 * VEC) {
                float xv[RB][VEC];
#pragma unroll
                for (int b = 0; b < RB; ++b) load_vec<VEC>(xv[b], xr[b] + c);
                if (j0 == 0) {
#pragma unroll
                    for (int b = 0; b < RB; ++b)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) mx = fmaxf(mx, fabsf(xv[b][v]));
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    if (j0 + jj < NJ) {
                        float uv[VEC];
                        load_vec<VEC>(uv, U + (j0 + jj) * K + c);
#pragma unroll
                        for (int b = 0; b < RB; ++b)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) part[b][jj] = fmaf(xv[b][v], uv[v], part[b][jj]);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const float tot = multi_sum<8>(part[b], lane);          // lane l holds column j0 + (l >> 3)
                const float t = __shfl(tot, (lane & 7) << 3, 64);       // every lane l now holds column j0 + (l & 7)
                if (lane >= j0 && lane < j0 + 8) mine[b] = t;           // lane j keeps column j
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b)
            if (lane < NJ && row0 + b < rows) out[static_cast<int64_t>(row0 + b) * NJ + lane] = mine[b];
    }
    if (jb.amax) hx2_amax_commit(mx, jb.amax);
}

// The same products on the fp32 matrix cores — out[16 rows][NJ <= 16] = X[16 x K] . U^T with v_mfma_f32_16x16x4_f32 (exact fp32
// products, fp32 accumulation): the VALU form above spends a multi-value wave reduction per 8 dot products and runs at 2.5x its
// traffic bound.  One wave per 16 rows; lane (i = lane & 15, q = lane >> 4) loads 16 bytes of row i per 16-k group (k = 16 g + 4 q ..
// + 3: the MFMA's k slot q carries these four k in four steps — any assignment works as long as A and B agree), eight groups in
// flight (two register batches); U sits in LDS as [NJ][Kp], Kp = 4 mod 8 so that the 16 columns' 16-byte reads fall on disjoint banks.  K % 4 == 0.
using f32x4_t = __attribute__((ext_vector_type(4))) float;
__device__ __host__ inline int row_dots_kp(int K) { return (K % 8 == 0) ? K + 4 : K; }
// The last nsplit blocks write the half planes of a and a^T (hx2_split_both_block): that pass needs only what
// the kernel in front of this one published, and as a launch of its own it cost 10 us of which 5 are kernel turn-around.
// The same dots for rows stored as bfloat16, on the bf16 matrix cores: a lane's 16-byte piece of a row IS an A fragment of
// v_mfma_f32_16x16x32_bf16 (row lane & 15, eight consecutive k), and the score vectors lie in LDS as TWO bf16 terms (u = hi + lo, relative
// error 2^-17: the rows themselves carry 2^-9), one 16-byte B fragment each — 2 MFMAs of 16 cycles per 32 columns where the float32 form
// issues eight 16x16x4 MFMAs of 32 (34 us for 62 MB at cfg 5: the kernel was bound by its own matrix instructions).
__device__ __forceinline__ int row_dots_b16_stride(int K) { return ((K / 8) & 1) ? K * 2 : K * 2 + 16; }      // bytes; (stride / 16) odd: 16 rows on distinct banks
__device__ __forceinline__ void row_dots_b16(const RowDotsJob& jb, int bid, const float* __restrict__ u, int32_t H, int32_t W, unsigned char* Usm) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    const uint16_t* __restrict__ X = reinterpret_cast<const uint16_t*>(jb.X);
    const int32_t* __restrict__ gather = jb.gather;
    float* __restrict__ out = jb.out;
    const int rows = jb.rows, K = jb.K, F = jb.F, off = jb.off, NJ = jb.NJ, nblocks = jb.nb;
    const int RS = row_dots_b16_stride(K);
    unsigned char* Uh = Usm;                                             // [NJ][RS] bf16 high terms
    unsigned char* Ul = Usm + 16 * RS;                                   // ... low terms
    constexpr int GU = 4;
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntiles = (rows + 15) / 16, G = (K + 31) / 32;
    const bool iu = i < NJ;
    float mx = 0.f;
    bool staged = false;
    for (int tile = bid * (kBlock / 64) + wave; tile < ntiles || !staged; tile += nblocks * (kBlock / 64)) {
        const bool live = tile < ntiles;
        const int row = min(tile * 16 + i, rows - 1);
        const int64_t xr = static_cast<int64_t>(gather ? gather[row] : row) * K;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        uint4 xv[2][GU];
        int cc[2][GU];
        auto request = [&](int g0, int buf) {
#pragma unroll
            for (int t = 0; t < GU; ++t) {                                // branch free: groups past K re-read column 0 and are zeroed
                const int c = 32 * (g0 + t) + 8 * q;
                const bool ok = c < K;
                cc[buf][t] = ok ? c : 0;
                const uint4 raw = *reinterpret_cast<const uint4*>(X + xr + cc[buf][t]);
                xv[buf][t] = ok ? raw : make_uint4(0, 0, 0, 0);
            }
        };
        request(0, 0);
        if (!staged) {
            for (int idx = threadIdx.x; idx < NJ * K; idx += kBlock) {
                const int j = idx / K, k = idx - j * K;
                const float v = u[static_cast<int64_t>(j % H) * W + (j / H) * F + off + k];
                const __bf16 hi = static_cast<__bf16>(v);
                const __bf16 lo = static_cast<__bf16>(v - static_cast<float>(hi));
                *reinterpret_cast<__bf16*>(Uh + j * RS + 2 * k) = hi;
                *reinterpret_cast<__bf16*>(Ul + j * RS + 2 * k) = lo;
            }
            __syncthreads();
            staged = true;
        }
        for (int g0 = 0; g0 < G; g0 += 2 * GU) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int gb = g0 + half * GU;
                if (gb >= G) break;                                       // wave-uniform
                if (gb + GU < G) request(gb + GU, half ^ 1);
#pragma unroll
                for (int t = 0; t < GU; ++t) {
                    const uint4 xr4 = xv[half][t];
                    const uint32_t w4[4] = {xr4.x, xr4.y, xr4.z, xr4.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        mx = fmaxf(mx, fmaxf(fabsf(__builtin_bit_cast(float, w4[d] << 16)), fabsf(__builtin_bit_cast(float, w4[d] & 0xffff0000u))));
                    uint4 bh = *reinterpret_cast<const uint4*>(Uh + (iu ? i : 0) * RS + 2 * cc[half][t]);
                    uint4 bl = *reinterpret_cast<const uint4*>(Ul + (iu ? i : 0) * RS + 2 * cc[half][t]);
                    if (!iu) { bh = make_uint4(0, 0, 0, 0); bl = make_uint4(0, 0, 0, 0); }
                    const bf16x8_t af = __builtin_bit_cast(bf16x8_t, xr4);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, bl), acc, 0, 0, 0);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                 // C layout: row 4 q + r, column i
                const int ro = tile * 16 + 4 * q + r;
                if (ro < rows && i < NJ) out[static_cast<int64_t>(ro) * NJ + i] = acc[r];
            }
        }
    }
    if (jb.amax) hx2_amax_commit(mx, jb.amax);
}

template <bool B16>
__global__ void __launch_bounds__(kBlock) k_row_dots_x(const RowDotsJob j0, const RowDotsJob j1, const float* __restrict__ u, int32_t H,
                                                       int32_t W, const Hx2SplitBoth sp, int32_t nsplit) {
    extern __shared__ __attribute__((aligned(16))) float U[];     // [NJ][Kp] (the launch reserves max NJ = 2 H rows)
    if (static_cast<int>(blockIdx.x) >= j0.nb + j1.nb) {           // last in the grid (first: 24.6 us for the launch against 21.2)
        const int lin = blockIdx.x - (j0.nb + j1.nb);
        hx2_split_both_block(sp, lin % sp.gx, (lin / sp.gx) % sp.batch, lin / (sp.gx * sp.batch));
        return;
    }
    const int bx = blockIdx.x;
    const bool second = bx >= j0.nb;
    const RowDotsJob& jb = second ? j1 : j0;
    if constexpr (B16) { row_dots_b16(jb, second ? bx - j0.nb : bx, u, H, W, reinterpret_cast<unsigned char*>(U)); return; }
    const float* __restrict__ X = jb.X;
    const int32_t* __restrict__ gather = jb.gather;
    float* __restrict__ out = jb.out;
    const int rows = jb.rows, K = jb.K, F = jb.F, off = jb.off, NJ = jb.NJ, Kp = row_dots_kp(jb.K);
    const int bid = second ? bx - j0.nb : bx, nblocks = jb.nb;
    // VW elements per lane and request: bfloat16 rows are read 16 bytes at a time as float32 rows are (8 elements: with 4 a request moved
    // half the bytes for the same issue slot — 40 us for 62 MB at cfg 5 — and a wave instruction costs the vector-memory path the same
    // whatever it moves); a group is then 4 VW columns and half as many groups are in a register batch
    constexpr int VW = B16 ? 8 : 4, GU = B16 ? 4 : 8;                    // groups per register batch, two batches in flight
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntiles = (rows + 15) / 16, G = (K + 4 * VW - 1) / (4 * VW);
    const bool iu = i < NJ;                                              // columns past NJ multiply by zero: U holds NJ rows only
    const float* Ui = U + (iu ? i : 0) * Kp;
    float mx = 0.f;
    bool staged = false;
    for (int tile = bid * (kBlock / 64) + wave; tile < ntiles || !staged; tile += nblocks * (kBlock / 64)) {
        const bool live = tile < ntiles;                                  // a wave without a tile still takes part in the staging
        const int row = min(tile * 16 + i, rows - 1);                    // rows past the end recompute the last row, not stored
        const int64_t xr = static_cast<int64_t>(gather ? gather[row] : row) * K;      // element offset of the row (the rows are float32 or bfloat16)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        // two register batches of GU groups: batch b+1 is requested in front of batch b's MFMAs (rows wider than 256 columns —
        // out_att-sized inputs — were one dependent round trip per batch: 150 us for 262 MB)
        float xv[2][GU][VW];
        int cc[2][GU];
        auto request = [&](int g0, int buf) {
#pragma unroll
            for (int t = 0; t < GU; ++t) {                                // branch free: groups past K re-read column 0 and are zeroed
                const int c = 4 * VW * (g0 + t) + VW * q;
                const bool ok = c < K;
                cc[buf][t] = ok ? c : 0;
                float xq[VW];
                if constexpr (B16) {
                    const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(X) + xr + cc[buf][t]);
                    const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) { xq[2 * d] = __builtin_bit_cast(float, w4[d] << 16); xq[2 * d + 1] = __builtin_bit_cast(float, w4[d] & 0xffff0000u); }
                } else {
                    load_in<4, false>(xq, X, xr + cc[buf][t]);
                }
#pragma unroll
                for (int e = 0; e < VW; ++e) xv[buf][t][e] = ok ? xq[e] : 0.f;
            }
        };
        request(0, 0);
        if (!staged) {                                                    // the score vectors arrive while the first rows are in flight
            for (int idx = threadIdx.x; idx < NJ * Kp; idx += kBlock) {
                const int j = idx / Kp, k = idx - j * Kp;
                U[idx] = k < K ? u[static_cast<int64_t>(j % H) * W + (j / H) * F + off + k] : 0.f;
            }
            __syncthreads();
            staged = true;
        }
        for (int g0 = 0; g0 < G; g0 += 2 * GU) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int gb = g0 + half * GU;
                if (gb >= G) break;                                       // wave-uniform
                if (gb + GU < G) request(gb + GU, half ^ 1);
#pragma unroll
                for (int t = 0; t < GU; ++t) {
#pragma unroll
                    for (int e0 = 0; e0 < VW; e0 += 4) {
                        f32x4_t uv = *reinterpret_cast<const f32x4_t*>(Ui + cc[half][t] + e0);
                        if (!iu) uv = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            mx = fmaxf(mx, fabsf(xv[half][t][e0 + e]));
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[half][t][e0 + e], uv[e], acc, 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                 // C layout: row 4 q + r, column i
                const int ro = tile * 16 + 4 * q + r;
                if (ro < rows && i < NJ) out[static_cast<int64_t>(ro) * NJ + i] = acc[r];
            }
        }
    }
    if (jb.amax) hx2_amax_commit(mx, jb.amax);
}

struct AtpFwdK {
    const int32_t* rowptr; const int32_t* src; const int32_t* eid;
    const float* x; const float* ee; const float* c_node; const float* c_rel; const float* keep;
    float* V; float* sigma; float* Z; float* Zk;
    int32_t N, E, F, R, H;
    float alpha;
    int32_t crel_by_row;        // 1: edge_embed is a table (recon_gat_atp_args.ee_index) and c_rel holds one entry per table ROW
    int32_t dst_shared;         // 1: the destination part of V (x_i Zk/Z) is written for head 0 only — without attention dropout Zk = Z and
                                // every head's copy is the same row; the GEMMs read head 0's (a_shared_k / a_shared_m)
    int32_t planes;             // 1 (2: with paired 16-byte stores, F % 8 == 0 and R % 8 == 0): V is written as half terms [H][N][2][W] of s_V * V (gemm_hx2.hip: high row | low row) instead of fp32 [N][H][W]
    Hx2Scale vs;                // s_V from max(|x|, |edge_embed|) * keep_max, an upper bound of |V| (V rows are k-weighted means)
    // hub rows (recon_graph): the first n_piece waves of the grid walk one piece each and leave UNNORMALISED partial sums in hubS
    // [n_piece][H][F + R] and hubZ [n_piece][2][H]; the wave of a node with more than hub_chunk slots does nothing, and
    // k_gat_atp_hub_fwd sums the pieces in table order, normalises and writes the node's V / Z / Zk.  hub_chunk = 0: off.
    int32_t hub_chunk, n_piece;
    const int4* piece;
    float* hubS; float* hubZ;
    int32_t* nan_flag;          // optional device word raised when a row sum is NaN / infinite (config.hip: recon_set_nan_flag)
    // row compaction (recon_graph.n_rows): N above is the number of ROWS — what rowptr, the pieces, V, Z and Zk are indexed by — and row r
    // aggregates into node row_node[r] (x and c_node are node tables); NULL: rows are nodes
    const int32_t* row_node;
};

// two half planes of VEC consecutive ALREADY SCALED values: 2 * VEC bytes per plane.  Values are clamped to half's range: the
// scale comes from a bound, not from V itself — max(|x|, |edge_embed|) bounds V exactly without dropout (rows of V are means),
// with dropout the caller's keep_max enters; one v_med3 per value guards against a wrong one.
__device__ __forceinline__ float hx2_clamp(float c) { return __builtin_amdgcn_fmed3f(c, -65504.f, 65504.f); }
template <int VEC>
__device__ __forceinline__ void store_planes(_Float16* hi_p, int64_t plane, const float (&c)[VEC]) {
    if constexpr (VEC == 4) {
        uint32_t hi[2], lo[2];
        hx2_split2(hx2_clamp(c[0]), hx2_clamp(c[1]), hi[0], lo[0]);
        hx2_split2(hx2_clamp(c[2]), hx2_clamp(c[3]), hi[1], lo[1]);
        *reinterpret_cast<uint2*>(hi_p) = make_uint2(hi[0], hi[1]);
        *reinterpret_cast<uint2*>(hi_p + plane) = make_uint2(lo[0], lo[1]);
    } else {
        static_assert(VEC == 2, "vector width");
        uint32_t hi, lo;
        hx2_split2(hx2_clamp(c[0]), hx2_clamp(c[1]), hi, lo);
        *reinterpret_cast<uint32_t*>(hi_p) = hi;
        *reinterpret_cast<uint32_t*>(hi_p + plane) = lo;
    }
}

// The same for VEC = 4 with 16-byte stores: lanes 2i and 2i+1 hold columns 8i .. 8i+7 between them; they swap halves (DPP
// quad_perm [1,0,3,2]) so that the even lane writes the 8 high terms into the high row and the odd lane the 8 low terms into
// the low row — one dwordx4 store instruction where store_planes issues two dwordx2.  Requires both lanes of a pair active
// together and 16-byte aligned parts: F % 8 == 0, R % 8 == 0.
__device__ __forceinline__ void store_planes_paired(_Float16* hi_p, int64_t plane, const float (&c)[4], bool odd) {
    uint32_t hi[2], lo[2];
    hx2_split2(hx2_clamp(c[0]), hx2_clamp(c[1]), hi[0], lo[0]);
    hx2_split2(hx2_clamp(c[2]), hx2_clamp(c[3]), hi[1], lo[1]);
    constexpr int kSwapPairs = 0xB1;                                  // quad_perm [1,0,3,2]
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t phi = __builtin_amdgcn_mov_dpp(hi[j], kSwapPairs, 0xf, 0xf, true);      // the partner's terms
        const uint32_t plo = __builtin_amdgcn_mov_dpp(lo[j], kSwapPairs, 0xf, 0xf, true);
        w[j] = odd ? plo : hi[j];                                     // even: own high terms (columns 8i..8i+3) | odd: partner's low terms
        w[2 + j] = odd ? lo[j] : phi;                                 // even: partner's high terms (8i+4..8i+7) | odd: own low terms
    }
    _Float16* dst = odd ? hi_p + plane - 4 : hi_p;                    // odd lane's own columns start at 8i+4
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
}

// wave = one destination node, HT heads (blockIdx.y selects the head group); lanes span the feature
// dimension: lane l owns columns (r*64 + l)*VEC .. +VEC of both the x row (F) and the relation row (R).
// PL: 0 = V as fp32 [N][H][W]; 1 = half-term rows [H][N][2][W] (gemm_hx2.hip), 8-byte stores; 2 = the same through paired 16-byte stores
// A wave can walk kK1NodesPerWave nodes (4 i + wave of its block's 4 * kK1NodesPerWave consecutive ones) with everything of node
// i+1 that does not depend on its edges — row pointers, the index vectors of its first 64 slots, its own row, its score term —
// requested while node i is being worked on.  Measured at cfg 2 with s_memtime stamps in the kernel (1.9 GHz under load): a wave
// lives 23.4 k cycles per node = 5.1 k until its index vectors are there (two dependent round trips), 9.2 k in the edge walk,
// 7.6 k converting and issuing its 24 stores, 0.9 k until they are acknowledged.  Taking the first 5.1 k off the chain (2 nodes per
// wave, one round of 4 096 waves) did NOT shorten the kernel: 38 us against 36 — what the kernel waits for is the memory system
// under 160 MB of writes, not its own chain (same for 4 rows in flight instead of 2, for 4 or 2 heads per wave, for 8- or 16-byte
// stores: 35.5 - 41 us).  Kept at 1.
constexpr int kK1NodesPerWave = 1;
template <int VEC, int KR, int HT, bool TRAIN, int PL, bool B16 = false>
__global__ void __launch_bounds__(kBlock, KR >= 8 ? 2 : 1) k_gat_atp_fwd(const AtpFwdK p) {      // KR = 8: 260 registers unbounded, one short of two waves per SIMD
    // edges in flight per wave: their rows are requested together, so a node of degree <= UNR costs ONE row round trip
    // (four edges in flight for one-row inputs, round 6: 30.9 against 29.6 us at cfg 2 — three waves per SIMD instead of four —, 63 against 69 us
    // on cfg 5's bfloat16 leg, 64 against 63 on its float32 leg: not kept)
    constexpr int UNR = (KR * HT >= 16 || KR >= 8) ? 1 : 2;       // KR = 8: two edges in flight are 128 registers of rows
    constexpr int NPW = kK1NodesPerWave;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(NPW == 1, "hub pieces are handed out one per wave");
    // pieces first (they are the longest rows of the launch), in blockIdx order so that they are dealt round over the XCDs; the
    // nodes behind them in XCD-contiguous ranges
    const int npb = (p.n_piece + kBlock / 64 - 1) / (kBlock / 64);
    const bool is_piece = static_cast<int>(blockIdx.x) < npb;
    const int widx = blockIdx.x * (kBlock / 64) + wave;                  // piece id when is_piece
    int first = xcd_block(blockIdx.x - npb, gridDim.x - npb) * ((kBlock / 64) * NPW) + wave, pbeg = 0, pend = 0;
    if (is_piece) {
        if (widx >= p.n_piece) return;
        const int4 pc = p.piece[widx]; first = pc.x; pbeg = pc.y; pend = pc.z;
    } else if (first >= p.N) return;
    const int F = p.F, R = p.R, H = p.H, W = 2 * F + R;
    const int h0 = blockIdx.y * HT;
    const int myh = h0 + (lane % HT);
    const bool hv = myh < H;
    const int myhc = hv ? myh : h0;
    int cf[KR]; bool aF[KR], aR[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { cf[r] = (r * 64 + lane) * VEC; aF[r] = cf[r] < F; aR[r] = cf[r] < R; }
    // Loads are branch free: a guarded load gets its own basic block and the value join makes the compiler wait for each
    // load on its own, so the row of x[src] and the row of r_e (and the score terms) would arrive one round trip after
    // the other.  Slots past the row re-read its last slot (weight forced to 0), lanes past F / R read column 0 (their
    // accumulators are never stored), lanes past H read head h0.
    int cfF[KR], cfR[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { cfF[r] = aF[r] ? cf[r] : 0; cfR[r] = aR[r] ? cf[r] : 0; }
    // The walk is a chain of dependent round trips (row pointers -> slot indices -> rows): the slot -> (source node, edge id)
    // indices of up to 64 slots come with ONE coalesced load per array (lane j holds slot beg + j) and are handed out with
    // v_readlane.
    int begs[NPW], ends[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int nd = min(first + (kBlock / 64) * i, p.N - 1);
        begs[i] = p.rowptr[nd]; ends[i] = p.rowptr[nd + 1];
    }
    if (is_piece) { begs[0] = pbeg; ends[0] = pend; }
    else if (p.hub_chunk && ends[0] - begs[0] > p.hub_chunk) return;    // a hub: its pieces and k_gat_atp_hub_fwd write this node
    struct NodeIn { int srcv, eidv; float cd; float xi[KR][VEC]; };
    auto request = [&](NodeIn& q, int row, int beg, int end) {          // row < N
        q.srcv = 0; q.eidv = 0;
        if (beg < end) { const int kk = beg + min(lane, min(64, end - beg) - 1); q.srcv = p.src[kk]; q.eidv = p.eid[kk]; }
        const int node = p.row_node ? p.row_node[row] : row;             // (wave-uniform) the node this row aggregates into
        q.cd = p.c_node[static_cast<int64_t>(node) * 2 * H + myhc];
#pragma unroll
        for (int r = 0; r < KR; ++r) load_in<VEC, B16>(q.xi[r], p.x, static_cast<int64_t>(node) * F + cfF[r]);
    };
    NodeIn cur, nxt;
    request(cur, first, begs[0], ends[0]);
    const float vscale = PL ? hx2_scale_wave(p.vs) : 1.f;             // one vector load + butterfly, in flight under the edge walk
    const bool odd = lane & 1;
    const int64_t vplane = W;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int node = first + (kBlock / 64) * i;
        if (node >= p.N) break;                                          // wave-uniform
        const int beg = begs[i], end = ends[i];
        if (i + 1 < NPW && node + (kBlock / 64) < p.N) request(nxt, node + (kBlock / 64), begs[i + 1 < NPW ? i + 1 : i], ends[i + 1 < NPW ? i + 1 : i]);
        int cn = min(64, end - beg);
        int srcv = cur.srcv, eidv = cur.eidv;
        const float cd = hv ? cur.cd : 0.f;
        float accS[HT][KR][VEC], accR[HT][KR][VEC];
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
            for (int r = 0; r < KR; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v) { accS[h][r][v] = 0.f; accR[h][r][v] = 0.f; }
        float Zl = 0.f, Zkl = 0.f;
        for (int c0 = beg; c0 < end; c0 += 64) {
            if (c0 > beg) {                                              // next chunk of a long row: new index vectors
                cn = min(64, end - c0);
                const int kk = c0 + min(lane, cn - 1);
                srcv = p.src[kk]; eidv = p.eid[kk];
            }
            for (int j0 = 0; j0 < cn; j0 += UNR) {
                float xs[UNR][KR][VEC], re[UNR][KR][VEC], sc[UNR], kf[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int j = min(j0 + u, cn - 1);
                    const int64_t k = c0 + j;
                    const int s = __builtin_amdgcn_readlane(srcv, j), e = __builtin_amdgcn_readlane(eidv, j);
                    const int64_t xr = static_cast<int64_t>(s) * F, rr = static_cast<int64_t>(e) * R;
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        load_in<VEC, B16>(xs[u][r], p.x, xr + cfF[r]);
                        load_in<VEC, B16>(re[u][r], p.ee, rr + cfR[r]);
                    }
                    sc[u] = cd + p.c_node[static_cast<int64_t>(s) * 2 * H + H + myhc] + p.c_rel[(p.crel_by_row ? static_cast<int64_t>(e) : k) * H + myhc];
                    kf[u] = p.keep ? p.keep[k * H + myhc] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (j0 + u < cn) {                                      // wave-uniform
                        const int64_t k = c0 + j0 + u;
                        const float sg = sc[u];
                        const float w = hv ? expf(-(sg > 0.f ? sg : p.alpha * sg)) : 0.f;
                        const float kw = kf[u] * w;
                        Zl += w;
                        Zkl += kw;
                        if constexpr (TRAIN) { if (hv && lane < HT) p.sigma[k * H + myh] = sg; }
#pragma unroll
                        for (int h = 0; h < HT; ++h) {
                            const float kwh = lane_bcast(kw, h);
#pragma unroll
                            for (int r = 0; r < KR; ++r)
#pragma unroll
                                for (int v = 0; v < VEC; ++v) {
                                    accS[h][r][v] = fmaf(kwh, xs[u][r][v], accS[h][r][v]);
                                    accR[h][r][v] = fmaf(kwh, re[u][r][v], accR[h][r][v]);
                                }
                        }
                    }
                }
            }
        }
        if (is_piece) {                                                  // wave-uniform: partial sums, nothing normalised
            if (hv && lane < HT) { p.hubZ[(static_cast<int64_t>(widx) * 2) * H + myh] = Zl; p.hubZ[(static_cast<int64_t>(widx) * 2 + 1) * H + myh] = Zkl; }
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                if (h0 + h < H) {
                    float* dst = p.hubS + (static_cast<int64_t>(widx) * H + h0 + h) * (F + R);
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        if (aF[r]) store_vec<VEC>(dst + cf[r], accS[h][r]);
                        if (aR[r]) store_vec<VEC>(dst + F + cf[r], accR[h][r]);
                    }
                }
            }
            return;
        }
        const float Zc = (Zl == 0.f) ? 1e-12f : Zl;                     // GAT/layers.py:152
        if (p.nan_flag && hv && lane < HT && !(fabsf(Zc) <= 3.0e38f)) *p.nan_flag = 1;      // the reference's asserts (:147, :167, :172)
        const float inv = 1.f / Zc;
        if constexpr (TRAIN) {
            if (hv && lane < HT) { p.Z[static_cast<int64_t>(node) * H + myh] = Zc; p.Zk[static_cast<int64_t>(node) * H + myh] = Zkl; }
        }
        // Every load has landed by now on every path; say so.  Without it the compiler meets "a load may still be in flight" at the
        // join behind each head's (uniform) guard below and drains the counter there — vmcnt counts stores too, so every head would
        // wait for the previous head's stores to COMPLETE.
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0), nothing else
        auto put = [&](float* Vr, _Float16* Vh, int col, const float (&o)[VEC]) {
            if constexpr (PL == 0) store_vec<VEC>(Vr + col, o);
            else if constexpr (PL == 2 && VEC == 4) store_planes_paired(Vh + col, vplane, o, odd);
            else store_planes<VEC>(Vh + col, vplane, o);
        };
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            if (h0 + h < H) {
                const float invh = lane_bcast(inv, h) * vscale;           // the operand scale rides on the normalisation
                const float zk = lane_bcast(Zkl, h) * invh;
                float* Vr = p.V + (static_cast<int64_t>(node) * H + h0 + h) * W;
                // half terms: HEAD-major [H][N][2][W] — a (node, head) still writes one contiguous 4 W-byte piece (high row | low
                // row), and the 128 rows a GEMM workgroup reads for one head are one compact 128 x 4 W-byte region instead of
                // 64-byte pieces strewn over H x 4 W-byte strides
                _Float16* Vh = reinterpret_cast<_Float16*>(p.V) + 2 * (static_cast<int64_t>(h0 + h) * p.N + node) * W;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    float o[VEC];
                    if (aF[r]) {
                        if (!(p.dst_shared && h0 + h > 0)) {                  // uniform
#pragma unroll
                            for (int v = 0; v < VEC; ++v) o[v] = cur.xi[r][v] * zk;
                            put(Vr, Vh, cf[r], o);
                        }
#pragma unroll
                        for (int v = 0; v < VEC; ++v) o[v] = accS[h][r][v] * invh;
                        put(Vr, Vh, F + cf[r], o);
                    }
                    if (aR[r]) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) o[v] = accR[h][r][v] * invh;
                        put(Vr, Vh, 2 * F + cf[r], o);
                    }
                }
            }
        }
        cur = nxt;
    }
}

// acc += sum over pieces q in [p0, p1) of the VEC floats at base + q * stride, in table order.  A hub of 9 000 edges has 142 pieces:
// one load at a time is 142 dependent round trips (measured on a 272 k-edge graph with such hubs: 78 us for the forward combine,
// 109 us for the source-side one), so sixteen pieces' loads are issued before the first add.
template <int VEC>
__device__ __forceinline__ void sum_pieces(float (&acc)[VEC], const float* __restrict__ base, int64_t stride, int p0, int p1) {
    constexpr int U = 16;
    int q = p0;
    for (; q + U <= p1; q += U) {
        float tv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) load_vec<VEC>(tv[u], base + static_cast<int64_t>(q + u) * stride);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += tv[u][v];
    }
    for (; q < p1; ++q) {
        float tv[VEC];
        load_vec<VEC>(tv, base + static_cast<int64_t>(q) * stride);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += tv[v];
    }
}

// The second half of a hub's forward: wave = one (hub, head).  Sums the pieces' partial sums in table order (fixed: results do not
// depend on scheduling), normalises as the epilogue above does and writes the node's V rows, Z and Zk.  8-byte plane stores: hubs are few.
template <int VEC, int PL, bool B16 = false>
__global__ void __launch_bounds__(kBlock) k_gat_atp_hub_fwd(const AtpFwdK p, const int32_t* __restrict__ hub_node, const int32_t* __restrict__ hub_ptr,
                                                            int32_t n_hub) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int idx = blockIdx.x * (kBlock / 64) + wave;
    const int F = p.F, R = p.R, H = p.H, W = 2 * F + R;
    if (idx >= n_hub * H) return;
    const int t = idx / H, h = idx - t * H;
    const int node = hub_node[t], p0 = hub_ptr[t], p1 = hub_ptr[t + 1];
    const float vscale = PL ? hx2_scale_wave(p.vs) : 1.f;
    // Z, Zk: lane q takes pieces p0 + q, p0 + q + 64, ..., then a butterfly — fixed order, and not two more serial walks of the pieces
    float zl = 0.f, zkl = 0.f;
    for (int q = p0 + lane; q < p1; q += 64) { zl += p.hubZ[(static_cast<int64_t>(q) * 2) * H + h]; zkl += p.hubZ[(static_cast<int64_t>(q) * 2 + 1) * H + h]; }
    const float Zl = group_sum<64>(zl), Zkl = group_sum<64>(zkl);
    const float Zc = (Zl == 0.f) ? 1e-12f : Zl;                         // GAT/layers.py:152
    if (p.nan_flag && lane == 0 && !(fabsf(Zc) <= 3.0e38f)) *p.nan_flag = 1;
    if (p.Z && lane == 0) { p.Z[static_cast<int64_t>(node) * H + h] = Zc; p.Zk[static_cast<int64_t>(node) * H + h] = Zkl; }
    const float invh = (1.f / Zc) * vscale, zk = Zkl * invh;
    float* Vr = p.V + (static_cast<int64_t>(node) * H + h) * W;
    _Float16* Vh = reinterpret_cast<_Float16*>(p.V) + 2 * (static_cast<int64_t>(h) * p.N + node) * W;
    auto put = [&](int col, const float (&o)[VEC]) {
        if constexpr (PL == 0) store_vec<VEC>(Vr + col, o);
        else store_planes<VEC>(Vh + col, W, o);
    };
    for (int c = lane * VEC; c < F + R; c += 64 * VEC) {                // partial sums [F | R] are columns F .. of V
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        sum_pieces<VEC>(acc, p.hubS + static_cast<int64_t>(h) * (F + R) + c, static_cast<int64_t>(H) * (F + R), p0, p1);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] *= invh;
        put(F + c, acc);
    }
    if (!(p.dst_shared && h > 0)) {
        for (int c = lane * VEC; c < F; c += 64 * VEC) {
            float xv[VEC];
            load_in<VEC, B16>(xv, p.x, static_cast<int64_t>(p.row_node ? p.row_node[node] : node) * F + c);
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[v] *= zk;
            put(c, xv);
        }
    }
}

struct AtpBwdK {
    const int32_t* rowptr; const int32_t* src; const int32_t* eid;
    const float* x; const float* ee; const float* keep; const float* sigma; const float* Z; const float* Zk;
    const float* q; const float* gV; const float* u;
    float* gsigma; float* Gs_dst; float* Gxs; float* gxd; float* g_ee;
    int32_t N, E, F, R, H;
    float alpha;
    // hub rows (recon_graph): the first n_piece waves walk one piece each and leave their sum of g_sigma in hubG [n_piece][H] (and
    // store no g_x row); the wave of a node with more than hub_chunk slots walks none of them (it still writes the direct part of
    // g_x and zero sums), and k_gat_atp_hub_bwd adds the pieces' sums in table order.  hub_chunk = 0: off.
    int32_t hub_chunk, n_piece;
    const int4* piece;
    float* hubG;
    int32_t gee_by_slot;        // edge_embed is a table read through `eid` (recon_gat_atp_args.ee_index): g_ee rows go by CSR slot
    int32_t gee_b16;            // (B16 instances, one head group per wave, rows by edge id) g_ee points to bfloat16 rows: rounded once, at the store
    int32_t persist;            // k_gat_atp_bwd: 1 = persistent waves over XCD-contiguous node ranges (graphs without hub pieces), 0 = one piece / node per wave
    // row compaction (recon_graph.n_rows): N is the number of ROWS — rowptr, the pieces, g_V, Z, Zk, q, gxd and Gs_dst go by row — and row r
    // belongs to node row_node[r] (x is a node table); NULL: rows are nodes
    const int32_t* row_node;
};

// ---- raw buffer access through wave-uniform ROW descriptors (round 6) -------------------------------------------------------------
// Every row k_gat_atp_bwd touches has a wave-uniform base (a node's block of g_V, the x / edge_embed row of an edge handed out by
// readlane, the per-slot rows of sigma / g_sigma / Gxs / g_edge_embed).  A descriptor per row — 4 SGPRs, built on the scalar unit —
// with num_records = the row's bytes makes the hardware's range check the mask: a lane whose byte offset lies past the row loads
// zeros and stores nothing, so the walk needs neither `lane < F ? v : 0` selects nor exec-masked store blocks, and an access costs
// ONE VGPR (the lane's byte offset, shared by all rows of an element size) instead of a 64-bit VGPR address (round 5: seven lane base
// pointers live across the walk, a v_lshl_add_u64 per access — the compiler re-associates `uniform row + lane offset` into
// `(tensor + lane offset) + uniform`, whatever the source says — and spills among them once anything else grew).
// On gfx9 soffset IS part of the check (a raw buffer access is dropped when voffset + inst_offset >= num_records - soffset: measured —
// a window of F x 4 bytes moved by soffset returned zeros for every row but the first), so where one descriptor serves several rows
// (a node's g_V block, a chunk's per-slot rows: the row / part goes into soffset) num_records covers them all and the lane mask
// rides in voffset itself: lanes past the row hold an offset no descriptor reaches (kK2Oob).
constexpr uint32_t kK2Oob = 0x7ffffff0u;
typedef uint32_t k2_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t k2_u32x2 __attribute__((ext_vector_type(2)));
#define K2_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(ptr)), 0, static_cast<int>(bytes), 0x00020000)
template <int VEC>
__device__ __forceinline__ void buf_load_f32(float (&r)[VEC], __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (VEC == 4) {
        // (elements copied to scalars first: __builtin_bit_cast(float, t.y) on the vector's element reads element 0 — clang 20 / ROCm 7.2)
        const k2_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        const uint32_t a = t.x, b = t.y, c = t.z, d = t.w;
        r[0] = __builtin_bit_cast(float, a); r[1] = __builtin_bit_cast(float, b); r[2] = __builtin_bit_cast(float, c); r[3] = __builtin_bit_cast(float, d);
    } else {
        const k2_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        const uint32_t a = t.x, b = t.y;
        r[0] = __builtin_bit_cast(float, a); r[1] = __builtin_bit_cast(float, b);
    }
}
template <int VEC, bool B16>
__device__ __forceinline__ void buf_load_in(float (&r)[VEC], __amdgpu_buffer_rsrc_t rs, uint32_t voff) {      // voff in bytes of the stored element type
    if constexpr (!B16) {
        buf_load_f32<VEC>(r, rs, voff, 0);
    } else if constexpr (VEC == 4) {
        const k2_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
        r[0] = __builtin_bit_cast(float, t.x << 16); r[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
        r[2] = __builtin_bit_cast(float, t.y << 16); r[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
    } else {
        const uint32_t t = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0);
        r[0] = __builtin_bit_cast(float, t << 16); r[1] = __builtin_bit_cast(float, t & 0xffff0000u);
    }
}
template <int VEC>
__device__ __forceinline__ void buf_store_f32(const float (&r)[VEC], __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff = 0) {
    if constexpr (VEC == 4)
        __builtin_amdgcn_raw_buffer_store_b128(k2_u32x4{__builtin_bit_cast(uint32_t, r[0]), __builtin_bit_cast(uint32_t, r[1]), __builtin_bit_cast(uint32_t, r[2]),
                                                        __builtin_bit_cast(uint32_t, r[3])}, rs, voff, soff, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b64(k2_u32x2{__builtin_bit_cast(uint32_t, r[0]), __builtin_bit_cast(uint32_t, r[1])}, rs, voff, soff, 0);
}

// wave = one destination node; head groups of HT are walked one after the other by the SAME wave so the
// per-edge outputs (g_edge_embed row, Gxs row) can be accumulated across groups without atomics.
// Two rows in flight per wave at three waves per SIMD (<= 168 registers) is the measured optimum at cfg 2: a four-deep ring needs
// 197 registers (two waves per SIMD: 101 us against 80), four deep at three waves spills (130 us; round 6, with the walk's registers down
// by the packed arithmetic: still 256 bytes of scratch inside the walk).
constexpr int kK2Ring = 2, kK2WavesPerSimd = 3;
// Wide rows (KR >= 4 register rows per lane: out_att-sized inputs) get the register budget of two waves per SIMD, KR = 8 of one: at
// three the KR = 8 form spilled 259 registers (0.73 ms per call at N = 8 192, F = R = 1 600).
// B16: x and edge_embed are stored as bfloat16 and read where they lie (recon_gat_atp_args.io_bf16; round 5: the backward no longer needs
// up-cast copies of them — at cfg 5, 62 MB read + 124 MB written by two cast kernels, and half of this kernel's gathered row bytes)
// LR (round 6; rows of at most 1 KiB: VEC = 4, KR = 1): the gathered rows x[src_e] / r_e of the next kK2LdsRing edges wait in LDS, copied
// there by LDS-DMA, instead of two edges' rows in registers.  Cycle stamps with the register ring: at cfg 2 (degree 4) the second pair of
// edges cost the walk a round trip (14.6 k of a node's 34 k cycles); at cfg 5's bfloat16 leg — per-edge rows streamed from HBM, rows of
// up to 64 slots per chunk — the walk was half of a wave's life (p50 22.6 k of 45 k cycles): two edges ahead cover L2's latency, not
// HBM's.  A slot is refilled as soon as its rows have been read into registers; ONE counted wait in front of a slot's reads — every
// iteration issues the same seven requests (the two copies of the rows of the edge RING ahead, three stores, the two score words of that
// edge; missing edges copy nothing through a zero-sized descriptor), so "all but the 5 + 7 (RING - 1) youngest" is exactly "this slot's
// copies have landed".  Eight registers less than the register ring.
constexpr int kK2LdsRingBytes = 8192;                                  // per wave: 4 slots of 2 x 1 KiB
// GE16 (B16 + LR instances, a separate one so that the usual instance carries none of it: with the branch inside, the cfg 5 bfloat16 walk went
// from 138 to 150 us): g_edge_embed rows are bfloat16, rounded where they are stored
template <int VEC, int KR, int HT, bool B16 = false, bool LR = false, bool GE16 = false>
__global__ void __launch_bounds__(kBlock, (KR * HT >= 8 && KR >= 8) ? 1 : ((KR >= 4 || KR * HT >= 8 && KR >= 2) ? 2 : kK2WavesPerSimd)) k_gat_atp_bwd(const AtpBwdK p) {
    static_assert(!LR || (VEC == 4 && KR == 1), "LDS row ring: rows of at most 1 KiB");
    extern __shared__ __attribute__((aligned(16))) float U[];          // [H][F + R]: u_dst | u_rel per head (u_src is k_gat_atp_src's business now)
#ifdef RECON_K2_STAMPS                                                   // cycle stamps of every wave, left in its gxd row instead of the gradient (tools/probe/k2_stamps.py)
    uint64_t stamp_[6];
    stamp_[0] = __builtin_readcyclecounter();
#endif
    const int F = p.F, R = p.R, H = p.H, W = 2 * F + R;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Two ways of handing out the work (p.persist, chosen per launch by the host):
    //  * graphs WITHOUT hub rows (batched sentence / context graphs: every node costs about the same) — PERSISTENT waves, round 6.  Cycle
    //    stamps of the one-node-per-wave form at cfg 2 (39 k cycles per wave): 9.4 k from entry until the row pointers, the slot indices
    //    (two dependent round trips) and the block's copy of u were there, 18.7 k for the node's rows of g_V, 10.6 k for the walk — a wave
    //    asked the memory system for nothing during half of its life.  A persistent wave stages u once and has the NEXT node's row
    //    pointers and index vectors in flight under the current node's rows: a node then starts with its requests.  XCD x (blocks x,
    //    x + 8, ...) owns the nodes [x, x + 1) * ceil(N / 8): the rows of x a batched graph's edges gather stay in ONE L2.
    //  * graphs WITH hub rows (knowledge graphs, power-law batches: a node costs anything between nothing and 64 slots) — one piece or
    //    node per wave, workgroups dealt by the hardware as they finish: pieces first (the longest rows of the launch, in blockIdx order so
    //    that they go round the XCDs), the nodes behind them in XCD-contiguous ranges.  Measured in between and not kept (K2' time per
    //    backward pass at cfg 5 float32 / stage A layer 1; this form: 156 / 44 us): persistent nodes with the pieces as a launch of their
    //    own (192 / 2 x 45: the pieces' 64-slot walks are a second kernel's worth of time), pieces dealt by wave number in front of a static
    //    share of nodes (156 / 62: the waves that got a piece run a whole share behind), pieces and nodes CLAIMED from counters (atomicAdd,
    //    one per XCD's range 256 bytes apart, claimed a node ahead, self-resetting: 197 / 91, cfg 2 73 against 53 — ~1 400 atomics per
    //    address cost more than the imbalance they removed).
    const bool persist = p.persist != 0;                                 // uniform over the launch
    const int npb = persist ? 0 : (p.n_piece + kBlock / 64 - 1) / (kBlock / 64);
    const int widx = blockIdx.x * (kBlock / 64) + wave;                  // piece id in a piece block
    const bool is_piece = static_cast<int>(blockIdx.x) < npb && widx < p.n_piece;
    const int xcd = blockIdx.x & 7, nbx = (gridDim.x + 7 - xcd) >> 3;    // (persist) blocks of this XCD, any grid size
    const int nchunk = (p.N + 7) >> 3;
    const int node_hi = persist ? min(p.N, (xcd + 1) * nchunk) : p.N, node_stride = nbx * (kBlock / 64);
    int node = persist ? xcd * nchunk + (blockIdx.x >> 3) * (kBlock / 64) + wave
                       : (static_cast<int>(blockIdx.x) < npb ? p.N : xcd_block(blockIdx.x - npb, gridDim.x - npb) * (kBlock / 64) + wave);
    // The walk is a chain of dependent round trips (row pointers -> slot indices -> rows); start it before anything else:
    // the slot -> (source node, edge id) indices of the first 64 slots come with ONE coalesced load per array (lane j holds
    // slot beg + j) and are handed out with v_readlane instead of an index load in front of every row load.
    int beg = 0, end = 0;
    if (is_piece) { const int4 pc = p.piece[widx]; node = pc.x; beg = pc.y; end = pc.z; }
    else if (node < node_hi) {
        beg = p.rowptr[node]; end = p.rowptr[node + 1];
        if (p.hub_chunk && end - beg > p.hub_chunk) end = beg;           // a hub: its pieces walk the row
    } else node = p.N;
    int srcv0 = 0, eidv0 = 0;
    if (beg < end) { const int kk = beg + min(lane, min(64, end - beg) - 1); srcv0 = p.src[kk]; eidv0 = p.eid[kk]; }
    const int UW = F + R;                                                // a head's row of U
    if (((F | R) & 3) == 0) {                                            // 16 bytes per request
        for (int idx = threadIdx.x; idx < (H * UW) >> 2; idx += kBlock) {
            const int h = (4 * idx) / UW, c = 4 * idx - h * UW;
            reinterpret_cast<float4*>(U)[idx] = *reinterpret_cast<const float4*>(p.u + h * W + (c < F ? c : F + c));
        }
    } else {
        for (int idx = threadIdx.x; idx < H * UW; idx += kBlock) { const int h = idx / UW, c = idx - h * UW; U[idx] = p.u[h * W + (c < F ? c : F + c)]; }
    }
    // (a copy instruction writes 64 lanes x 16 bytes whatever the row's length — lanes past the row write zeros: a row's room is 1 KiB)
    constexpr int RING = 4, ROWB = 1024, SLOTB = 2 * ROWB;                              // (LR) slots per wave, bytes per row / per slot
    static_assert(RING == 4, "the walk names the ring's slots");
    unsigned char* const ringb = reinterpret_cast<unsigned char*>(U + H * UW) + (kBlock / 64) * 512 + (LR ? wave * kK2LdsRingBytes : 0);
    if constexpr (LR) {                                                  // lanes past a row's end read what was here before the row: zeros, never NaN bit patterns
#pragma unroll
        for (int i = 0; i < kK2LdsRingBytes / 1024; ++i) reinterpret_cast<float4*>(ringb + 1024 * i)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (node >= p.N) return;
    constexpr uint32_t ES = B16 ? 2u : 4u;                               // bytes per stored element of x / edge_embed
    auto in_row = [](const float* base, int64_t row, int width) -> const float* {      // uniform base of row `row` of a [.][width] table of input elements
        if constexpr (B16) return reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(base) + row * width);
        else return base + row * width;
    };
    // the lane's columns: byte offsets into fp32 rows of F / R columns (voF / voR) and input rows (viF / viR), out of every range for lanes
    // past the row; uF / uR: the same columns clamped for the LDS rows of u
    uint32_t voF[KR], voR[KR], viF[KR], viR[KR];
    int uF[KR], uR[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
        const int c = (r * 64 + lane) * VEC;
        voF[r] = c < F ? static_cast<uint32_t>(c) * 4u : kK2Oob; voR[r] = c < R ? static_cast<uint32_t>(c) * 4u : kK2Oob;
        viF[r] = c < F ? static_cast<uint32_t>(c) * ES : kK2Oob; viR[r] = c < R ? static_cast<uint32_t>(c) * ES : kK2Oob;
        uF[r] = c < F ? c : 0; uR[r] = c < R ? c : 0;
    }
#pragma unroll 1
    for (;;) {                                                           // ---- one node (or piece) per iteration
#ifdef RECON_K2_STAMPS
    stamp_[1] = __builtin_readcyclecounter();
#endif
    node = __builtin_amdgcn_readfirstlane(node); beg = __builtin_amdgcn_readfirstlane(beg); end = __builtin_amdgcn_readfirstlane(end);
    const int cn0 = min(64, end - beg);
    // the next node of this wave: its row pointers are requested now, its index vectors below (behind this node's first requests)
    const int nnode = persist ? node + node_stride : p.N;
    const bool more = nnode < node_hi;                                   // wave-uniform
    int nbeg = 0, nend = 0;
    if (more) { nbeg = p.rowptr[nnode]; nend = p.rowptr[nnode + 1]; }
    // its index vectors wait in LDS (2 x 256 bytes per wave behind the rows of u), not in registers: two more live registers across the
    // walk were spilled the moment they arrived — a vmcnt(0) each, in front of the g_V requests
    int* const nidx = reinterpret_cast<int*>(U + H * UW) + wave * 128 + lane;
    float xi[KR][VEC], gxd[KR][VEC];
    {
        const auto rxi = K2_RSRC(in_row(p.x, p.row_node ? p.row_node[node] : node, F), F * ES);      // x is a NODE table (row compaction: AtpBwdK.row_node)
#pragma unroll
        for (int r = 0; r < KR; ++r) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) gxd[r][v] = 0.f;
            buf_load_in<VEC, B16>(xi[r], rxi, viF[r]);                   // lanes past F: zeros
        }
    }
    // lane l works for head h0 + (l >> SH) (the layout multi_sum leaves its totals in); lane h << SH speaks for head h
    constexpr int SH = HT == 8 ? 3 : HT == 4 ? 4 : HT == 2 ? 5 : 6;
    const int hl = lane >> SH;
    const bool writer = (lane & ((1 << SH) - 1)) == 0;
    const int ngroups = (H + HT - 1) / HT;
    for (int hg = 0; hg < ngroups; ++hg) {
        const int h0 = hg * HT;
        const int myh = h0 + hl;
        const bool hv = myh < H;
        // ring of PF register slots (static indices: the loop body is unrolled PF times) holding the rows / score of the next
        // PF edges; filled for the first edges BEFORE the g_V rows are requested, so both are in flight together
        // (LR: the walk is unrolled over the ring's slots — the two score words of an edge travel at the ring's distance too, in registers:
        // the compiler waits for ITS loads by counting the requests it knows, and a score word requested behind a copy and needed two edges
        // later would make it wait for that copy — in-order completion — whatever the ring's depth)
        constexpr int PF = LR ? 4 : (KR == 1 ? kK2Ring : (KR == 2 ? 2 : 1));
        const uint32_t mh4 = static_cast<uint32_t>(hv ? myh : h0) * 4u;
        int c0 = beg, cn = cn0, srcv = srcv0, eidv = eidv0;
        const float* keepp = p.keep ? p.keep : p.sigma;
        // per-slot rows (sigma, keep, g_sigma, Gxs, g_edge_embed by slot) of a CHUNK of up to 64 slots share one descriptor each, built once
        // per chunk: base = the chunk's first row, the slot's row in soffset (<= 64 rows: far below 4 GB), num_records = the chunk.
        // Per edge that leaves the two gathered rows to describe.
        const uint32_t Hb = static_cast<uint32_t>(H) * 4u, Fb = static_cast<uint32_t>(F) * 4u, Rb = static_cast<uint32_t>(R) * 4u;
        auto rsg = K2_RSRC(p.sigma + static_cast<int64_t>(c0) * H, 64 * Hb), rkf = K2_RSRC(keepp + static_cast<int64_t>(c0) * H, 64 * Hb);
        auto rgs = K2_RSRC(p.gsigma + static_cast<int64_t>(c0) * H, 64 * Hb);
        auto rGx = K2_RSRC(p.Gxs + static_cast<int64_t>(c0) * F, 64 * Fb);
        auto rGe = K2_RSRC((p.g_ee && p.gee_by_slot) ? p.g_ee + static_cast<int64_t>(c0) * R : p.Gxs, (p.g_ee && p.gee_by_slot) ? 64 * Rb : 0u);
        float xs_r[LR ? 1 : PF][KR][VEC], re_r[LR ? 1 : PF][KR][VEC], sg_r[PF], kf_r[PF];
        auto fetch_scores = [&](int slot, int j) {                       // the edge's score word and keep factor for this lane's head
            const uint32_t so = static_cast<uint32_t>(j) * Hb;
            sg_r[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsg, mh4, so, 0));
            kf_r[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rkf, mh4, so, 0));     // eval: re-reads sigma, replaced by 1 at the use (no branch)
        };
        auto fetch_edge = [&](int slot, int j) {                         // register ring.  j: position inside the chunk (uniform), clamped by the caller
            const int s_ = __builtin_amdgcn_readlane(srcv, j), e_ = __builtin_amdgcn_readlane(eidv, j);
            const auto rx = K2_RSRC(in_row(p.x, s_, F), F * ES);
            const auto re = K2_RSRC(in_row(p.ee, e_, R), R * ES);
#pragma unroll
            for (int r = 0; r < KR; ++r) {                               // lanes past F / R: zeros
                buf_load_in<VEC, B16>(xs_r[LR ? 0 : slot][r], rx, viF[r]);
                buf_load_in<VEC, B16>(re_r[LR ? 0 : slot][r], re, viR[r]);
            }
            fetch_scores(slot, j);
        };
        // LDS ring: both rows of the edge at position j of the chunk into slot j % RING (two copies; a position past the chunk copies nothing)
        const uint32_t vd = static_cast<uint32_t>(lane) * 16u;
        auto fill_rows = [&](int j) {
            const bool live = j < cn;                                    // wave-uniform
            const int jc = min(j, cn - 1);
            const int s_ = __builtin_amdgcn_readlane(srcv, jc), e_ = __builtin_amdgcn_readlane(eidv, jc);
            const dma_u32x4 dx = dma_descriptor(in_row(p.x, s_, F), live ? static_cast<uint32_t>(F) * ES : 0u);
            const dma_u32x4 de = dma_descriptor(in_row(p.ee, e_, R), live ? static_cast<uint32_t>(R) * ES : 0u);
            unsigned char* slot = ringb + (j & (RING - 1)) * SLOTB;
            dma16_buffer_to_lds(dx, vd, slot);
            dma16_buffer_to_lds(de, vd, slot + ROWB);
        };
        if (beg < end) {
            if constexpr (LR) {
#pragma unroll
                for (int u = 0; u < RING; ++u) fill_rows(u);
#pragma unroll
                for (int u = 0; u < PF; ++u) fetch_scores(u, min(u, cn - 1));
            } else {
#pragma unroll
                for (int u = 0; u < PF; ++u) fetch_edge(u, min(u, cn - 1));
            }
        }
        // Z, Zk, q of the node's heads through one descriptor each (lanes of missing heads read head h0's and are masked at the use): requested
        // BEHIND the rows of g_V below — in front of them the three words were spilled as they arrived, a vmcnt(0) each, and the batch of
        // g_V requests left a round trip late
        const auto rZ = K2_RSRC(p.Z + static_cast<int64_t>(node) * H, H * 4), rZk = K2_RSRC(p.Zk + static_cast<int64_t>(node) * H, H * 4);
        const auto rq = K2_RSRC(p.q + static_cast<int64_t>(node) * H, H * 4);
        float Zl, Zkl, ql;
        auto load_norms = [&]() {
            Zl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rZ, mh4, 0, 0));
            Zkl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rZk, mh4, 0, 0));
            ql = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, mh4, 0, 0));
        };
        float gVs[HT][KR][VEC], gVr[HT][KR][VEC];
        float pdv[HT];
        if (hg == 0 && more && p.hub_chunk && nend - nbeg > p.hub_chunk) nend = nbeg;      // (wave-uniform) a hub: its pieces walk the row
        const bool nfetch = hg == 0 && more && nbeg < nend;              // wave-uniform
        int nsrcv = 0, neidv = 0;
        // the node's rows of g_V: ONE descriptor over the group's rows, the row / part of a request in soffset, the lane mask in voffset
        const float* gvn = p.gV + (static_cast<int64_t>(node) * H + h0) * W;
        const auto rgv = K2_RSRC(gvn, min(HT, H - h0) * W * 4);
        if (h0 + HT <= H) {
            // Every head of the group exists (the common case): no guard per head, all 3 HT rows requested together.  A guard around
            // one head's loads is a basic block of its own, and at the join behind it the compiler drains the load counter: eight
            // dependent round trips per node, 20 k of a wave's 45 k cycles (s_memtime stamps at cfg 2: 5.8 k until the score vectors
            // are staged, 20.2 k for these rows, 17.1 k in the edge loop, 2.2 k to the end).  Straight-line, the compiler interleaves
            // requests and arithmetic; 85 -> 76.6 us (two batches of four heads: 77.2).  The asm keeps the batch's arithmetic in front
            // of what follows.
            // (Measured on top and not kept: the edge loop without its `break` / guarded refill, so that no vmcnt(0) is left at the top
            // of an edge — 80 us, 2 spills; touching every line of the node's rows up front so that the guarded form hits L2 — 90 us.)
            constexpr int HB = HT >= 8 ? 8 : HT;
#pragma unroll
            for (int hb = 0; hb < HT; hb += HB) {
                float gd[HB][KR][VEC];
#pragma unroll
                for (int t = 0; t < HB; ++t)
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const uint32_t row = static_cast<uint32_t>((hb + t) * W) * 4u;
                        buf_load_f32<VEC>(gd[t][r], rgv, voF[r], row);
                        buf_load_f32<VEC>(gVs[hb + t][r], rgv, voF[r], row + static_cast<uint32_t>(F) * 4u);
                        buf_load_f32<VEC>(gVr[hb + t][r], rgv, voR[r], row + static_cast<uint32_t>(2 * F) * 4u);
                    }
                if (hb == 0) load_norms();
                if (hb == 0 && nfetch) { const int kk = nbeg + min(lane, min(64, nend - nbeg) - 1); nsrcv = p.src[kk]; neidv = p.eid[kk]; }
                // every request of the batch is issued before the first of them is waited for: left to itself the scheduler (it works
                // towards the three-waves-per-SIMD register budget) loads a head's three rows, waits, folds the destination row away, and
                // only then asks for the next head — eight dependent round trips per node (cycle stamps: 21.6 k of a wave's 38.6 k cycles)
                __builtin_amdgcn_sched_barrier(0);
                if (hb == 0) { Zl = hv ? Zl : 1.f; Zkl = hv ? Zkl : 0.f; ql = hv ? ql : 0.f; }
                if (hb == 0 && nfetch) { nidx[0] = nsrcv; nidx[64] = neidv; }
                const float zrl = Zkl / Zl;
#pragma unroll
                for (int t = 0; t < HB; ++t) {
                    const int h = hb + t;
                    const float zr = lane_bcast(zrl, h << SH);
                    float pd = 0.f;
#pragma unroll
                    for (int r = 0; r < KR; ++r)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            pd = fmaf(gd[t][r][v], xi[r][v], pd);
                            gxd[r][v] = fmaf(zr, gd[t][r][v], gxd[r][v]);
                        }
                    pdv[h] = pd;
                }
                asm volatile("" : "+v"(pdv[hb + HB - 1]) : : "memory");
            }
        } else {
        // (Requesting all 3 HT rows at once — branch free, the destination part staged through LDS-DMA — was measured: the
        // extra live registers spill at 3 waves per SIMD and the kernel gets slower, 80 -> 91 us at cfg 2; so was a persistent
        // node-pipelined form with the next node's rows in flight: 107 us at the one wave per SIMD its 346 registers allow.
        // PMC: 1 580 VALU instructions per node keep a SIMD's issue port busy for 42 us of the kernel's 80 as it was in round 5.)
        load_norms();
        if (nfetch) { const int kk = nbeg + min(lane, min(64, nend - nbeg) - 1); nidx[0] = p.src[kk]; nidx[64] = p.eid[kk]; }
        Zl = hv ? Zl : 1.f; Zkl = hv ? Zkl : 0.f; ql = hv ? ql : 0.f;
        const float zrl = Zkl / Zl;
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            float pd = 0.f;
            const bool hok = h0 + h < H;                                  // wave-uniform
            const float zr = lane_bcast(zrl, h << SH);
#pragma unroll
            for (int r = 0; r < KR; ++r) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) { gVs[h][r][v] = 0.f; gVr[h][r][v] = 0.f; }
                if (hok) {                                                // wave-uniform
                    const uint32_t row = static_cast<uint32_t>(h * W) * 4u;
                    float gd[VEC];
                    buf_load_f32<VEC>(gd, rgv, voF[r], row);
                    buf_load_f32<VEC>(gVs[h][r], rgv, voF[r], row + static_cast<uint32_t>(F) * 4u);
                    buf_load_f32<VEC>(gVr[h][r], rgv, voR[r], row + static_cast<uint32_t>(2 * F) * 4u);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        pd = fmaf(gd[v], xi[r][v], pd);
                        gxd[r][v] = fmaf(zr, gd[v], gxd[r][v]);           // direct path: V_dst = x_i Zk/Z
                    }
                }
            }
            pdv[h] = pd;
        }
        }
        const float invl = 1.f / Zl;
        const float tdl = multi_sum<HT>(pdv, lane);
#ifdef RECON_K2_STAMPS
        stamp_[2] = __builtin_readcyclecounter();
#endif
        const float gZl = -ql * invl;                                     // d loss / d Z   (every part of V is ~ 1/Z)
        float sum_gs = 0.f;
        // The walk.  FULL = every head of the group exists (wave-uniform, the common case): the body of an edge is then straight-line code
        // apart from the ring refill.  Round 6: (1) the rows of u come out of LDS branch-free (clamped columns) — under `if (lane < F)` /
        // `if (h0 + h < H)` every one of the 16 ds_read_b128 of an edge sat in a block of its own with `s_waitcnt lgkmcnt(0)` right
        // behind it: sixteen exposed LDS round trips per edge at three waves per SIMD; (2) the dots and the row updates are packed
        // (v_pk_fma_f32: two fp32 lanes per instruction, the rate the 157 TF/s vector peak is quoted at); (3) the 8-value reduction is
        // VALU only (multi_sum: v_permlane32/16_swap instead of ds_bpermute); (4) the score path's share of the row bound for
        // x[src_e], g_sigma[e][h] u_src[h], left this kernel: summed over the edges of a source it is Gs_src[j][h] u_src[h] — once per
        // node in k_gat_atp_src instead of once per edge here (8 LDS reads and 32 FMAs per edge); (5) rows through buffer descriptors.
        auto walk = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        // one edge: scores, gradient of its score, its two output rows.  j: position in the chunk, e: its edge id, xs / re: its gathered rows
        auto edge = [&](int j, int e, const float (&xs)[KR][VEC], const float (&re)[KR][VEC], float sg, float kf) {
                float part[HT];
#pragma unroll
                for (int h = 0; h < HT; ++h) {
                    f32x2 acc = {0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < KR; ++r)
#pragma unroll
                        for (int v = 0; v < VEC; v += 2) {
                            acc = pk_fma(f32x2{gVs[h][r][v], gVs[h][r][v + 1]}, f32x2{xs[r][v], xs[r][v + 1]}, acc);
                            acc = pk_fma(f32x2{gVr[h][r][v], gVr[h][r][v + 1]}, f32x2{re[r][v], re[r][v + 1]}, acc);
                        }
                    part[h] = acc.x + acc.y;
                }
                const float tl = multi_sum<HT>(part, lane);
                const float w = hv ? __expf(-(sg > 0.f ? sg : p.alpha * sg)) : 0.f;      // v_exp_f32: 2 ulp, enough for a gradient factor
                const float gw = fmaf(kf * (tl + tdl), invl, gZl);
                const float gs = hv ? -gw * w * (sg > 0.f ? 1.f : p.alpha) : 0.f;
                const float al_ = kf * w * invl;
                sum_gs += gs;
                if (hv && writer) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, gs), rgs, mh4, static_cast<uint32_t>(j) * Hb, 0);
                f32x2 gxs[KR][VEC / 2], gr[KR][VEC / 2];
#pragma unroll
                for (int r = 0; r < KR; ++r)
#pragma unroll
                    for (int v = 0; v < VEC / 2; ++v) { gxs[r][v] = f32x2{0.f, 0.f}; gr[r][v] = f32x2{0.f, 0.f}; }
                // the rows of u are the same for every edge: left alone, the compiler hoists all HT KR reads out of the walk and then
                // spills them.  An offset it cannot see through keeps them here.
                // (a head's two factors are wave-uniform — readlane — and reach v_pk_fma_f32 as scalar operands broadcast by op_sel)
                int u_off = 0;
                asm volatile("" : "+v"(u_off));
#pragma unroll
                for (int h = 0; h < HT; ++h) {
                    if (FULL || h0 + h < H) {                         // (not FULL: wave-uniform)
                        const f32x2 ab = {lane_bcast(al_, h << SH), lane_bcast(gs, h << SH)};
                        const f32x2 ah2 = __builtin_shufflevector(ab, ab, 0, 0), bh2 = __builtin_shufflevector(ab, ab, 1, 1);
                        const float* uh = U + (h0 + h) * UW + F + u_off;
#pragma unroll
                        for (int r = 0; r < KR; ++r) {
                            // (the score path's share of the row bound for x[src_e], g_sigma[e][h] u_src[h], is NOT added here: summed
                            // over the edges of a source it is Gs_src[j][h] u_src[h] — once per node in k_gat_atp_src instead of once
                            // per edge here: 8 LDS reads and 32 FMAs less per edge, and the registers they took)
                            float ur[VEC];
                            load_vec<VEC>(ur, uh + uR[r]);
#pragma unroll
                            for (int v = 0; v < VEC; v += 2) {
                                gxs[r][v / 2] = pk_fma(ah2, f32x2{gVs[h][r][v], gVs[h][r][v + 1]}, gxs[r][v / 2]);
                                gr[r][v / 2] = pk_fma(ah2, f32x2{gVr[h][r][v], gVr[h][r][v + 1]}, pk_fma(bh2, f32x2{ur[v], ur[v + 1]}, gr[r][v / 2]));
                            }
                        }
                    }
                }
                // g_edge_embed rows go by slot (a table read through an index: the caller sums them per table row) or by edge id
                const bool ge_edge = p.g_ee && !p.gee_by_slot;       // wave-uniform
                constexpr bool ge16 = GE16;                           // bfloat16 rows of g_edge_embed (2 R bytes each): rows by edge id, one head group per wave (host)
                const auto rGee = ge16 ? K2_RSRC(reinterpret_cast<uint16_t*>(p.g_ee) + static_cast<int64_t>(e) * R, Rb / 2)
                                       : (ge_edge ? K2_RSRC(p.g_ee + static_cast<int64_t>(e) * R, Rb) : rGe);
                const uint32_t sox = static_cast<uint32_t>(j) * Fb, soe = ge_edge ? 0u : static_cast<uint32_t>(j) * Rb;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    float ox[VEC], orr[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; v += 2) { ox[v] = gxs[r][v / 2].x; ox[v + 1] = gxs[r][v / 2].y; orr[v] = gr[r][v / 2].x; orr[v + 1] = gr[r][v / 2].y; }
                    if (hg > 0) {                                    // wave-uniform: a later head group adds to the rows the first one wrote
                        float o1[VEC], o2[VEC];
                        buf_load_f32<VEC>(o1, rGx, voF[r], sox);
                        buf_load_f32<VEC>(o2, rGee, voR[r], soe);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) { ox[v] += o1[v]; orr[v] += o2[v]; }
                    }
                    buf_store_f32<VEC>(ox, rGx, voF[r], sox);       // lanes past F / R (and every lane without g_edge_embed): dropped by the range check
                    if constexpr (GE16) {
                        {                                            // (one head group per wave here: nothing was added above) round to nearest even, 2 bytes per value
                            uint32_t w2[VEC / 2];
#pragma unroll
                            for (int v = 0; v < VEC; v += 2) {
                                const __bf16 b0 = static_cast<__bf16>(orr[v]), b1 = static_cast<__bf16>(orr[v + 1]);
                                w2[v / 2] = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, b0)) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, b1)) << 16);
                            }
                            const uint32_t vo16 = voR[r] == kK2Oob ? kK2Oob : voR[r] / 2;
                            if constexpr (VEC == 4) __builtin_amdgcn_raw_buffer_store_b64(k2_u32x2{w2[0], w2[1]}, rGee, vo16, 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b32(w2[0], rGee, vo16, 0, 0);
                            continue;
                        }
                    }
                    buf_store_f32<VEC>(orr, rGee, voR[r], soe);
                }
        };
        while (c0 < end) {
            if constexpr (LR) {
                // the chunk's first RING edges were requested in front of everything this wave has waited for since (first chunk: the node's
                // g_V rows); later chunks: their fills have to land
                // (the builtin, not inline asm: the compiler then KNOWS nothing of its own is in flight at the loop's entry — otherwise a load it
                // issued on the way in (the chunk's first score words) may still be pending in registers the loop body overwrites, and the
                // wait for it lands inside the loop header: a vmcnt(0) per group of edges)
                __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), nothing else
                asm volatile("" ::: "memory");
                // whole groups of RING edges run without a branch (a run-time condition around an edge's requests makes the compiler count
                // the FEWEST requests any path issues, and its waits for the score words then reach into the copies in flight: with
                // `if (j < cn)` around every edge it waited for all but seven requests where seventeen were its own); the chunk's last
                // 1 .. RING - 1 edges follow, guarded
                auto lr_edge = [&](int j, auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const int e = __builtin_amdgcn_readlane(eidv, j);
                    float xs[KR][VEC], re[KR][VEC];
                    dma_wait<5 + 7 * (RING - 1)>();                      // this slot's copies (issued RING edges ago: five requests and RING - 1 whole edges behind them) have landed
                    asm volatile("" ::: "memory");
                    const unsigned char* slot = ringb + u * SLOTB;       // (j % RING == u: chunks start at slot 0)
                    if constexpr (B16) {
                        const uint2 tx = reinterpret_cast<const uint2*>(slot)[lane], te = reinterpret_cast<const uint2*>(slot + ROWB)[lane];
                        xs[0][0] = __builtin_bit_cast(float, tx.x << 16); xs[0][1] = __builtin_bit_cast(float, tx.x & 0xffff0000u);
                        xs[0][2] = __builtin_bit_cast(float, tx.y << 16); xs[0][3] = __builtin_bit_cast(float, tx.y & 0xffff0000u);
                        re[0][0] = __builtin_bit_cast(float, te.x << 16); re[0][1] = __builtin_bit_cast(float, te.x & 0xffff0000u);
                        re[0][2] = __builtin_bit_cast(float, te.y << 16); re[0][3] = __builtin_bit_cast(float, te.y & 0xffff0000u);
                    } else {
                        const float4 tx = reinterpret_cast<const float4*>(slot)[lane], te = reinterpret_cast<const float4*>(slot + ROWB)[lane];
                        xs[0][0] = tx.x; xs[0][1] = tx.y; xs[0][2] = tx.z; xs[0][3] = tx.w;
                        re[0][0] = te.x; re[0][1] = te.y; re[0][2] = te.z; re[0][3] = te.w;
                    }
                    // the slot is free once these reads have executed: the next copies into it are issued behind them
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[0][0]), "+v"(re[0][0]) : : "memory");
                    const float sg = sg_r[u], kf = p.keep ? kf_r[u] : 1.f;
                    // in this order (the count and order of an edge's requests are part of the waits): the copies of the edge RING ahead, the
                    // arithmetic, the two score words of that edge — BEHIND the last use of this edge's (one register each for the ring's
                    // whole life: requested earlier they were live beside this edge's words and had to be copied into their names at the
                    // loop's back edge, behind a vmcnt(0)) — and behind the copies (in-order completion: a word the compiler waits for takes
                    // every older request with it)
                    fill_rows(j + RING);                                 // two copies; a position past the chunk copies nothing
                    edge(j, e, xs, re, sg, kf);
                    asm volatile("" ::: "memory");
                    fetch_scores(u, min(j + RING, cn - 1));
                };
                int j0 = 0;
                for (; j0 + RING <= cn; j0 += RING) {
                    lr_edge(j0, std::integral_constant<int, 0>{}); lr_edge(j0 + 1, std::integral_constant<int, 1>{});
                    lr_edge(j0 + 2, std::integral_constant<int, 2>{}); lr_edge(j0 + 3, std::integral_constant<int, 3>{});
                }
                if (j0 < cn) lr_edge(j0, std::integral_constant<int, 0>{});
                if (j0 + 1 < cn) lr_edge(j0 + 1, std::integral_constant<int, 1>{});
                if (j0 + 2 < cn) lr_edge(j0 + 2, std::integral_constant<int, 2>{});
            } else {
            for (int j0 = 0; j0 < cn; j0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int j = j0 + u;
                    if (j >= cn) break;                                  // wave-uniform
                    const int e = __builtin_amdgcn_readlane(eidv, j);
                    float xs[KR][VEC], re[KR][VEC];
#pragma unroll
                    for (int r = 0; r < KR; ++r)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) { xs[r][v] = xs_r[u][r][v]; re[r][v] = re_r[u][r][v]; }
                    const float sg = sg_r[u], kf = p.keep ? kf_r[u] : 1.f;
                    if (j + PF < cn) fetch_edge(u, j + PF);              // wave-uniform; refills the slot just consumed
                    edge(j, e, xs, re, sg, kf);
                }
            }
            }
            c0 += 64;
            if (c0 < end) {                                              // next chunk of a long row: new index vectors, refill the ring
                cn = min(64, end - c0);
                const int kk = c0 + min(lane, cn - 1);
                srcv = p.src[kk]; eidv = p.eid[kk];
                rsg = K2_RSRC(p.sigma + static_cast<int64_t>(c0) * H, 64 * Hb); rkf = K2_RSRC(keepp + static_cast<int64_t>(c0) * H, 64 * Hb);
                rgs = K2_RSRC(p.gsigma + static_cast<int64_t>(c0) * H, 64 * Hb);
                rGx = K2_RSRC(p.Gxs + static_cast<int64_t>(c0) * F, 64 * Fb);
                rGe = K2_RSRC((p.g_ee && p.gee_by_slot) ? p.g_ee + static_cast<int64_t>(c0) * R : p.Gxs, (p.g_ee && p.gee_by_slot) ? 64 * Rb : 0u);
                if constexpr (LR) {
                    // (the previous chunk's last iterations copied nothing into the slots: they are free)
#pragma unroll
                    for (int u = 0; u < RING; ++u) fill_rows(u);
#pragma unroll
                    for (int u = 0; u < PF; ++u) fetch_scores(u, min(u, cn - 1));
                } else {
#pragma unroll
                    for (int u = 0; u < PF; ++u) fetch_edge(u, min(u, cn - 1));
                }
            }
        }
        };
        if (h0 + HT <= H) walk(std::true_type{}); else walk(std::false_type{});
#ifdef RECON_K2_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (stores included)
        stamp_[3] = __builtin_readcyclecounter();
#endif
        if (hv && writer) {
            if (is_piece) p.hubG[static_cast<int64_t>(widx) * H + myh] = sum_gs;
            else p.Gs_dst[static_cast<int64_t>(node) * 2 * H + myh] = sum_gs;
        }
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            if (h0 + h < H) {
                const float sh = lane_bcast(sum_gs, h << SH);
                const float* uh = U + (h0 + h) * UW;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    float ud[VEC];
                    load_vec<VEC>(ud, uh + uF[r]);                      // lanes past F: column 0, never stored
#pragma unroll
                    for (int v = 0; v < VEC; ++v) gxd[r][v] = fmaf(sh, ud[v], gxd[r][v]);
                }
            }
        }
    }
    if (is_piece) return;                                                // k_gat_atp_hub_bwd adds the pieces' share to the node's row (one piece per wave)
#ifdef RECON_K2_STAMPS
    stamp_[4] = __builtin_readcyclecounter();
    stamp_[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);             // HW_ID: wave, SIMD, CU, SE, XCC
    if (lane == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(p.gxd + static_cast<int64_t>(node) * F);
        for (int i = 0; i < 6; ++i) o[i] = stamp_[i];
        o[6] = static_cast<uint64_t>(end - beg);
    }
    if (!more) return;
    node = nnode; beg = nbeg; end = nend; srcv0 = nidx[0]; eidv0 = nidx[64];       // (zero-degree nodes: never read)
    stamp_[0] = __builtin_readcyclecounter();
    continue;
#endif
    const auto rgx = K2_RSRC(p.gxd + static_cast<int64_t>(node) * F, F * 4);
#pragma unroll
    for (int r = 0; r < KR; ++r) buf_store_f32<VEC>(gxd[r], rgx, voF[r]);
    if (!more) return;
    node = nnode; beg = nbeg; end = nend; srcv0 = nidx[0]; eidv0 = nidx[64];       // (zero-degree nodes: never read)
    }
}

// The second half of a hub's backward: wave = one hub.  Adds the pieces' sums of g_sigma in table order to Gs_dst (the node's own
// wave wrote zeros) and their share sum_h S_h u_dst[h] to the node's g_x row.  H <= 64 (one head per lane).
template <int VEC>
__global__ void __launch_bounds__(kBlock) k_gat_atp_hub_bwd(const AtpBwdK p, const int32_t* __restrict__ hub_node, const int32_t* __restrict__ hub_ptr,
                                                            int32_t n_hub) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * (kBlock / 64) + wave;
    const int F = p.F, R = p.R, H = p.H, W = 2 * F + R;
    if (t >= n_hub) return;
    const int node = hub_node[t], p0 = hub_ptr[t], p1 = hub_ptr[t + 1];
    float tot = 0.f;
    if (lane < H) {
        float tt[1] = {0.f};
        sum_pieces<1>(tt, p.hubG + lane, H, p0, p1);
        tot = tt[0];
        p.Gs_dst[static_cast<int64_t>(node) * 2 * H + lane] += tot;
    }
    for (int c = lane * VEC; c < F; c += 64 * VEC) {
        float o[VEC];
        load_vec<VEC>(o, p.gxd + static_cast<int64_t>(node) * F + c);
        for (int h = 0; h < H; ++h) {
            const float sh = lane_bcast(tot, h);
            float ud[VEC];
            load_vec<VEC>(ud, p.u + static_cast<int64_t>(h) * W + c);
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = fmaf(sh, ud[v], o[v]);
        }
        store_vec<VEC>(p.gxd + static_cast<int64_t>(node) * F + c, o);
    }
}

// CSC walk: g_x[j] = gxd[j] + sum_{e: src_e = j} Gxs[slot];  Gs_src[j][h] = sum gsigma[slot][h]
struct AtpSrcK {
    const int32_t* rowptr_src; const int32_t* slot_by_src; const float* Gxs; const float* gxd; const float* gsigma;
    float* g_x; float* Gs_src;
    const float* u; int32_t W;  // score vectors [H][W]: the row of node j also takes sum_h Gs_src[j][h] u_src[h] (k_gat_atp_bwd leaves that term to this pass)
    int32_t N, F, H;
    // hub rows of the CSC view (recon_graph): the first n_piece waves sum one piece each into hubP [n_piece][F + H] (g_x part | sums of
    // g_sigma); the wave of a node with more than hub_chunk positions leaves its row to k_gat_atp_hub_src.  hub_chunk = 0: off.
    int32_t hub_chunk, n_piece;
    const int4* piece;
    float* hubP;
    const int32_t* node_row;    // row compaction (recon_graph.node_row): gxd is indexed by ROW; a node without a row (-1) has no destination part.  NULL: gxd[node]
};
template <int VEC, int KR>
__global__ void __launch_bounds__(kBlock) k_gat_atp_src(const AtpSrcK p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npb = (p.n_piece + kBlock / 64 - 1) / (kBlock / 64);       // piece blocks first, dealt round the XCDs (see k_gat_atp_fwd)
    const int widx = blockIdx.x * (kBlock / 64) + wave;
    const bool is_piece = static_cast<int>(blockIdx.x) < npb;
    int node = xcd_block(blockIdx.x - npb, gridDim.x - npb) * (kBlock / 64) + wave;
    int beg, end;
    if (is_piece) {
        if (widx >= p.n_piece) return;
        const int4 pc = p.piece[widx]; node = pc.x; beg = pc.y; end = pc.z;
    } else {
        if (node >= p.N) return;
        beg = p.rowptr_src[node]; end = p.rowptr_src[node + 1];
        if (p.hub_chunk && end - beg > p.hub_chunk) return;
    }
    const int F = p.F, H = p.H;
    const int drow = (p.node_row && !is_piece) ? p.node_row[node] : node;   // (wave-uniform) the node's row of gxd, -1: it has none
    float acc[KR][VEC];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[r][v] = 0.f;
        const int c = (r * 64 + lane) * VEC;
        if (!is_piece && c < F && p.g_x && drow >= 0) load_vec<VEC>(acc[r], p.gxd + static_cast<int64_t>(drow) * F + c);
    }
    float gs = 0.f;
    // The CSC position -> slot indices of up to 64 positions come with one coalesced load and are handed out with v_readlane; four
    // edges per iteration, every load issued before the first use and branch free (positions past the row re-read its last one
    // and are skipped at the accumulate; lanes past F / H read column 0 / head 0 and are never stored)
    constexpr int U = 4;
    const int hl = lane < H ? lane : 0;
    for (int c0 = beg; c0 < end; c0 += 64) {
        const int cn = min(64, end - c0);
        const int slotv = p.slot_by_src[c0 + min(lane, cn - 1)];
        for (int j0 = 0; j0 < cn; j0 += U) {
            int slot[U];
#pragma unroll
            for (int u = 0; u < U; ++u) slot[u] = __builtin_amdgcn_readlane(slotv, min(j0 + u, cn - 1));
            float g4[U], t[U][KR][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g4[u] = p.gsigma[static_cast<int64_t>(slot[u]) * H + hl];
                if (p.g_x) {                                                // uniform
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const int c = (r * 64 + lane) * VEC;
                        load_vec<VEC>(t[u][r], p.Gxs + static_cast<int64_t>(slot[u]) * F + (c < F ? c : 0));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j0 + u < cn) {                                           // uniform; fixed summation order
                    gs += g4[u];
                    if (p.g_x) {
#pragma unroll
                        for (int r = 0; r < KR; ++r)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[r][v] += t[u][r][v];
                    }
                }
            }
        }
    }
    float* gs_out = is_piece ? p.hubP + static_cast<int64_t>(widx) * (F + H) + F : p.Gs_src + static_cast<int64_t>(node) * 2 * H + H;
    float* row_out = is_piece ? p.hubP + static_cast<int64_t>(widx) * (F + H) : p.g_x + static_cast<int64_t>(node) * F;
    if (lane < H) gs_out[lane] = gs;
    if (!is_piece && p.g_x) {                                            // uniform.  + sum_h Gs_src[node][h] u_src[h][:], heads in order (H <= 64)
        constexpr int HBK = KR <= 2 ? 8 : 2;                             // rows of u requested together (L2 hits: one round trip per batch, not per head)
        for (int hb = 0; hb < H; hb += HBK) {
            float us[HBK][KR][VEC];
#pragma unroll
            for (int t = 0; t < HBK; ++t) {
                const float* uh = p.u + static_cast<int64_t>(min(hb + t, H - 1)) * p.W + F;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const int c = (r * 64 + lane) * VEC;
                    load_vec<VEC>(us[t][r], uh + (c < F ? c : 0));
                }
            }
#pragma unroll
            for (int t = 0; t < HBK; ++t) {
                const float sh = hb + t < H ? lane_bcast(gs, min(hb + t, 63)) : 0.f;      // heads in order; past H: + 0
#pragma unroll
                for (int r = 0; r < KR; ++r)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[r][v] = fmaf(sh, us[t][r][v], acc[r][v]);
            }
        }
    }
    if (p.g_x) {
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int c = (r * 64 + lane) * VEC;
            if (c < F) {
                if (is_piece && ((F + H) % VEC)) {                            // uniform: a piece's row starts at a multiple of F + H floats
#pragma unroll
                    for (int v = 0; v < VEC; ++v) row_out[c + v] = acc[r][v];
                } else store_vec<VEC>(row_out + c, acc[r]);
            }
        }
    }
}

// The second half of a source hub: wave = one (hub, 64-column stripe); g_x row = its direct part + the pieces' rows in table order,
// Gs_src likewise (by the wave of stripe 0).
__global__ void __launch_bounds__(kBlock) k_gat_atp_hub_src(const AtpSrcK p, const int32_t* __restrict__ hub_node, const int32_t* __restrict__ hub_ptr,
                                                            int32_t n_hub) {
    const int lane = threadIdx.x & 63;
    const int F = p.F, H = p.H;
    const int stripes = (F + 63) / 64;
    const int idx = blockIdx.x * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (idx >= n_hub * stripes) return;
    const int t = idx / stripes, c = (idx - t * stripes) * 64 + lane;
    const int node = hub_node[t], p0 = hub_ptr[t], p1 = hub_ptr[t + 1];
    // every wave of the hub needs all H sums (its stripe of the row takes sum_h Gs_src[h] u_src[h]): lane h < H adds them up, in table order
    float gsl[1] = {0.f};
    if (lane < H) sum_pieces<1>(gsl, p.hubP + F + lane, F + H, p0, p1);
    if (c < H) p.Gs_src[static_cast<int64_t>(node) * 2 * H + H + c] = gsl[0];      // stripe 0: c == lane
    if (!p.g_x) return;
    const int drow = p.node_row ? p.node_row[node] : node;
    float a[1] = {(c < F && drow >= 0) ? p.gxd[static_cast<int64_t>(drow) * F + c] : 0.f};
    if (c < F) sum_pieces<1>(a, p.hubP + c, F + H, p0, p1);
    for (int h = 0; h < H; ++h) a[0] = fmaf(lane_bcast(gsl[0], h), p.u[static_cast<int64_t>(h) * p.W + F + (c < F ? c : 0)], a[0]);
    if (c < F) p.g_x[static_cast<int64_t>(node) * F + c] = a[0];
}


// Tall-skinny transposed product with fixed-order reduction (replaces 128x128-tile GEMMs whose M is 8..32):
//   out[j][c] = sum_r G[r*ldg + j] * X[row(r)*K + c],   j < NJ (<= 16), c < K,  row(r) = gather ? gather[r] : r
// pass 1: block b sums its slice of rows into partial[b][j][c]; pass 2 adds the slices in order.
// Up to two independent products per launch (blocks [0, j0.nb) run job 0, the rest job 1), like k_row_dots.
struct SkinnyJob { const float* G; const float* X; const int32_t* gather; float* partial; int32_t ldg, nj, rows, K, rpb, nb; };
template <int NJ, bool B16 = false>
__global__ void __launch_bounds__(256) k_skinny_tn_partial(const SkinnyJob j0, const SkinnyJob j1, const SkinnyJob j2) {
    // 4 waves split the block's rows; lane l owns columns 4l..4l+3 of a 256-column stripe; fixed-order LDS combine
    __shared__ float red[3][NJ][256];
    const int which = static_cast<int>(blockIdx.x) < j0.nb ? 0 : (static_cast<int>(blockIdx.x) < j0.nb + j1.nb ? 1 : 2);
    const SkinnyJob& jb = which == 0 ? j0 : (which == 1 ? j1 : j2);
    const float* __restrict__ G = jb.G;
    const float* __restrict__ X = jb.X;
    const int32_t* __restrict__ gather = jb.gather;
    float* __restrict__ partial = jb.partial;
    const int ldg = jb.ldg, nj = jb.nj, rows = jb.rows, K = jb.K, rows_per_block = jb.rpb;
    const int bid = blockIdx.x - (which == 0 ? 0 : (which == 1 ? j0.nb : j0.nb + j1.nb));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = bid * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    for (int c0 = 0; c0 < K; c0 += 256) {
        const int c = c0 + 4 * lane;
        float acc[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
        // gathered row ids are fetched one iteration ahead (lane q holds the id of row rb+q), so the row loads do not
        // wait behind an index load
        int gi = 0;
        if (gather && lane < 4 && r0 + wave * 4 + lane < r1) gi = gather[r0 + wave * 4 + lane];
        for (int rb = r0 + wave * 4; rb < r1; rb += 16) {
            int gnext = 0;
            if (gather && lane < 4 && rb + 16 + lane < r1) gnext = gather[rb + 16 + lane];
            float xv[4][4];
            // the 4 x NJ coefficients of this iteration arrive with ONE load (lane q*NJ + j holds G[rb+q][j]) and are
            // broadcast per use; one scalar-address load per coefficient made the kernel issue bound
            float gl = 0.f;
            {
                const int q = lane / NJ, j = lane % NJ;
                if (q < 4 && rb + q < r1 && j < nj) gl = G[static_cast<int64_t>(rb + q) * ldg + j];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rb + q;
                xv[q][0] = xv[q][1] = xv[q][2] = xv[q][3] = 0.f;
                if (r < r1) {
                    const int64_t row = gather ? __builtin_amdgcn_readlane(gi, q) : r;
                    if constexpr (B16) {                                  // rows of bfloat16 (K % 8 == 0: a quad is inside the row or past it)
                        if (c + 3 < K) load_in<4, true>(xv[q], X, row * K + c);
                    } else {
                        if (c + 3 < K) { const float4 t = *reinterpret_cast<const float4*>(X + row * K + c); xv[q][0] = t.x; xv[q][1] = t.y; xv[q][2] = t.z; xv[q][3] = t.w; }
                        else { for (int v = 0; v < 4; ++v) if (c + v < K) xv[q][v] = X[row * K + c + v]; }
                    }
                }
            }
            gi = gnext;
            float gv[4][NJ];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < NJ; ++j) gv[q][j] = (q * NJ + j < 64) ? lane_bcast(gl, (q * NJ + j) & 63) : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[j][v] = fmaf(gv[q][j], xv[q][v], acc[j][v]);
        }
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[wave - 1][j][4 * lane + v] = acc[j][v];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (j < nj)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (c + v < K)
                            partial[(static_cast<int64_t>(bid) * nj + j) * K + c + v] =
                                ((acc[j][v] + red[0][j][4 * lane + v]) + red[1][j][4 * lane + v]) + red[2][j][4 * lane + v];
        }
    }
}
// (A matrix-core form of these partial products — v_mfma_f32_16x16x4_f32 with the contraction over the rows, both H-column halves
// of the node-side product in one pass over x — was measured at cfg 2: 17.7 - 18.9 us against 16.7 for this kernel.  Like the
// score dots it is a few tens of MB behind a ~5 us kernel turn-around: the arithmetic is not what it waits for.)
// out[(j % P)*S1 + (j / P)*S2 + c] = sum_b partial[b][j][c]   (16 elements x 64 slice groups per block, fixed order)
struct SkinnyRedJob { const float* partial; float* out; int64_t S1, S2; int32_t nb, nj, K, P, nblocks; };
__global__ void __launch_bounds__(1024) k_skinny_reduce(const SkinnyRedJob j0, const SkinnyRedJob j1, const SkinnyRedJob j2) {
    __shared__ float red[64][17];
    const int which = static_cast<int>(blockIdx.x) < j0.nblocks ? 0 : (static_cast<int>(blockIdx.x) < j0.nblocks + j1.nblocks ? 1 : 2);
    const SkinnyRedJob& jb = which == 0 ? j0 : (which == 1 ? j1 : j2);
    const float* __restrict__ partial = jb.partial;
    float* __restrict__ out = jb.out;
    const int nb = jb.nb, nj = jb.nj, K = jb.K, P = jb.P;
    const int64_t S1 = jb.S1, S2 = jb.S2;
    const int bid = blockIdx.x - (which == 0 ? 0 : (which == 1 ? j0.nblocks : j0.nblocks + j1.nblocks));
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int idx = bid * 16 + e;
    const int tot = nj * K;
    const int per = (nb + 63) / 64;
    const int b0 = grp * per, b1 = min(nb, (grp + 1) * per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < tot) {
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            s0 += partial[static_cast<int64_t>(b) * tot + idx];
            s1 += partial[static_cast<int64_t>(b + 1) * tot + idx];
            s2 += partial[static_cast<int64_t>(b + 2) * tot + idx];
            s3 += partial[static_cast<int64_t>(b + 3) * tot + idx];
        }
        for (; b < b1; ++b) s0 += partial[static_cast<int64_t>(b) * tot + idx];
    }
    red[grp][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && idx < tot) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 64; ++g) t += red[g][e];
        const int j = idx / K, c = idx % K;
        out[(j % P) * S1 + (j / P) * S2 + c] = t;
    }
}

struct AtpShape { int vec, kr, ht; };
bool atp_shape(int F, int R, int H, AtpShape* s) {
    const int mx = F > R ? F : R;
    int vec;
    if (F % 4 == 0 && R % 4 == 0) vec = 4;
    else if (F % 2 == 0 && R % 2 == 0) vec = 2;
    else return false;
    int kr = (mx + 64 * vec - 1) / (64 * vec);
    if (kr > 2 && kr <= 4) kr = 4; else if (kr > 4 && kr <= 8) kr = 8; else if (kr > 8) return false;
    int ht = 1;
    while (ht < H && ht < 8) ht <<= 1;
    while (ht > 1 && 2 * ht * kr * vec > 96) ht >>= 1;           // accumulator budget (VGPRs per lane)
    if (2 * ht * kr * vec > 96) return false;
    s->vec = vec; s->kr = kr; s->ht = ht;
    return true;
}
bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

#define ATP_DISPATCH(S, CALL)                                                                              \
    do {                                                                                                   \
        const int key_ = (S).vec * 1000 + (S).kr * 10 + ((S).ht == 8 ? 3 : (S).ht == 4 ? 2 : (S).ht == 2 ? 1 : 0); \
        switch (key_) {                                                                                    \
            case 4010: CALL(4, 1, 1); break; case 4011: CALL(4, 1, 2); break; case 4012: CALL(4, 1, 4); break; case 4013: CALL(4, 1, 8); break; \
            case 4020: CALL(4, 2, 1); break; case 4021: CALL(4, 2, 2); break; case 4022: CALL(4, 2, 4); break;   \
            case 4040: CALL(4, 4, 1); break; case 4041: CALL(4, 4, 2); break; case 4080: CALL(4, 8, 1); break;   \
            case 2010: CALL(2, 1, 1); break; case 2011: CALL(2, 1, 2); break; case 2012: CALL(2, 1, 4); break; case 2013: CALL(2, 1, 8); break; \
            case 2020: CALL(2, 2, 1); break; case 2021: CALL(2, 2, 2); break; case 2022: CALL(2, 2, 4); break; case 2023: CALL(2, 2, 8); break; \
            case 2040: CALL(2, 4, 1); break; case 2041: CALL(2, 4, 2); break; case 2042: CALL(2, 4, 4); break;   \
            case 2080: CALL(2, 8, 1); break; case 2081: CALL(2, 8, 2); break;                                   \
            default: return RECON_ERR_UNSUPPORTED;                                                         \
        }                                                                                                  \
    } while (0)

int check_atp(const recon_graph* g, const recon_gat_atp_args* a) {
    if (!g || !a) return RECON_ERR_INVALID;
    if (a->N != g->N || a->E != g->E) return RECON_ERR_INVALID;
    if (a->N < 0 || a->E < 0 || a->F <= 0 || a->R <= 0 || a->D <= 0 || a->H <= 0) return RECON_ERR_INVALID;
    if (a->ld_out < a->H * a->D) return RECON_ERR_INVALID;
    if (!a->x || !a->a || !a->a_2 || !a->u || !a->c_node || !a->V || !a->out) return RECON_ERR_INVALID;
    if (a->E > 0 && (!a->edge_embed || !a->c_rel)) return RECON_ERR_INVALID;
    if (a->H > 4096) return RECON_ERR_UNSUPPORTED;
    if (a->ee_index && a->E > 0 && a->ee_rows <= 0) return RECON_ERR_INVALID;
    return RECON_OK;
}

}  // namespace
}  // namespace recon

using namespace recon;

// workgroups of k_gat_atp_bwd a device keeps resident: waves per SIMD as its launch bounds allow (1 / 2 / 3: see the kernel), capped by the
// LDS a workgroup's copy of u takes; a multiple of 8
static int64_t k2_resident_blocks(int kr, int ht, size_t lds_bytes) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    const int waves_per_simd = (kr * ht >= 8 && kr >= 8) ? 1 : ((kr >= 4 || (kr * ht >= 8 && kr >= 2)) ? 2 : 3);
    int per_cu = waves_per_simd;                                         // 4 SIMDs x waves / 4 waves per workgroup
    const int by_lds = static_cast<int>((160 * 1024) / (lds_bytes + 256));
    if (by_lds < per_cu) per_cu = by_lds < 1 ? 1 : by_lds;
    return static_cast<int64_t>(cus) * per_cu / 8 * 8;
}

extern "C" int recon_gat_atp_supported(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H) {
    (void)N; (void)E; (void)D;
    AtpShape s;
    if (F <= 0 || R <= 0 || H <= 0 || H > 32) return 0;
    if (!atp_shape(F, R, H, &s)) return 0;
    if (2 * H > 64) return 0;                                       // k_row_dots keeps one result per lane
    // LDS stages (64 KiB per workgroup without opting into more): K2' keeps u [H][2F+R], the score dots their 2H (H) vectors of
    // F (R) columns, padded by 4.  Callers fall back to the project-then-aggregate kernels (gat_heads: 'proj').
    const int64_t W = 2LL * F + R, mx = F > R ? F : R;
    if (4 * H * W > 64 * 1024 || 8LL * H * (mx + 4) > 64 * 1024) return 0;
    return 1;
}

// a_split layout: planes of a  [3][H][D][kp(W)]  then planes of a^T [3][H][W][kp(D)]  (bf16), each part 16-byte aligned
static size_t split_part_bytes(int64_t rows, int32_t K, int32_t H) { return align_up(static_cast<size_t>(3) * H * rows * bx3_kp(K) * 2, 256); }
extern "C" size_t recon_gat_atp_split_bytes(int32_t F, int32_t R, int32_t D, int32_t H) {
    if (F <= 0 || R <= 0 || D <= 0 || H <= 0) return 256;
    const int32_t W = 2 * F + R;
    return split_part_bytes(D, W, H) + split_part_bytes(W, D, H);
}

// f16 x 2 mode (gemm_hx2.hip): V, g_h, a and a^T live as two half terms each and all three large products run on them
// (V as [H][N][2][W], g_h as [H][N][2][D]: head-major, a row's high and low terms side by side — the producers write one
// contiguous piece per (node, head) and a GEMM workgroup's 128 rows of one head are one compact region).
// All or nothing: the planes replace the fp32 tensors, so the decision must be the same in every stage of a step.
//   aux (recon_hx2_aux_bytes(), 256-byte aligned): amax quantities [0] a, [1] x, [2] edge_embed, [3] grad_out (kHx2Slots hashed
//   slots each), then a page of zeros
//   a_split: planes of a [2][H][D][kp(W)], then (at the bf16 x 3 layout's offset) planes of a^T [2][H][W][kp(D)]
static bool atp_hx2_shape(int32_t F, int32_t R, int32_t D, int32_t H) {
    const int64_t W = 2LL * F + R;
    if (F <= 0 || R <= 0 || D <= 0 || H <= 0 || (W & 7) || (D & 7)) return false;
    const int64_t pa = static_cast<int64_t>(H) * D * hx2_kp(static_cast<int32_t>(W)), pt = static_cast<int64_t>(H) * W * hx2_kp(D);
    return 2 * (pa > pt ? pa : pt) < (1LL << 31) && W < (1 << 24);
}
static bool atp_hx2(const recon_gat_atp_args* a) {
    if (a->split_mode != RECON_SPLIT_F16X2 || !a->a_split || !a->aux) return false;
    if (!atp_hx2_shape(a->F, a->R, a->D, a->H) || (a->ld_out & 3)) return false;
    return !((reinterpret_cast<uintptr_t>(a->a_split) & 15) || (reinterpret_cast<uintptr_t>(a->aux) & 255) || (reinterpret_cast<uintptr_t>(a->V) & 15));
}
extern "C" int recon_gat_atp_f16x2_supported(int32_t F, int32_t R, int32_t D, int32_t H) { return atp_hx2_shape(F, R, D, H) ? 1 : 0; }
// bfloat16 x / edge_embed read in place by the forward (recon_gat_atp_args.io_bf16): the f16 x 2 family with 16-byte plane stores and the
// matrix-core score dots — F % 8 == 0, R % 8 == 0, at most 8 heads
extern "C" int recon_gat_atp_bf16_io_supported(int32_t F, int32_t R, int32_t D, int32_t H) {
    if (!atp_hx2_shape(F, R, D, H) || (F % 8) || (R % 8) || 2 * H > 16) return 0;
    recon::AtpShape s;
    recon::atp_shape(F, R, H, &s);
    return s.vec == 4 ? 1 : 0;
}
// The destination part of V is one row for all heads when there is no attention dropout (Zk = Z): K1' writes it for head 0 only and
// the projection / weight-gradient GEMMs read head 0's copy for every head (52 MB less written and, twice, less read at cfg 2).
// F % 8 == 0 so that the shared columns are whole 16-byte groups; the heads' planes lie within 2^31 elements.
static int32_t atp_dst_shared(const recon_gat_atp_args* a) {
    const int64_t W = 2LL * a->F + a->R;
    return (atp_hx2(a) && !a->keep && a->H > 1 && (a->F % 8) == 0 && 2 * W * a->N * a->H < (1LL << 31)) ? a->F : 0;
}
// Row compaction (recon_graph.n_rows): the node-parallel stages run over the destination rows that have edges.  n = rows, rowptr = their
// row pointers, row_node / node_row = the maps between rows and nodes (NULL: rows are nodes, n = N).
struct AtpRows { int32_t n; const int32_t* rowptr; const int32_t* row_node; const int32_t* node_row; };
static AtpRows atp_rows(const recon_graph* g, const recon_gat_atp_args* a) {
    if (g->n_rows > 0 && g->n_rows < a->N && g->row_node && g->rowptr_rows && g->node_row) return AtpRows{g->n_rows, g->rowptr_rows, g->row_node, g->node_row};
    return AtpRows{a->N, g->rowptr_dst, nullptr, nullptr};
}
// hub tables present and complete (recon_graph_hubs_fill)
static bool atp_hubs(const recon_graph* g) {
    return g->hub_chunk > 0 && g->n_hub > 0 && g->n_piece > 0 && g->hub_node && g->hub_ptr && g->piece && g->hub_ws;
}
static uint32_t* atp_q(const recon_gat_atp_args* a, int q) { return static_cast<uint32_t*>(a->aux) + q * kHx2QuantityWords; }
static Hx2Scale atp_scale_a(const recon_gat_atp_args* a) { return Hx2Scale{atp_q(a, 0), nullptr, 1.f}; }
static Hx2Scale atp_scale_v(const recon_gat_atp_args* a) {
    const float km = (a->keep && a->keep_max > 1.f) ? a->keep_max : 1.f;
    return Hx2Scale{atp_q(a, 1), atp_q(a, 2), km};
}
static Hx2Scale atp_scale_g(const recon_gat_atp_args* a) { return Hx2Scale{atp_q(a, 3), nullptr, 1.f}; }

static int atp_fwd_common(const recon_graph* g, const recon_gat_atp_args* a, AtpShape* s) {
    int rc = check_atp(g, a);
    if (rc != RECON_OK) return rc;
    if (!recon_gat_atp_supported(a->N, a->E, a->F, a->R, a->D, a->H)) return RECON_ERR_UNSUPPORTED;
    const bool train = a->Z != nullptr;
    if (train && (!a->Zk || (a->E > 0 && !a->sigma))) return RECON_ERR_INVALID;
    if (a->keep && !train) return RECON_ERR_INVALID;
    atp_shape(a->F, a->R, a->H, s);
    if (a->io_bf16) {                                               // x / edge_embed stored as bfloat16: the f16 x 2 family's forward kernels only
        if (!recon_gat_atp_bf16_io_supported(a->F, a->R, a->D, a->H) || !atp_hx2(a) || a->ee_index) return RECON_ERR_UNSUPPORTED;
        if (!al(a->x, 16) || !al(a->edge_embed, 16) || !al(a->V, 16) || !al(a->u, 16)) return RECON_ERR_UNSUPPORTED;     // rows are read 16 bytes at a time
        return RECON_OK;
    }
    if (!al(a->x, 4 * s->vec) || !al(a->edge_embed, 4 * s->vec) || !al(a->V, 16) || !al(a->u, 16)) return RECON_ERR_UNSUPPORTED;
    return RECON_OK;
}

// stage 1: score vectors u = a_2^T a, then c_node = x.[u_dst | u_src], c_rel = edge_embed[eid].u_rel
extern "C" int recon_gat_atp_scores(const recon_graph* g, const recon_gat_atp_args* a, recon_stream_t stream) {
    AtpShape s;
    int rc = atp_fwd_common(g, a, &s);
    if (rc != RECON_OK) return rc;
    if (a->N == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    const int32_t N = a->N, E = a->E, F = a->F, R = a->R, D = a->D, H = a->H, W = 2 * F + R;
    const bool hx2 = atp_hx2(a);
    const int nblk_a = static_cast<int>(ceil_div64(W, 64)) * H;
    if (hx2 && nblk_a > kHx2BlkMaxWords) return RECON_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_score_vec, dim3(static_cast<unsigned>(ceil_div64(W, 64)), static_cast<unsigned>(H)), dim3(1024), 0, st, a->a,
                       a->a_2, D, W, a->u, hx2 ? atp_q(a, 0) : nullptr);
    const bool dots_x = s.vec == 4 && 2 * H <= 16;                // score dots on the matrix cores (k_row_dots_x)
    Hx2SplitBoth sp{};
    int nsplit = 0;
    if (hx2) {                                                    // half planes of s_a a and s_a a^T
        char* ws = static_cast<char*>(a->a_split);
        if (!hx2_split_both_args(a->a, static_cast<int64_t>(D) * W, D, W, H, ws, ws + split_part_bytes(D, W, H), &sp)) return RECON_ERR_INVALID;
        sp.sc = atp_scale_a(a); sp.blkmax_quantity = atp_q(a, 0); sp.nblk = nblk_a;
        if (dots_x) nsplit = sp.gx * H * 2;                       // rides in the score dots' launch
        else {
            rc = hx2_split_planes_both(a->a, static_cast<int64_t>(D) * W, D, W, H, ws, ws + split_part_bytes(D, W, H), atp_scale_a(a), st,
                                       atp_q(a, 0), nblk_a);
            if (rc != RECON_OK) return rc;
        }
    } else if (a->a_split) {                                      // bf16 term planes of a and a^T for the split-precision GEMMs
        if (reinterpret_cast<uintptr_t>(a->a_split) & 15) return RECON_ERR_INVALID;
        char* ws = static_cast<char*>(a->a_split);
        rc = bx3_split_planes(a->a, W, static_cast<int64_t>(D) * W, false, D, W, H, ws, st);
        if (rc != RECON_OK) return rc;
        rc = bx3_split_planes(a->a, W, static_cast<int64_t>(D) * W, true, W, D, H, ws + split_part_bytes(D, W, H), st);
        if (rc != RECON_OK) return rc;
    }
    {
        RowDotsJob jn, je;
        jn.X = a->x; jn.gather = nullptr; jn.rows = N; jn.K = F; jn.F = F; jn.off = 0; jn.NJ = 2 * H; jn.out = a->c_node;
        jn.amax = hx2 ? atp_q(a, 1) : nullptr; je.amax = hx2 ? atp_q(a, 2) : nullptr;
        // rows per block, measured at cfg 2 (every block first stages its score vectors in LDS): 16 / 32 rows 24.7 us, 32 / 64 rows
        // 18.7 us, 64 / 128 rows 25.4 us
        constexpr int rd_n = 32, rd_e = 64;
        jn.nb = static_cast<int>(ceil_div64(N, rd_n) < 2048 ? ceil_div64(N, rd_n) : 2048);
        // table mode: one score term per table ROW (looked up by K1' through ee_index), not one per edge
        const int32_t erows = a->ee_index ? (E > 0 ? a->ee_rows : 0) : E;
        je.X = a->edge_embed; je.gather = a->ee_index ? nullptr : g->eid; je.rows = erows; je.K = R; je.F = 0; je.off = 2 * F; je.NJ = H; je.out = a->c_rel;
        je.nb = erows > 0 ? static_cast<int>(ceil_div64(erows, rd_e) < 2048 ? ceil_div64(erows, rd_e) : 2048) : 0;
        const size_t lds_n = static_cast<size_t>(2) * H * F * sizeof(float), lds_e = static_cast<size_t>(H) * R * sizeof(float);
        const size_t lds = lds_n > lds_e ? lds_n : lds_e;
        if (lds > 64 * 1024) return RECON_ERR_UNSUPPORTED;
        const dim3 grid(static_cast<unsigned>(jn.nb + je.nb));
        if (dots_x) {                                                 // 16 rows per wave, 64 per block
            jn.nb = static_cast<int>(ceil_div64(N, 64) < 4096 ? ceil_div64(N, 64) : 4096);
            je.nb = erows > 0 ? static_cast<int>(ceil_div64(erows, 64) < 8192 ? ceil_div64(erows, 64) : 8192) : 0;
            const int kpm = row_dots_kp(F) > row_dots_kp(R) ? row_dots_kp(F) : row_dots_kp(R);
            if (a->io_bf16) {
                const int kx = F > R ? F : R;                          // two bf16 term planes of 16 score vectors (row stride: row_dots_b16_stride)
                hipLaunchKernelGGL(k_row_dots_x<true>, dim3(static_cast<unsigned>(jn.nb + je.nb + nsplit)), dim3(kBlock), 2 * 16 * (static_cast<size_t>(kx) * 2 + 16), st,
                                   jn, je, a->u, H, W, sp, nsplit);
            }
            else hipLaunchKernelGGL(k_row_dots_x<false>, dim3(static_cast<unsigned>(jn.nb + je.nb + nsplit)), dim3(kBlock), sizeof(float) * (2 * H) * kpm, st,
                                    jn, je, a->u, H, W, sp, nsplit);
        } else if (s.vec == 4) hipLaunchKernelGGL((k_row_dots<4>), grid, dim3(kBlock), lds, st, jn, je, a->u, H, W);
        else hipLaunchKernelGGL((k_row_dots<2>), grid, dim3(kBlock), lds, st, jn, je, a->u, H, W);
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

// stage 2 (K1'): edge stage, V = normalised aggregated raw features.  HBM bound.
extern "C" int recon_gat_atp_aggregate(const recon_graph* g, const recon_gat_atp_args* a, recon_stream_t stream) {
    AtpShape s;
    int rc = atp_fwd_common(g, a, &s);
    if (rc != RECON_OK) return rc;
    if (a->N == 0) return RECON_OK;
    const bool train = a->Z != nullptr;
    hipStream_t st = as_stream(stream);
    AtpFwdK p;
    const AtpRows rw = atp_rows(g, a);
    p.rowptr = rw.rowptr; p.row_node = rw.row_node; p.src = g->src; p.eid = a->ee_index ? a->ee_index : g->eid;       // the row of edge_embed a slot reads
    p.x = a->x; p.ee = a->edge_embed; p.c_node = a->c_node; p.c_rel = a->c_rel; p.keep = a->keep;
    p.V = a->V; p.sigma = a->sigma; p.Z = a->Z; p.Zk = a->Zk;
    p.nan_flag = nan_flag();
    p.N = rw.n; p.E = a->E; p.F = a->F; p.R = a->R; p.H = a->H; p.alpha = a->alpha;
    p.crel_by_row = a->ee_index ? 1 : 0;
    p.planes = atp_hx2(a) ? ((a->F % 8 == 0 && a->R % 8 == 0) ? 2 : 1) : 0;
    p.dst_shared = atp_dst_shared(a) ? 1 : 0;
    p.vs = p.planes ? atp_scale_v(a) : Hx2Scale{nullptr, nullptr, 1.f};
    const bool hubs = atp_hubs(g);
    if (hubs && static_cast<size_t>(g->hub_ws_floats) < recon_graph_hub_ws_floats(g, a->F, a->R, a->H)) return RECON_ERR_WORKSPACE;
    p.hub_chunk = hubs ? g->hub_chunk : 0; p.n_piece = hubs ? g->n_piece : 0;
    p.piece = reinterpret_cast<const int4*>(g->piece);
    p.hubS = g->hub_ws; p.hubZ = hubs ? g->hub_ws + static_cast<size_t>(g->n_piece) * a->H * (a->F + a->R) : nullptr;
    dim3 grid(static_cast<unsigned>(ceil_div64(rw.n, (kBlock / 64) * kK1NodesPerWave) + ceil_div64(p.n_piece, kBlock / 64)),
              static_cast<unsigned>(ceil_div64(a->H, s.ht)));
#define CALL_FWD(V_, K_, H_)                                                                                   \
    do {                                                                                                       \
        if (p.planes == 2 && V_ == 4 && a->io_bf16) { if (train) hipLaunchKernelGGL((k_gat_atp_fwd<4, K_, H_, true, 2, true>), grid, dim3(kBlock), 0, st, p);  \
                            else hipLaunchKernelGGL((k_gat_atp_fwd<4, K_, H_, false, 2, true>), grid, dim3(kBlock), 0, st, p); }     \
        else if (p.planes == 2 && V_ == 4) { if (train) hipLaunchKernelGGL((k_gat_atp_fwd<V_, K_, H_, true, 2>), grid, dim3(kBlock), 0, st, p);  \
                            else hipLaunchKernelGGL((k_gat_atp_fwd<V_, K_, H_, false, 2>), grid, dim3(kBlock), 0, st, p); }     \
        else if (p.planes) { if (train) hipLaunchKernelGGL((k_gat_atp_fwd<V_, K_, H_, true, 1>), grid, dim3(kBlock), 0, st, p);  \
                            else hipLaunchKernelGGL((k_gat_atp_fwd<V_, K_, H_, false, 1>), grid, dim3(kBlock), 0, st, p); }     \
        else { if (train) hipLaunchKernelGGL((k_gat_atp_fwd<V_, K_, H_, true, 0>), grid, dim3(kBlock), 0, st, p);               \
               else hipLaunchKernelGGL((k_gat_atp_fwd<V_, K_, H_, false, 0>), grid, dim3(kBlock), 0, st, p); }                  \
    } while (0)
    ATP_DISPATCH(s, CALL_FWD);
#undef CALL_FWD
    if (hubs) {
        const dim3 gh(static_cast<unsigned>(ceil_div64(1LL * g->n_hub * a->H, kBlock / 64)));
        if (s.vec == 4 && a->io_bf16) hipLaunchKernelGGL((k_gat_atp_hub_fwd<4, 1, true>), gh, dim3(kBlock), 0, st, p, g->hub_node, g->hub_ptr, g->n_hub);
        else if (s.vec == 4) { if (p.planes) hipLaunchKernelGGL((k_gat_atp_hub_fwd<4, 1>), gh, dim3(kBlock), 0, st, p, g->hub_node, g->hub_ptr, g->n_hub);
                          else hipLaunchKernelGGL((k_gat_atp_hub_fwd<4, 0>), gh, dim3(kBlock), 0, st, p, g->hub_node, g->hub_ptr, g->n_hub); }
        else { if (p.planes) hipLaunchKernelGGL((k_gat_atp_hub_fwd<2, 1>), gh, dim3(kBlock), 0, st, p, g->hub_node, g->hub_ptr, g->n_hub);
               else hipLaunchKernelGGL((k_gat_atp_hub_fwd<2, 0>), gh, dim3(kBlock), 0, st, p, g->hub_node, g->hub_ptr, g->n_hub); }
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

// stage 3: out[:, h*D:(h+1)*D] = act(V[:, h, :] . a[h]^T)      one batched MFMA GEMM over the heads
extern "C" int recon_gat_atp_project(const recon_graph* g, const recon_gat_atp_args* a, recon_stream_t stream) {
    AtpShape s;
    int rc = atp_fwd_common(g, a, &s);
    if (rc != RECON_OK) return rc;
    if (a->N == 0) return RECON_OK;
    const int32_t W = 2 * a->F + a->R;
    const int32_t NR = atp_rows(g, a).n;                                 // rows of V and of `out` (row compaction: the nodes with edges)
    OperandDesc A = plain_operand(a->V, static_cast<int64_t>(a->H) * W);
    OperandDesc B = plain_operand(a->a, W);
    OutputDesc C = plain_output(a->out, a->ld_out);
    GemmBatch bt;
    bt.batch = a->H; bt.a_bs = W; bt.b_bs = static_cast<int64_t>(a->D) * W; bt.c_bs = a->D;
    bt.epilogue = a->concat ? 1 : 0;
    if (atp_hx2(a))
        return gemm_hx2_batched(NR, a->D, W, a->V, W, 2LL * W, 2LL * NR * W, a->a_split, C, bt, atp_scale_v(a), atp_scale_a(a),
                                as_stream(stream), atp_dst_shared(a));  // V terms [H][N][2][W]: plane stride W, row stride 2 W, head stride 2 N W
    if (a->a_split && bx3_supported(A, W, bt)) return gemm_bx3_batched(NR, a->D, W, A, a->a_split, C, bt, as_stream(stream));
    return gemm_f32_batched(NR, a->D, W, A, true, B, true, C, bt, 1, nullptr, as_stream(stream));
}

// out[n][:] = rows[node_row[n]][:], or zeros where the node has no row: the output of a row-compacted layer call back in node order
// (GAT/layers.py:152-158: a node without in-edges gets 0 / 1e-12 = 0 and elu(0) = 0).  One wave per node.
namespace {
__global__ void __launch_bounds__(kBlock) k_rows_expand(const float* __restrict__ rows, int32_t ld_rows, const int32_t* __restrict__ node_row, int32_t N,
                                                        int32_t width, float* __restrict__ out, int32_t ld_out, int32_t vec4) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (n >= N) return;
    const int r = node_row[n];                                           // wave-uniform
    float* o = out + static_cast<int64_t>(n) * ld_out;
    const float* in = rows + static_cast<int64_t>(r < 0 ? 0 : r) * ld_rows;
    if (vec4) {
        for (int c = 4 * lane; c < width; c += 256) {
            const float4 v = r >= 0 ? *reinterpret_cast<const float4*>(in + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(o + c) = v;
        }
    } else {
        for (int c = lane; c < width; c += 64) o[c] = r >= 0 ? in[c] : 0.f;
    }
}
}  // namespace
extern "C" int recon_rows_expand(const float* rows, int32_t ld_rows, const int32_t* node_row, int32_t N, int32_t width, float* out, int32_t ld_out,
                                 recon_stream_t stream) {
    if (N < 0 || width < 0 || ld_rows < width || ld_out < width || (N > 0 && (!node_row || !out)) || (N > 0 && width > 0 && !rows)) return RECON_ERR_INVALID;
    if (N == 0 || width == 0) return RECON_OK;
    const int vec4 = ((width | ld_rows | ld_out) & 3) == 0 && !((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(out)) & 15);
    hipLaunchKernelGGL(k_rows_expand, dim3(static_cast<unsigned>(ceil_div64(N, kBlock / 64))), dim3(kBlock), 0, as_stream(stream), rows, ld_rows, node_row, N, width, out,
                       ld_out, vec4);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_gat_atp_fwd(const recon_graph* g, const recon_gat_atp_args* a, recon_stream_t stream) {
    int rc = recon_gat_atp_scores(g, a, stream);
    if (rc != RECON_OK) return rc;
    rc = recon_gat_atp_aggregate(g, a, stream);
    if (rc != RECON_OK) return rc;
    return recon_gat_atp_project(g, a, stream);
}

extern "C" int recon_gat_atp_bwd_gee_bf16_supported(int32_t F, int32_t R, int32_t D, int32_t H) {
    recon::AtpShape s;
    if (!recon_gat_atp_bf16_io_supported(F, R, D, H) || !recon::atp_shape(F, R, H, &s)) return 0;
    // one head group per wave (no read-modify-write of the stored rows), and the LDS-ring instance of the edge pass (the one built with the bfloat16 store)
    const size_t lds_k2 = static_cast<size_t>(H) * (F + R) * sizeof(float) + (kBlock / 64) * 512;
    const bool lring = s.vec == 4 && s.kr == 1 && lds_k2 + (kBlock / 64) * kK2LdsRingBytes <= 64 * 1024 && cfg_char(CFG_K2_LDS_RING) != '0';
    return (s.ht >= H && lring) ? 1 : 0;
}

extern "C" size_t recon_gat_atp_bwd_split_bytes(int32_t N, int32_t D, int32_t H) {
    if (N <= 0 || D <= 0 || H <= 0) return 256;
    return align_up(static_cast<size_t>(3) * N * bx3_kp(H * D) * 2, 256);
}

extern "C" size_t recon_gat_atp_bwd_partial_floats(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H) {
    const int64_t W = 2LL * F + R;
    int s1 = gemm_pick_split_k(static_cast<int32_t>(W), D, N, H);
    const int s2 = bx3_kmajor_splits(N, bx3_kmajor_split_k(static_cast<int32_t>(W), D, N, H));   // split-precision form
    if (s2 > s1) s1 = s2;
    size_t need = static_cast<size_t>(s1) * H * D * W;                         // g_a^T = V^T g_h, batched over heads
    (void)E; (void)F; (void)R;
    return need > 0 ? need : 1;
}

// scratch of the skinny score-gradient products (<= kSkinnySlices row slices x 16 columns); separate from `partial` so that
// the weight-gradient GEMM may run concurrently on another stream
constexpr int kSkinnySlices = 1024;                              // row slices of the skinny products (first pass blocks)
extern "C" size_t recon_gat_atp_bwd_partial2_floats(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H) {
    (void)N; (void)E; (void)D; (void)H;
    const size_t mx = static_cast<size_t>(F > R ? F : R);
    return static_cast<size_t>(3) * kSkinnySlices * 16 * mx;       // three products in flight
}

extern "C" int recon_gat_atp_bwd(const recon_graph* g, const recon_gat_atp_bwd_args* b, recon_stream_t stream) {
    return recon_gat_atp_bwd_phase(g, b, RECON_ATP_BWD_ALL, stream);
}

// The backward in four independently launchable phases (bit mask), so that a caller can overlap the
// MFMA-bound weight-gradient GEMM (WEIGHTS) with the HBM-bound edge chain (INPUTS) on two streams:
//   PREPARE -> { INPUTS , WEIGHTS } -> FINISH      (INPUTS and WEIGHTS only read what PREPARE wrote)
extern "C" int recon_gat_atp_bwd_phase(const recon_graph* g, const recon_gat_atp_bwd_args* b, int32_t phases, recon_stream_t stream) {
    if (!b) return RECON_ERR_INVALID;
    const recon_gat_atp_args* a = &b->fwd;
    int rc = check_atp(g, a);
    if (rc != RECON_OK) return rc;
    if (!recon_gat_atp_supported(a->N, a->E, a->F, a->R, a->D, a->H)) return RECON_ERR_UNSUPPORTED;
    if (!a->Z || !a->Zk || !b->grad_out || !b->g_V || !b->gxd || !b->Gs || !b->g_u || !b->partial || !b->partial2 || !b->q) return RECON_ERR_INVALID;
    if (a->E > 0 && (!a->sigma || !b->g_sigma || !b->Gxs)) return RECON_ERR_INVALID;
    const bool hx2 = atp_hx2(a);
    if ((a->concat || atp_rows(g, a).row_node) && !b->g_h && !hx2) return RECON_ERR_INVALID;   // (compacted rows: g_h is also the row-order copy of grad_out)
    if (hx2 && (!b->gh_split || (reinterpret_cast<uintptr_t>(b->gh_split) & 15) || (b->ld_gout & 3))) return RECON_ERR_INVALID;
    if (b->ld_gout < a->H * a->D) return RECON_ERR_INVALID;
    if (a->N == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    const int32_t N = a->N, E = a->E, F = a->F, R = a->R, D = a->D, H = a->H, W = 2 * F + R;
    // NR rows (row compaction: the destination nodes with edges; else N): `out`, g_h, q, V, g_V, Z, Zk, gxd and the destination half of Gs
    // go by row; x, c_node, g_x and the source half of Gs by node; grad_out is read through row_node
    const AtpRows rw = atp_rows(g, a);
    const int32_t NR = rw.n;
    const int64_t HD = 1LL * H * D;
    AtpShape s;
    atp_shape(F, R, H, &s);
    const size_t lds_u = static_cast<size_t>(H) * (F + R) * sizeof(float);                  // k_gat_atp_bwd's copy of u: the dst and rel parts
    if (lds_u > 60 * 1024) return RECON_ERR_UNSUPPORTED;
    const size_t lds_k2 = lds_u + (kBlock / 64) * 512;                 // + the next node's index vectors, per wave (k_gat_atp_bwd)

    // (0) through the ELU, and q = g_h . h per (node, head)
    const float* gh = b->grad_out;
    int32_t ld_gh = b->ld_gout;
    // bf16 term planes of g_h [3][N][kp(HD)] for the split-precision weight-gradient GEMM (written by the same pass)
    const int64_t ld_ghp = bx3_kp(static_cast<int32_t>(HD));
    const bool gh_planes = hx2 || (b->gh_split && a->a_split && !(reinterpret_cast<uintptr_t>(b->gh_split) & 15) && (D % 8) == 0 &&
                                   ((b->ld_gout | a->ld_out) & 3) == 0);
    uint16_t* ghp = gh_planes ? static_cast<uint16_t*>(b->gh_split) : nullptr;
    // f16 x 2 with heads of at most 256 columns: the planes of g_h carry PER-ROW scales (k_elu_grad_q finds a row's maximum in the wave that
    // holds the row), so no pass over grad_out for a tensor-wide maximum runs in front (k_hx2_amax: 52 MB, 14 us + a launch at cfg 2).  The
    // inverse scales [H][ldi] live behind the two planes, in the room of the third plane that only the bf16 x 3 family uses.
    const int64_t ldi = (static_cast<int64_t>(NR) + 7) / 8 * 8;
    const size_t inv_off = align_up(static_cast<size_t>(2) * N * bx3_kp(static_cast<int32_t>(HD)) * 2, 256);
    const bool row_scaled = hx2 && D <= 256 && ((a->ld_out | b->ld_gout) & 3) == 0 && cfg_char(CFG_ATP_ROW_SCALE) != '0' &&
                            inv_off + static_cast<size_t>(H) * ldi * sizeof(float) <= recon_gat_atp_bwd_split_bytes(N, D, H);
    float* row_inv = row_scaled ? reinterpret_cast<float*>(static_cast<char*>(b->gh_split) + inv_off) : nullptr;
    if (phases & RECON_ATP_BWD_PREPARE) {
        if (hx2 && !row_scaled) {                                 // max |grad_out| bounds |g_h| (|elu'| <= 1): the scale of the g_h planes
            // quantity 3 is zero here: the scores stage cleared all of aux, and a backward pass clears it again when it is done with it
            // (k_score_vec_bwd) — a fill of its own cost 5 us per step
            rc = hx2_amax(b->grad_out, N, static_cast<int32_t>(HD), b->ld_gout, atp_q(a, 3), st);
            if (rc != RECON_OK) return rc;
        }
        hipLaunchKernelGGL(k_elu_grad_q, dim3(static_cast<unsigned>(ceil_div64(1LL * NR * H, 16))), dim3(256), 0, st, b->grad_out, b->ld_gout, a->out,
                           a->ld_out, NR, H, D, a->concat, ((a->concat || rw.row_node) && !hx2) ? b->g_h : nullptr, b->q, ghp, hx2 ? 2LL * D : ld_ghp,
                           hx2 ? static_cast<int64_t>(D) : static_cast<int64_t>(NR) * ld_ghp, hx2 ? 1 : 0,
                           hx2 ? atp_scale_g(a) : Hx2Scale{nullptr, nullptr, 1.f}, row_inv, ldi, row_scaled ? atp_q(a, 3) : nullptr, rw.row_node);
    }
    if (a->concat || rw.row_node) { gh = b->g_h; ld_gh = static_cast<int32_t>(HD); }
    GemmBatch bt;
    bt.batch = H; bt.epilogue = 0;
    if (phases & RECON_ATP_BWD_INPUTS) {
    // (1) g_V[:, h, :] = g_h[:, h, :] . a[h]
    {
        OperandDesc A = plain_operand(gh, ld_gh);
        OperandDesc B = plain_operand(a->a, W);
        OutputDesc C = plain_output(b->g_V, static_cast<int64_t>(H) * W);
        bt.a_bs = D; bt.b_bs = static_cast<int64_t>(D) * W; bt.c_bs = W;
        if (hx2)
            rc = gemm_hx2_batched(NR, W, D, b->gh_split, D, 2LL * D, 2LL * NR * D,           // g_h terms [H][N][2][D]
                                  static_cast<const char*>(a->a_split) + split_part_bytes(D, W, H), C, bt,
                                  row_scaled ? Hx2Scale{nullptr, nullptr, 1.f} : atp_scale_g(a), atp_scale_a(a), st, 0, row_inv, ldi);
        else if (a->a_split && bx3_supported(A, D, bt))
            rc = gemm_bx3_batched(NR, W, D, A, static_cast<const char*>(a->a_split) + split_part_bytes(D, W, H), C, bt, st);
        else
            rc = gemm_f32_batched(NR, W, D, A, true, B, false, C, bt, 1, nullptr, st);
        if (rc != RECON_OK) return rc;
    }
    // (2) edge pass over the destination CSR
    {
        AtpBwdK p;
        p.rowptr = rw.rowptr; p.row_node = rw.row_node; p.src = g->src; p.eid = a->ee_index ? a->ee_index : g->eid;
        p.gee_by_slot = a->ee_index ? 1 : 0;
        p.gee_b16 = b->g_ee_bf16 ? 1 : 0;
        if (b->g_ee_bf16 && !(a->io_bf16 && !a->ee_index && b->g_edge_embed && recon_gat_atp_bwd_gee_bf16_supported(F, R, D, H))) return RECON_ERR_UNSUPPORTED;
        p.x = a->x; p.ee = a->edge_embed; p.keep = a->keep; p.sigma = a->sigma; p.Z = a->Z; p.Zk = a->Zk;
        p.q = b->q; p.gV = b->g_V; p.u = a->u;
        p.gsigma = b->g_sigma; p.Gs_dst = b->Gs; p.Gxs = b->Gxs; p.gxd = b->gxd; p.g_ee = b->g_edge_embed;
        p.N = NR; p.E = E; p.F = F; p.R = R; p.H = H; p.alpha = a->alpha;
        const bool hubs = atp_hubs(g);
        if (hubs && static_cast<size_t>(g->hub_ws_floats) < recon_graph_hub_ws_floats(g, F, R, H)) return RECON_ERR_WORKSPACE;
        p.hub_chunk = hubs ? g->hub_chunk : 0; p.n_piece = hubs ? g->n_piece : 0;
        p.piece = reinterpret_cast<const int4*>(g->piece); p.hubG = g->hub_ws;
        // persistent waves (no hub pieces): as many workgroups as the kernel's occupancy keeps resident (k2_resident_blocks), a multiple of 8
        // (one share per XCD) and never more than the nodes need; with hub pieces: one piece / node per wave, the pieces' blocks in front
        dim3 grid;
        // the LDS row ring (LR) for rows of at most 1 KiB wherever it fits the 64 KiB a launch gets without asking; RECON_K2_LDS_RING=0: off
        const bool lring = s.vec == 4 && s.kr == 1 && lds_k2 + (kBlock / 64) * kK2LdsRingBytes <= 64 * 1024 && cfg_char(CFG_K2_LDS_RING) != '0';
        const size_t lds_run = lds_k2 + (lring ? (kBlock / 64) * kK2LdsRingBytes : 0);
#define CALL_BWD(V_, K_, H_) do { constexpr bool LRC = V_ == 4 && K_ == 1;                                                                   \
                                  if (V_ == 4 && a->io_bf16) { if (LRC && lring && p.gee_b16) hipLaunchKernelGGL((k_gat_atp_bwd<4, K_, H_, true, LRC, LRC>), grid, dim3(kBlock), lds_run, st, p); \
                                                               else if (LRC && lring) hipLaunchKernelGGL((k_gat_atp_bwd<4, K_, H_, true, LRC>), grid, dim3(kBlock), lds_run, st, p); \
                                                               else hipLaunchKernelGGL((k_gat_atp_bwd<4, K_, H_, true>), grid, dim3(kBlock), lds_run, st, p); }          \
                                  else if (LRC && lring) hipLaunchKernelGGL((k_gat_atp_bwd<V_, K_, H_, false, LRC>), grid, dim3(kBlock), lds_run, st, p);              \
                                  else hipLaunchKernelGGL((k_gat_atp_bwd<V_, K_, H_>), grid, dim3(kBlock), lds_run, st, p); } while (0)
        p.persist = (p.n_piece == 0 && cfg_char(CFG_K2_PERSIST) != '0') ? 1 : 0;
        if (p.persist) {
            const int64_t need = ceil_div64(NR, kBlock / 64);
            int64_t nb = k2_resident_blocks(s.kr, s.ht, lds_run);
            if (nb > need) nb = (need + 7) / 8 * 8;
            grid = dim3(static_cast<unsigned>(nb));
        } else {
            grid = dim3(static_cast<unsigned>(ceil_div64(NR, kBlock / 64) + ceil_div64(p.n_piece, kBlock / 64)));
        }
        ATP_DISPATCH(s, CALL_BWD);
#undef CALL_BWD
        if (hubs) {
            const dim3 gh(static_cast<unsigned>(ceil_div64(g->n_hub, kBlock / 64)));
            if (s.vec == 4) hipLaunchKernelGGL((k_gat_atp_hub_bwd<4>), gh, dim3(kBlock), 0, st, p, g->hub_node, g->hub_ptr, g->n_hub);
            else hipLaunchKernelGGL((k_gat_atp_hub_bwd<2>), gh, dim3(kBlock), 0, st, p, g->hub_node, g->hub_ptr, g->n_hub);
        }
        RECON_CHECK_LAUNCH();
    }
    // (3) source-side sums over the CSC view
    {
        AtpSrcK p;
        p.rowptr_src = g->rowptr_src; p.slot_by_src = g->slot_by_src; p.Gxs = b->Gxs; p.gxd = b->gxd; p.gsigma = b->g_sigma;
        p.g_x = b->g_x; p.Gs_src = b->Gs; p.u = a->u; p.W = W;
        p.N = N; p.F = F; p.H = H; p.node_row = rw.node_row;
        const bool hubs = g->hub_chunk > 0 && g->n_hub_src > 0 && g->n_piece_src > 0 && g->hub_node_src && g->hub_ptr_src && g->piece_src && g->hub_ws;
        if (hubs && static_cast<size_t>(g->hub_ws_floats) < recon_graph_hub_ws_floats(g, F, R, H)) return RECON_ERR_WORKSPACE;
        p.hub_chunk = hubs ? g->hub_chunk : 0; p.n_piece = hubs ? g->n_piece_src : 0;
        p.piece = reinterpret_cast<const int4*>(g->piece_src); p.hubP = g->hub_ws;
        dim3 grid(static_cast<unsigned>(ceil_div64(N, kBlock / 64) + ceil_div64(p.n_piece, kBlock / 64)));
        const int key = s.vec * 10 + s.kr;
        switch (key) {
            case 41: hipLaunchKernelGGL((k_gat_atp_src<4, 1>), grid, dim3(kBlock), 0, st, p); break;
            case 42: hipLaunchKernelGGL((k_gat_atp_src<4, 2>), grid, dim3(kBlock), 0, st, p); break;
            case 44: hipLaunchKernelGGL((k_gat_atp_src<4, 4>), grid, dim3(kBlock), 0, st, p); break;
            case 48: hipLaunchKernelGGL((k_gat_atp_src<4, 8>), grid, dim3(kBlock), 0, st, p); break;
            case 21: hipLaunchKernelGGL((k_gat_atp_src<2, 1>), grid, dim3(kBlock), 0, st, p); break;
            case 22: hipLaunchKernelGGL((k_gat_atp_src<2, 2>), grid, dim3(kBlock), 0, st, p); break;
            case 24: hipLaunchKernelGGL((k_gat_atp_src<2, 4>), grid, dim3(kBlock), 0, st, p); break;
            default: hipLaunchKernelGGL((k_gat_atp_src<2, 8>), grid, dim3(kBlock), 0, st, p); break;
        }
        if (hubs) hipLaunchKernelGGL(k_gat_atp_hub_src, dim3(static_cast<unsigned>(ceil_div64(1LL * g->n_hub_src * ceil_div64(F, 64), kBlock / 64))), dim3(kBlock), 0, st, p,
                                     g->hub_node_src, g->hub_ptr_src, g->n_hub_src);
        RECON_CHECK_LAUNCH();
    }
    }   // INPUTS (its score-gradient products follow below)
    if (b->g_a || b->g_a_2) {
        if (!b->g_a) return RECON_ERR_INVALID;                          // g_a_2 is produced together with g_a
        // (4) g_a[h] = g_h[:, h, :]^T . V[:, h, :], computed as its transpose V^T g_h so that D (<= 208 at cfg 2) is the
        //     tile's column dimension; the split-K second pass stores it back transposed
        if (phases & RECON_ATP_BWD_WEIGHTS) {
            OperandDesc A = plain_operand(a->V, static_cast<int64_t>(H) * W);      // major = k (node), minor = m (w)
            OperandDesc B = plain_operand(gh, ld_gh);                              // major = k (node), minor = n (d)
            OutputDesc C = plain_output(b->g_a, W);
            GemmBatch bw = bt;
            bw.a_bs = W; bw.b_bs = D; bw.c_bs = static_cast<int64_t>(D) * W; bw.c_transpose = 1;
            const int64_t ldv = static_cast<int64_t>(H) * W;
            if (hx2) {                                                                     // f16 x 2: both operands are half planes
                const int sk = bx3_kmajor_splits(NR, bx3_kmajor_split_k(W, D, NR, H));
                rc = gemm_hx2_kmajor_batched(W, D, NR, a->V, 2LL * W, W, 2LL * NR * W, b->gh_split, 2LL * D, D, 2LL * NR * D,
                                             H, sk, b->partial, static_cast<const char*>(a->aux) + kHx2ZeroPageOffset, atp_scale_v(a), atp_scale_g(a), st,
                                             atp_dst_shared(a), row_inv, ldi);
                // its second pass runs in FINISH, fused with the score path's terms (k_atp_weights_finish) — unless the caller wants G now
                if (rc == RECON_OK && (phases & RECON_ATP_BWD_EARLY_SUM)) rc = splitk_reduce(b->partial, sk, W, D, C, bw.c_bs, H, 0, true, st);
            } else if (gh_planes && bx3_kmajor_supported(a->V, ldv, W, ld_ghp, D, W, D)) {    // split-precision MFMA, both operands k-major
                const int sk = bx3_kmajor_splits(NR, bx3_kmajor_split_k(W, D, NR, H));
                rc = gemm_bx3_kmajor_batched(W, D, NR, a->V, ldv, W, b->gh_split, ld_ghp, static_cast<int64_t>(NR) * ld_ghp, D, H, sk, b->partial, st);
                if (rc == RECON_OK) rc = splitk_reduce(b->partial, sk, W, D, C, bw.c_bs, H, 0, true, st);
            } else {
                const int sk = gemm_pick_split_k(W, D, NR, H);
                rc = gemm_f32_batched(W, D, NR, A, false, B, false, C, bw, sk, b->partial, st);
            }
            if (rc != RECON_OK) return rc;
        }
        // (5) g_u = [Gs_dst | Gs_src]^T x   and   gsigma^T edge_embed[eid]   (skinny products, fixed-order reduce)
        if (phases & RECON_ATP_BWD_INPUTS) {
            // Each product: out[(j % P)*S1 + (j / P)*S2 + :] = sum_r G[r][j] * X[row(r)][:].  Up to THREE products share one
            // launch for the partial sums and one for the fixed-order reduce (these kernels are a few MB each and latency
            // bound: every launch saved is ~10 us); each third of `partial2` serves one product.
            struct Prod { const float* G; int ldg, nj; const float* X; const int32_t* gather; int rows, K, P; int64_t S1, S2; float* out; };
            const int32_t* ee_gather = a->ee_index ? a->ee_index : g->eid;
            const size_t third = recon_gat_atp_bwd_partial2_floats(N, E, F, R, D, H) / 3;
            auto run_jobs = [&](const Prod* pr, int count) {
                SkinnyJob sj[3];
                SkinnyRedJob rj[3];
                int nj_max = 0, nb_total = 0, nr_total = 0;
                for (int i = 0; i < 3; ++i) {
                    sj[i] = SkinnyJob{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 1, 0};
                    rj[i] = SkinnyRedJob{nullptr, nullptr, 0, 0, 0, 0, 0, 1, 0};
                    if (i >= count) continue;
                    const Prod& q = pr[i];
                    if (q.rows <= 0) {
                        for (int j = 0; j < q.nj; ++j) (void)hipMemsetAsync(q.out + (j % q.P) * q.S1 + (j / q.P) * q.S2, 0, sizeof(float) * q.K, st);
                        continue;
                    }
                    int rpb = static_cast<int>(ceil_div64(q.rows, kSkinnySlices));
                    if (rpb < 32) rpb = 32;                             // >= 8 rows per wave
                    const int nb = static_cast<int>(ceil_div64(q.rows, rpb));
                    float* part = b->partial2 + i * third;
                    sj[i] = SkinnyJob{q.G, q.X, q.gather, part, q.ldg, q.nj, q.rows, q.K, rpb, nb};
                    rj[i] = SkinnyRedJob{part, q.out, q.S1, q.S2, nb, q.nj, q.K, q.P, static_cast<int32_t>(ceil_div64(1LL * q.nj * q.K, 16))};
                    if (q.nj > nj_max) nj_max = q.nj;
                    nb_total += nb; nr_total += rj[i].nblocks;
                }
                if (nb_total == 0) return;
                const dim3 gp(static_cast<unsigned>(nb_total));
                if (a->io_bf16) {                                        // (at most 8 heads there: recon_gat_atp_bf16_io_supported)
                    if (nj_max <= 8) hipLaunchKernelGGL((k_skinny_tn_partial<8, true>), gp, dim3(256), 0, st, sj[0], sj[1], sj[2]);
                    else hipLaunchKernelGGL((k_skinny_tn_partial<16, true>), gp, dim3(256), 0, st, sj[0], sj[1], sj[2]);
                } else if (nj_max <= 8) hipLaunchKernelGGL((k_skinny_tn_partial<8>), gp, dim3(256), 0, st, sj[0], sj[1], sj[2]);
                else hipLaunchKernelGGL((k_skinny_tn_partial<16>), gp, dim3(256), 0, st, sj[0], sj[1], sj[2]);
                hipLaunchKernelGGL(k_skinny_reduce, dim3(static_cast<unsigned>(nr_total)), dim3(1024), 0, st, rj[0], rj[1], rj[2]);
            };
            if (H <= 8) {
                // the common case, all three in one <8> launch: node-side products (Gs is [N][2H], dst sums | src sums: column
                // (s, h) lands in g_u[h][s*F ...]) as two H-column jobs, and the edge-side product g_sigma^T edge_embed[eid]
                const Prod pr[3] = {{b->Gs, 2 * H, H, a->x, rw.row_node, NR, F, H, W, 0, b->g_u},      // destination sums go by row: x through row_node
                                    {b->Gs + H, 2 * H, H, a->x, nullptr, N, F, H, W, 0, b->g_u + F},
                                    {b->g_sigma, H, H, a->edge_embed, ee_gather, E, R, H, W, 0, b->g_u + 2 * F}};
                run_jobs(pr, 3);
            } else {
                for (int j = 0; j < 2 * H; ++j) {                        // more than 8 heads: one column at a time keeps the map simple
                    const Prod one = {b->Gs + j, 2 * H, 1, a->x, j < H ? rw.row_node : nullptr, j < H ? NR : N, F, 1, W, 0,
                                      b->g_u + static_cast<int64_t>(j % H) * W + (j / H) * F};
                    run_jobs(&one, 1);
                }
                for (int h0 = 0; h0 < H; h0 += 16) {
                    const int nh = H - h0 < 16 ? H - h0 : 16;
                    const Prod one = {b->g_sigma + h0, H, nh, a->edge_embed, ee_gather, E, R, nh, W, 0, b->g_u + static_cast<int64_t>(h0) * W + 2 * F};
                    run_jobs(&one, 1);
                }
            }
            RECON_CHECK_LAUNCH();
        }
        // (6) through u = a_2^T a
        if ((phases & RECON_ATP_BWD_FINISH) && hx2 && !(phases & RECON_ATP_BWD_EARLY_SUM)) {
            const int sk = bx3_kmajor_splits(NR, bx3_kmajor_split_k(W, D, NR, H));
            const int ntile = static_cast<int>(ceil_div64(W, 32) * ceil_div64(D, 32)) * H;
            hipLaunchKernelGGL(k_atp_weights_finish, dim3(static_cast<unsigned>(ntile + ceil_div64(1LL * H * D, 8))), dim3(256), 0, st, b->partial, sk, W,
                               D, H, a->a, a->a_2, b->g_u, b->g_a, b->g_a_2, atp_q(a, 3), kHx2QuantityWords);
        } else if (phases & RECON_ATP_BWD_FINISH)
        hipLaunchKernelGGL(k_score_vec_bwd, dim3(static_cast<unsigned>(D), static_cast<unsigned>(H)), dim3(256), 0, st, a->a, a->a_2, b->g_u,
                           D, W, b->g_a, b->g_a_2, atp_hx2(a) ? atp_q(a, 3) : nullptr, kHx2QuantityWords);
        RECON_CHECK_LAUNCH();
    } else if ((phases & RECON_ATP_BWD_FINISH) && atp_hx2(a)) {        // input gradients only: no last kernel to clear the slots in
        if (hipMemsetAsync(atp_q(a, 3), 0, sizeof(uint32_t) * kHx2QuantityWords, st) != hipSuccess) return RECON_ERR_LAUNCH;
    }
    return RECON_OK;
}
