// bfloat16 GraphConvolution (models/layers.py:57-63 with bfloat16 tensors; BASELINE.json configs[2], SURVEY 8d cfg 3a):
//     out = relu(adj @ (x @ W) + bias),   bf16 storage, fp32 accumulation.
// x @ W, g_support @ W^T and x^T @ g_support run on the bf16 MFMA GEMMs of gemm_b16.hip; the per-graph aggregate adj @ support
// (n x n x out per graph, n <= 32 in the reference's regime) stays on v_mfma_f32_16x16x4_f32 with bf16 loads / stores — it is
// bandwidth bound and fp32 products cost nothing there.  Activations carry a row stride (ldx, lds, ldo ... multiples of 8
// elements = 16 bytes) so that feature counts like 300 need no repacking between layers: columns past the feature count are
// written as zeros and read as "times zero".
#include <stdlib.h>
#include <type_traits>
#include "recon_common.h"

namespace recon {
int32_t b16_kp(int32_t K);
int b16_pad_planes(const void* src, int64_t ld, bool transposed, int32_t rows, int32_t K, void* dst, hipStream_t st);
int gemm_b16(int32_t M, int32_t N, int32_t K, const void* A, int64_t lda, const void* Bp, void* C, int64_t ldc, bool out_bf16, hipStream_t st);
int b16_kmajor_splits(int32_t M, int32_t N, int32_t K);
int b16_pad_planes_both(const void* src, int64_t ld, int32_t M, int32_t N, void* dst_t, void* dst_n, hipStream_t st);
int gemm_b16_kmajor(int32_t M, int32_t N, int32_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* out, int64_t ldo, float* partial,
                    const void* zeros, hipStream_t st, const B16ReduceJob* extra);
int b16_pad_planes_both_multi(int32_t count, const void* const* src, const int32_t* M, const int32_t* N, void* const* dst_t, void* const* dst_n, hipStream_t st);
int b16_kmajor_splits_multi(int32_t M, int32_t N, int32_t K, int32_t count);
int gemm_b16_kmajor_multi(int32_t count, const B16KmProduct* pr, int32_t K, const void* zeros, hipStream_t st, const B16ReduceJob* extra, int32_t n_extra);

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4_g = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2_g = __attribute__((ext_vector_type(2))) uint32_t;
__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, static_cast<uint32_t>(v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(v)); }
// two floats -> two bf16 in one register: ONE v_cvt_pk_bf16_f32 (converted one by one and combined by hand the compiler emits two
// conversions, a shift and an or: the epilogues of the fused kernels are mostly this)
typedef __bf16 bf16x2_g __attribute__((ext_vector_type(2)));
typedef float f32x2_g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_g{a, b}, bf16x2_g)); }
// byte extents of buffer descriptors: integer arithmetic only (`min<int64_t>(v, 0x7fffffff)` resolves to a double-precision minimum in
// the HIP headers: the extent then lives in vector registers and EVERY access through the descriptor is wrapped in a readfirstlane loop)
__device__ __forceinline__ int sat_i32(int64_t v) { return v < 0x7fffffffLL ? static_cast<int>(v) : 0x7fffffff; }

// Y[b][i][o] = epilogue( sum_j M[i][j] * Xin[b][j][o] ),  M = adj[b] (TRANS = false) or adj[b]^T; block = (64 columns, graph,
// 32-row tile), the contraction walked in chunks of 32 through LDS on the bf16 matrix cores (any n).  MASK: Xin = gout * (fwd_out > 0);
// EPI: + bias, ReLU.
// Columns O <= o < ldy are written as zeros.
// colsum (backward only): per-graph column sums of the masked operand, [B][O] floats — the bias gradient's first pass rides on the tile that is
// in LDS anyway (row tile 0 of each graph adds the chunks up in k order: fixed order) instead of a pass of its own over grad_out and out.
template <bool TRANS, bool MASK, bool EPI>
__global__ void __launch_bounds__(256) k_gcn_b16_aggregate(const uint16_t* __restrict__ adj, const uint16_t* __restrict__ Xin, int64_t ldx,
                                                           const uint16_t* __restrict__ fwd_out, int64_t ldf, const uint16_t* __restrict__ bias,
                                                           int32_t n, int32_t O, uint16_t* __restrict__ Y, int64_t ldy, float* __restrict__ colsum = nullptr,
                                                           const int32_t* __restrict__ node_ptr = nullptr, const int64_t* __restrict__ adj_ptr = nullptr) {
    // Round 5: both tiles staged as bf16 (no widening), ONE v_mfma_f32_16x16x32_bf16 per row tile and chunk of 32 contraction rows (the first
    // form widened to fp32 in LDS and issued sixteen 16x16x4 fp32 MFMAs behind 24 four-byte LDS reads per chunk: 31-36 us per layer on the
    // ragged cfg 5 batch for 0.75 GFLOP).  M's chunk lies in LDS as [i][32 k] (row stride 80 B: the sixteen 16-byte fragment reads of a quarter
    // wave fall on distinct banks) whichever way it is read from memory — TRANS costs nothing; X's chunk as [k][64 o] (row stride 144 B), its
    // fragments (8 consecutive k of one column) through the transposing read.
    constexpr int RSA = 80, RSX = 144;
    __shared__ __attribute__((aligned(16))) unsigned char As[32 * RSA];
    __shared__ __attribute__((aligned(16))) unsigned char Xs[32 * RSX];
    const int b = blockIdx.y, o0 = blockIdx.x * 64, i0 = blockIdx.z * 32;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // RAGGED batch (node_ptr): graph b owns node rows node_ptr[b] .. node_ptr[b + 1] and a dense n_b x n_b adjacency at adj + adj_ptr[b]
    int64_t rowbase = static_cast<int64_t>(b) * n;
    const uint16_t* A = adj + static_cast<int64_t>(b) * n * n;
    if (node_ptr) {
        rowbase = node_ptr[b];
        n = node_ptr[b + 1] - node_ptr[b];
        A = adj + adj_ptr[b];
        if (i0 >= n) return;                                           // uniform for the block
    }
    const int li = lane & 15, lq = lane >> 4;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float csum = 0.f;
    const bool do_colsum = colsum != nullptr && blockIdx.z == 0 && t < 64;
    // EIGHT chunks requested at once, in registers: with one MFMA per row tile a chunk's arithmetic covers nothing, so every chunk requested
    // one ahead still exposed a whole memory round trip (22.8 us per layer at cfg 5); a graph of <= 256 nodes now pays one
    constexpr int NCH = 8;
    uint16_t mraw[NCH][4];
    uint2 xraw[NCH][2], fraw[NCH][2];
    auto tr_frag = [](const unsigned char* lo_p, const unsigned char* hi_p) {
        typedef short i16x4 __attribute__((ext_vector_type(4)));
        const i16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lo_p));
        const i16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(hi_p));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const int tr_row = 8 * lq + (li >> 2), tr_col = 16 * w + 4 * (li & 3);
    bool first = true;
    for (int kg = 0; kg < n; kg += 32 * NCH) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k0 = kg + 32 * c;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // the lanes of a wave walk the CONTIGUOUS index of the adjacency (k for adj, i for adj^T): 32 two-byte elements = one or two
                // cache lines per request instead of 32 (the element-wise gather of adj rows was what the kernel waited for)
                const int idx = t + 256 * u, k = TRANS ? idx >> 5 : idx & 31, i = TRANS ? idx & 31 : idx >> 5;
                mraw[c][u] = (k0 + k < n && i0 + i < n) ? (TRANS ? A[static_cast<int64_t>(k0 + k) * n + i0 + i] : A[static_cast<int64_t>(i0 + i) * n + k0 + k])
                                                        : static_cast<uint16_t>(0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {                                  // 4 consecutive columns per thread: 8-byte loads (row strides are
                const int idx = t + 256 * u, k = idx >> 4, oq = o0 + 4 * (idx & 15);  // multiples of 8 elements, so a quad never straddles a row)
                const bool ok = k0 + k < n && oq < ldx;
                const int64_t row = rowbase + k0 + k;
                xraw[c][u] = ok ? *reinterpret_cast<const uint2*>(Xin + row * ldx + oq) : make_uint2(0, 0);
                if constexpr (MASK) fraw[c][u] = (ok && oq < ldf) ? *reinterpret_cast<const uint2*>(fwd_out + row * ldf + oq) : make_uint2(0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (kg + 32 * c >= n) break;                                  // uniform for the block
            if (!first) __syncthreads();
            first = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = t + 256 * u, k = TRANS ? idx >> 5 : idx & 31, i = TRANS ? idx & 31 : idx >> 5;
                *reinterpret_cast<uint16_t*>(As + i * RSA + 2 * k) = mraw[c][u];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = t + 256 * u, k = idx >> 4, oq = o0 + 4 * (idx & 15);
                uint32_t v[2] = {xraw[c][u].x, xraw[c][u].y};
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    // columns in [O, ldx) hold zeros or are cleared here; MASK: gradient kept where the forward output is positive (bf16: sign
                    // clear and not zero)
                    bool k0_ = oq + 2 * d < O, k1_ = oq + 2 * d + 1 < O;
                    if constexpr (MASK) {
                        const uint32_t m = d == 0 ? fraw[c][u].x : fraw[c][u].y;
                        k0_ = k0_ && (m & 0x8000u) == 0 && (m & 0x7fffu) != 0;
                        k1_ = k1_ && (m & 0x80000000u) == 0 && (m & 0x7fff0000u) != 0;
                    }
                    v[d] &= (k0_ ? 0xffffu : 0u) | (k1_ ? 0xffff0000u : 0u);
                }
                *reinterpret_cast<uint2*>(Xs + k * RSX + 8 * (idx & 15)) = make_uint2(v[0], v[1]);
            }
            __syncthreads();
            if (do_colsum) {
#pragma unroll 8
                for (int k = 0; k < 32; ++k) csum += bf2f(*reinterpret_cast<const uint16_t*>(Xs + k * RSX + 2 * t));
            }
            const bf16x8 bfrag = tr_frag(Xs + tr_row * RSX + 2 * tr_col, Xs + (tr_row + 4) * RSX + 2 * tr_col);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(As + (16 * rt + li) * RSA + 16 * lq);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, acc[rt], 0, 0, 0);
            }
        }
    }
    if (do_colsum && o0 + t < O) colsum[static_cast<int64_t>(b) * O + o0 + t] = csum;
    const int o = o0 + 16 * w + li;                                 // C layout: col = lane & 15, row = (lane >> 4) * 4 + r
    if (o < ldy) {
        const float bv = (EPI && bias && o < O) ? bf2f(bias[o]) : 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 16 * tt + 4 * lq + r;
                if (i < n) {
                    float v = acc[tt][r];
                    if constexpr (EPI) { v += bv; v = v > 0.f ? v : 0.f; }
                    Y[(rowbase + i) * ldy + o] = o < O ? f2bf(v) : static_cast<uint16_t>(0);
                }
            }
    }
}

// g_adj[b][i][j] = sum_o gpre[b][i][o] * support[b][j][o]
__global__ void __launch_bounds__(256) k_gcn_b16_grad_adj(const uint16_t* __restrict__ gout, int64_t ldg, const uint16_t* __restrict__ fwd_out, int64_t ldf,
                                                          const uint16_t* __restrict__ sup, int64_t lds, int32_t n, int32_t O, uint16_t* __restrict__ gadj,
                                                          const int32_t* __restrict__ node_ptr = nullptr, const int64_t* __restrict__ adj_ptr = nullptr) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int64_t rowbase = static_cast<int64_t>(b) * n, abase = static_cast<int64_t>(b) * n * n;
    if (node_ptr) { rowbase = node_ptr[b]; n = node_ptr[b + 1] - node_ptr[b]; abase = adj_ptr[b]; }
    if (idx >= n * n) return;
    const int i = idx / n, j = idx % n;
    const int64_t ri = rowbase + i, rj = rowbase + j;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(bf2f(fwd_out[ri * ldf + o]) > 0.f ? bf2f(gout[ri * ldg + o]) : 0.f, bf2f(sup[rj * lds + o]), s);
    gadj[abase + idx] = f2bf(s);
}

// g_bias[o] = sum over graphs of the per-graph column sums of gpre (left in `partial` by the aggregate kernel): 16 columns x 64 slice groups
// per block, fixed order
__global__ void __launch_bounds__(1024) k_gcn_b16_bias_reduce(const float* __restrict__ partial, int32_t nb, int32_t O, uint16_t* __restrict__ gbias) {
    __shared__ float red[64][17];
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + e;
    const int per = (nb + 63) / 64;
    const int b0 = grp * per, b1 = min(nb, (grp + 1) * per);
    float s = 0.f;
    if (o < O)
        for (int b = b0; b < b1; ++b) s += partial[static_cast<int64_t>(b) * O + o];
    red[grp][e] = s;
    __syncthreads();
    if (grp == 0 && o < O) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 64; ++g) t += red[g][e];
        gbias[o] = f2bf(t);
    }
}
// ------------------------------------------------------------------------------------------------ fused forward (n <= 32)
// out = relu(adj @ (x @ W) + bias) for graphs of at most 32 nodes in ONE kernel: a wave owns one graph (32 rows), a workgroup four.
//   1. support [32 x O] = x_b [32 x I] . W on v_mfma_f32_16x16x32_bf16: A fragments (row i, 8 consecutive k) are 16-byte global loads
//      into registers, the W^T planes [O][kp(I)] (k contiguous, zero padded) go through a double-buffered LDS slab per K step, shared by
//      the four waves; the product stays in the accumulators (2 row tiles x NT column tiles).
//   2. out^T [O x 32] = support^T . adj^T, again on the matrix cores and WITHOUT moving `support`: in the C layout a lane holds, for
//      column o, rows j = 4 lq + r of both row tiles — read as an MFMA operand (row o = lane & 15 of tile t, eight k slots = nodes
//      4 lq .. + 3 and 16 + 4 lq .. + 3, rounded to bf16 as torch.mm would round `support`) that is exactly an A fragment under a
//      permutation of the contraction index; adj's fragments are fetched under the same permutation (two 8-byte loads per lane).
//      The result's C layout has four consecutive columns o per node: + bias, ReLU, one 8-byte store.
// Traffic per layer: x once, out once, adj, W from L2 — 41 MB at cfg 3a (B = 1024, n = 32, D = 300) against 19.7 MB x 4 + the support
// round trip of the GEMM + aggregate pair.  `support` is not written (the backward recomputes nothing from it unless d adj is wanted,
// and then the unfused path runs).
constexpr int kFusedNT = 20;                      // column tiles of 16: out_features <= 320

struct GcnFusedK {
    const uint16_t* x; int64_t ldx;
    const uint16_t* adj; const uint16_t* wt; const uint16_t* bias;      // wt: W^T planes [O][Ip]
    uint16_t* out; int64_t ldo;
    int32_t B, n, I, Ip, O, nt;                                          // nt = ceil(ldo / 16) <= kFusedNT
    int32_t adj_vec, bias_vec;                                           // 8-byte loads of adj rows / bias are possible
};

__device__ __forceinline__ int gf_lds_off(int row, int kq) { return row * 64 + (((kq + 2 * (row >> 3)) & 3) << 4); }

// NS = column parts per graph: wave w works on graph slot w & 3 of the workgroup and on column tiles [part NTP, (part + 1) NTP), part = w >> 2,
// NTP = kFusedNT / NS — both products are local to a column range (out[:, cols] = adj . support[:, cols]), so the parts never talk; the
// x fragments are fetched by every part (L1 hits), the W^T slab is staged once per workgroup.  More, lighter waves per SIMD (NS = 4: 40
// accumulator registers, four waves per SIMD) hide what one wave per SIMD with 160 accumulator registers could not.
template <int NS, bool VEC>
__global__ void __launch_bounds__(256 * NS, 1) k_gcn_b16_fused_fwd(const GcnFusedK p) {
    constexpr int SLABF = kFusedNT * 16 * 64;
    __shared__ __attribute__((aligned(16))) unsigned char Wsf[2 * SLABF + 16];     // one K step of W^T: [o][32 k], 2 x 20 KiB; 16 bytes nobody reads
    // (pieces / column tiles that do not exist are written there / computed anyway: a guard per piece or tile is a basic block each, and
    // the compiler drains its counters at every join — the K step had been ten times  LDS read -> wait -> two MFMAs  behind branches)
    constexpr int NTP = kFusedNT / NS, NTHR = 256 * NS;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int g = blockIdx.x * 4 + (w & 3);                              // this wave's graph (may be past B: loads come back as zeros)
    const int c_lo = (w >> 2) * NTP;                                     // its first column tile
    const int n = p.n, nt = p.nt;
    const int64_t rows_total = static_cast<int64_t>(p.B) * n;
    // x: rows of graph g, out-of-range rows / columns read as zeros through the descriptor (the tensor's own extent)
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0,
                                                      sat_i32(rows_total * p.ldx * 2), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wt), 0, p.O * p.Ip * 2, 0x00020000);
    uint32_t xoff[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int i = 16 * rt + li;
        xoff[rt] = (g < p.B && i < n) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.ldx + 8 * lq) * 2) : 0xfffffff0u;
    }
    // W^T slab staging: piece s = t + NTHR q -> (row o = s >> 2, k group s & 3) of the K step; rows past O come back as zeros
    constexpr int WQ = (kFusedNT * 16 * 4 + NTHR - 1) / NTHR;
    uint32_t woff[WQ]; int wlds[WQ];
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int s = t + NTHR * q, o = s >> 2, kq = s & 3;
        woff[q] = (o < 16 * nt) ? static_cast<uint32_t>((o * p.Ip + 8 * kq) * 2) : 0xfffffff0u;
        wlds[q] = o < kFusedNT * 16 ? gf_lds_off(o, kq) : -1;
    }
    // ---- adj^T fragments of this graph under the k permutation (slots 0..3: j = 4 lq .., slots 4..7: j = 16 + 4 lq ..), output node i' = li + 16 it,
    // and this lane's bias values — requested HERE, used after the K loop: behind it each would be a dependent round trip of its own
    // (20 column tiles x one bias load each had been 20 serial L2 latencies: most of the kernel)
    uint32_t adjraw[2][2][4];
    u32x2_g adjv[2][2];
    const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.adj), 0,
                                                      sat_i32(static_cast<int64_t>(p.B) * n * n * 2), 0x00020000);
    constexpr bool adj_vec = VEC;                                        // n % 4 == 0 and adj 8-byte aligned: one 8-byte load per four nodes
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 16 * it + li;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j0 = 16 * h + 4 * lq;
            const uint32_t base = static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * n + j0) * 2);
            if constexpr (adj_vec) {
                adjv[it][h] = __builtin_amdgcn_raw_buffer_load_b64(ra, (g < p.B && i < n && j0 < n) ? base : 0xfffffff0u, 0, 0);
            } else {
                // n is arbitrary (<= 32): element-wise 2-byte loads keep rows of odd length and their tails exact
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    adjraw[it][h][q] = __builtin_amdgcn_raw_buffer_load_b16(ra, (g < p.B && i < n && j0 + q < n) ? base + 2u * q : 0xfffffff0u, 0, 0);
            }
        }
    }
    const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.bias ? p.bias : p.x), 0, p.bias ? p.O * 2 : 0, 0x00020000);
    uint32_t braw[NTP][4];
    u32x2_g bvec[NTP];
    constexpr bool bias_vec = VEC;                                       // O % 4 == 0 and bias 8-byte aligned
#pragma unroll
    for (int c = 0; c < NTP; ++c) {
        const uint32_t bo = static_cast<uint32_t>((16 * (c_lo + c) + 4 * lq) * 2);
        if constexpr (bias_vec) bvec[c] = __builtin_amdgcn_raw_buffer_load_b64(rb, bo, 0, 0);
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) braw[c][q] = __builtin_amdgcn_raw_buffer_load_b16(rb, bo + 2u * q, 0, 0);
        }
    }
    const int nks = p.Ip >> 5;
    // Two K steps of requests in flight: the pieces of W^T for step s are requested during step s - 2 (register set s & 1), written to the
    // LDS slab s & 1 at the end of step s - 1 (its last readers passed the barrier of step s - 2) and read in step s; x fragments likewise.
    // One step of MFMAs is ~300 cycles, an L2 round trip several times that: with one step of distance every step waited for its operands.
    u32x4_g wreg[2][WQ], areg[2][2];
    auto load_w = [&](auto SET, int ks) {
        constexpr int S_ = decltype(SET)::value;
        const uint32_t dead = ks < nks ? 0u : 0xfffffff0u;               // past the end: out of range, zeros, no branch
#pragma unroll
        for (int q = 0; q < WQ; ++q) wreg[S_][q] = __builtin_amdgcn_raw_buffer_load_b128(rw, (woff[q] + 64u * ks) | dead, 0, 0);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) areg[S_][rt] = __builtin_amdgcn_raw_buffer_load_b128(rx, (xoff[rt] + 64u * ks) | dead, 0, 0);
    };
    auto store_w = [&](auto SET) {
        constexpr int S_ = decltype(SET)::value;
#pragma unroll
        for (int q = 0; q < WQ; ++q) *reinterpret_cast<u32x4_g*>(Wsf + (wlds[q] >= 0 ? S_ * SLABF + wlds[q] : 2 * SLABF)) = wreg[S_][q];
    };
    f32x4 acc[2][NTP];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < NTP; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // columns in_features .. kp(in_features) of a row are whatever lies behind it (pad columns of a padded row, the next row's first elements
    // of an unpadded one): this lane's x fragment of the LAST K step is masked to the columns that exist, so their content never matters
    u32x4_g tailmask;
    {
        const int k0 = 32 * (nks - 1) + 8 * lq;
        uint32_t m[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = (k0 + 2 * d < p.I ? 0x0000ffffu : 0u) | (k0 + 2 * d + 1 < p.I ? 0xffff0000u : 0u);
        tailmask = u32x4_g{m[0], m[1], m[2], m[3]};
    }
    const u32x4_g allmask = u32x4_g{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    load_w(S0{}, 0);
    load_w(S1{}, 1);
    bf16x8 a_cur[2] = {__builtin_bit_cast(bf16x8, areg[0][0]), __builtin_bit_cast(bf16x8, areg[0][1])};
    store_w(S0{});
    __syncthreads();
    const int b_rd = gf_lds_off(li, lq) + 1024 * c_lo;
    // step ks out of slab / set SET; the other set holds step ks + 1 (requested a step ago)
    auto step = [&](auto SET, auto OTHER, int ks) {
        constexpr int S_ = decltype(SET)::value, O_ = decltype(OTHER)::value;
        const u32x4_g km = ks == nks - 1 ? tailmask : allmask;
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4_g, a_cur[0]) & km), a1 = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4_g, a_cur[1]) & km);
        const bf16x8 n0 = __builtin_bit_cast(bf16x8, areg[O_][0]), n1 = __builtin_bit_cast(bf16x8, areg[O_][1]);      // x fragments of step ks + 1
        store_w(OTHER);                                                  // slab of step ks + 1: its buffer was last read in step ks - 1, a barrier ago
        load_w(SET, ks + 2);                                             // set S_ is free: its W pieces are in LDS, its x fragments in a_cur
        const unsigned char* slab = Wsf + S_ * SLABF;
        constexpr int CG = NTP < 4 ? NTP : 4;                            // column tiles read ahead of their MFMAs (tiles past nt read zero rows of the slab)
#pragma unroll
        for (int c0 = 0; c0 < NTP; c0 += CG) {
            bf16x8 bq[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) if (c0 + c < NTP) bq[c] = *reinterpret_cast<const bf16x8*>(slab + b_rd + 1024 * (c0 + c));
#pragma unroll
            for (int c = 0; c < CG; ++c)
                if (c0 + c < NTP) {
                    acc[0][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bq[c], acc[0][c0 + c], 0, 0, 0);
                    acc[1][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bq[c], acc[1][c0 + c], 0, 0, 0);
                }
        }
        a_cur[0] = n0; a_cur[1] = n1;
        __syncthreads();
    };
    for (int ks = 0; ks < nks; ks += 2) {
        step(S0{}, S1{}, ks);
        if (ks + 1 < nks) step(S1{}, S0{}, ks + 1);
    }
    bf16x8 adjf[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        uint32_t v[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (adj_vec) { v[2 * h] = adjv[it][h].x; v[2 * h + 1] = adjv[it][h].y; }
            else {
                v[2 * h] = (adjraw[it][h][0] & 0xffffu) | (adjraw[it][h][1] << 16);
                v[2 * h + 1] = (adjraw[it][h][2] & 0xffffu) | (adjraw[it][h][3] << 16);
            }
        }
        adjf[it] = __builtin_bit_cast(bf16x8, u32x4_g{v[0], v[1], v[2], v[3]});
    }
    // ---- out^T tile = support^T . adj^T ; + bias, ReLU, 8-byte stores (pad columns O .. ldo are written as zeros)
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, sat_i32(rows_total * p.ldo * 2), 0x00020000);
    auto pack2 = [](float a, float b) { return pack_bf2(a, b); };
#pragma unroll
    for (int c = 0; c < NTP; ++c)
        if (c_lo + c < nt) {
            const bf16x8 sf = __builtin_bit_cast(bf16x8, u32x4_g{pack2(acc[0][c][0], acc[0][c][1]), pack2(acc[0][c][2], acc[0][c][3]),
                                                                   pack2(acc[1][c][0], acc[1][c][1]), pack2(acc[1][c][2], acc[1][c][3])});
            const int o0 = 16 * (c_lo + c) + 4 * lq;
            float bv[4];
            if constexpr (bias_vec) {
                bv[0] = bf2f(static_cast<uint16_t>(bvec[c].x & 0xffffu)); bv[1] = bf2f(static_cast<uint16_t>(bvec[c].x >> 16));
                bv[2] = bf2f(static_cast<uint16_t>(bvec[c].y & 0xffffu)); bv[3] = bf2f(static_cast<uint16_t>(bvec[c].y >> 16));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[q] = bf2f(static_cast<uint16_t>(braw[c][q]));
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                f32x4 r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, adjf[it], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const int i = 16 * it + li;                              // C layout: column = node i, rows o0 .. o0 + 3
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = r[q] + bv[q]; v[q] = (v[q] > 0.f && o0 + q < p.O) ? v[q] : 0.f; }
                const uint32_t off = (g < p.B && i < n && o0 < p.ldo) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.ldo + o0) * 2) : 0xfffffff0u;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_g{pack2(v[0], v[1]), pack2(v[2], v[3])}, ro, off, 0, 0);
            }
        }
}

// ------------------------------------------------------------------------------------------------ fused STACK forward (n <= 32, inference)
// L GraphConvolutions with one adjacency (models/layers.py:57-63 called L times, as the reference's stacks do) in ONE kernel: the fused
// forward above, looped over the layers with the activations of a graph resident in LDS between them — [K step][32 nodes][64 B] images
// (the A fragments of the next layer's x @ W are 16-byte LDS reads of exactly the bits the per-layer kernel would have written to and
// re-read from HBM: results are bit-equal to L calls of it).  HBM traffic: x once, the last layer's result once, adj once; the W^T planes
// of every layer through the LDS slab from L2 ("fused 3-hop, H resident in LDS, 19 KB per graph: 42 MB" at cfg 3a against
// 125 MB layer by layer).  Workgroup = 4 graphs x NS column parts as above; between the K loop of a layer and the writes of its result into
// the image, and again before the next layer reads it, a workgroup barrier.
constexpr int kMaxStack = 8;
struct GcnStackK {
    const uint16_t* x; int64_t ldx;
    const uint16_t* adj;
    const uint16_t* wt[kMaxStack];                                       // W_l^T planes [D][kp(in_l)]
    const uint16_t* bias[kMaxStack];
    uint16_t* out; int64_t ldo;
    uint16_t* save[kMaxStack];                                           // training: the result of layer l < L - 1, [B*n][ldo] (null: not kept)
    uint16_t* xcopy; int64_t ldxc;                                       // training: a copy of x with 16-byte aligned rows (operand of layer 0's weight gradient), or null
    int32_t B, n, I0, D, L, nt;                                          // in_0 = I0, every layer's out = in_{l+1} = D; nt = ceil(ldo / 16)
};

// Every operand arrives by range-checked LDS-DMA (buffer_load ... lds: no staging registers, rows / columns that do not exist come back as
// zeros): the graphs' x tiles straight into the activation images at the start, the W^T slabs of ALL layers through a ring of three
// 20-KiB slabs two K steps ahead (counted vmcnt, raw s_barrier: the copies stay in flight across the barriers; the first form of this
// kernel staged W through registers one step ahead and waited for it at every step).  One K-loop body for every layer; NS = 4 column parts
// = 16 waves per CU fit without spills because nothing but accumulators and fragments lives in registers.
// GPW = graphs per wave: a wave that works on two graphs reads each W^T fragment ONCE for both — the K step of the 16-wave form is bound by
// LDS reads (16 waves x (2 x-fragments + 5 W^T fragments) x 1 KiB = 112 KiB per step at 128 B per clock: ~900 of a step's ~1 800 cycles, in
// lockstep with the barrier, against 640 cycles of matrix pipe); 8 waves x (2 x 2 + 5) KiB = 72 KiB.
template <int NS, int GPW>
__global__ void __launch_bounds__(256 * NS / GPW, 1) k_gcn_b16_stack_fwd(const GcnStackK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char st_sm[];      // W^T slabs [3][20 KiB] | activations [4 graphs][NKI][2 KiB] | scratch 1 KiB
    constexpr int NTP = kFusedNT / NS, NGS = 4 / GPW, NW = NGS * NS, SLAB = kFusedNT * 16 * 64, NRING = 3;      // NGS = graph slots of waves
    constexpr int SP = kFusedNT;                                         // pieces (16 rows x 64 B) of a slab
    constexpr int NDW = (SP + NW - 1) / NW;                               // copy instructions per wave and slab (waves past the last piece: scratch)
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int gw = (w % NGS) * GPW, g0 = blockIdx.x * 4 + gw;            // this wave's first graph (slot in the workgroup, index in the batch)
    const int c_lo = (w / NGS) * NTP;
    const int n = p.n, nt = p.nt, D = p.D;
    const int nk0 = (p.I0 + 31) >> 5, nkh = (D + 31) >> 5, nki = nk0 > nkh ? nk0 : nkh;      // K steps of layer 0 / of the others / of an image
    unsigned char* Hg = st_sm + NRING * SLAB + gw * nki * 2048;
    const int scratch = NRING * SLAB + 4 * nki * 2048;
    const int64_t rows_total = static_cast<int64_t>(p.B) * n;
    // copy lanes: LDS slot s = 64 piece + lane of a [16 rows][64 B] piece -> (row s >> 2, k group under the slot rotation of gf_lds_off)
    const int c_row = lane >> 2, c_kq = ((lane & 3) - 2 * (c_row >> 3)) & 3;
    // ---- x tiles of the workgroup's four graphs into the images: piece = (graph, K step, row half); rows past n and graphs past B read as zeros
    {
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0,
                                                          sat_i32(rows_total * p.ldx * 2), 0x00020000);
        const int npc = 4 * nk0 * 2;
        for (int pc = w; pc < npc; pc += NW) {                          // wave-uniform
            const int gq = pc / (2 * nk0), r = pc - gq * 2 * nk0, ks = r >> 1, half = r & 1;
            const int gg = blockIdx.x * 4 + gq, i = 16 * half + c_row;
            const uint32_t off = (gg < p.B && i < n) ? static_cast<uint32_t>(((static_cast<int64_t>(gg) * n + i) * p.ldx + 32 * ks + 8 * c_kq) * 2) : 0xfffffff0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(st_sm + NRING * SLAB + gq * nki * 2048 + ks * 2048 + half * 1024), 16, off, 0, 0, 0);
        }
    }
    // ---- the W^T slab of flat step f = (layer, K step): this wave's pieces.  Everything that does not change from step to step is set up
    // per LAYER (descriptor, step count) or once (this lane's offsets under the two plane widths, the pieces' places in a slab): the K step
    // itself advances a scalar offset.  (Computed inside every step, pointer load and branches included, the issue was ~400 of a step's
    // ~1 900 cycles — cycle stamps — on every wave's path between the barrier and its fragment reads.)  Steps past the last layer run
    // on a descriptor of zero bytes: zeros into a slab nobody reads.
    int il = 0, ik = 0, ib = 0;                                          // issue side: layer, K step, ring slot
    const int Ip0 = (p.I0 + 31) & ~31, Iph = (D + 31) & ~31;
    uint32_t voff0[NDW], voffh[NDW];
#pragma unroll
    for (int q = 0; q < NDW; ++q) {
        const int pc = q * NW + w, o = 16 * pc + c_row;
        const bool ok = pc < SP && o < 16 * nt;
        voff0[q] = ok ? static_cast<uint32_t>((o * Ip0 + 8 * c_kq) * 2) : 0xfffffff0u;
        voffh[q] = ok ? static_cast<uint32_t>((o * Iph + 8 * c_kq) * 2) : 0xfffffff0u;
    }
    const uint16_t* wcur = p.wt[0];                                      // the layer being copied: planes and their extent (a pointer and an int, not
    int wbytes = D * Ip0 * 2, nk_issue = nk0;                            // the descriptor: a variable of that type does not survive the host pass)
    auto issue = [&]() {
        const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wcur), 0, wbytes, 0x00020000);
#pragma unroll
        for (int q = 0; q < NDW; ++q) {
            const int pc = q * NW + w;                                   // wave-uniform
            const int vo = static_cast<int>(il == 0 ? voff0[q] : voffh[q]), so = 64 * ik;      // (as locals: passed as expressions the host pass drops the kernel)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(st_sm + (pc < SP ? ib * SLAB + 1024 * pc : scratch)), 16, vo, so, 0, 0);
        }
        if (++ik >= nk_issue) {
            ik = 0; ++il; nk_issue = nkh;
            const bool live = il < p.L;
            wcur = p.wt[live ? il : 0]; wbytes = live ? D * Iph * 2 : 0;
        }
        ib = ib + 1 == NRING ? 0 : ib + 1;
    };
    // every layer's bias row into LDS (behind the scratch KiB): [L][kFusedNT * 16] bf16, zeros where a layer has none / past D
    const int bias_sm = scratch + 1024;
    for (int i = t; i < p.L * kFusedNT * 16; i += 64 * NW) {
        const int l = i / (kFusedNT * 16), o = i - l * (kFusedNT * 16);
        reinterpret_cast<uint16_t*>(st_sm + bias_sm)[i] = (p.bias[l] && o < D) ? p.bias[l][o] : static_cast<uint16_t>(0);
    }
    issue();
    issue();
    // adj^T fragments (8-byte loads: n % 4 == 0, adj 8-byte aligned — checked by the host), kept for all layers
    bf16x8 adjf[GPW][2];
    {
        const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.adj), 0,
                                                          sat_i32(static_cast<int64_t>(p.B) * n * n * 2), 0x00020000);
        u32x2_g adjv[GPW][2][2];
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = 16 * it + li, g = g0 + gi;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j0 = 16 * h + 4 * lq;
                    const uint32_t base = static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * n + j0) * 2);
                    adjv[gi][it][h] = __builtin_amdgcn_raw_buffer_load_b64(ra, (g < p.B && i < n && j0 < n) ? base : 0xfffffff0u, 0, 0);
                }
            }
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
            for (int it = 0; it < 2; ++it)
                adjf[gi][it] = __builtin_bit_cast(bf16x8, u32x4_g{adjv[gi][it][0].x, adjv[gi][it][0].y, adjv[gi][it][1].x, adjv[gi][it][1].y});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the x tiles, the first two slabs, adj
    if (p.xcopy) {
        // every wave writes the x pieces it copied itself back out with aligned rows (what lies behind a row's last feature goes along:
        // those columns of the weight-gradient GEMM's operand only reach rows of the product that are never stored)
        const auto rc = __builtin_amdgcn_make_buffer_rsrc(p.xcopy, 0, sat_i32(rows_total * p.ldxc * 2), 0x00020000);
        const int npc = 4 * nk0 * 2;
        for (int pc = w; pc < npc; pc += NW) {
            const int gq = pc / (2 * nk0), r = pc - gq * 2 * nk0, ks = r >> 1, half = r & 1;
            const int gg = blockIdx.x * 4 + gq, i = 16 * half + c_row, col = 32 * ks + 8 * c_kq;
            const u32x4_g v = *reinterpret_cast<const u32x4_g*>(st_sm + NRING * SLAB + gq * nki * 2048 + ks * 2048 + half * 1024 + lane * 16);
            const uint32_t off = (gg < p.B && i < n && col < p.ldxc) ? static_cast<uint32_t>(((static_cast<int64_t>(gg) * n + i) * p.ldxc + col) * 2) : 0xfffffff0u;
            __builtin_amdgcn_raw_buffer_store_b128(v, rc, off, 0, 0);
        }
    }
    const u32x4_g allmask = u32x4_g{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    const int b_rd = gf_lds_off(li, lq) + 1024 * c_lo;
    const int a_rd0 = gf_lds_off(li, lq), a_rd1 = gf_lds_off(16 + li, lq);
    auto pack2 = [](float a, float b) { return pack_bf2(a, b); };
    int cb = 0;                                                          // ring slot of the step being computed
    bool started = false;
#ifdef RECON_STAMPS
    uint64_t stamp[48]; int ns_ = 0;
    stamp[ns_++] = __builtin_readcyclecounter();
#endif

#pragma unroll 1
    for (int l = 0; l < p.L; ++l) {
        const bool last = l == p.L - 1;
        const int I = l == 0 ? p.I0 : D, nks = l == 0 ? nk0 : nkh;
        // where this layer's result goes in memory: the last one to `out`, the others to their `save` buffer when the caller keeps them
        // (training: the backward's ReLU masks and the weight gradients' operands); a descriptor of zero bytes drops every store
        uint16_t* const dstp = last ? p.out : p.save[l];
        const auto ro = __builtin_amdgcn_make_buffer_rsrc(dstp ? dstp : p.out, 0, dstp ? sat_i32(rows_total * p.ldo * 2) : 0, 0x00020000);
        u32x2_g bvec[NTP];
        u32x4_g tailmask;
        {
            const int k0 = 32 * (nks - 1) + 8 * lq;
            uint32_t m[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) m[d] = (k0 + 2 * d < I ? 0x0000ffffu : 0u) | (k0 + 2 * d + 1 < I ? 0xffff0000u : 0u);
            tailmask = u32x4_g{m[0], m[1], m[2], m[3]};
        }
        f32x4 acc[GPW][2][NTP];
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int c = 0; c < NTP; ++c) acc[gi][rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            // this wave's copies of this step have landed (those of the next step may still travel) ...
            if (started) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDW) : "memory");
            started = true;
            // ... everybody's have, everybody is past the previous step's reads of the slab that is requested next, and (first step of a
            // layer) the image written by the previous layer's epilogue is complete
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef RECON_STAMPS
            if (ns_ < 46) stamp[ns_++] = __builtin_readcyclecounter();
#endif
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifdef RECON_STAMPS
            if (ns_ < 46) stamp[ns_++] = __builtin_readcyclecounter();
#endif
            const unsigned char* slab = st_sm + cb * SLAB;
            const u32x4_g km = ks == nks - 1 ? tailmask : allmask;
            bf16x8 af[GPW][2];
#pragma unroll
            for (int gi = 0; gi < GPW; ++gi) {
                const u32x4_g c0 = *reinterpret_cast<const u32x4_g*>(Hg + gi * nki * 2048 + ks * 2048 + a_rd0);
                const u32x4_g c1 = *reinterpret_cast<const u32x4_g*>(Hg + gi * nki * 2048 + ks * 2048 + a_rd1);
                af[gi][0] = __builtin_bit_cast(bf16x8, c0 & km); af[gi][1] = __builtin_bit_cast(bf16x8, c1 & km);
            }
            bf16x8 bq[NTP];
#pragma unroll
            for (int c = 0; c < NTP; ++c) bq[c] = *reinterpret_cast<const bf16x8*>(slab + b_rd + 1024 * c);
            // the copies of step ks + 2 are requested BEHIND this step's fragment reads: a slab's 20 KiB pass the CU's vector-memory path in
            // ~300 cycles (cycle stamps; 64 B per clock), which in front of the reads was dead time for every wave — here it runs under the
            // reads' latency and the first matrix instructions
#ifdef RECON_STAMPS
            if (ns_ < 46) stamp[ns_++] = __builtin_readcyclecounter();
#endif
            issue();
#ifdef RECON_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ns_ < 46) stamp[ns_++] = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (int c = 0; c < NTP; ++c)
#pragma unroll
                for (int gi = 0; gi < GPW; ++gi) {
                    acc[gi][0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[gi][0], bq[c], acc[gi][0][c], 0, 0, 0);
                    acc[gi][1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[gi][1], bq[c], acc[gi][1][c], 0, 0, 0);
                }
            cb = cb + 1 == NRING ? 0 : cb + 1;
        }
        // the layer's bias values out of LDS (copied there once, at the start: requested here from memory, they made this point a vmcnt(0) that
        // drained the slab ring at the end of every layer)
#pragma unroll
        for (int c = 0; c < NTP; ++c) bvec[c] = *reinterpret_cast<const u32x2_g*>(st_sm + bias_sm + l * (kFusedNT * 32) + (16 * (c_lo + c) + 4 * lq) * 2);
        // every wave must be past its last read of the image before the result overwrites it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < NTP; ++c) {
            const int o0 = 16 * (c_lo + c) + 4 * lq;
            float bv[4];
            bv[0] = bf2f(static_cast<uint16_t>(bvec[c].x & 0xffffu)); bv[1] = bf2f(static_cast<uint16_t>(bvec[c].x >> 16));
            bv[2] = bf2f(static_cast<uint16_t>(bvec[c].y & 0xffffu)); bv[3] = bf2f(static_cast<uint16_t>(bvec[c].y >> 16));
#pragma unroll
            for (int gi = 0; gi < GPW; ++gi) {
                const int g = g0 + gi;
                const bf16x8 sf = __builtin_bit_cast(bf16x8, u32x4_g{pack2(acc[gi][0][c][0], acc[gi][0][c][1]), pack2(acc[gi][0][c][2], acc[gi][0][c][3]),
                                                                       pack2(acc[gi][1][c][0], acc[gi][1][c][1]), pack2(acc[gi][1][c][2], acc[gi][1][c][3])});
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    f32x4 r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, adjf[gi][it], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const int i = 16 * it + li;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = r[q] + bv[q]; v[q] = (v[q] > 0.f && o0 + q < D) ? v[q] : 0.f; }
                    const u32x2_g pk = u32x2_g{pack2(v[0], v[1]), pack2(v[2], v[3])};
                    // the last layer's result to memory (masked by an out-of-range offset), every other one into the image (tiles past it: scratch).
                    // Rows of nodes past n hold relu(bias): the per-layer kernel never stores them and reads them back as zeros; adj's columns
                    // for them are zero (out-of-range loads), so they never reach a result
                    const uint32_t off = (g < p.B && i < n && o0 < p.ldo) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.ldo + o0) * 2) : 0xfffffff0u;
                    __builtin_amdgcn_raw_buffer_store_b64(pk, ro, off, 0, 0);
                    unsigned char* hd = (!last && o0 < 32 * nkh) ? Hg + gi * nki * 2048 + (o0 >> 5) * 2048 + gf_lds_off(i, (o0 & 31) >> 3) + 2 * (o0 & 7) : st_sm + scratch;
                    *reinterpret_cast<u32x2_g*>(hd) = pk;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // copies of steps past the end target this workgroup's LDS
#ifdef RECON_STAMPS
    stamp[ns_++] = __builtin_readcyclecounter();
#ifndef RECON_STAMP_WAVE
#define RECON_STAMP_WAVE 1
#endif
    if (blockIdx.x == 7 && t == 64 * RECON_STAMP_WAVE) { uint64_t* o = reinterpret_cast<uint64_t*>(p.out); for (int i = 0; i < ns_; ++i) o[i] = stamp[i]; o[47] = ns_; }
#endif
}

// Round 5, measured and not kept (tools/probe/gcn_stack_scan.py, gcn_stack_stamps.py):
//  * the W^T fragments requested straight into registers (no slab ring, no barrier inside a layer): 88 us against 41 — one K step of
//    lookahead in 128 registers does not cover the round trip that two steps of LDS-DMA cover;
//  * the kernel is bound by a workgroup's LATENCY, not by throughput: 256 graphs (64 workgroups) take 37 us, 1 024 take 43; a layer costs
//    11.5 us = 10 K steps of ~1 800 cycles (640 of them the matrix pipe: four waves per SIMD x 10 MFMAs; the rest the barrier, the
//    fragment reads behind it and the copy wait — cycle stamps of one wave: ~900 cycles at the barrier, ~900 from it to the next) + ~10 k
//    cycles at each layer boundary (the epilogue's dependent MFMA -> add -> ReLU -> image write chains, four waves per SIMD).  Halving the
//    barriers needs 64-wide K steps, whose double buffer (2 x 40 KiB) + four images (80 KiB) is exactly the CU's 160 KiB.
// ------------------------------------------------------------------------------------------------ fused backward (n <= 32)
// g_support = adj^T (grad_out . [out > 0])  and  g_x = g_support W^T  in ONE kernel, the mirror image of the fused forward: the small
// product comes first here, so its operand has to be TRANSPOSED on the way in — the masked gradient tile of a graph ([32 nodes][out] bf16,
// 20 KiB) is staged in LDS and read back through ds_read_b64_tr_b16 (rows = the contraction index j), as is adj:
//   1. per K step of the big product (32 columns o = two 16-row tiles of g_support^T [o][i]): 2 x 2 MFMAs g_support^T = gpre^T . adj, their
//      accumulators (column = node i, rows = 4 consecutive o) are exactly B fragments of the big product under the k permutation of the
//      forward (slots 0..3: o = 4 lq .., slots 4..7: o = 16 + 4 lq ..); the same accumulators leave as 8-byte stores of g_support (for the
//      weight-gradient GEMM), one part of the graph's waves per K step;
//   2. g_x^T [f][i] += W [f][o] . g_support^T [o][i]: A fragments of W out of the double-buffered LDS slab of the planes W [in][kp(out)] (read
//      under the same permutation: two 8-byte pieces per lane), 8-byte stores of g_x (four consecutive features per node).
// The per-graph column sums of the masked gradient (bias gradient, first pass) are taken from the staged tile.
struct GcnFusedBwdK {
    const uint16_t* gout; int64_t ldg;
    const uint16_t* fout; int64_t ldf;
    const uint16_t* adj; const uint16_t* wn;                              // wn: W planes [I][Op]
    uint16_t* gsup; int64_t lds;
    uint16_t* gx; int64_t ldgx;
    float* colsum;                                                       // [B][O] or null
    int32_t B, n, I, O, Op, nt;                                          // Op = kp(out); nt = ceil(I8 / 16) <= kFusedNT
};

constexpr int kRSG = 2 * kFusedNT * 16 + 16;                            // bytes of a row of the staged gradient tile (rows 8 apart on different banks)
constexpr int kRSA = 2 * 32 + 16;                                       // ... of the adj tile

template <int NS>
__global__ void __launch_bounds__(256 * NS, 1) k_gcn_b16_fused_bwd(const GcnFusedBwdK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_sm[];
    unsigned char* Ws = fb_sm;                                           // [2][kFusedNT * 16 * 64]
    unsigned char* Gs = Ws + 2 * kFusedNT * 16 * 64;                     // [4][32][kRSG]
    unsigned char* As = Gs + 4 * 32 * kRSG;                              // [4][32][kRSA]
    constexpr int NTP = kFusedNT / NS, NTHR = 256 * NS;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int gsl = w & 3, part = w >> 2;
    const int g = blockIdx.x * 4 + gsl;
    const int c_lo = part * NTP;
    const int n = p.n, nt = p.nt;
    const int64_t rows_total = static_cast<int64_t>(p.B) * n;
    const int nks = p.Op >> 5;
    unsigned char* Gg = Gs + gsl * 32 * kRSG;
    unsigned char* Ag = As + gsl * 32 * kRSA;
    // ---- stage the masked gradient tile and adj of this graph (its NS waves share the rows)
    {
        const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.gout), 0, sat_i32(rows_total * p.ldg * 2), 0x00020000);
        const auto rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.fout), 0, sat_i32(rows_total * p.ldf * 2), 0x00020000);
        const int npr = kFusedNT * 2;                                    // 16-byte pieces per staged row (320 columns)
        for (int q = part * 64 + lane; q < 32 * npr; q += 64 * NS) {
            const int j = q / npr, pc = q - j * npr, o0 = 8 * pc;
            const bool ok = g < p.B && j < n;
            const uint32_t og = (ok && o0 < p.ldg) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + j) * p.ldg + o0) * 2) : 0xfffffff0u;
            const uint32_t of = (ok && o0 < p.ldf) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + j) * p.ldf + o0) * 2) : 0xfffffff0u;
            const u32x4_g gv = __builtin_amdgcn_raw_buffer_load_b128(rg, og, 0, 0), fv = __builtin_amdgcn_raw_buffer_load_b128(rf, of, 0, 0);
            uint32_t m[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t gq = gv[d], fq = fv[d];
                // forward output > 0 (bf16: sign clear and not zero) and the column exists: keep the gradient, else zero
                const bool k0 = (fq & 0x8000u) == 0 && (fq & 0x7fffu) != 0 && o0 + 2 * d < p.O;
                const bool k1 = (fq & 0x80000000u) == 0 && (fq & 0x7fff0000u) != 0 && o0 + 2 * d + 1 < p.O;
                m[d] = (k0 ? gq & 0xffffu : 0u) | (k1 ? gq & 0xffff0000u : 0u);
            }
            *reinterpret_cast<u32x4_g*>(Gg + j * kRSG + 16 * pc) = u32x4_g{m[0], m[1], m[2], m[3]};
        }
        const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.adj), 0, sat_i32(static_cast<int64_t>(p.B) * n * n * 2), 0x00020000);
        for (int q = part * 64 + lane; q < 32 * 32; q += 64 * NS) {      // adj[j][i], element-wise: any n <= 32
            const int j = q >> 5, i = q & 31;
            const uint32_t v = (g < p.B && j < n && i < n) ? __builtin_amdgcn_raw_buffer_load_b16(ra, static_cast<uint32_t>(((static_cast<int64_t>(g) * n + j) * n + i) * 2), 0, 0) : 0u;
            *reinterpret_cast<uint16_t*>(Ag + j * kRSA + 2 * i) = static_cast<uint16_t>(v);
        }
    }
    // ---- W slab staging (planes W [I][Op], k = o contiguous): piece s -> (row f = s >> 2, k group s & 3) of the K step
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wn), 0, p.I * p.Op * 2, 0x00020000);
    constexpr int WQ = (kFusedNT * 16 * 4 + NTHR - 1) / NTHR;
    uint32_t woff[WQ]; int wlds[WQ];
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int s_ = t + NTHR * q, f = s_ >> 2, kq = s_ & 3;
        woff[q] = (f < 16 * nt && f < p.I) ? static_cast<uint32_t>((f * p.Op + 8 * kq) * 2) : 0xfffffff0u;
        wlds[q] = f < kFusedNT * 16 ? gf_lds_off(f, kq) : -1;
    }
    u32x4_g wreg[2][WQ];
    auto load_w = [&](auto SET, int ks) {
        constexpr int S_ = decltype(SET)::value;
        const uint32_t dead = ks < nks ? 0u : 0xfffffff0u;
#pragma unroll
        for (int q = 0; q < WQ; ++q) wreg[S_][q] = __builtin_amdgcn_raw_buffer_load_b128(rw, (woff[q] + 64u * ks) | dead, 0, 0);
    };
    auto store_w = [&](auto SET) {
        constexpr int S_ = decltype(SET)::value;
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            if (wlds[q] >= 0) *reinterpret_cast<u32x4_g*>(Ws + S_ * (kFusedNT * 16 * 64) + wlds[q]) = wreg[S_][q];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    load_w(S0{}, 0);
    load_w(S1{}, 1);
    store_w(S0{});
    __syncthreads();                                                     // tiles and the first slab are in place
    // ---- bias gradient, first pass: column sums of the staged tile over the graph's nodes (fixed order), by the part-0 wave of each graph
    if (p.colsum && part == 0 && g < p.B) {
        for (int o = lane; o < p.O; o += 64) {
            float sacc = 0.f;
#pragma unroll 8
            for (int j = 0; j < 32; ++j) sacc += bf2f(*reinterpret_cast<const uint16_t*>(Gg + j * kRSG + 2 * o));
            p.colsum[static_cast<int64_t>(g) * p.O + o] = sacc;
        }
    }
    // transposing reads: the 16 lanes of group lq address a 4 (k) x 16 (m) block of halves, lane ip at row ip >> 2, columns 4 (ip & 3) ..; lane ip
    // receives column ip of the four rows; two reads (k rows 8 lq .. + 3 and + 4 .. + 7) make a fragment of 8 consecutive k
    auto tr_frag = [](const unsigned char* lo_p, const unsigned char* hi_p) {
        typedef short i16x4 __attribute__((ext_vector_type(4)));
        const i16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lo_p));
        const i16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(hi_p));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const int tr_row = 8 * lq + (li >> 2), tr_col = 4 * (li & 3);
    bf16x8 adjf[2];                                                      // B fragments of adj: k = node j, column = node i of tile it
#pragma unroll
    for (int it = 0; it < 2; ++it) adjf[it] = tr_frag(Ag + tr_row * kRSA + (16 * it + tr_col) * 2, Ag + (tr_row + 4) * kRSA + (16 * it + tr_col) * 2);
    f32x4 acc[NTP][2];
#pragma unroll
    for (int c = 0; c < NTP; ++c)
#pragma unroll
        for (int it = 0; it < 2; ++it) acc[c][it] = f32x4{0.f, 0.f, 0.f, 0.f};
    const auto rs_ = __builtin_amdgcn_make_buffer_rsrc(p.gsup, 0, sat_i32(rows_total * p.lds * 2), 0x00020000);
    auto pack2 = [](float a, float b) { return pack_bf2(a, b); };
    // W fragment of row f under the k permutation: k = 4 lq .. + 3 is half (lq & 1) of slot lq >> 1, k = 16 + 4 lq .. of slot 2 + (lq >> 1)
    const int wrow = li, whalf = 8 * (lq & 1), wk1 = lq >> 1, wk2 = 2 + (lq >> 1);
    auto step = [&](auto SET, auto OTHER, int ks) {
        constexpr int S_ = decltype(SET)::value;
        store_w(OTHER);                                                  // slab of step ks + 1 (its buffer was last read in step ks - 1)
        load_w(SET, ks + 2);
        // 1. g_support^T tiles of this K step
        bf16x8 bfr[2];
        {
            f32x4 sa[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int ot = 2 * ks + tt;
                const bf16x8 af = tr_frag(Gg + tr_row * kRSG + (16 * ot + tr_col) * 2, Gg + (tr_row + 4) * kRSG + (16 * ot + tr_col) * 2);
#pragma unroll
                for (int it = 0; it < 2; ++it) sa[tt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, adjf[it], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const uint32_t q0 = pack2(sa[0][it][0], sa[0][it][1]), q1 = pack2(sa[0][it][2], sa[0][it][3]);
                const uint32_t q2 = pack2(sa[1][it][0], sa[1][it][1]), q3 = pack2(sa[1][it][2], sa[1][it][3]);
                bfr[it] = __builtin_bit_cast(bf16x8, u32x4_g{q0, q1, q2, q3});
                if ((ks % NS) == part) {                                 // this part writes the K step's columns of g_support (uniform)
                    const int i = 16 * it + li;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int o0 = 16 * (2 * ks + tt) + 4 * lq;
                        const uint32_t off = (g < p.B && i < n && o0 < p.lds) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.lds + o0) * 2) : 0xfffffff0u;
                        __builtin_amdgcn_raw_buffer_store_b64(tt == 0 ? u32x2_g{q0, q1} : u32x2_g{q2, q3}, rs_, off, 0, 0);
                    }
                }
            }
        }
        // 2. g_x^T += W . g_support^T
        const unsigned char* slab = Ws + S_ * (kFusedNT * 16 * 64);
#pragma unroll
        for (int c = 0; c < NTP; ++c)
            if (c_lo + c < nt) {                                         // uniform
                const int f = 16 * (c_lo + c) + wrow;
                const u32x2_g a1 = *reinterpret_cast<const u32x2_g*>(slab + gf_lds_off(f, wk1) + whalf);
                const u32x2_g a2 = *reinterpret_cast<const u32x2_g*>(slab + gf_lds_off(f, wk2) + whalf);
                const bf16x8 af = __builtin_bit_cast(bf16x8, u32x4_g{a1.x, a1.y, a2.x, a2.y});
                acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[0], acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[1], acc[c][1], 0, 0, 0);
            }
        __syncthreads();
    };
    for (int ks = 0; ks < nks; ks += 2) {
        step(S0{}, S1{}, ks);
        if (ks + 1 < nks) step(S1{}, S0{}, ks + 1);
    }
    // ---- g_x: C layout column = node i, rows = four consecutive features
    const auto rx_ = __builtin_amdgcn_make_buffer_rsrc(p.gx, 0, sat_i32(rows_total * p.ldgx * 2), 0x00020000);
#pragma unroll
    for (int c = 0; c < NTP; ++c)
        if (c_lo + c < nt) {
            const int f0 = 16 * (c_lo + c) + 4 * lq;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = 16 * it + li;
                const uint32_t off = (g < p.B && i < n && f0 < p.ldgx) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.ldgx + f0) * 2) : 0xfffffff0u;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_g{pack2(acc[c][it][0], acc[c][it][1]), pack2(acc[c][it][2], acc[c][it][3])}, rx_, off, 0, 0);
            }
        }
}

// ------------------------------------------------------------------------------------------------ fused STACK backward (n <= 32)
// The backward of L GraphConvolutions over one adjacency in ONE kernel — the fused backward above looped over the layers, last to first, with
// the gradient of a graph resident in LDS between them (the mirror image of k_gcn_b16_stack_fwd):
//   layer l:  gpre_l = g_l . [act_l > 0]            (the staged tile: from grad_out for the last layer, from the previous step's accumulators else)
//             g_support_l = adj^T gpre_l            -> memory (operand of the weight gradient x_l^T g_support_l, a k-major GEMM behind this kernel)
//             g_{l-1} = g_support_l W_l^T           -> masked by act_{l-1} (8-byte loads requested at the top of the layer) into the staged tile;
//                                                      layer 0: to memory as g_x when wanted
// W planes [in_l][kp(hidden)] of every layer arrive by LDS-DMA through a ring of three slabs two K steps ahead, exactly as in the forward; a
// K step ends with ONE 8-byte store per lane of g_support (each of the graph's four parts owns one 16 x 16 tile of the step), so the request
// counter at the loop-head wait never holds more than the two copies of the next slab + that store.  Results are bit-equal to the per-layer
// kernels (same products, same roundings, same summation orders).
struct GcnStackBwdK {
    const uint16_t* gout; int64_t ldg;
    const uint16_t* act[kMaxStack]; int64_t lda;                         // act[l] = output of layer l, [B*n][lda]
    const uint16_t* adj;
    const uint16_t* wn[kMaxStack];                                       // W_l planes [in_l][Op]
    uint16_t* gsup[kMaxStack]; int64_t lds;
    uint16_t* gx; int64_t ldgx;                                          // null: not wanted
    float* colsum[kMaxStack];                                            // [B][D] per layer, or null
    int32_t B, n, I0, D, Op, L;
};

__global__ void __launch_bounds__(1024, 1) k_gcn_b16_stack_bwd(const GcnStackBwdK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_sm[];      // W slabs [3][20 KiB] | gradient tiles [4][32][kRSG] | adj [4][32][kRSA] | scratch 1 KiB
    constexpr int NS = 4, NTP = kFusedNT / NS, NW = 4 * NS, SLAB = kFusedNT * 16 * 64, NRING = 3, SP = kFusedNT;
    constexpr int NDW = (SP + NW - 1) / NW;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int gsl = w & 3, part = w >> 2;
    const int g = blockIdx.x * 4 + gsl;
    const int c_lo = part * NTP;
    const int n = p.n, D = p.D;
    const int64_t rows_total = static_cast<int64_t>(p.B) * n;
    const int nks = p.Op >> 5;
    unsigned char* Gg = sb_sm + NRING * SLAB + gsl * 32 * kRSG;
    unsigned char* Ag = sb_sm + NRING * SLAB + 4 * 32 * kRSG + gsl * 32 * kRSA;
    const int scratch = NRING * SLAB + 4 * 32 * kRSG + 4 * 32 * kRSA;
    const int c_row = lane >> 2, c_kq = ((lane & 3) - 2 * (c_row >> 3)) & 3;
    // ---- the W slab of flat step (layer L-1 .. 0, K step): this wave's pieces; per-layer descriptor, per-lane offsets computed once, the K
    // step as a scalar offset (see the forward); steps past the end run on a descriptor of zero bytes
    // (as inline asm: through the builtin the compiler drains the request counter in front of the transposing LDS reads of every K step)
    int il = p.L - 1, ik = 0, ib = 0;
    uint32_t voff[NDW];
#pragma unroll
    for (int q = 0; q < NDW; ++q) {
        const int pc = q * NW + w, f = 16 * pc + c_row;
        voff[q] = pc < SP ? static_cast<uint32_t>((f * p.Op + 8 * c_kq) * 2) : 0xfffffff0u;       // rows past in_l: out of range, zeros
    }
    auto layer_desc = [&](int l) {
        const bool live = l >= 0;
        return dma_descriptor(p.wn[live ? l : 0], live ? static_cast<uint32_t>((l == 0 ? p.I0 : D) * p.Op * 2) : 0u);
    };
    dma_u32x4 rw = layer_desc(il);
    auto issue = [&]() {
#pragma unroll
        for (int q = 0; q < NDW; ++q) {
            const int pc = q * NW + w;                                   // wave-uniform
            dma16_buffer_to_lds(rw, voff[q], sb_sm + (pc < SP ? ib * SLAB + 1024 * pc : scratch), static_cast<uint32_t>(64 * ik));
        }
        if (++ik >= nks) { ik = 0; --il; rw = layer_desc(il); }
        ib = ib + 1 == NRING ? 0 : ib + 1;
    };
    issue();
    issue();
    // ---- stage the last layer's masked gradient tile and adj of this graph (its four waves share the rows)
    {
        const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.gout), 0, sat_i32(rows_total * p.ldg * 2), 0x00020000);
        const auto rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.act[p.L - 1]), 0, sat_i32(rows_total * p.lda * 2), 0x00020000);
        const int npr = kFusedNT * 2;                                    // 16-byte pieces per staged row (320 columns)
        for (int q = part * 64 + lane; q < 32 * npr; q += 64 * NS) {
            const int j = q / npr, pc = q - j * npr, o0 = 8 * pc;
            const bool ok = g < p.B && j < n;
            const uint32_t og = (ok && o0 < p.ldg) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + j) * p.ldg + o0) * 2) : 0xfffffff0u;
            const uint32_t of = (ok && o0 < p.lda) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + j) * p.lda + o0) * 2) : 0xfffffff0u;
            const u32x4_g gv = __builtin_amdgcn_raw_buffer_load_b128(rg, og, 0, 0), fv = __builtin_amdgcn_raw_buffer_load_b128(rf, of, 0, 0);
            uint32_t m[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t gq = gv[d], fq = fv[d];
                const bool k0 = (fq & 0x8000u) == 0 && (fq & 0x7fffu) != 0 && o0 + 2 * d < D;
                const bool k1 = (fq & 0x80000000u) == 0 && (fq & 0x7fff0000u) != 0 && o0 + 2 * d + 1 < D;
                m[d] = (k0 ? gq & 0xffffu : 0u) | (k1 ? gq & 0xffff0000u : 0u);
            }
            *reinterpret_cast<u32x4_g*>(Gg + j * kRSG + 16 * pc) = u32x4_g{m[0], m[1], m[2], m[3]};
        }
        const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.adj), 0, sat_i32(static_cast<int64_t>(p.B) * n * n * 2), 0x00020000);
        for (int q = part * 64 + lane; q < 32 * 32; q += 64 * NS) {      // adj[j][i], element-wise: any n <= 32
            const int j = q >> 5, i = q & 31;
            const uint32_t v = (g < p.B && j < n && i < n) ? __builtin_amdgcn_raw_buffer_load_b16(ra, static_cast<uint32_t>(((static_cast<int64_t>(g) * n + j) * n + i) * 2), 0, 0) : 0u;
            *reinterpret_cast<uint16_t*>(Ag + j * kRSA + 2 * i) = static_cast<uint16_t>(v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // the first two slabs too
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    auto tr_frag = [](const unsigned char* lo_p, const unsigned char* hi_p) {
        typedef short i16x4 __attribute__((ext_vector_type(4)));
        const i16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lo_p));
        const i16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(hi_p));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const int tr_row = 8 * lq + (li >> 2), tr_col = 4 * (li & 3);
    bf16x8 adjf[2];                                                      // B fragments of adj: k = node j, column = node i of tile it
#pragma unroll
    for (int it = 0; it < 2; ++it) adjf[it] = tr_frag(Ag + tr_row * kRSA + (16 * it + tr_col) * 2, Ag + (tr_row + 4) * kRSA + (16 * it + tr_col) * 2);
    auto pack2 = [](float a, float b) { return pack_bf2(a, b); };
    const int wrow = li, whalf = 8 * (lq & 1), wk1 = lq >> 1, wk2 = 2 + (lq >> 1);
    const int my_tt = part >> 1, my_it = part & 1;                       // the 16 x 16 tile of a K step's g_support this part stores
    int cb = 0;
    bool started = false;
    f32x4 acc[NTP][2];

#pragma unroll 1
    for (int l = p.L - 1; l >= 0; --l) {
#pragma unroll
        for (int c = 0; c < NTP; ++c)
#pragma unroll
            for (int it = 0; it < 2; ++it) acc[c][it] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the ReLU mask of the NEXT tile (output of layer l - 1) for this lane's accumulators: requested now, used behind the K loop
        u32x2_g fm[NTP][2];
        {
            const uint16_t* ap = l > 0 ? p.act[l - 1] : p.adj;
            const auto rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ap), 0, l > 0 ? sat_i32(rows_total * p.lda * 2) : 0, 0x00020000);
#pragma unroll
            for (int c = 0; c < NTP; ++c)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int i = 16 * it + li, f0 = 16 * (c_lo + c) + 4 * lq;
                    const uint32_t off = (g < p.B && i < n && f0 < p.lda) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.lda + f0) * 2) : 0xfffffff0u;
                    fm[c][it] = __builtin_amdgcn_raw_buffer_load_b64(rf, off, 0, 0);
                }
        }
        const auto rs_ = __builtin_amdgcn_make_buffer_rsrc(p.gsup[l], 0, sat_i32(rows_total * p.lds * 2), 0x00020000);
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            if (started) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDW) : "memory");      // this wave's copies of this step (and everything older) have landed
            started = true;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // everybody's have; the tile written by the previous layer's epilogue is complete
            asm volatile("" ::: "memory");
            issue();
            // 1. g_support^T tiles of this K step (32 columns o)
            bf16x8 bfr[2];
            {
                f32x4 sa[2][2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int ot = 2 * ks + tt;
                    const bf16x8 af = tr_frag(Gg + tr_row * kRSG + (16 * ot + tr_col) * 2, Gg + (tr_row + 4) * kRSG + (16 * ot + tr_col) * 2);
#pragma unroll
                    for (int it = 0; it < 2; ++it) sa[tt][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, adjf[it], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
                uint32_t q[2][2][2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int it = 0; it < 2; ++it) { q[tt][it][0] = pack2(sa[tt][it][0], sa[tt][it][1]); q[tt][it][1] = pack2(sa[tt][it][2], sa[tt][it][3]); }
#pragma unroll
                for (int it = 0; it < 2; ++it) bfr[it] = __builtin_bit_cast(bf16x8, u32x4_g{q[0][it][0], q[0][it][1], q[1][it][0], q[1][it][1]});
                const int i = 16 * my_it + li, o0 = 16 * (2 * ks + my_tt) + 4 * lq;
                const uint32_t off = (g < p.B && i < n && o0 < p.lds) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.lds + o0) * 2) : 0xfffffff0u;
                const u32x2_g sv = my_tt == 0 ? (my_it == 0 ? u32x2_g{q[0][0][0], q[0][0][1]} : u32x2_g{q[0][1][0], q[0][1][1]})
                                              : (my_it == 0 ? u32x2_g{q[1][0][0], q[1][0][1]} : u32x2_g{q[1][1][0], q[1][1][1]});
                __builtin_amdgcn_raw_buffer_store_b64(sv, rs_, off, 0, 0);
            }
            // 2. g_x^T += W . g_support^T
            const unsigned char* slab = sb_sm + cb * SLAB;
#pragma unroll
            for (int c = 0; c < NTP; ++c) {
                const int f = 16 * (c_lo + c) + wrow;
                const u32x2_g a1 = *reinterpret_cast<const u32x2_g*>(slab + gf_lds_off(f, wk1) + whalf);
                const u32x2_g a2 = *reinterpret_cast<const u32x2_g*>(slab + gf_lds_off(f, wk2) + whalf);
                const bf16x8 af = __builtin_bit_cast(bf16x8, u32x4_g{a1.x, a1.y, a2.x, a2.y});
                acc[c][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[0], acc[c][0], 0, 0, 0);
                acc[c][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[1], acc[c][1], 0, 0, 0);
            }
            cb = cb + 1 == NRING ? 0 : cb + 1;
        }
        // bias gradient, first pass: column sums of the staged tile over the graph's nodes (fixed order), each part its 80 columns
        if (p.colsum[l] && g < p.B) {
            for (int o = 16 * c_lo + lane; o < 16 * (c_lo + NTP) && o < D; o += 64) {
                float sacc = 0.f;
#pragma unroll 8
                for (int j = 0; j < 32; ++j) sacc += bf2f(*reinterpret_cast<const uint16_t*>(Gg + j * kRSG + 2 * o));
                p.colsum[l][static_cast<int64_t>(g) * D + o] = sacc;
            }
        }
        // every wave must be past its last read of the tile before the next one overwrites it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (l > 0) {
#pragma unroll
            for (int c = 0; c < NTP; ++c)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int i = 16 * it + li, f0 = 16 * (c_lo + c) + 4 * lq;
                    uint32_t v0 = pack2(acc[c][it][0], acc[c][it][1]), v1 = pack2(acc[c][it][2], acc[c][it][3]);
                    const uint32_t m0 = fm[c][it].x, m1 = fm[c][it].y;
                    v0 = (((m0 & 0x8000u) == 0 && (m0 & 0x7fffu) != 0) ? v0 & 0xffffu : 0u) | (((m0 & 0x80000000u) == 0 && (m0 & 0x7fff0000u) != 0) ? v0 & 0xffff0000u : 0u);
                    v1 = (((m1 & 0x8000u) == 0 && (m1 & 0x7fffu) != 0) ? v1 & 0xffffu : 0u) | (((m1 & 0x80000000u) == 0 && (m1 & 0x7fff0000u) != 0) ? v1 & 0xffff0000u : 0u);
                    *reinterpret_cast<u32x2_g*>(Gg + i * kRSG + 2 * f0) = u32x2_g{v0, v1};
                }
        } else if (p.gx) {
            const auto rx_ = __builtin_amdgcn_make_buffer_rsrc(p.gx, 0, sat_i32(rows_total * p.ldgx * 2), 0x00020000);
#pragma unroll
            for (int c = 0; c < NTP; ++c)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int i = 16 * it + li, f0 = 16 * (c_lo + c) + 4 * lq;
                    const uint32_t off = (g < p.B && i < n && f0 < p.ldgx) ? static_cast<uint32_t>(((static_cast<int64_t>(g) * n + i) * p.ldgx + f0) * 2) : 0xfffffff0u;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2_g{pack2(acc[c][it][0], acc[c][it][1]), pack2(acc[c][it][2], acc[c][it][3])}, rx_, off, 0, 0);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // copies of steps past the end target this workgroup's LDS
}

bool gcn_fused_ok(const recon_gcn_b16_args* a) {
    const bool off = cfg_char(CFG_GCN_FUSED) == '0';
    return !off && a->n <= 32 && a->ldo <= kFusedNT * 16 && (a->ldo & 3) == 0 &&
           static_cast<int64_t>(a->B) * a->n * a->ldx * 2 < 0x7fffffffLL && static_cast<int64_t>(a->B) * a->n * a->ldo * 2 < 0x7fffffffLL;
}

int check(const recon_gcn_b16_args* a) {
    if (!a || a->B < 0 || a->n <= 0 || a->in_features <= 0 || a->out_features <= 0) return RECON_ERR_INVALID;
    if (!a->x || !a->adj || !a->weight || !a->out || !a->w_planes) return RECON_ERR_INVALID;
    if (a->node_ptr && (!a->adj_ptr || !a->support || a->total_rows < 0)) return RECON_ERR_INVALID;      // ragged batch: the two-kernel form
    if (!a->support && !gcn_fused_ok(a)) return RECON_ERR_INVALID;    // only the fused forward (n <= 32, out <= 320) does without it
    const int64_t i8 = (a->in_features + 7) / 8 * 8, o8 = (a->out_features + 7) / 8 * 8;
    const bool fused = !a->support;                                   // the fused forward masks the K tail: rows need no padding, only 4-byte alignment
    if (fused ? ((a->ldx & 1) || a->ldx < a->in_features) : ((a->ldx & 7) || a->ldx < i8)) return RECON_ERR_INVALID;
    if ((a->ldo & 7) || a->ldo < o8 || (a->support && ((a->lds & 7) || a->lds < o8))) return RECON_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(a->x) | reinterpret_cast<uintptr_t>(a->support) | reinterpret_cast<uintptr_t>(a->out) |
         reinterpret_cast<uintptr_t>(a->w_planes)) & 15) return RECON_ERR_INVALID;
    if (a->B > 65535 || a->n > 65535 * 32) return RECON_ERR_UNSUPPORTED;
    return RECON_OK;
}
size_t planes_part(int32_t rows, int32_t K) { return align_up(static_cast<size_t>(rows) * b16_kp(K) * 2, 256); }

}  // namespace
}  // namespace recon

using namespace recon;

extern "C" size_t recon_gcn_b16_planes_bytes(int32_t in_features, int32_t out_features) {
    if (in_features <= 0 || out_features <= 0) return 256;
    return planes_part(out_features, in_features) + planes_part(in_features, out_features);
}

extern "C" int recon_gcn_b16_fwd(const recon_gcn_b16_args* a, recon_stream_t stream) {
    int rc = check(a);
    if (rc != RECON_OK) return rc;
    if (a->B == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    if (a->node_ptr && (a->total_rows == 0 || a->total_rows > 0x7fffffffLL)) return a->total_rows == 0 ? RECON_OK : RECON_ERR_UNSUPPORTED;
    const int32_t rows = a->node_ptr ? static_cast<int32_t>(a->total_rows) : a->B * a->n, I = a->in_features, O = a->out_features;
    char* wp = static_cast<char*>(a->w_planes);
    // W^T [O][kp(I)] for this product, W [I][kp(O)] for g_x in the backward (both zero padded along k)
    if (!a->w_planes_valid) {
        rc = b16_pad_planes_both(a->weight, O, I, O, wp, wp + planes_part(O, I), st);      // one launch for both layouts
        if (rc != RECON_OK) return rc;
    }
    if (!a->support) {                                               // fused: one kernel, `support` stays in registers
        GcnFusedK k;
        k.x = static_cast<const uint16_t*>(a->x); k.ldx = a->ldx; k.adj = static_cast<const uint16_t*>(a->adj);
        k.wt = reinterpret_cast<const uint16_t*>(wp); k.bias = static_cast<const uint16_t*>(a->bias);
        k.out = static_cast<uint16_t*>(a->out); k.ldo = a->ldo;
        k.B = a->B; k.n = a->n; k.I = I; k.Ip = b16_kp(I); k.O = O; k.nt = static_cast<int32_t>(ceil_div64(a->ldo, 16));
        k.adj_vec = (a->n % 4 == 0 && (reinterpret_cast<uintptr_t>(a->adj) & 7) == 0) ? 1 : 0;
        k.bias_vec = (!a->bias || (O % 4 == 0 && (reinterpret_cast<uintptr_t>(a->bias) & 7) == 0)) ? 1 : 0;
        const int ns = [] { const int v = cfg_int(CFG_GCN_FUSED_PARTS, 4); return v == 1 || v == 2 ? v : 4; }();
        const dim3 fg(static_cast<unsigned>(ceil_div64(a->B, 4)));
        const bool vec = k.adj_vec && k.bias_vec;
#define CALL_F(N_) do { if (vec) hipLaunchKernelGGL((k_gcn_b16_fused_fwd<N_, true>), fg, dim3(256 * N_), 0, st, k); \
                        else hipLaunchKernelGGL((k_gcn_b16_fused_fwd<N_, false>), fg, dim3(256 * N_), 0, st, k); } while (0)
        if (ns == 1) CALL_F(1); else if (ns == 2 || !vec) CALL_F(2); else CALL_F(4);      // the element-wise loads of odd shapes need the registers of the two-part form
#undef CALL_F
        RECON_CHECK_LAUNCH();
        return RECON_OK;
    }
    // support = x @ W      (models/layers.py:58)
    rc = gemm_b16(rows, O, I, a->x, a->ldx, wp, a->support, a->lds, true, st);
    if (rc != RECON_OK) return rc;
    // out = relu(adj @ support + bias)     (:59-63)
    dim3 grid(static_cast<unsigned>(ceil_div64(a->ldo, 64)), static_cast<unsigned>(a->B), static_cast<unsigned>(ceil_div64(a->n, 32)));
    hipLaunchKernelGGL((k_gcn_b16_aggregate<false, false, true>), grid, dim3(256), 0, st, static_cast<const uint16_t*>(a->adj),
                       static_cast<const uint16_t*>(a->support), a->lds, nullptr, 0, static_cast<const uint16_t*>(a->bias), a->n, O,
                       static_cast<uint16_t*>(a->out), a->ldo, nullptr, a->node_ptr, a->adj_ptr);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

static size_t gcn_b16_gw_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t out_features) {
    return static_cast<size_t>(b16_kmajor_splits(in_features, out_features, B * n)) * in_features * out_features;
}

extern "C" size_t recon_gcn_b16_bwd_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t out_features) {
    // [split-K partials of g_W][per-graph column sums for g_bias]: both second passes run in one launch at the end
    return gcn_b16_gw_partial_floats(B, n, in_features, out_features) + static_cast<size_t>(B > 0 ? B : 0) * (out_features > 0 ? out_features : 0) + 1;
}

extern "C" int recon_gcn_b16_bwd(const recon_gcn_b16_bwd_args* b, recon_stream_t stream) {
    if (!b) return RECON_ERR_INVALID;
    const recon_gcn_b16_args* a = &b->fwd;
    int rc = check(a);
    if (rc != RECON_OK) return rc;
    if (!b->grad_out || !b->g_support || !b->partial || !b->zeros || (b->ldg & 7) || (b->ldgx & 7)) return RECON_ERR_INVALID;
    if (a->B == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    if (a->node_ptr && (a->total_rows == 0 || a->total_rows > 0x7fffffffLL)) return a->total_rows == 0 ? RECON_OK : RECON_ERR_UNSUPPORTED;
    const int32_t rows = a->node_ptr ? static_cast<int32_t>(a->total_rows) : a->B * a->n, I = a->in_features, O = a->out_features, n = a->n;
    const uint16_t* gout = static_cast<const uint16_t*>(b->grad_out);
    const uint16_t* fout = static_cast<const uint16_t*>(a->out);
    // [B][O] per-graph column sums of gpre, behind the split-K partials (ragged batch: the partials of a product over total_rows rows)
    float* colsum = b->partial + (a->node_ptr ? gcn_b16_gw_partial_floats(1, rows, I, O) : gcn_b16_gw_partial_floats(a->B, n, I, O));
    const bool fused_bwd_off = cfg_char(CFG_GCN_FUSED_BWD) == '0';
    const int64_t i8f = (I + 7) / 8 * 8;
    const bool fused_bwd = !fused_bwd_off && !a->node_ptr && !b->g_adj && b->g_x && n <= 32 && a->lds <= kFusedNT * 16 && i8f <= kFusedNT * 16 && b->ldgx >= i8f &&
                           (b->ldgx & 3) == 0 && (a->lds & 3) == 0 && (b->ldg & 7) == 0 && (a->ldo & 7) == 0 &&
                           ((reinterpret_cast<uintptr_t>(b->g_x) | reinterpret_cast<uintptr_t>(b->grad_out) | reinterpret_cast<uintptr_t>(a->out)) & 15) == 0 &&
                           static_cast<int64_t>(rows) * (a->lds > b->ldgx ? a->lds : b->ldgx) * 2 < 0x7fffffffLL &&
                           static_cast<int64_t>(rows) * (b->ldg > a->ldo ? b->ldg : a->ldo) * 2 < 0x7fffffffLL;
    if (fused_bwd) {
        GcnFusedBwdK k;
        k.gout = gout; k.ldg = b->ldg; k.fout = fout; k.ldf = a->ldo; k.adj = static_cast<const uint16_t*>(a->adj);
        k.wn = reinterpret_cast<const uint16_t*>(static_cast<const char*>(a->w_planes) + planes_part(O, I));
        k.gsup = static_cast<uint16_t*>(b->g_support); k.lds = a->lds; k.gx = static_cast<uint16_t*>(b->g_x); k.ldgx = b->ldgx;
        k.colsum = b->g_bias ? colsum : nullptr;
        k.B = a->B; k.n = n; k.I = I; k.O = O; k.Op = b16_kp(O); k.nt = static_cast<int32_t>(ceil_div64(i8f, 16));
        constexpr int NSB = 4;
        const size_t fl = 2ull * kFusedNT * 16 * 64 + 4ull * 32 * kRSG + 4ull * 32 * kRSA;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gcn_b16_fused_bwd<NSB>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(fl));
        hipLaunchKernelGGL(k_gcn_b16_fused_bwd<NSB>, dim3(static_cast<unsigned>(ceil_div64(a->B, 4))), dim3(256 * NSB), fl, st, k);
        RECON_CHECK_LAUNCH();
    } else {
    // g_support = adj^T @ (grad_out * (out > 0)); pad columns zeroed (it is the A operand of the next product)
    dim3 grid(static_cast<unsigned>(ceil_div64(a->lds, 64)), static_cast<unsigned>(a->B), static_cast<unsigned>(ceil_div64(n, 32)));
    hipLaunchKernelGGL((k_gcn_b16_aggregate<true, true, false>), grid, dim3(256), 0, st, static_cast<const uint16_t*>(a->adj), gout, b->ldg, fout, a->ldo,
                       nullptr, n, O, static_cast<uint16_t*>(b->g_support), a->lds, b->g_bias ? colsum : nullptr, a->node_ptr, a->adj_ptr);
    }
    if (b->g_adj)
        hipLaunchKernelGGL(k_gcn_b16_grad_adj, dim3(static_cast<unsigned>(ceil_div64(n * n, 256)), static_cast<unsigned>(a->B)), dim3(256), 0, st, gout,
                           b->ldg, fout, a->ldo, static_cast<const uint16_t*>(a->support), a->lds, n, O, static_cast<uint16_t*>(b->g_adj), a->node_ptr,
                           a->adj_ptr);
    RECON_CHECK_LAUNCH();
    // g_bias: second pass over the per-graph column sums — in the launch of g_W's second pass when there is one
    const B16ReduceJob bias_job{colsum, static_cast<uint16_t*>(b->g_bias), O, a->B, 1, O};
    // g_x = g_support @ W^T : B operand = W [I][kp(O)] (k = out contiguous)
    if (b->g_x && !fused_bwd) {
        const int64_t i8 = (I + 7) / 8 * 8;
        if (b->ldgx < i8 || (reinterpret_cast<uintptr_t>(b->g_x) & 15)) return RECON_ERR_INVALID;
        rc = gemm_b16(rows, I, O, b->g_support, a->lds, static_cast<const char*>(a->w_planes) + planes_part(O, I), b->g_x, b->ldgx, true, st);
        if (rc != RECON_OK) return rc;                                // (pad columns of g_x stay unwritten: every reader of a gradient stops at the feature count)
    }
    // g_W = x^T @ g_support   (k-major, split-K over the B*n rows, fixed-order second pass)
    if (b->g_weight) {
        rc = gemm_b16_kmajor(I, O, rows, a->x, a->ldx, b->g_support, a->lds, b->g_weight, O, b->partial, b->zeros, st, b->g_bias ? &bias_job : nullptr);
        if (rc != RECON_OK) return rc;
    } else if (b->g_bias) {
        hipLaunchKernelGGL(k_gcn_b16_bias_reduce, dim3(static_cast<unsigned>(ceil_div64(O, 16))), dim3(1024), 0, st, colsum, a->B, O,
                           static_cast<uint16_t*>(b->g_bias));
        RECON_CHECK_LAUNCH();
    }
    return RECON_OK;
}

// four column parts per graph, one graph per wave (16 waves); RECON_GCN_STACK_GPW=2: two graphs per wave (8 waves; measured 36 us against 34 at cfg 3a)
static void launch_stack_fwd(const GcnStackK& k, size_t lds, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(ceil_div64(k.B, 4)));
    if (cfg_int(CFG_GCN_STACK_GPW, 1) == 5) {                           // experiment: five column parts x two graph slots = ten waves, 80 KiB of fragment reads per K step
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gcn_b16_stack_fwd<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((k_gcn_b16_stack_fwd<5, 2>), grid, dim3(640), lds, st, k);
    } else if (cfg_int(CFG_GCN_STACK_GPW, 1) != 2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gcn_b16_stack_fwd<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((k_gcn_b16_stack_fwd<4, 1>), grid, dim3(1024), lds, st, k);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gcn_b16_stack_fwd<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((k_gcn_b16_stack_fwd<4, 2>), grid, dim3(512), lds, st, k);
    }
}

extern "C" int recon_gcn_b16_stack_fwd(const recon_gcn_b16_stack_args* a, recon_stream_t stream) {
    if (!a || a->B < 0 || a->n <= 0 || a->in_features <= 0 || a->hidden <= 0 || a->L < 1 || a->L > kMaxStack) return RECON_ERR_INVALID;
    if (!a->x || !a->adj || !a->out || !a->w_planes) return RECON_ERR_INVALID;
    for (int l = 0; l < a->L; ++l) if (!a->w_planes[l]) return RECON_ERR_INVALID;
    const int64_t o8 = (a->hidden + 7) / 8 * 8;
    if ((a->ldx & 1) || a->ldx < a->in_features || (a->ldo & 7) || a->ldo < o8) return RECON_ERR_INVALID;
    if (a->in_features > 16 * kFusedNT * 2) return RECON_ERR_UNSUPPORTED;
    // the shapes of the fused single-layer forward, with 8-byte adjacency / bias loads: n % 4 == 0, hidden % 4 == 0
    if (a->n > 32 || (a->n & 3) || a->ldo > kFusedNT * 16 || (a->hidden & 3) || a->B > 65535 * 4) return RECON_ERR_UNSUPPORTED;
    if (static_cast<int64_t>(a->B) * a->n * a->ldx * 2 >= 0x7fffffffLL || static_cast<int64_t>(a->B) * a->n * a->ldo * 2 >= 0x7fffffffLL) return RECON_ERR_UNSUPPORTED;
    uintptr_t al = reinterpret_cast<uintptr_t>(a->x) | reinterpret_cast<uintptr_t>(a->out);
    for (int l = 0; l < a->L; ++l) al |= reinterpret_cast<uintptr_t>(a->w_planes[l]);
    if ((al & 15) || (reinterpret_cast<uintptr_t>(a->adj) & 7)) return RECON_ERR_UNSUPPORTED;
    for (int l = 0; l < a->L; ++l) if (a->bias && a->bias[l] && (reinterpret_cast<uintptr_t>(a->bias[l]) & 7)) return RECON_ERR_UNSUPPORTED;
    if (a->B == 0) return RECON_OK;
    GcnStackK k{};
    k.x = static_cast<const uint16_t*>(a->x); k.ldx = a->ldx; k.adj = static_cast<const uint16_t*>(a->adj);
    for (int l = 0; l < a->L; ++l) {
        k.wt[l] = static_cast<const uint16_t*>(a->w_planes[l]);
        k.bias[l] = (a->bias && a->bias[l]) ? static_cast<const uint16_t*>(a->bias[l]) : nullptr;
    }
    k.out = static_cast<uint16_t*>(a->out); k.ldo = a->ldo;
    k.B = a->B; k.n = a->n; k.I0 = a->in_features; k.D = a->hidden; k.L = a->L; k.nt = static_cast<int32_t>(ceil_div64(a->ldo, 16));
    const int64_t nki = ((a->in_features > a->hidden ? a->in_features : a->hidden) + 31) / 32;
    const size_t lds = 3ull * kFusedNT * 16 * 64 + static_cast<size_t>(kMaxStack) * kFusedNT * 32 + 4ull * nki * 2048 + 1024;
    if (lds > 160 * 1024) return RECON_ERR_UNSUPPORTED;                 // in_features > 384: layer by layer
    // column parts per graph: 4 (sixteen waves per CU; nothing but accumulators and fragments lives in registers)
    const int ns = cfg_int(CFG_GCN_STACK_PARTS, 4) == 2 ? 2 : 4;
    if (ns == 4) {
        launch_stack_fwd(k, lds, as_stream(stream));
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gcn_b16_stack_fwd<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL((k_gcn_b16_stack_fwd<2, 1>), dim3(static_cast<unsigned>(ceil_div64(a->B, 4))), dim3(512), lds, as_stream(stream), k);
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

// ---- the stack with gradients: forward = the kernel above keeping every layer's result, backward = k_gcn_b16_stack_bwd + one k-major GEMM per layer
static int stack_train_check(const recon_gcn_b16_stack_train_args* a, bool bwd) {
    if (!a || a->B < 0 || a->n <= 0 || a->in_features <= 0 || a->hidden <= 0 || a->L < 1 || a->L > kMaxStack) return RECON_ERR_INVALID;
    if (!a->x || !a->adj || !a->weight || !a->planes || !a->acts) return RECON_ERR_INVALID;
    const int64_t o8 = (a->hidden + 7) / 8 * 8, i8 = (a->in_features + 7) / 8 * 8;
    if ((a->ldo & 7) || a->ldo < o8) return RECON_ERR_INVALID;
    // x either has rows the weight-gradient GEMM can read (stride % 8 == 0), or the forward leaves a copy with such rows in x_rows
    if (a->x_rows ? ((a->ldx & 1) || a->ldx < a->in_features || (a->ldxr & 7) || a->ldxr < i8) : ((a->ldx & 7) || a->ldx < i8)) return RECON_ERR_INVALID;
    uintptr_t al = reinterpret_cast<uintptr_t>(a->x) | reinterpret_cast<uintptr_t>(a->x_rows);
    if (a->x_rows && static_cast<int64_t>(a->B) * a->n * a->ldxr * 2 >= 0x7fffffffLL) return RECON_ERR_UNSUPPORTED;
    for (int l = 0; l < a->L; ++l) {
        if (!a->weight[l] || !a->planes[l] || !a->acts[l]) return RECON_ERR_INVALID;
        al |= reinterpret_cast<uintptr_t>(a->planes[l]) | reinterpret_cast<uintptr_t>(a->acts[l]);
        if (a->bias && a->bias[l] && (reinterpret_cast<uintptr_t>(a->bias[l]) & 7)) return RECON_ERR_UNSUPPORTED;
    }
    if (a->n > 32 || (a->n & 3) || a->ldo > kFusedNT * 16 || (a->hidden & 3) || a->B > 65535 * 4) return RECON_ERR_UNSUPPORTED;
    if (a->in_features > 16 * kFusedNT) return RECON_ERR_UNSUPPORTED;                 // g_x of layer 0 is a tile of at most 320 columns
    if (static_cast<int64_t>(a->B) * a->n * a->ldx * 2 >= 0x7fffffffLL || static_cast<int64_t>(a->B) * a->n * a->ldo * 2 >= 0x7fffffffLL) return RECON_ERR_UNSUPPORTED;
    if ((al & 15) || (reinterpret_cast<uintptr_t>(a->adj) & 7)) return RECON_ERR_UNSUPPORTED;
    if (bwd) {
        // grad_out is read in place through masked 16-byte loads: any even row stride >= hidden, 4-byte aligned
        if (!a->grad_out || !a->g_support || !a->partial || !a->zeros || (a->ldg & 1) || a->ldg < a->hidden) return RECON_ERR_INVALID;
        // g_x leaves the kernel as 8-byte stores of four features: any row stride % 4 == 0 >= in_features rounded up to 4, 8-byte aligned
        if (a->g_x && ((a->ldgx & 3) || a->ldgx < (a->in_features + 3) / 4 * 4)) return RECON_ERR_INVALID;
        if ((reinterpret_cast<uintptr_t>(a->grad_out) & 3) || (reinterpret_cast<uintptr_t>(a->g_x) & 7)) return RECON_ERR_UNSUPPORTED;
        uintptr_t bl = reinterpret_cast<uintptr_t>(a->zeros);
        for (int l = 0; l < a->L; ++l) {
            if (!a->g_support[l]) return RECON_ERR_INVALID;
            bl |= reinterpret_cast<uintptr_t>(a->g_support[l]);
        }
        if (bl & 15) return RECON_ERR_UNSUPPORTED;
        if (static_cast<int64_t>(a->B) * a->n * a->ldg * 2 >= 0x7fffffffLL || (a->g_x && static_cast<int64_t>(a->B) * a->n * a->ldgx * 2 >= 0x7fffffffLL)) return RECON_ERR_UNSUPPORTED;
    }
    return RECON_OK;
}

extern "C" int recon_gcn_b16_stack_train_fwd(const recon_gcn_b16_stack_train_args* a, recon_stream_t stream) {
    int rc = stack_train_check(a, false);
    if (rc != RECON_OK) return rc;
    if (a->B == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    const int32_t D = a->hidden;
    GcnStackK k{};
    k.x = static_cast<const uint16_t*>(a->x); k.ldx = a->ldx; k.adj = static_cast<const uint16_t*>(a->adj);
    // every layer's W^T [D][kp(I)] (this pass) and W [I][kp(D)] (the backward) in one launch
    int32_t pm[kMaxStack], pn[kMaxStack];
    void* pt[kMaxStack]; void* pnn[kMaxStack];
    for (int l = 0; l < a->L; ++l) {
        pm[l] = l == 0 ? a->in_features : D; pn[l] = D;
        pt[l] = a->planes[l]; pnn[l] = static_cast<char*>(a->planes[l]) + planes_part(D, pm[l]);
    }
    rc = b16_pad_planes_both_multi(a->L, a->weight, pm, pn, pt, pnn, st);
    if (rc != RECON_OK) return rc;
    for (int l = 0; l < a->L; ++l) {
        char* wp = static_cast<char*>(a->planes[l]);
        k.wt[l] = reinterpret_cast<const uint16_t*>(wp);
        k.bias[l] = (a->bias && a->bias[l]) ? static_cast<const uint16_t*>(a->bias[l]) : nullptr;
        k.save[l] = l < a->L - 1 ? static_cast<uint16_t*>(a->acts[l]) : nullptr;
    }
    k.out = static_cast<uint16_t*>(a->acts[a->L - 1]); k.ldo = a->ldo;
    k.xcopy = static_cast<uint16_t*>(a->x_rows); k.ldxc = a->ldxr;
    k.B = a->B; k.n = a->n; k.I0 = a->in_features; k.D = D; k.L = a->L; k.nt = static_cast<int32_t>(ceil_div64(a->ldo, 16));
    const int64_t nki = ((a->in_features > D ? a->in_features : D) + 31) / 32;
    const size_t lds = 3ull * kFusedNT * 16 * 64 + static_cast<size_t>(kMaxStack) * kFusedNT * 32 + 4ull * nki * 2048 + 1024;
    if (lds > 160 * 1024) return RECON_ERR_UNSUPPORTED;
    launch_stack_fwd(k, lds, st);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

// floats of the split-K partials of the stack's weight gradients: [layer 0: sk x in x hidden][layers 1 ..: sk x hidden x hidden each]
// (sized for the largest split count any subset of the layers' products may run with: a frozen layer drops out of the launch)
static int stack_gw_splits(int32_t B, int32_t n, int32_t in_features, int32_t hidden, int32_t L) {
    int sk = 1;
    for (int32_t Mx : {in_features > hidden ? in_features : hidden, hidden})
        for (int c = 1; c <= L; ++c) {
            const int v = c == 1 ? b16_kmajor_splits(Mx, hidden, B * n) : b16_kmajor_splits_multi(Mx, hidden, B * n, c);
            sk = v > sk ? v : sk;
        }
    return sk;
}
static size_t stack_gw_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t hidden, int32_t L) {
    const size_t sk = static_cast<size_t>(stack_gw_splits(B, n, in_features, hidden, L));
    return sk * in_features * hidden + static_cast<size_t>(L - 1) * sk * hidden * hidden;
}

extern "C" size_t recon_gcn_b16_stack_bwd_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t hidden, int32_t L) {
    if (B <= 0 || n <= 0 || in_features <= 0 || hidden <= 0 || L <= 0) return 1;
    // [split-K partials of every layer's g_W][L x per-graph column sums for the g_bias]
    return stack_gw_partial_floats(B, n, in_features, hidden, L) + static_cast<size_t>(L) * B * hidden + 1;
}

extern "C" int recon_gcn_b16_stack_train_bwd(const recon_gcn_b16_stack_train_args* a, recon_stream_t stream) {
    int rc = stack_train_check(a, true);
    if (rc != RECON_OK) return rc;
    if (a->B == 0) return RECON_OK;
    hipStream_t st = as_stream(stream);
    const int32_t D = a->hidden, rows = a->B * a->n;
    float* colsum = a->partial + stack_gw_partial_floats(a->B, a->n, a->in_features, D, a->L);
    GcnStackBwdK k{};
    k.gout = static_cast<const uint16_t*>(a->grad_out); k.ldg = a->ldg; k.lda = a->ldo; k.adj = static_cast<const uint16_t*>(a->adj);
    k.lds = a->ldo; k.gx = static_cast<uint16_t*>(a->g_x); k.ldgx = a->ldgx;
    k.B = a->B; k.n = a->n; k.I0 = a->in_features; k.D = D; k.Op = b16_kp(D); k.L = a->L;
    for (int l = 0; l < a->L; ++l) {
        const int32_t I = l == 0 ? a->in_features : D;
        k.act[l] = static_cast<const uint16_t*>(a->acts[l]);
        k.wn[l] = reinterpret_cast<const uint16_t*>(static_cast<const char*>(a->planes[l]) + planes_part(D, I));
        k.gsup[l] = static_cast<uint16_t*>(a->g_support[l]);
        k.colsum[l] = (a->g_bias && a->g_bias[l]) ? colsum + static_cast<size_t>(l) * a->B * D : nullptr;
    }
    const size_t lds = 3ull * kFusedNT * 16 * 64 + 4ull * 32 * kRSG + 4ull * 32 * kRSA + 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gcn_b16_stack_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(k_gcn_b16_stack_bwd, dim3(static_cast<unsigned>(ceil_div64(a->B, 4))), dim3(1024), lds, st, k);
    RECON_CHECK_LAUNCH();
    // g_W_l = x_l^T @ g_support_l for every layer in ONE k-major launch (split-K over the B*n rows) + one fixed-order second pass that also
    // finishes the bias gradients
    B16KmProduct pr[kMaxStack];
    B16ReduceJob bj[kMaxStack];
    int np = 0, nbj = 0;
    const size_t sk = static_cast<size_t>(stack_gw_splits(a->B, a->n, a->in_features, D, a->L));
    float* part = a->partial;
    for (int l = 0; l < a->L; ++l) {
        const int32_t I = l == 0 ? a->in_features : D;
        if (a->g_weight && a->g_weight[l])
            pr[np++] = B16KmProduct{l == 0 ? (a->x_rows ? a->x_rows : a->x) : a->acts[l - 1], l == 0 ? (a->x_rows ? a->ldxr : a->ldx) : a->ldo, a->g_support[l], a->ldo,
                                    a->g_weight[l], D, part, I, D};
        part += sk * I * D;
        if (a->g_bias && a->g_bias[l]) bj[nbj++] = B16ReduceJob{k.colsum[l], static_cast<uint16_t*>(a->g_bias[l]), D, a->B, 1, D};
    }
    if (np > 0) {
        rc = gemm_b16_kmajor_multi(np, pr, rows, a->zeros, st, nbj ? bj : nullptr, nbj);
        if (rc != RECON_OK) return rc;
    } else {
        for (int j = 0; j < nbj; ++j) {
            hipLaunchKernelGGL(k_gcn_b16_bias_reduce, dim3(static_cast<unsigned>(ceil_div64(D, 16))), dim3(1024), 0, st, bj[j].partial, a->B, D, bj[j].out);
            RECON_CHECK_LAUNCH();
        }
    }
    return RECON_OK;
}

extern "C" int recon_gcn_b16_transposed_planes(const void* weight, int32_t in_features, int32_t out_features, void* planes, recon_stream_t stream) {
    if (!weight || !planes || in_features <= 0 || out_features <= 0) return RECON_ERR_INVALID;
    return b16_pad_planes(weight, out_features, true, out_features, in_features, planes, as_stream(stream));     // W^T [out][kp(in)], zero padded along k
}
