// The TransE margin loss of the stage-A loop — GAT/main.py:344-376 (batch_gat_loss) with gat_loss_func = nn.MarginRankingLoss(margin):
//     pos = train_indices[:n_pos].repeat(reps, 1),  neg = train_indices[n_pos:]                     (reps = 2 * valid_invalid_ratio_gat)
//     x_p = entity[pos[:, 0]] + relation[pos[:, 1]] - entity[pos[:, 2]],  pos_norm = |x_p|_1       (the same for neg)
//     loss = mean_j max(0, pos_norm_j - neg_norm_j + margin)                                        (y = -1)
// As the reference writes it this is 6 gathers, 4 adds, 2 norms, the ranking loss (28 launches) and, backwards, 6 sort-based index
// gradients (~75 launches with recon_amd.gather_rows' segment sums, one CSR per gather).  Here:
//   recon_transe_margin_fwd   TWO launches: a wave per pair j reads its six rows, writes term_j and the segment keys of the backward
//                             (entity rows: pos heads | pos tails | neg heads | neg tails, relation rows: pos | neg); one workgroup
//                             adds the terms in index order (fixed order: deterministic) and divides by their number;
//   recon_transe_margin_bwd   ONE launch: the six gradient rows of every pair (+- w sign(x), w = g_loss / pairs where the term is
//                             active); the tables' gradients are then two fixed-order segment sums by key (recon_spmm_rowsum_fwd).
#include "recon_common.h"

namespace recon {
namespace {

struct TransE {
    const float* ent; const float* rel; const int64_t* tri;       // tri [T][3] = (head, relation, tail); T = n_pos (1 + reps)
    int64_t n_pos, pairs; int32_t D, vec4; float margin;
};

// x = h + r - t over the lane's columns; returns |x|_1 of the lane's part (the caller reduces), sign(x) optionally kept
template <bool KEEP>
__device__ __forceinline__ float row_l1(const TransE& p, int64_t t, int lane, float (&sg)[8]) {
    const int64_t h = p.tri[3 * t], r = p.tri[3 * t + 1], tl = p.tri[3 * t + 2];
    const float* eh = p.ent + h * p.D; const float* er = p.rel + r * p.D; const float* et = p.ent + tl * p.D;
    float s = 0.f;
    if (p.vec4) {                                                   // D % 4 == 0, 16-byte aligned tables, D <= 512: two float4 per lane
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = 4 * lane + 256 * u;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, d = a;
            if (c < p.D) { a = *reinterpret_cast<const float4*>(eh + c); b = *reinterpret_cast<const float4*>(er + c); d = *reinterpret_cast<const float4*>(et + c); }
            const float x[4] = {a.x + b.x - d.x, a.y + b.y - d.y, a.z + b.z - d.z, a.w + b.w - d.w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                s += fabsf(x[v]);
                if (KEEP) sg[4 * u + v] = x[v] > 0.f ? 1.f : (x[v] < 0.f ? -1.f : 0.f);
            }
        }
    } else {
        for (int c = lane; c < p.D; c += 64) s += fabsf(eh[c] + er[c] - et[c]);
    }
    return s;
}

__global__ void __launch_bounds__(256) k_transe_margin_fwd(const TransE p, float* __restrict__ terms, int64_t* __restrict__ ent_key,
                                                            int64_t* __restrict__ rel_key, int32_t* __restrict__ nan_word, int64_t key_ld, int64_t rel_off) {
    // key_ld: row stride of the key tensors (0: each its own, 4 P / 2 P); rel_off: added to the relation keys — one [2][6 P] tensor over both
    // tables (entity rows first, relation rows from rel_off): ONE sort and ONE segment sum for both table gradients (recon_transe_margin_fwd_keys)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 4 + w;
    float sg[8];
    if (j < p.pairs) {
        const int64_t tp = j % p.n_pos, tn = p.n_pos + j;
        const float pn = group_sum<64>(row_l1<false>(p, tp, lane, sg)), nn = group_sum<64>(row_l1<false>(p, tn, lane, sg));
        if (lane == 0) {
            // clamp_min of the reference propagates NaN (GAT/main.py:374 then asserts on the loss); fmaxf alone would return the 0
            const float v = pn - nn + p.margin;
            terms[j] = (v != v) ? v : fmaxf(0.f, v);
            if (nan_word && !(fabsf(v) <= 3.0e38f)) *nan_word = 1;
            const int64_t P = p.pairs;
            if (ent_key) {                                          // both rows of the [2][4 P] key tensor (row 1 is ignored by the row sum)
                const int64_t k[4] = {p.tri[3 * tp], p.tri[3 * tp + 2], p.tri[3 * tn], p.tri[3 * tn + 2]};
#pragma unroll
                for (int q = 0; q < 4; ++q) { ent_key[q * P + j] = k[q]; ent_key[(key_ld ? key_ld : 4 * P) + q * P + j] = k[q]; }
            }
            if (rel_key) {
                const int64_t k[2] = {p.tri[3 * tp + 1], p.tri[3 * tn + 1]};
#pragma unroll
                for (int q = 0; q < 2; ++q) { rel_key[q * P + j] = k[q] + rel_off; rel_key[(key_ld ? key_ld : 2 * P) + q * P + j] = k[q] + rel_off; }
            }
        }
    }
}

// loss = mean of the terms, added in index order by ONE workgroup (fixed order: deterministic).  A launch of its own: the "last workgroup
// reduces" form in the kernel above cost every one of its 2 128 workgroups a device-scope release fence (an L2 write-back each): 208 us
// for a kernel whose work takes 10.
__global__ void __launch_bounds__(256) k_transe_mean(const float* __restrict__ terms, int64_t pairs, float* __restrict__ loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < pairs; i += 256) s += terms[i];          // thread t: terms t, t + 256, ... in order
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / static_cast<float>(pairs);
}

// gradient rows: g_ent [4 P][D] in the key order of the forward (pos heads | pos tails | neg heads | neg tails), g_rel [2 P][D] (pos | neg)
__global__ void __launch_bounds__(256) k_transe_margin_bwd(const TransE p, const float* __restrict__ terms, const float* __restrict__ g_loss,
                                                            float* __restrict__ g_ent, float* __restrict__ g_rel) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 4 + w;
    if (j >= p.pairs) return;
    const int64_t tp = j % p.n_pos, tn = p.n_pos + j, P = p.pairs;
    const float wgt = terms[j] > 0.f ? g_loss[0] / static_cast<float>(P) : 0.f;
    float sp[8], sn[8];
    (void)row_l1<true>(p, tp, lane, sp);
    (void)row_l1<true>(p, tn, lane, sn);
    if (p.vec4) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c >= p.D) continue;
            const float4 gp = make_float4(wgt * sp[4 * u], wgt * sp[4 * u + 1], wgt * sp[4 * u + 2], wgt * sp[4 * u + 3]);
            const float4 gn = make_float4(-wgt * sn[4 * u], -wgt * sn[4 * u + 1], -wgt * sn[4 * u + 2], -wgt * sn[4 * u + 3]);
            const float4 mp = make_float4(-gp.x, -gp.y, -gp.z, -gp.w), mn = make_float4(-gn.x, -gn.y, -gn.z, -gn.w);
            *reinterpret_cast<float4*>(g_ent + (0 * P + j) * p.D + c) = gp;
            *reinterpret_cast<float4*>(g_ent + (1 * P + j) * p.D + c) = mp;
            *reinterpret_cast<float4*>(g_ent + (2 * P + j) * p.D + c) = gn;
            *reinterpret_cast<float4*>(g_ent + (3 * P + j) * p.D + c) = mn;
            *reinterpret_cast<float4*>(g_rel + (0 * P + j) * p.D + c) = gp;
            *reinterpret_cast<float4*>(g_rel + (1 * P + j) * p.D + c) = gn;
        }
    } else {
        const int64_t hp = p.tri[3 * tp], rp = p.tri[3 * tp + 1], tlp = p.tri[3 * tp + 2];
        const int64_t hn = p.tri[3 * tn], rn = p.tri[3 * tn + 1], tln = p.tri[3 * tn + 2];
        for (int c = lane; c < p.D; c += 64) {
            const float xp = p.ent[hp * p.D + c] + p.rel[rp * p.D + c] - p.ent[tlp * p.D + c];
            const float xn = p.ent[hn * p.D + c] + p.rel[rn * p.D + c] - p.ent[tln * p.D + c];
            const float gp = wgt * (xp > 0.f ? 1.f : (xp < 0.f ? -1.f : 0.f)), gn = -wgt * (xn > 0.f ? 1.f : (xn < 0.f ? -1.f : 0.f));
            g_ent[(0 * P + j) * p.D + c] = gp; g_ent[(1 * P + j) * p.D + c] = -gp;
            g_ent[(2 * P + j) * p.D + c] = gn; g_ent[(3 * P + j) * p.D + c] = -gn;
            g_rel[(0 * P + j) * p.D + c] = gp; g_rel[(1 * P + j) * p.D + c] = gn;
        }
    }
}

bool fill(TransE* p, const float* ent, const float* rel, const int64_t* tri, int64_t n_pos, int32_t reps, int32_t D, float margin) {
    if (!ent || !rel || !tri || n_pos <= 0 || reps <= 0 || D <= 0) return false;
    p->ent = ent; p->rel = rel; p->tri = tri; p->n_pos = n_pos; p->pairs = n_pos * reps; p->D = D; p->margin = margin;
    p->vec4 = (D % 4 == 0 && D <= 512 && !((reinterpret_cast<uintptr_t>(ent) | reinterpret_cast<uintptr_t>(rel)) & 15)) ? 1 : 0;
    return true;
}
}  // namespace
}  // namespace recon

extern "C" int recon_transe_margin_fwd(const float* entity, const float* relation, const int64_t* triples, int64_t n_pos, int32_t reps, int32_t D,
                                       float margin, float* terms, float* loss, int64_t* ent_key, int64_t* rel_key, uint32_t* counter,
                                       recon_stream_t stream) {
    recon::TransE p;
    if (n_pos == 0 && reps > 0 && D > 0) return RECON_ERR_INVALID;      // the mean of nothing (the reference returns nan here)
    if (!recon::fill(&p, entity, relation, triples, n_pos, reps, D, margin) || !terms || !loss) return RECON_ERR_INVALID;
    if (p.pairs > (1LL << 31) - 4) return RECON_ERR_UNSUPPORTED;
    (void)counter;
    hipLaunchKernelGGL(recon::k_transe_margin_fwd, dim3(static_cast<unsigned>(ceil_div64(p.pairs, 4))), dim3(256), 0, as_stream(stream), p, terms, ent_key, rel_key,
                       recon::nan_flag(), static_cast<int64_t>(0), static_cast<int64_t>(0));
    hipLaunchKernelGGL(recon::k_transe_mean, dim3(1), dim3(256), 0, as_stream(stream), terms, p.pairs, loss);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_transe_margin_fwd_keys(const float* entity, const float* relation, const int64_t* triples, int64_t n_pos, int32_t reps, int32_t D,
                                            float margin, float* terms, float* loss, int64_t* keys, int64_t n_entities, recon_stream_t stream) {
    recon::TransE p;
    if (n_pos == 0 && reps > 0 && D > 0) return RECON_ERR_INVALID;
    if (!recon::fill(&p, entity, relation, triples, n_pos, reps, D, margin) || !terms || !loss || !keys || n_entities <= 0) return RECON_ERR_INVALID;
    if (p.pairs > ((1LL << 31) - 4) / 6) return RECON_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(recon::k_transe_margin_fwd, dim3(static_cast<unsigned>(ceil_div64(p.pairs, 4))), dim3(256), 0, as_stream(stream), p, terms, keys,
                       keys + 4 * p.pairs, recon::nan_flag(), 6 * p.pairs, n_entities);
    hipLaunchKernelGGL(recon::k_transe_mean, dim3(1), dim3(256), 0, as_stream(stream), terms, p.pairs, loss);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_transe_margin_bwd(const float* entity, const float* relation, const int64_t* triples, int64_t n_pos, int32_t reps, int32_t D,
                                       const float* terms, const float* g_loss, float* g_ent_rows, float* g_rel_rows, recon_stream_t stream) {
    recon::TransE p;
    if (!recon::fill(&p, entity, relation, triples, n_pos, reps, D, 0.f) || !terms || !g_loss || !g_ent_rows || !g_rel_rows) return RECON_ERR_INVALID;
    if (p.vec4 && ((reinterpret_cast<uintptr_t>(g_ent_rows) | reinterpret_cast<uintptr_t>(g_rel_rows)) & 15)) return RECON_ERR_INVALID;
    hipLaunchKernelGGL(recon::k_transe_margin_bwd, dim3(static_cast<unsigned>(ceil_div64(p.pairs, 4))), dim3(256), 0, as_stream(stream), p, terms, g_loss,
                       g_ent_rows, g_rel_rows);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
