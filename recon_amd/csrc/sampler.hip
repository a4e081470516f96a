// N1 — the batch builders of the stage-A loop on the device, a fixed handful of launches per batch:
//   Corpus.get_batch_adj_data            GAT/create_batch.py:391-436   recon_kg_adj_count  + recon_kg_adj_fill
//   Corpus.get_batch_nhop_neighbors_all  :871-895 over the 2-hop neighbourhoods of bfs / get_further_neighbors (:788-869)
//                                                                       recon_kg_nhop (count pass, write pass)
// recon_amd/sampler.py used to assemble the same results from torch primitives (ragged aranges, sorts, searchsorted, uniques): 194 launches
// and several host round trips per batch — more than the model's forward.  The knowledge graph arrives as the grouped CSR that
// KGNeighbourSampler builds once: pairs (source, target) grouped by source in order of first appearance (pair_ptr [Ne + 1], pair_tgt [P],
// pair_first_rel [P], not_loop [P]), the relations of a pair contiguous in insertion order (rel_ptr [P + 1], rel_sorted [T]).
// Semantics (pinned by tests/golden/sampler*.npz, produced by the reference's Corpus): see recon_amd/sampler.py.
#include "recon_common.h"

namespace recon {
namespace {

struct KG {
    const int64_t* pair_ptr; const int64_t* pair_tgt; const int64_t* pair_first_rel; const uint8_t* not_loop;
    const int64_t* rel_ptr; const int64_t* rel_sorted; int64_t Ne;
};

// exclusive scan of one value per thread over a 1024-thread workgroup (16 waves); `carry` is the running total across calls
__device__ __forceinline__ int64_t block_scan_1024(int64_t v, int64_t* wsum, int64_t& carry) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int64_t o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int64_t base = 0, tot = 0;
    for (int i = 0; i < 16; ++i) { const int64_t s = wsum[i]; if (i < w) base += s; tot += s; }
    __syncthreads();
    const int64_t excl = carry + base + x - v;
    carry += tot;
    return excl;
}

// ---- 1-hop: per batch position b its number of edges (relations of its non-loop pairs); marks of the entities and targets seen.  The
// workgroup that arrives last scans the counts (edge offsets), compacts the marks into sorted unique id lists, and clears the marks.
__global__ void __launch_bounds__(1024) k_kg_adj_count(const KG g, const int64_t* __restrict__ ents, int32_t B, int64_t* __restrict__ nrel,
                                                        uint8_t* __restrict__ ent_mark, uint8_t* __restrict__ tgt_mark, int64_t* __restrict__ rel_off,
                                                        int64_t* __restrict__ uniq_ent, int64_t* __restrict__ uniq_tgt, int64_t* __restrict__ totals,
                                                        uint32_t* __restrict__ counter) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 16 + w;
    if (b < B) {
        const int64_t e = ents[b];
        int64_t nr = 0;
        for (int64_t p = g.pair_ptr[e] + lane; p < g.pair_ptr[e + 1]; p += 64)
            if (g.not_loop[p]) { nr += g.rel_ptr[p + 1] - g.rel_ptr[p]; tgt_mark[g.pair_tgt[p]] = 1; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nr += __shfl_xor(nr, off, 64);
        if (lane == 0) { nrel[b] = nr; ent_mark[e] = 1; }
    }
    __shared__ uint32_t ticket;
    __shared__ int64_t wsum[16];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(counter, 1u);
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    __threadfence();
    int64_t carry = 0;
    for (int i0 = 0; i0 < B; i0 += 1024) {                              // edge offsets
        const int i = i0 + threadIdx.x;
        const int64_t v = i < B ? nrel[i] : 0;
        const int64_t ex = block_scan_1024(v, wsum, carry);
        if (i < B) rel_off[i] = ex;
    }
    if (threadIdx.x == 0) totals[0] = carry;
    for (int which = 0; which < 2; ++which) {                           // sorted unique ids out of the marks
        uint8_t* mark = which ? tgt_mark : ent_mark;
        int64_t* out = which ? uniq_tgt : uniq_ent;
        carry = 0;
        for (int64_t i0 = 0; i0 < g.Ne; i0 += 1024) {
            const int64_t i = i0 + threadIdx.x;
            const int64_t v = (i < g.Ne && mark[i]) ? 1 : 0;
            const int64_t ex = block_scan_1024(v, wsum, carry);
            if (v) { out[ex] = i; mark[i] = 0; }
        }
        if (threadIdx.x == 0) totals[1 + which] = carry;
    }
    if (threadIdx.x == 0) *counter = 0u;
}

// edges of batch position b from rel_off[b] on: its non-loop pairs in order, a pair's relations in order
__global__ void __launch_bounds__(256) k_kg_adj_fill(const KG g, const int64_t* __restrict__ ents, int32_t B, const int64_t* __restrict__ rel_off,
                                                      int64_t E, int64_t* __restrict__ edge, int64_t* __restrict__ edge_type) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t e = ents[b], p0 = g.pair_ptr[e], p1 = g.pair_ptr[e + 1];
    int64_t o = rel_off[b];
    for (int64_t pc = p0; pc < p1; pc += 64) {
        const int64_t p = pc + lane;
        const bool ok = p < p1 && g.not_loop[p];
        const int64_t r0 = ok ? g.rel_ptr[p] : 0;
        const int cnt = ok ? static_cast<int>(g.rel_ptr[p + 1] - r0) : 0;
        int x = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(x, off, 64); if (lane >= off) x += t; }
        const int tot = __shfl(x, 63, 64);
        if (ok) {
            const int64_t tgt = g.pair_tgt[p];
            int64_t at = o + x - cnt;
            for (int r = 0; r < cnt; ++r, ++at) { edge[at] = tgt; edge[E + at] = e; edge_type[at] = g.rel_sorted[r0 + r]; }
        }
        o += tot;
    }
}

// ---- 2-hop: one wave per source with the visited set of its BFS as a bitmap in LDS (the source, its level-1 nodes, then every target as it
// is discovered).  Parents in pair order, a parent's neighbours in pair order (their targets are distinct): first visit wins, as in
// Corpus.bfs.  WRITE = false counts, true writes the quadruples (source, first relation source -> parent, first relation parent -> target,
// target) from quad_off[b] on; k_kg_scan turns the counts into offsets.
template <bool WRITE>
__global__ void __launch_bounds__(64) k_kg_nhop(const KG g, const int64_t* __restrict__ srcs, int32_t S, int32_t partial, int64_t* __restrict__ qcount,
                                                 const int64_t* __restrict__ quad_off, int64_t* __restrict__ quads) {
    extern __shared__ uint32_t seen[];
    const int lane = threadIdx.x, b = blockIdx.x;
    const int words = static_cast<int>((g.Ne + 63) >> 6) * 2;          // a multiple of two: the parent tables behind it stay 8-byte aligned
    const int64_t s = srcs[b], p0 = g.pair_ptr[s], p1 = g.pair_ptr[s + 1];
    for (int i = lane; i < words; i += 64) seen[i] = 0u;
    __syncthreads();
    if (lane == 0) atomicOr(&seen[s >> 5], 1u << (s & 31));
    for (int64_t p = p0 + lane; p < p1; p += 64)
        if (g.not_loop[p]) { const int64_t u = g.pair_tgt[p]; atomicOr(&seen[u >> 5], 1u << (u & 31)); }
    __syncthreads();
    int64_t count = 0;
    const int64_t base = WRITE ? quad_off[b] : 0;
    const int64_t limit = WRITE ? qcount[b] : (partial ? 1 : (1LL << 62));
    // Parents in batches of 64 (lane i requests parent i's target, first relation and pair range: two dependent round trips per batch), and the
    // batch's candidates — parent-major, pair order: the BFS's discovery order — as ONE flat sequence walked 64 at a time, whatever the
    // parents' degrees (one parent at a time, a source with a few hundred low-degree parents was a chain of as many round trips: 0.5 ms
    // per batch of 128 sources).  A parent's targets are distinct; the same target under two parents of one chunk is resolved by lane
    // order (equal-value peers through ballots: the lowest lane is the first visit).
    int32_t* poff = reinterpret_cast<int32_t*>(seen + words);           // [64] first candidate of parent i within the batch
    int64_t* pc0 = reinterpret_cast<int64_t*>(poff + 64);               // [64] its pair range's start
    int64_t* pr1 = pc0 + 64;                                            // [64] its first relation
    int vbits = 1;
    while (vbits < 63 && (1LL << vbits) < g.Ne) ++vbits;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t pb = p0; pb < p1 && count < limit; pb += 64) {
        const int64_t pa_l = pb + lane;
        const bool ok_l = pa_l < p1 && g.not_loop[pa_l];
        const int64_t u_l = ok_l ? g.pair_tgt[pa_l] : 0;
        const int64_t c0_l = ok_l ? g.pair_ptr[u_l] : 0, c1_l = ok_l ? g.pair_ptr[u_l + 1] : 0;
        const int deg = static_cast<int>(c1_l - c0_l);
        int x = deg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(x, off, 64); if (lane >= off) x += t; }
        const int T = __shfl(x, 63, 64);
        __syncthreads();                                                 // the previous batch's readers are done with the tables
        poff[lane] = x - deg; pc0[lane] = c0_l; pr1[lane] = ok_l ? g.pair_first_rel[pa_l] : 0;
        __syncthreads();
        for (int x0 = 0; x0 < T && count < limit; x0 += 64) {            // wave-uniform
            const int xi = x0 + lane;
            const bool valid = xi < T;
            int lo = 0;                                                  // the last parent whose first candidate is <= xi (parents without candidates share an offset with the next)
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) if (lo + step < 64 && poff[lo + step] <= xi) lo += step;
            const int64_t c = pc0[lo] + (xi - poff[lo]);
            const int64_t v = valid ? g.pair_tgt[c] : 0;
            const uint32_t bit = 1u << (v & 31);
            const bool unseen = valid && !(seen[v >> 5] & bit);
            unsigned long long peers = __ballot(unseen);                 // unseen lanes with the same target: the lowest one is the first visit
            for (int bb = 0; bb < vbits; ++bb) {
                const bool on = (v >> bb) & 1;
                const unsigned long long m = __ballot(unseen && on);
                peers &= on ? m : ~m;
            }
            const bool fresh = unseen && (peers & lt) == 0;
            if (fresh) atomicOr(&seen[v >> 5], bit);
            const uint64_t m = __ballot(fresh);
            const int before = __popcll(m & lt);
            if (WRITE && fresh && count + before < limit) {
                int64_t* q = quads + 4 * (base + count + before);
                q[0] = s; q[1] = pr1[lo]; q[2] = g.pair_first_rel[c]; q[3] = v;
            }
            count += __popcll(m);
        }
    }
    if (WRITE) return;
    if (count > limit) count = limit;
    if (lane == 0) qcount[b] = count;
}

// offsets of the sources' quadruples: exclusive scan of the counts by one wave (a launch of its own: "last workgroup scans" costs every
// workgroup a device-scope fence)
__global__ void __launch_bounds__(64) k_kg_scan(const int64_t* __restrict__ qcount, int32_t S, int64_t* __restrict__ quad_off, int64_t* __restrict__ total) {
    const int lane = threadIdx.x;
    int64_t carry = 0;
    for (int i0 = 0; i0 < S; i0 += 64) {
        const int i = i0 + lane;
        const int64_t v = i < S ? qcount[i] : 0;
        int64_t x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int64_t o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
        if (i < S) quad_off[i] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
    if (lane == 0) total[0] = carry;
}

bool kg_ok(const recon_kg* k) {
    return k && k->num_entities > 0 && k->pair_ptr && k->pair_tgt && k->pair_first_rel && k->not_loop && k->rel_ptr && k->rel_sorted;
}
KG kg_of(const recon_kg* k) {
    return KG{k->pair_ptr, k->pair_tgt, k->pair_first_rel, k->not_loop, k->rel_ptr, k->rel_sorted, k->num_entities};
}
}  // namespace
}  // namespace recon

// the visited bitmap of one source + the parent tables of a batch (64 x (4 + 8 + 8) bytes, behind an 8-byte boundary)
extern "C" size_t recon_kg_nhop_lds_bytes(int64_t num_entities) { return static_cast<size_t>((num_entities + 63) / 64) * 8 + 64 * 20; }

extern "C" int recon_kg_adj_count(const recon_kg* kg, const int64_t* entities, int32_t B, int64_t* nrel, uint8_t* ent_mark, uint8_t* tgt_mark,
                                  int64_t* rel_off, int64_t* uniq_ent, int64_t* uniq_tgt, int64_t* totals, uint32_t* counter, recon_stream_t stream) {
    if (!recon::kg_ok(kg) || B < 0) return RECON_ERR_INVALID;
    if (B == 0) return RECON_OK;
    if (!entities || !nrel || !ent_mark || !tgt_mark || !rel_off || !uniq_ent || !uniq_tgt || !totals || !counter) return RECON_ERR_INVALID;
    hipLaunchKernelGGL(recon::k_kg_adj_count, dim3(static_cast<unsigned>(ceil_div64(B, 16))), dim3(1024), 0, as_stream(stream), recon::kg_of(kg), entities, B,
                       nrel, ent_mark, tgt_mark, rel_off, uniq_ent, uniq_tgt, totals, counter);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_kg_adj_fill(const recon_kg* kg, const int64_t* entities, int32_t B, const int64_t* rel_off, int64_t E, int64_t* edge,
                                 int64_t* edge_type, recon_stream_t stream) {
    if (!recon::kg_ok(kg) || B < 0 || E < 0) return RECON_ERR_INVALID;
    if (B == 0 || E == 0) return RECON_OK;
    if (!entities || !rel_off || !edge || !edge_type) return RECON_ERR_INVALID;
    hipLaunchKernelGGL(recon::k_kg_adj_fill, dim3(static_cast<unsigned>(ceil_div64(B, 4))), dim3(256), 0, as_stream(stream), recon::kg_of(kg), entities, B, rel_off,
                       E, edge, edge_type);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_kg_nhop(const recon_kg* kg, const int64_t* sources, int32_t S, int32_t partial_2hop, int32_t write, int64_t* qcount, int64_t* quad_off,
                             int64_t* quads, int64_t* total, uint32_t* counter, recon_stream_t stream) {
    if (!recon::kg_ok(kg) || S < 0) return RECON_ERR_INVALID;
    if (S == 0) return RECON_OK;
    (void)counter;
    if (!sources || !qcount || !quad_off || (write ? !quads : !total)) return RECON_ERR_INVALID;
    const size_t lds = recon_kg_nhop_lds_bytes(kg->num_entities);
    if (lds > 160 * 1024 - 64) return RECON_ERR_UNSUPPORTED;             // the visited set of one source must fit a CU's LDS (5.2 M entities)
    const void* kern = write ? reinterpret_cast<const void*>(recon::k_kg_nhop<true>) : reinterpret_cast<const void*>(recon::k_kg_nhop<false>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return RECON_ERR_LAUNCH;
    if (write) hipLaunchKernelGGL(recon::k_kg_nhop<true>, dim3(static_cast<unsigned>(S)), dim3(64), lds, as_stream(stream), recon::kg_of(kg), sources, S, partial_2hop,
                                  qcount, quad_off, quads);
    else {
        hipLaunchKernelGGL(recon::k_kg_nhop<false>, dim3(static_cast<unsigned>(S)), dim3(64), lds, as_stream(stream), recon::kg_of(kg), sources, S, partial_2hop,
                           qcount, quad_off, quads);
        hipLaunchKernelGGL(recon::k_kg_scan, dim3(1), dim3(64), 0, as_stream(stream), qcount, S, quad_off, total);
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
