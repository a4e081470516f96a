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
    // sorted unique ids out of the marks: 16 marks per thread and pass (one 16-byte load where the array allows it), so that an FB15k-sized
    // entity set is ONE scan per array — a mark per thread and pass was 15 dependent passes of a single workgroup per array, 45 of the
    // launch's 59 us
    __shared__ int32_t wsum32[16];
    for (int which = 0; which < 2; ++which) {
        uint8_t* mark = which ? tgt_mark : ent_mark;
        int64_t* out = which ? uniq_tgt : uniq_ent;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(mark) & 15) == 0;
        int32_t carry32 = 0;
        for (int64_t i0 = 0; i0 < g.Ne; i0 += 16 * 1024) {
            const int64_t i = i0 + 16 * static_cast<int64_t>(threadIdx.x);
            const bool full = vec_ok && i + 16 <= g.Ne;
            uint32_t wd[4] = {0u, 0u, 0u, 0u};
            if (full) { const uint4 q = *reinterpret_cast<const uint4*>(mark + i); wd[0] = q.x; wd[1] = q.y; wd[2] = q.z; wd[3] = q.w; }
            else {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (i + k < g.Ne && mark[i + k]) wd[k >> 2] |= 1u << (8 * (k & 3));
            }
            int32_t cnt = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) cnt += ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu) != 0u;
            // exclusive scan of cnt over the workgroup
            int32_t x = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
            if (lane == 63) wsum32[w] = x;
            __syncthreads();
            int32_t base = 0, tot = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const int32_t sj = wsum32[j]; if (j < w) base += sj; tot += sj; }
            __syncthreads();
            int64_t at = carry32 + base + x - cnt;
            carry32 += tot;
            if (cnt) {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if ((wd[k >> 2] >> (8 * (k & 3))) & 0xffu) { out[at++] = i + k; if (!full) mark[i + k] = 0; }
                if (full) *reinterpret_cast<uint4*>(mark + i) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (threadIdx.x == 0) totals[1 + which] = carry32;
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
                                                 const int64_t* __restrict__ quad_off, int64_t* __restrict__ quads, const int64_t* __restrict__ s_dev = nullptr) {
    extern __shared__ uint32_t seen[];
    const int lane = threadIdx.x, b = blockIdx.x;
    // s_dev (count pass launched BEFORE the host knows how many sources there are — recon_kg_nhop_count_early): the grid covers the most there
    // can be, the sources past the count on the device have none
    if (s_dev && b >= *s_dev) { if (lane == 0) qcount[b] = 0; return; }
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
    // The walk of one source is a chain: a chunk's candidates are looked up in `seen` after the chunk before it has marked its own.  What the
    // chain does NOT order are the loads: where a candidate lies (parent tables) and what it is (its target, its first relation) depend on the
    // knowledge graph alone.  So the targets of the next kChunks chunks are requested before a chunk is resolved, and the next 64 parents'
    // pair ranges before a batch's candidates are walked — a long source (a hub among its parents: tens of thousands of candidates) was one
    // exposed round trip per 64 candidates, 0.78 ms for the slowest of 128 sources.
    constexpr int kChunks = 4;
    // parents two stages ahead: stage 1 (a parent's target, loop mark, first relation) for the batch after the next, stage 2 (the target's
    // pair range — needs stage 1's answer as its address) for the next; raw values only, combined where they are used, so that nothing in
    // front of a batch's walk waits for them.  Lanes past the source's last pair re-read its first pair and count as loops.
    struct Stage1 { bool in; uint8_t nl; int64_t u, r1; };
    struct Stage2 { int64_t c0, c1; };
    auto request1 = [&](int64_t pb) {
        Stage1 r;
        const int64_t pa_l = pb + lane;
        r.in = pa_l < p1;
        const int64_t pa_c = r.in ? pa_l : p0;
        r.nl = g.not_loop[pa_c]; r.u = g.pair_tgt[pa_c]; r.r1 = g.pair_first_rel[pa_c];
        return r;
    };
    auto request2 = [&](const Stage1& a) { return Stage2{g.pair_ptr[a.u], g.pair_ptr[a.u + 1]}; };
    Stage1 a1{false, 0, 0, 0}, b1{false, 0, 0, 0};
    Stage2 a2{0, 0};
    if (p0 < p1) { a1 = request1(p0); b1 = request1(p0 + 64); a2 = request2(a1); }
    for (int64_t pb = p0; pb < p1 && count < limit; pb += 64) {
        const bool ok = a1.in && a1.nl;
        const int deg = ok ? static_cast<int>(a2.c1 - a2.c0) : 0;
        int x = deg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(x, off, 64); if (lane >= off) x += t; }
        const int T = __shfl(x, 63, 64);
        __syncthreads();                                                 // the previous batch's readers are done with the tables
        poff[lane] = x - deg; pc0[lane] = ok ? a2.c0 : 0; pr1[lane] = a1.r1;
        __syncthreads();
        a1 = b1; a2 = request2(b1); b1 = request1(pb + 128);            // under this batch's walk
        int64_t v_r[kChunks], f_r[kChunks]; int lo_r[kChunks];
        auto request = [&](int slot, int x0) {                           // chunks past T: every lane invalid, reads candidate 0 of parent 0's range
            const int xi = x0 + lane;
            const bool valid = xi < T;
            // the last parent whose first candidate is <= xi (parents without candidates share an offset with the next): a 4-ary search over
            // the non-decreasing offsets, three dependent LDS round trips instead of a binary search's six
            int lo;
            { const int a = poff[16], b = poff[32], c = poff[48]; lo = c <= xi ? 48 : (b <= xi ? 32 : (a <= xi ? 16 : 0)); }
            { const int a = poff[lo + 4], b = poff[lo + 8], c = poff[lo + 12]; lo += c <= xi ? 12 : (b <= xi ? 8 : (a <= xi ? 4 : 0)); }
            { const int a = poff[lo + 1], b = poff[lo + 2], c = poff[lo + 3]; lo += c <= xi ? 3 : (b <= xi ? 2 : (a <= xi ? 1 : 0)); }
            const int64_t c = valid ? pc0[lo] + (xi - poff[lo]) : 0;
            v_r[slot] = g.pair_tgt[c]; f_r[slot] = g.pair_first_rel[c]; lo_r[slot] = lo;
        };
#pragma unroll
        for (int d = 0; d < kChunks; ++d) request(d, 64 * d);
        for (int x0 = 0; x0 < T && count < limit; x0 += 64 * kChunks) {  // wave-uniform
#pragma unroll
            for (int d = 0; d < kChunks; ++d) {
                __builtin_amdgcn_sched_barrier(0);                       // a chunk's wait stays in front of that chunk
                const int xi = x0 + 64 * d + lane;
                const bool valid = xi < T && count < limit;
                const int64_t v = valid ? v_r[d] : 0, fr = f_r[d];
                const int lo = lo_r[d];
                const uint32_t bit = 1u << (v & 31);
                const bool unseen = valid && !(seen[v >> 5] & bit);
                // mark; the lane whose atomic found the bit clear is A first visit of its target.  It is THE first visit (the lowest lane, the
                // BFS's order) unless two unseen lanes of this chunk carry the same target — then fewer lanes won than were unseen, and the
                // equal-value peers are found through ballots over the value's bits (rare: a target is unseen once)
                bool fresh = unseen && !(atomicOr(&seen[v >> 5], unseen ? bit : 0u) & bit);
                const unsigned long long un = __ballot(unseen);
                if (__popcll(un) != __popcll(__ballot(fresh))) {         // wave-uniform
                    unsigned long long peers = un;
                    for (int bb = 0; bb < vbits; ++bb) {
                        const bool on = (v >> bb) & 1;
                        const unsigned long long m = __ballot(unseen && on);
                        peers &= on ? m : ~m;
                    }
                    fresh = unseen && (peers & lt) == 0;
                }
                const uint64_t m = __ballot(fresh);
                const int before = __popcll(m & lt);
                if (WRITE && fresh && count + before < limit) {
                    int64_t* q = quads + 4 * (base + count + before);
                    q[0] = s; q[1] = pr1[lo]; q[2] = fr; q[3] = v;
                }
                count += __popcll(m);
                request(d, x0 + 64 * (d + kChunks));                     // behind the slot's last use: no copies of the ring (they would wait for the loads)
            }
        }
    }
    if (WRITE) return;
    if (count > limit) count = limit;
    if (lane == 0) qcount[b] = count;
}

// The same walk by a WORKGROUP of four waves per source (the default where the tables fit in LDS): a round resolves 256 candidates, and
// the slowest source of a batch — what a launch lasts — needs a quarter of the rounds.  First visit = the LOWEST candidate index that meets
// an unseen target: every unseen lane posts its index with an LDS atomicMin into own[target] (one word per entity, all-ones at the
// start; a word is consulted only in the round its target is first met, so it never needs resetting), and behind a barrier the lane
// that reads its own index back is the first visit.  Positions: per-wave ballots + the waves' counts through LDS.
constexpr int kNhopWaves = 4;
template <bool WRITE>
__global__ void __launch_bounds__(64 * kNhopWaves) k_kg_nhop_wg(const KG g, const int64_t* __restrict__ srcs, int32_t S, int32_t partial,
                                                                 int64_t* __restrict__ qcount, const int64_t* __restrict__ quad_off, int64_t* __restrict__ quads,
                                                                 const int64_t* __restrict__ s_dev = nullptr) {
    extern __shared__ uint32_t seen[];
    constexpr int NTH = 64 * kNhopWaves;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, b = blockIdx.x;
    if (s_dev && b >= *s_dev) { if (t == 0) qcount[b] = 0; return; }  // (see k_kg_nhop)
    const int words = static_cast<int>((g.Ne + 63) >> 6) * 2;
    const int ne2 = static_cast<int>((g.Ne + 1) & ~1LL);               // own[] padded to an even count: the tables behind it stay 8-byte aligned
    uint32_t* own = seen + words;
    int32_t* poff = reinterpret_cast<int32_t*>(own + ne2);               // [64] first candidate of parent i within the batch
    int64_t* pc0 = reinterpret_cast<int64_t*>(poff + 64);               // [64] its pair range's start
    int64_t* pr1 = pc0 + 64;                                            // [64] its first relation
    int32_t* wsum = reinterpret_cast<int32_t*>(pr1 + 64);               // [kNhopWaves] first visits per wave of the current round
    const int64_t s = srcs[b], p0 = g.pair_ptr[s], p1 = g.pair_ptr[s + 1];
    for (int i = t; i < words; i += NTH) seen[i] = 0u;
    for (int i = t; i < ne2; i += NTH) own[i] = 0xffffffffu;
    __syncthreads();
    if (t == 0) atomicOr(&seen[s >> 5], 1u << (s & 31));
    for (int64_t p = p0 + t; p < p1; p += NTH)
        if (g.not_loop[p]) { const int64_t u = g.pair_tgt[p]; atomicOr(&seen[u >> 5], 1u << (u & 31)); }
    __syncthreads();
    int64_t count = 0;                                                  // the same in every thread
    const int64_t base = WRITE ? quad_off[b] : 0;
    const int64_t limit = WRITE ? qcount[b] : (partial ? 1 : (1LL << 62));
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    constexpr int kChunks = 4;
    // parents two stages ahead, as in k_kg_nhop; every wave requests the same 64 parents (lane i: parent i) and scans their degrees itself
    struct Stage1 { bool in; uint8_t nl; int64_t u, r1; };
    struct Stage2 { int64_t c0, c1; };
    auto request1 = [&](int64_t pb) {
        Stage1 r;
        const int64_t pa_l = pb + lane;
        r.in = pa_l < p1;
        const int64_t pa_c = r.in ? pa_l : p0;
        r.nl = g.not_loop[pa_c]; r.u = g.pair_tgt[pa_c]; r.r1 = g.pair_first_rel[pa_c];
        return r;
    };
    auto request2 = [&](const Stage1& a) { return Stage2{g.pair_ptr[a.u], g.pair_ptr[a.u + 1]}; };
    Stage1 a1{false, 0, 0, 0}, b1{false, 0, 0, 0};
    Stage2 a2{0, 0};
    if (p0 < p1) { a1 = request1(p0); b1 = request1(p0 + 64); a2 = request2(a1); }
    for (int64_t pb = p0; pb < p1 && count < limit; pb += 64) {
        const bool ok = a1.in && a1.nl;
        const int deg = ok ? static_cast<int>(a2.c1 - a2.c0) : 0;
        int x = deg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
        const int T = __shfl(x, 63, 64);
        __syncthreads();                                                 // the previous batch's readers are done with the tables
        if (w == 0) { poff[lane] = x - deg; pc0[lane] = ok ? a2.c0 : 0; pr1[lane] = a1.r1; }
        __syncthreads();
        a1 = b1; a2 = request2(b1); b1 = request1(pb + 128);            // under this batch's walk
        int64_t v_r[kChunks], f_r[kChunks]; int lo_r[kChunks];
        auto request = [&](int slot, int x0) {                           // rounds past T: every lane invalid, reads candidate 0 of parent 0's range
            const int xi = x0 + t;
            const bool valid = xi < T;
            int lo;
            { const int a = poff[16], bq = poff[32], c = poff[48]; lo = c <= xi ? 48 : (bq <= xi ? 32 : (a <= xi ? 16 : 0)); }
            { const int a = poff[lo + 4], bq = poff[lo + 8], c = poff[lo + 12]; lo += c <= xi ? 12 : (bq <= xi ? 8 : (a <= xi ? 4 : 0)); }
            { const int a = poff[lo + 1], bq = poff[lo + 2], c = poff[lo + 3]; lo += c <= xi ? 3 : (bq <= xi ? 2 : (a <= xi ? 1 : 0)); }
            const int64_t c = valid ? pc0[lo] + (xi - poff[lo]) : 0;
            v_r[slot] = g.pair_tgt[c]; f_r[slot] = g.pair_first_rel[c]; lo_r[slot] = lo;
        };
#pragma unroll
        for (int d = 0; d < kChunks; ++d) request(d, NTH * d);
        for (int x0 = 0; x0 < T && count < limit; x0 += NTH * kChunks) {  // uniform over the workgroup
#pragma unroll
            for (int d = 0; d < kChunks; ++d) {
                __builtin_amdgcn_sched_barrier(0);                       // a round's wait stays in front of that round
                const int xi = x0 + NTH * d + t;
                const bool valid = xi < T && count < limit;
                const int64_t v = valid ? v_r[d] : 0, fr = f_r[d];
                const int lo = lo_r[d];
                const uint32_t bit = 1u << (v & 31);
                const bool unseen = valid && !(seen[v >> 5] & bit);
                if (unseen) atomicMin(&own[v], static_cast<uint32_t>(xi));
                __syncthreads();
                const bool fresh = unseen && own[v] == static_cast<uint32_t>(xi);
                if (fresh) atomicOr(&seen[v >> 5], bit);
                const unsigned long long m = __ballot(fresh);
                if (lane == 0) wsum[w] = __popcll(m);
                __syncthreads();                                         // the marks and the counts are in; nobody reads own[] of this round any more
                int before = __popcll(m & lt), tot = 0;
#pragma unroll
                for (int ww = 0; ww < kNhopWaves; ++ww) { const int c = wsum[ww]; tot += c; before += ww < w ? c : 0; }
                if (WRITE && fresh && count + before < limit) {
                    int64_t* q = quads + 4 * (base + count + before);
                    q[0] = s; q[1] = pr1[lo]; q[2] = fr; q[3] = v;
                }
                count += tot;
                request(d, x0 + NTH * (d + kChunks));                    // behind the slot's last use
            }
        }
    }
    if (WRITE) return;
    if (count > limit) count = limit;
    if (t == 0) qcount[b] = count;
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
// ---- dead-row pruning of a batch graph (SpKBGATModified: GAT/models.py:167-178) ----------------------------------------------------
// The reference evaluates both attention layers for every entity and then keeps, through `mask`, the rows of the batch entities only:
//   out = entities_upgraded + mask.unsqueeze(-1) * out_entity_1.
// A row of the second layer outside the mask is multiplied by zero, forward and backward; and a row of the first layer matters only
// where the second layer reads it — as x_i of a kept row or as x_j of an edge into one.  So the edges both layers need are those whose
// DESTINATION lies in   need = mask  ∪  { src(e) : mask[dst(e)] } ;  every other edge aggregates into a row nobody reads.  (In the
// reference's batches the destinations are the batch entities' NEIGHBOURS, Corpus.get_batch_adj_data: of ~29 000 edges into ~9 000 rows
// a few hundred edges into a few dozen rows survive.)  ONE workgroup: need <- mask; mark the sources of the edges into masked rows;
// count the surviving edges of each list per thread (a thread owns a run of consecutive edges), scan, write them IN ORDER — the kept edges
// of a row keep their relative order, so the row's sums are the sums the unpruned layer forms.  pos (optional): the surviving edges'
// positions in the CONCATENATED input list, 1-hop then n-hop (the per-edge dropout factors of a recorded run are looked up through them).
// BATCH (recon_edges_prune_batch, what SpKBGATModified calls): the mask itself is made here from the batch's entity ids (GAT/models.py:167-172:
// zeros, mask[unique(batch_entities)] = 1), the n-hop list arrives as the [E2][4] quadruples (source, rel_1, rel_2, target) it is derived from
// (:145-148: edge = (target, source), type = (rel_1, rel_2)), and both lists' survivors go into ONE [2][E1 + E2] buffer, 1-hop first — the
// concatenation the layers build anyway (GAT/layers.py:124-127) — so that the caller's three tensors (1-hop, n-hop, both) are views of it.
template <bool BATCH>
__global__ void __launch_bounds__(1024) k_edges_prune(const int64_t* __restrict__ edge, const int64_t* __restrict__ type, int64_t E1,
                                                      const int64_t* __restrict__ edge_nhop, const int64_t* __restrict__ type_nhop, int64_t E2,
                                                      const float* __restrict__ mask_in, const int64_t* __restrict__ batch, int32_t B,
                                                      float* __restrict__ mask_out, int32_t N, uint8_t* __restrict__ need,
                                                      int64_t* __restrict__ out_edge, int64_t* __restrict__ out_type, int64_t* __restrict__ out_edge_nhop,
                                                      int64_t* __restrict__ out_type_nhop, int64_t* __restrict__ pos, int64_t* __restrict__ counts,
                                                      int64_t* __restrict__ ext_index, int64_t ext_base) {
    // ext_index (BATCH, optional): the row of the layers' edge-embedding TABLE each surviving edge reads — a 1-hop edge its relation id, the j-th
    // surviving n-hop edge row ext_base + j (the n-hop rows are appended to the relation table, gat_layers.cat_edge_embed) — as one [n1 + n2] list
    // The host waits for the counts (the output sizes): this launch is on the iteration's critical path.  Every pass asks for sixteen
    // edges per thread at once (one edge in flight per thread: 102 us for a 30 000-edge batch; this form: 65 us).  What is left is ONE compute
    // unit's rate of scattered accesses — ~4 per edge (the row's mask word, the source's flag, the row's flag, the survivor's words) —, not a
    // chain of round trips: batching the survivors' loads as well (every edge's source and types, kept or not) took 117 us.  Spreading the
    // passes over many workgroups needs a grid-wide order between marking and reading the flags (two launches, or a look-back scan).
    // Outputs (!BATCH): the destination row of a list at out[0 ..), its source row at out[E ..) — the caller's [2, n] tensor is a view with row stride E.
    __shared__ int32_t wsum[2][16];
    __shared__ int32_t s_bad;
    constexpr int U = 16;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* mask = BATCH ? mask_out : mask_in;
    if constexpr (BATCH) {
        if (t == 0) s_bad = 0;
        for (int n = t; n < N; n += 1024) { mask_out[n] = 0.f; need[n] = 0; }
        __syncthreads();
        for (int b = t; b < B; b += 1024) {
            const int64_t v = batch[b];
            if (v >= 0 && v < N) { mask_out[v] = 1.f; need[v] = 1; } else s_bad = 1;      // the reference's `mask[mask_indices] = 1.0` raises on such an id: reported
        }
        __threadfence_block();
    } else {
        for (int n0 = t; n0 < N; n0 += 1024 * U) {                    // (sixteen flags per thread in flight: one at a time was 15 round trips for 14 541 entities)
            float m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) m[u] = n0 + 1024 * u < N ? mask[n0 + 1024 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) if (n0 + 1024 * u < N) need[n0 + 1024 * u] = m[u] != 0.f;
        }
    }
    __syncthreads();
    auto in_range = [&](int64_t v) { return v >= 0 && v < N; };       // ids out of range: the edge is kept, and the graph build reports them as it always does
    // the two lists through one set of accessors: list 0 = edge [2][E1] / type [E1]; list 1 = edge_nhop [2][E2] / type_nhop [E2][2], or (BATCH) the quadruples
    auto dst_of = [&](int list, int64_t i) { return list == 0 ? edge[i] : (BATCH ? edge_nhop[4 * i + 3] : edge_nhop[i]); };
    auto src_of = [&](int list, int64_t i) { return list == 0 ? edge[E1 + i] : (BATCH ? edge_nhop[4 * i] : edge_nhop[E2 + i]); };
    auto mark = [&](int list, int64_t E) {
        for (int64_t i0 = t; i0 < E; i0 += 1024 * U) {
            int32_t d[U], s_[U];
            float m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + 1024 * u;
                const int64_t dv = i < E ? dst_of(list, i) : -1, sv = i < E ? src_of(list, i) : -1;
                d[u] = in_range(dv) ? static_cast<int32_t>(dv) : -1; s_[u] = in_range(sv) ? static_cast<int32_t>(sv) : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) m[u] = d[u] >= 0 ? mask[d[u]] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) if (m[u] != 0.f && s_[u] >= 0) need[s_[u]] = 1;
        }
    };
    mark(0, E1);
    mark(1, E2);
    __threadfence_block();
    __syncthreads();
    // ordered write: a thread owns U CONSECUTIVE edges of a 16 384-edge tile; positions = kept in the tiles before + in the waves before + in the
    // lanes before (a wave scan of the threads' counts) + among the thread's own
    int par = 0;
    int64_t ext_off = 0;                                                 // where the n-hop survivors' entries of ext_index start: the 1-hop survivors' count
    auto write = [&](int list, int64_t E, int64_t* __restrict__ oe, int64_t row_stride, int64_t* __restrict__ ot, int64_t pos_off) -> int64_t {
        int64_t run = 0;
        for (int64_t b = 0; b < E; b += 1024 * U, par ^= 1) {
            const int64_t i0 = b + static_cast<int64_t>(U) * t;
            int32_t d[U];                                                // ids as 32-bit values, -1 = outside [0, N) (64-bit copies of 32 ids per thread spilled)
            uint8_t nd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int64_t v = i0 + u < E ? dst_of(list, i0 + u) : 0; d[u] = in_range(v) ? static_cast<int32_t>(v) : -1; }
#pragma unroll
            for (int u = 0; u < U; ++u) nd[u] = d[u] >= 0 ? need[d[u]] : 1;
            int c = 0;
            uint32_t kmask = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) { const bool k = i0 + u < E && nd[u] != 0; kmask |= static_cast<uint32_t>(k) << u; c += k; }
            int x = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
            if (lane == 63) wsum[par][w] = x;
            __syncthreads();                                             // (one barrier per tile: the sums alternate between two rows)
            int wb = 0, tile = 0;
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) { const int v = wsum[par][ww]; tile += v; wb += ww < w ? v : 0; }
            int64_t o = run + wb + x - c;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if ((kmask >> u) & 1) {                                  // (the survivors are few: their sources and types are fetched here)
                    const int64_t i = i0 + u;
                    oe[o] = dst_of(list, i); oe[row_stride + o] = src_of(list, i);
                    if (list == 0) { ot[o] = type[i]; if (BATCH && ext_index) ext_index[o] = type[i]; }
                    else if (BATCH) { ot[2 * o] = edge_nhop[4 * i + 1]; ot[2 * o + 1] = edge_nhop[4 * i + 2]; if (ext_index) ext_index[ext_off + o] = ext_base + o; }
                    else { ot[2 * o] = type_nhop[2 * i]; ot[2 * o + 1] = type_nhop[2 * i + 1]; }
                    if (pos) pos[pos_off + o] = pos_off + i;
                    ++o;
                }
            }
            run += tile;
        }
        return run;
    };
    const int64_t n1 = write(0, E1, out_edge, BATCH ? E1 + E2 : E1, out_type, 0);
    ext_off = n1;
    __syncthreads();
    // (the n-hop positions follow the 1-hop ones at E1: the caller cuts the two runs out of pos with the counts)
    const int64_t n2 = write(1, E2, BATCH ? out_edge + n1 : out_edge_nhop, BATCH ? E1 + E2 : E2, out_type_nhop, E1);
    if (t == 0) { counts[0] = n1; counts[1] = n2; if (BATCH) counts[2] = s_bad; }
}

KG kg_of(const recon_kg* k) {
    return KG{k->pair_ptr, k->pair_tgt, k->pair_first_rel, k->not_loop, k->rel_ptr, k->rel_sorted, k->num_entities};
}
}  // namespace
}  // namespace recon

// the visited bitmap of one source + the parent tables of a batch (64 x (4 + 8 + 8) bytes, behind an 8-byte boundary)
extern "C" size_t recon_kg_nhop_lds_bytes(int64_t num_entities) { return static_cast<size_t>((num_entities + 63) / 64) * 8 + 64 * 20; }
// the workgroup form's: bitmap + one word per entity + the parent tables + the waves' counts
static size_t nhop_wg_lds_bytes(int64_t num_entities) {
    return static_cast<size_t>((num_entities + 63) / 64) * 8 + static_cast<size_t>((num_entities + 1) & ~1LL) * 4 + 64 * 20 + 16;
}

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
    const size_t lds_wg = nhop_wg_lds_bytes(kg->num_entities);
    if (lds_wg <= 72 * 1024 && recon::cfg_char(recon::CFG_KG_NHOP) != 'w') {   // four waves per source (two workgroups per CU still fit): ~18 k entities
        const void* kw = write ? reinterpret_cast<const void*>(recon::k_kg_nhop_wg<true>) : reinterpret_cast<const void*>(recon::k_kg_nhop_wg<false>);
        if (lds_wg > 48 * 1024 && hipFuncSetAttribute(kw, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_wg)) != hipSuccess) return RECON_ERR_LAUNCH;
        const dim3 blk(64 * recon::kNhopWaves);
        if (write) hipLaunchKernelGGL(recon::k_kg_nhop_wg<true>, dim3(static_cast<unsigned>(S)), blk, lds_wg, as_stream(stream), recon::kg_of(kg), sources, S, partial_2hop,
                                      qcount, quad_off, quads);
        else {
            hipLaunchKernelGGL(recon::k_kg_nhop_wg<false>, dim3(static_cast<unsigned>(S)), blk, lds_wg, as_stream(stream), recon::kg_of(kg), sources, S, partial_2hop,
                               qcount, quad_off, quads);
            hipLaunchKernelGGL(recon::k_kg_scan, dim3(1), dim3(64), 0, as_stream(stream), qcount, S, quad_off, total);
        }
        RECON_CHECK_LAUNCH();
        return RECON_OK;
    }
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

// The count pass of recon_kg_nhop for sources whose NUMBER is still on the device (the unique-entity list recon_kg_adj_count has just written: up to
// S_max entries, *s_dev of them valid): launched behind it, so that the host reads the 1-hop sizes and the 2-hop total in ONE round trip.
extern "C" int recon_kg_nhop_count_early(const recon_kg* kg, const int64_t* sources, int32_t S_max, const int64_t* s_dev, int32_t partial_2hop, int64_t* qcount,
                                         int64_t* quad_off, int64_t* total, recon_stream_t stream) {
    if (!recon::kg_ok(kg) || S_max <= 0 || !sources || !s_dev || !qcount || !quad_off || !total) return RECON_ERR_INVALID;
    const size_t lds_wg = nhop_wg_lds_bytes(kg->num_entities);
    if (lds_wg <= 72 * 1024 && recon::cfg_char(recon::CFG_KG_NHOP) != 'w') {
        const void* kw = reinterpret_cast<const void*>(recon::k_kg_nhop_wg<false>);
        if (lds_wg > 48 * 1024 && hipFuncSetAttribute(kw, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_wg)) != hipSuccess) return RECON_ERR_LAUNCH;
        hipLaunchKernelGGL(recon::k_kg_nhop_wg<false>, dim3(static_cast<unsigned>(S_max)), dim3(64 * recon::kNhopWaves), lds_wg, as_stream(stream), recon::kg_of(kg), sources,
                           S_max, partial_2hop, qcount, quad_off, static_cast<int64_t*>(nullptr), s_dev);
    } else {
        const size_t lds = recon_kg_nhop_lds_bytes(kg->num_entities);
        if (lds > 160 * 1024 - 64) return RECON_ERR_UNSUPPORTED;
        const void* kern = reinterpret_cast<const void*>(recon::k_kg_nhop<false>);
        if (lds > 48 * 1024 && hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return RECON_ERR_LAUNCH;
        hipLaunchKernelGGL(recon::k_kg_nhop<false>, dim3(static_cast<unsigned>(S_max)), dim3(64), lds, as_stream(stream), recon::kg_of(kg), sources, S_max, partial_2hop,
                           qcount, quad_off, static_cast<int64_t*>(nullptr), s_dev);
    }
    hipLaunchKernelGGL(recon::k_kg_scan, dim3(1), dim3(64), 0, as_stream(stream), qcount, S_max, quad_off, total);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_edges_prune(const int64_t* edge, const int64_t* type, int64_t E1, const int64_t* edge_nhop, const int64_t* type_nhop, int64_t E2,
                                 const float* mask, int32_t N, uint8_t* need, int64_t* out_edge, int64_t* out_type, int64_t* out_edge_nhop,
                                 int64_t* out_type_nhop, int64_t* pos, int64_t* counts, recon_stream_t stream) {
    if (E1 < 0 || E2 < 0 || N <= 0 || !mask || !need || !counts) return RECON_ERR_INVALID;
    if (E1 > 0 && (!edge || !type || !out_edge || !out_type)) return RECON_ERR_INVALID;
    if (E2 > 0 && (!edge_nhop || !type_nhop || !out_edge_nhop || !out_type_nhop)) return RECON_ERR_INVALID;
    hipLaunchKernelGGL(recon::k_edges_prune<false>, dim3(1), dim3(1024), 0, as_stream(stream), edge, type, E1, edge_nhop, type_nhop, E2, mask, nullptr, 0, nullptr, N,
                       need, out_edge, out_type, out_edge_nhop, out_type_nhop, pos, counts, nullptr, static_cast<int64_t>(0));
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_edges_prune_batch(const int64_t* batch_entities, int32_t B, const int64_t* edge, const int64_t* type, int64_t E1, const int64_t* quads,
                                       int64_t E2, int32_t N, float* mask, uint8_t* need, int64_t* out_edge, int64_t* out_type, int64_t* out_type_nhop,
                                       int64_t* pos, int64_t* counts, int64_t* ext_index, int64_t ext_base, recon_stream_t stream) {
    if (B < 0 || E1 < 0 || E2 < 0 || N <= 0 || !mask || !need || !counts || (B > 0 && !batch_entities)) return RECON_ERR_INVALID;
    if (E1 > 0 && (!edge || !type || !out_type)) return RECON_ERR_INVALID;
    if (E2 > 0 && (!quads || !out_type_nhop)) return RECON_ERR_INVALID;
    if (E1 + E2 > 0 && !out_edge) return RECON_ERR_INVALID;
    hipLaunchKernelGGL(recon::k_edges_prune<true>, dim3(1), dim3(1024), 0, as_stream(stream), edge, type, E1, quads, nullptr, E2, nullptr, batch_entities, B, mask, N,
                       need, out_edge, out_type, nullptr, out_type_nhop, pos, counts, ext_index, ext_base);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
