// K3 — graph preparation for the GAT hot path: int64 COO edge list -> destination-CSR and
// source-CSC index arrays (int32), via a hand-written stable LSD radix sort (8-bit digits,
// wave-ballot ranking).  Replaces the implicit coalesce / sort of torch.sparse.sum at
// GAT/layers.py:56-58.  Integer work, HBM/latency bound; cached per edge tensor by the caller.
#include "recon_common.h"

namespace {

constexpr int kTile = 1024;      // items per block per pass
constexpr int kFoldScanBlocks = 128;   // up to this many tiles (131 072 items) the scatter kernel scans the histogram itself: a launch less per pass
constexpr int kThreads = 256;    // 4 waves, each ranks 256 consecutive items

__global__ void __launch_bounds__(kThreads) k_radix_hist(const int32_t* __restrict__ keys, int32_t n, int shift,
                                                         int32_t* __restrict__ blockhist, int32_t nblocks) {
    __shared__ int32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kTile;
#pragma unroll
    for (int j = 0; j < kTile / kThreads; ++j) {
        int i = base + j * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255], 1);
    }
    __syncthreads();
    blockhist[threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// key conversion (dst -> int32 key, iota value) or source gather (src of slot -> key, slot value) of one 1 024-item tile TOGETHER with the
// tile's first-pass digit histogram: one launch instead of two at the head of each sort (a graph build is a chain of ~5 us turn-arounds)
template <bool GATHER>
__global__ void __launch_bounds__(kThreads) k_keys_hist(const int64_t* __restrict__ in, const int32_t* __restrict__ eid, int32_t n,
                                                        int32_t* __restrict__ src_out, int32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                        int32_t* __restrict__ blockhist, int32_t nblocks, int32_t N, int32_t* __restrict__ bad,
                                                        int32_t* __restrict__ nexthist, const int32_t* __restrict__ lb_keys = nullptr,
                                                        int32_t* __restrict__ lb_rowptr = nullptr, int32_t* __restrict__ zero4 = nullptr) {
    if (zero4 && blockIdx.x == 0 && threadIdx.x < 6) zero4[threadIdx.x] = 0;      // the hub counts of recon_graph_build_counted (+ flag word, + the count of rows with edges)
    if (static_cast<int>(blockIdx.x) >= nblocks) {                       // trailing blocks: the row pointers of the sort that just finished
        const int r = (blockIdx.x - nblocks) * kThreads + threadIdx.x;  // (k_rowptr_lower_bound's work: a launch less per build)
        if (r > N) return;
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (lb_keys[mid] < r) lo = mid + 1; else hi = mid; }
        lb_rowptr[r] = lo;
        return;
    }
    __shared__ int32_t hist[256];
    hist[threadIdx.x] = 0;
    if (nexthist) nexthist[threadIdx.x * nblocks + blockIdx.x] = 0;     // the second pass' histogram, counted by the first pass' scatter
    __syncthreads();
    const int base = blockIdx.x * kTile;
#pragma unroll
    for (int j = 0; j < kTile / kThreads; ++j) {
        const int i = base + j * kThreads + threadIdx.x;
        if (i < n) {
            const int64_t v64 = GATHER ? in[eid[i]] : in[i];
            if (bad && (v64 < 0 || v64 >= N)) *bad = 1;                // an id outside [0, N): reported with the hub counts (no round trip of its own)
            const int32_t v = static_cast<int32_t>(v64);
            if (GATHER) src_out[i] = v;
            keys[i] = v; vals[i] = i;
            atomicAdd(&hist[v & 255], 1);
        }
    }
    __syncthreads();
    blockhist[threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// single-block exclusive scan (in place); n is at most a few hundred thousand counters.  Every thread owns a contiguous chunk (a multiple
// of four counters, read and written 16 bytes at a time with all of a chunk's loads in flight together: one counter per dependent
// load made this kernel 127 us at 372 k edges — most of a graph build).
__global__ void __launch_bounds__(1024) k_scan_exclusive(int32_t* __restrict__ data, int32_t n) {
    __shared__ int32_t sums[1024];
    const int t = threadIdx.x;
    const int chunk = (((n + 1023) / 1024) + 3) & ~3;
    const int lo = min(t * chunk, n), hi = min(lo + chunk, n);
    const bool full = hi - lo == chunk && !(reinterpret_cast<uintptr_t>(data) & 15);
    int32_t s = 0;
    if (full) {
        const int4* d4 = reinterpret_cast<const int4*>(data + lo);
#pragma unroll 8
        for (int i = 0; i < chunk / 4; ++i) { const int4 v = d4[i]; s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (int i = lo; i < hi; ++i) s += data[i];
    }
    sums[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan
        int32_t v = (t >= off) ? sums[t - off] : 0;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    int32_t run = sums[t] - s;
    if (full) {
        int4* d4 = reinterpret_cast<int4*>(data + lo);
#pragma unroll 8
        for (int i = 0; i < chunk / 4; ++i) {
            const int4 v = d4[i];
            int4 o;
            o.x = run; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
            run = o.w + v.w;
            d4[i] = o;
        }
    } else {
        for (int i = lo; i < hi; ++i) { int32_t v = data[i]; data[i] = run; run += v; }
    }
}

// SCANNED: `blockhist` holds the exclusive scan over (digit, block) (k_scan_exclusive ran in front).  Otherwise it holds the raw per-block
// digit counts and every workgroup derives its own bases from them — digit t's total over all blocks, the digits' exclusive scan, plus
// digit t's counts of the blocks in front of this one: nblocks loads per thread instead of a launch (kFoldScanBlocks: where that is cheap).
// `nexthist` (zeroed by k_keys_hist): the per-tile digit histogram of the NEXT pass, counted here where every item's new position is known
// — integer atomics, so the counts are the same on every run — instead of by a launch of k_radix_hist between the passes.
template <bool SCANNED>
__global__ void __launch_bounds__(kThreads) k_radix_scatter(const int32_t* __restrict__ keys_in,
                                                            const int32_t* __restrict__ vals_in,
                                                            int32_t* __restrict__ keys_out,
                                                            int32_t* __restrict__ vals_out, int32_t n, int shift,
                                                            const int32_t* __restrict__ blockhist_scanned,
                                                            int32_t nblocks, int32_t* __restrict__ nexthist = nullptr) {
    __shared__ int32_t wcount[4][256];
    __shared__ int32_t dig[256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += kThreads) (&wcount[0][0])[i] = 0;
    // unscanned: digit t's counts of every tile are requested here, together with the items — they depend on the previous launch only, and
    // behind the ranking they were a round trip of their own in every pass
    int32_t tot = 0, before = 0;
    if constexpr (!SCANNED) {
        const int32_t* row = blockhist_scanned + threadIdx.x * nblocks;
        const int me = static_cast<int>(blockIdx.x);
        auto take = [&](int b, int32_t c) { if (b < nblocks) { tot += c; before += b < me ? c : 0; } };
        if ((nblocks & 3) == 0 && !(reinterpret_cast<uintptr_t>(blockhist_scanned) & 15)) {          // sixteen counts per round trip
            for (int b0 = 0; b0 < nblocks; b0 += 16) {
                int4 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = *reinterpret_cast<const int4*>(row + min(b0 + 4 * u, nblocks - 4));
#pragma unroll
                for (int u = 0; u < 4; ++u) { take(b0 + 4 * u, c[u].x); take(b0 + 4 * u + 1, c[u].y); take(b0 + 4 * u + 2, c[u].z); take(b0 + 4 * u + 3, c[u].w); }
            }
        } else {
            for (int b0 = 0; b0 < nblocks; b0 += 8) {
                int32_t c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) c[u] = row[min(b0 + u, nblocks - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) take(b0 + u, c[u]);
            }
        }
    }
    const int base = blockIdx.x * kTile + w * 256;
    int32_t key[4], val[4], rank[4];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                        // (in front of the barrier: it waits for every request made so far)
        const int i = base + j * 64 + lane;
        key[j] = i < n ? keys_in[i] : 0;
        val[j] = i < n ? vals_in[i] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = base + j * 64 + lane;
        const bool valid = i < n;
        const int digit = (key[j] >> shift) & 255;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1;
            const unsigned long long m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        rank[j] = 0;
        if (valid) {
            const int prior = wcount[w][digit];               // every peer reads before the leader adds
            rank[j] = prior + __popcll(peers & lt);
            if ((peers & lt) == 0) wcount[w][digit] = prior + __popcll(peers);   // lowest peer lane
        }
    }
    __syncthreads();
    {   // thread t owns digit t: global base of this block + exclusive prefix over the 4 waves
        const int t = threadIdx.x;
        int32_t run;
        if constexpr (SCANNED) {
            run = blockhist_scanned[t * nblocks + blockIdx.x];
        } else {
            int32_t x = tot;                                             // inclusive scan over the digits: within the wave, then over the four waves' sums
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
            if (lane == 63) dig[w] = x;
            __syncthreads();
            int32_t wbase = 0;
#pragma unroll
            for (int ww = 0; ww < 3; ++ww) wbase += ww < w ? dig[ww] : 0;
            run = wbase + x - tot + before;
        }
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { int32_t c = wcount[ww][t]; wcount[ww][t] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = base + j * 64 + lane;
        if (i < n) {
            const int pos = wcount[w][(key[j] >> shift) & 255] + rank[j];
            keys_out[pos] = key[j];
            vals_out[pos] = val[j];
            if (nexthist) atomicAdd(&nexthist[((key[j] >> (shift + 8)) & 255) * nblocks + pos / kTile], 1);
        }
    }
}

// rowptr[r] = first slot whose (sorted) key is >= r, r in [0, N]
__global__ void k_rowptr_lower_bound(const int32_t* __restrict__ sorted_keys, int32_t E, int32_t N,
                                     int32_t* __restrict__ rowptr) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > N) return;
    int lo = 0, hi = E;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (sorted_keys[mid] < r) lo = mid + 1; else hi = mid; }
    rowptr[r] = lo;
}

// The last launch of a full build when the caller wants the hub-table sizes as well (recon_graph_build_counted): the source row pointers
// as above and, from them and the finished destination row pointers, counts[4] = (hubs, pieces) of both sides — what k_hub_scan<false>
// counts in a launch of its own.  A thread's upper neighbour's result comes through LDS (the block's last thread searches once more);
// integer atomics into zeroed words: the same counts on every run.
__global__ void __launch_bounds__(256) k_rowptr_lower_bound_count(const int32_t* __restrict__ sorted_keys, int32_t E, int32_t N, int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ rowptr_other, int32_t chunk, int32_t* __restrict__ counts,
                                                                  const int32_t* __restrict__ bad) {
    __shared__ int32_t lb[257];
    const int t = threadIdx.x, r = blockIdx.x * 256 + t;
    if (blockIdx.x == 0 && t == 0) counts[4] = bad ? *bad : 0;           // the build's id-range flag (set by earlier launches) travels with the counts: one copy back
    auto search = [&](int key) { int lo = 0, hi = E; while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid; } return lo; };
    const int mine = r <= N ? search(r) : E;
    lb[t] = mine;
    if (t == 255) lb[256] = r + 1 <= N ? search(r + 1) : E;
    if (r <= N) rowptr[r] = mine;
    const int deg_o = r < N ? rowptr_other[r + 1] - rowptr_other[r] : 0;
    const int live = __syncthreads_count(deg_o > 0);                     // (the barrier the LDS exchange needs) destination rows WITH edges: counts[5], one atomic per block
    if (t == 0 && live) atomicAdd(&counts[5], live);
    if (r < N) {
        const int deg = lb[t + 1] - mine;
        if (deg_o > chunk) { atomicAdd(&counts[0], 1); atomicAdd(&counts[1], (deg_o + chunk - 1) / chunk); }
        if (deg > chunk) { atomicAdd(&counts[2], 1); atomicAdd(&counts[3], (deg + chunk - 1) / chunk); }
    }
}

__global__ void k_copy_i32(const int32_t* __restrict__ in, int32_t* __restrict__ out, int32_t n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

int bits_for(int32_t n) { int b = 1; while (b < 31 && (1 << b) < n) ++b; return b; }

struct SortWs { int32_t *kA, *kB, *vA, *vB, *hist, *hist2; int32_t nblocks; };

// stable sort of (kA, vA); the LAST pass scatters into (kfinal, vfinal) when given (a copy launch each less per sort: a graph build is
// launch bound at the stage-A batch sizes), else the result ends up in (*kout, *vout), which point into the ping-pong buffers
int radix_sort_pairs(SortWs& ws, int32_t n, int32_t key_range, int32_t** kout, int32_t** vout, hipStream_t st, int32_t* kfinal = nullptr,
                     int32_t* vfinal = nullptr, bool first_hist_done = false) {
    const int passes = (bits_for(key_range) + 7) / 8;
    int32_t *ki = ws.kA, *vi = ws.vA, *ko = ws.kB, *vo = ws.vB;
    // two passes with the first histogram already counted (k_keys_hist, which also zeroed hist2): the first scatter counts the second
    // pass' histogram into hist2 — four launches per sort instead of five
    const bool chain = passes == 2 && first_hist_done && ws.hist2;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        if (p == passes - 1) { if (kfinal) ko = kfinal; if (vfinal) vo = vfinal; }
        int32_t* hist = (chain && p == 1) ? ws.hist2 : ws.hist;
        int32_t* next = (chain && p == 0) ? ws.hist2 : nullptr;
        if (!(chain && p == 1) && (p > 0 || !first_hist_done))
            hipLaunchKernelGGL(k_radix_hist, dim3(ws.nblocks), dim3(kThreads), 0, st, ki, n, shift, hist, ws.nblocks);
        if (ws.nblocks <= kFoldScanBlocks) {
            hipLaunchKernelGGL((k_radix_scatter<false>), dim3(ws.nblocks), dim3(kThreads), 0, st, ki, vi, ko, vo, n, shift, hist, ws.nblocks, next);
        } else {
            hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, hist, 256 * ws.nblocks);
            hipLaunchKernelGGL((k_radix_scatter<true>), dim3(ws.nblocks), dim3(kThreads), 0, st, ki, vi, ko, vo, n, shift, hist, ws.nblocks, next);
        }
        int32_t* t;
        t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    *kout = ki; *vout = vi;
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

// Hub tables (recon_hip.h): destination rows longer than `chunk` slots, in node order, cut into pieces of <= chunk slots.
// ONE workgroup: every thread counts the hubs / pieces of its own run of consecutive nodes, a block-wide exclusive scan hands out the
// table positions, and the thread fills them in node order — ordered tables, the same on every run.
template <bool FILL>
__global__ void __launch_bounds__(1024) k_hub_scan(const int32_t* __restrict__ rowptr, int32_t N, int32_t chunk, int32_t* __restrict__ counts,
                                                   int32_t* __restrict__ hub_node, int32_t* __restrict__ hub_ptr, int4* __restrict__ piece) {
    __shared__ int32_t sh[1024], sp[1024];
    __shared__ int32_t s_live;
    const int t = threadIdx.x;
    if (t == 0) s_live = 0;
    const bool count_live = !FILL && blockIdx.x == 0 && hub_ptr;          // count form behind a build, destination side: rows with edges -> counts[5]
    if (!FILL && blockIdx.x == 0 && t == 0 && hub_ptr) counts[4] = hub_node ? *hub_node : 0;   // count form behind a build: its id-range flag rides in `hub_node` (hub_ptr: non-null marks that form)
    if (!FILL && blockIdx.x == 1) {                                      // count form, second workgroup: the other row pointer rides in `piece`
        rowptr = reinterpret_cast<const int32_t*>(piece);
        counts += 2;
        if (!rowptr) { if (t == 0) { counts[0] = 0; counts[1] = 0; } return; }
    }
    const int per = (N + 1023) / 1024;                                   // every thread owns `per` consecutive nodes
    const int lo = min(t * per, N), hi = min(lo + per, N);
    int h = 0, p = 0, live = 0;
    for (int i = lo; i < hi; ++i) {
        const int deg = rowptr[i + 1] - rowptr[i];
        if (deg > chunk) { ++h; p += (deg + chunk - 1) / chunk; }
        live += deg > 0;
    }
    sh[t] = h; sp[t] = p;
    __syncthreads();
    if (count_live && live) atomicAdd(&s_live, live);
    for (int off = 1; off < 1024; off <<= 1) {                           // Hillis-Steele inclusive scans
        const int vh = (t >= off) ? sh[t - off] : 0, vp = (t >= off) ? sp[t - off] : 0;
        __syncthreads();
        sh[t] += vh; sp[t] += vp;
        __syncthreads();
    }
    if (FILL) {
        int oh = sh[t] - h, op = sp[t] - p;
        for (int i = lo; i < hi; ++i) {
            const int beg = rowptr[i], deg = rowptr[i + 1] - beg;
            if (deg > chunk) {
                const int np = (deg + chunk - 1) / chunk;
                hub_node[oh] = i; hub_ptr[oh] = op;
                for (int q = 0; q < np; ++q) {
                    const int b = beg + q * chunk;
                    piece[op + q] = make_int4(i, b, min(b + chunk, beg + deg), oh);
                }
                ++oh; op += np;
            }
        }
        if (t == 1023) hub_ptr[sh[t]] = sp[t];
    } else if (t == 1023) { counts[0] = sh[t]; counts[1] = sp[t]; if (count_live) counts[5] = s_live; }
}

// Row compaction (recon_graph.n_rows): the destination rows WITH edges, in node order — row_node[r] = node, rowptr_rows[r] = the row's
// first slot (rowptr_rows[n_rows] = E), node_row[node] = r or -1.  ONE workgroup, the scheme of k_hub_scan: every thread counts the live
// rows of its own run of consecutive nodes, a block-wide exclusive scan hands out the positions.  A knowledge-graph batch aggregates
// into the batch's ~128 entities of a 14 541-entity table: every node-parallel stage of the attention layer then runs over the rows
// that have something to aggregate instead of over the table.
__global__ void __launch_bounds__(1024) k_rows_compact(const int32_t* __restrict__ rowptr, int32_t N, int32_t* __restrict__ row_node,
                                                       int32_t* __restrict__ rowptr_rows, int32_t* __restrict__ node_row) {
    __shared__ int32_t sc[1024];
    const int t = threadIdx.x;
    const int per = (N + 1023) / 1024;
    const int lo = min(t * per, N), hi = min(lo + per, N);
    int live = 0;
    for (int i = lo; i < hi; ++i) live += rowptr[i + 1] > rowptr[i];
    sc[t] = live;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (t >= off) ? sc[t - off] : 0;
        __syncthreads();
        sc[t] += v;
        __syncthreads();
    }
    int o = sc[t] - live;
    for (int i = lo; i < hi; ++i) {
        const int b = rowptr[i];
        if (rowptr[i + 1] > b) { row_node[o] = i; rowptr_rows[o] = b; node_row[i] = o; ++o; }
        else node_row[i] = -1;
    }
    if (t == 1023) rowptr_rows[sc[t]] = rowptr[N];
}

// ---- the whole build of a SMALL graph in ONE launch (E <= kSmallE, N <= kSmallN): the chain of 14 launches above is a chain of turn-arounds on
// the device and ~0.1 ms of launch work on the host per graph.  One workgroup of 16 waves:
//   counts[key] in LDS -> exclusive scan -> rowptr;   then the stable LSD radix sort, a pass = every wave counts the digits of ITS
//   contiguous segment (LDS), one scan over (digit, wave), and every wave walks its segment again in order, 64 items a round, ranking
//   equal digits with ballots (as k_radix_scatter) — positions are a pure function of the input: the same tables on every run.
// Buffers written in one pass are read in the next by other waves of the same workgroup (same CU, behind a barrier).
// Measured on the stage-A batches (E = 17 k .. 34 k): 86 us rows-only / 239 us with the source view against ~35 / ~75 us for the chain —
// one workgroup is latency bound on its scattered stores — so the limit sits where the one launch still wins (E <= 8 192: ~15 us).
constexpr int kSmallE = 8192, kSmallN = 32768, kSmallUnroll = 8;

// (kA, vA) is the input; the passes ping-pong between the two buffer pairs, the last one writes (k_final, v_final)
__device__ __forceinline__ void small_sort(int32_t E, int passes, int32_t* kA, int32_t* vA, int32_t* kB, int32_t* vB, int32_t* k_final, int32_t* v_final,
                                           int32_t* wc, int32_t* part) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int seg = ((E + 15) / 16 + 63) & ~63;                          // items per wave, a multiple of 64
    const int lo = w * seg, hi = min(E, lo + seg);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int32_t* ki = kA; const int32_t* vi = vA;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        int32_t* ko = (p == passes - 1) ? k_final : ((p & 1) ? kA : kB);
        int32_t* vo = (p == passes - 1) ? v_final : ((p & 1) ? vA : vB);
        for (int i = t; i < 4096; i += 1024) wc[i] = 0;
        __syncthreads();
        // (rounds of 64 items, kSmallUnroll rounds' loads in flight together: one round per dependent load made this kernel a chain of
        // ~250 L2 round trips — 0.1 ms per graph, slower than the launches it replaces)
        for (int i0 = lo; i0 < hi; i0 += 64 * kSmallUnroll) {
            int32_t kk[kSmallUnroll];
#pragma unroll
            for (int u = 0; u < kSmallUnroll; ++u) { const int i = i0 + 64 * u + lane; kk[u] = i < hi ? ki[i] : -1; }
#pragma unroll
            for (int u = 0; u < kSmallUnroll; ++u) if (kk[u] >= 0) atomicAdd(&wc[((kk[u] >> shift) & 255) * 16 + w], 1);      // [digit][wave]
        }
        __syncthreads();
        {   // exclusive scan over the 4 096 (digit, wave) counters: four per thread
            int32_t c[4], s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = wc[4 * t + j]; s += c[j]; }
            int32_t x = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
            if (lane == 63) part[w] = x;
            __syncthreads();
            int32_t base = 0;
            for (int i = 0; i < w; ++i) base += part[i];
            int32_t run = base + x - s;
#pragma unroll
            for (int j = 0; j < 4; ++j) { wc[4 * t + j] = run; run += c[j]; }
        }
        __syncthreads();
        for (int j0 = lo; j0 < hi; j0 += 64 * kSmallUnroll) {            // wave-uniform trip counts
          int32_t kk[kSmallUnroll], vv[kSmallUnroll];
#pragma unroll
          for (int u = 0; u < kSmallUnroll; ++u) { const int i = j0 + 64 * u + lane; kk[u] = i < hi ? ki[i] : 0; vv[u] = i < hi ? vi[i] : 0; }
#pragma unroll
          for (int u = 0; u < kSmallUnroll; ++u) {
            const int i0 = j0 + 64 * u;
            if (i0 >= hi) break;
            const int i = i0 + lane;
            const bool valid = i < hi;
            const int32_t key = kk[u], val = vv[u];
            const int digit = (key >> shift) & 255;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (digit >> b) & 1;
                const unsigned long long m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
            if (valid) {
                const int32_t prior = wc[digit * 16 + w];               // every peer reads before the leader adds (one wave: LDS operations in order)
                const int rank = __popcll(peers & lt);
                if (rank == 0) wc[digit * 16 + w] = prior + __popcll(peers);
                ko[prior + rank] = key;
                vo[prior + rank] = val;
            }
          }
        }
        __syncthreads();
        ki = ko; vi = vo;
    }
}

template <bool WITH_SRC>
__global__ void __launch_bounds__(1024) k_graph_build_small(const int64_t* __restrict__ edge_dst, const int64_t* __restrict__ edge_src, int32_t N, int32_t E,
                                                             int32_t* __restrict__ rowptr_dst, int32_t* __restrict__ eid, int32_t* __restrict__ src_out,
                                                             int32_t* __restrict__ dst_out, int32_t* __restrict__ rowptr_src,
                                                             int32_t* __restrict__ slot_by_src, int32_t* __restrict__ kA, int32_t* __restrict__ vA,
                                                             int32_t* __restrict__ kB, int32_t* __restrict__ vB, int32_t* __restrict__ bad) {
    extern __shared__ int32_t lds_i[];
    int32_t* wc = lds_i;                                                // [256][16]
    int32_t* part = lds_i + 4096;                                       // [16] (+ padding)
    int32_t* counts = lds_i + 4096 + 64;                                // [N + 1]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int bits = 1;
    while (bits < 31 && (1 << bits) < N) ++bits;
    const int passes = (bits + 7) / 8;
    const int per = (N + 1 + 1023) / 1024;                              // counters per thread in the scans below
    for (int side = 0; side < (WITH_SRC ? 2 : 1); ++side) {
        int32_t* rowptr = side ? rowptr_src : rowptr_dst;
        for (int i = t; i <= N; i += 1024) counts[i] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < E; i0 += 1024 * kSmallUnroll) {
            int32_t k[kSmallUnroll], e[kSmallUnroll];
#pragma unroll
            for (int u = 0; u < kSmallUnroll; ++u) { const int i = i0 + 1024 * u + t; e[u] = (side && i < E) ? eid[i] : i; }
#pragma unroll
            for (int u = 0; u < kSmallUnroll; ++u) {
                const int i = i0 + 1024 * u + t;
                const int64_t v64 = i < E ? (side ? edge_src[e[u]] : edge_dst[i]) : 0;
                const bool oob = v64 < 0 || v64 >= N;                   // reported (`bad`), and kept away from the LDS counters
                if (oob && bad) *bad = 1;
                k[u] = oob ? 0 : static_cast<int32_t>(v64);
            }
#pragma unroll
            for (int u = 0; u < kSmallUnroll; ++u) {
                const int i = i0 + 1024 * u + t;
                if (i >= E) break;
                if (side) src_out[i] = k[u];
                kA[i] = k[u]; vA[i] = i;
                atomicAdd(&counts[k[u]], 1);
            }
        }
        __syncthreads();
        {   // rowptr = exclusive scan of the counts
            const int a = min(t * per, N + 1), b = min(a + per, N + 1);
            int32_t s = 0;
            for (int i = a; i < b; ++i) s += counts[i];
            int32_t x = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
            if (lane == 63) part[w] = x;
            __syncthreads();
            int32_t base = 0;
            for (int i = 0; i < w; ++i) base += part[i];
            int32_t run = base + x - s;
            for (int i = a; i < b; ++i) { rowptr[i] = run; run += counts[i]; }
        }
        __syncthreads();
        if (side == 0) small_sort(E, passes, kA, vA, kB, vB, dst_out, eid, wc, part);
        else small_sort(E, passes, kA, vA, kB, vB, (passes & 1) ? kB : kA, slot_by_src, wc, part);      // the sorted source keys are not kept: into the buffer the last pass does not read
    }
}


// out32[k] = out64[k] = index[eid[k]]: an index tensor in CSR-slot order, both widths in one launch (was an index, a cast and eid's own cast)
__global__ void k_slot_index(const int64_t* __restrict__ index, const int32_t* __restrict__ eid, int32_t E, int32_t* __restrict__ out32, int64_t* __restrict__ out64) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const int64_t v = index[eid[k]];
    out32[k] = static_cast<int32_t>(v);
    out64[k] = v;
}
}  // namespace

extern "C" int recon_graph_hubs_count(const recon_graph* g, int32_t chunk, void* workspace, int32_t* counts, recon_stream_t stream) {
    return recon_graph_hubs_count_checked(g, chunk, workspace, counts, nullptr, nullptr, stream);
}

extern "C" int recon_graph_hubs_count_checked(const recon_graph* g, int32_t chunk, void* workspace, int32_t* counts, const int32_t* bad, int32_t* bad_host,
                                              recon_stream_t stream) {
    if (!g || !counts || chunk <= 0 || g->N < 0 || !g->rowptr_dst || (bad && !bad_host)) return RECON_ERR_INVALID;
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (bad_host) *bad_host = 0;
    if (g->N == 0 || g->E <= chunk) {
        if (bad && (hipMemcpyAsync(bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost, as_stream(stream)) != hipSuccess ||
                    hipStreamSynchronize(as_stream(stream)) != hipSuccess)) return RECON_ERR_LAUNCH;
        return RECON_OK;
    }
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 3)) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    int32_t* d = static_cast<int32_t*>(workspace);
    // workgroup 0 counts the destination side, workgroup 1 the source side (or writes zeros): one launch
    hipLaunchKernelGGL((k_hub_scan<false>), dim3(2), dim3(1024), 0, st, g->rowptr_dst, g->N, chunk, d, nullptr, nullptr, reinterpret_cast<int4*>(g->rowptr_src));
    RECON_CHECK_LAUNCH();
    // the range-check flag of the build rides in the same round trip (second copy, one synchronisation)
    if (bad && hipMemcpyAsync(bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess) return RECON_ERR_LAUNCH;
    if (hipMemcpyAsync(counts, d, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return RECON_ERR_LAUNCH;
    return RECON_OK;
}

extern "C" int recon_graph_hubs_fill(const recon_graph* g, recon_stream_t stream) {
    if (!g || g->N < 0 || !g->rowptr_dst || (g->n_hub_src > 0 && !g->rowptr_src) || g->hub_chunk <= 0 || g->n_hub < 0 || g->n_piece < 0 || g->n_hub_src < 0 ||
        g->n_piece_src < 0) return RECON_ERR_INVALID;
    if (g->n_hub > 0) {
        if (!g->hub_node || !g->hub_ptr || !g->piece || (reinterpret_cast<uintptr_t>(g->piece) & 15)) return RECON_ERR_INVALID;
        // (rows compacted — recon_graph_rows_compact ran in front on this stream: the tables name ROWS)
        const bool rows = g->n_rows > 0 && g->rowptr_rows;
        hipLaunchKernelGGL((k_hub_scan<true>), dim3(1), dim3(1024), 0, as_stream(stream), rows ? g->rowptr_rows : g->rowptr_dst, rows ? g->n_rows : g->N, g->hub_chunk,
                           nullptr, g->hub_node, g->hub_ptr, reinterpret_cast<int4*>(g->piece));
    }
    if (g->n_hub_src > 0) {
        if (!g->hub_node_src || !g->hub_ptr_src || !g->piece_src || (reinterpret_cast<uintptr_t>(g->piece_src) & 15)) return RECON_ERR_INVALID;
        hipLaunchKernelGGL((k_hub_scan<true>), dim3(1), dim3(1024), 0, as_stream(stream), g->rowptr_src, g->N, g->hub_chunk, nullptr,
                           g->hub_node_src, g->hub_ptr_src, reinterpret_cast<int4*>(g->piece_src));
    }
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

// destination side, per piece and head: the partial sums of the source and relation parts (F + R) and of Z, Zk (forward; the backward
// needs H per piece); source side, per piece: a g_x row and H sums of g_sigma.  One scratch serves both: the walks are stream ordered.
extern "C" size_t recon_graph_hub_ws_floats(const recon_graph* g, int32_t F, int32_t R, int32_t H) {
    if (!g || F <= 0 || R <= 0 || H <= 0) return 0;
    const size_t d = g->n_piece > 0 ? static_cast<size_t>(g->n_piece) * H * (static_cast<size_t>(F) + R + 2) : 0;
    const size_t s = g->n_piece_src > 0 ? static_cast<size_t>(g->n_piece_src) * (static_cast<size_t>(F) + H) : 0;
    return d > s ? d : s;
}

extern "C" size_t recon_graph_workspace_bytes(int32_t N, int32_t E) {
    (void)N;
    const size_t e = align_up(static_cast<size_t>(E > 0 ? E : 1) * sizeof(int32_t), 256);
    const size_t nblocks = static_cast<size_t>(ceil_div64(E > 0 ? E : 1, kTile));
    return 4 * e + 2 * align_up(256 * nblocks * sizeof(int32_t), 256) + 256;     // two histograms (the second pass' is counted while the first pass scatters) + the hub counts
}
static int32_t* graph_ws_counts(void* workspace, int32_t N, int32_t E) {
    return reinterpret_cast<int32_t*>(static_cast<char*>(workspace) + recon_graph_workspace_bytes(N, E) - 256);
}

extern "C" int recon_graph_build(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace,
                                 size_t workspace_bytes, recon_stream_t stream) {
    return recon_graph_build_checked(edge_dst, edge_src, g, workspace, workspace_bytes, nullptr, stream);
}

static int graph_build(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace, size_t workspace_bytes, int32_t* bad, int32_t chunk,
                       recon_stream_t stream);
extern "C" int recon_graph_build_checked(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace,
                                         size_t workspace_bytes, int32_t* bad, recon_stream_t stream) {
    return graph_build(edge_dst, edge_src, g, workspace, workspace_bytes, bad, 0, stream);
}
// The build with the hub-table sizes for `chunk` counted on the way (its last launch, or one k_hub_scan launch behind the single-launch
// build): they stay in the workspace and recon_graph_hubs_read() fetches them — a launch less per build than build + hubs_count.
extern "C" int recon_graph_build_counted(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace, size_t workspace_bytes,
                                         int32_t* bad, int32_t chunk, recon_stream_t stream) {
    if (chunk <= 0 || !g || !g->rowptr_src) return RECON_ERR_INVALID;
    return graph_build(edge_dst, edge_src, g, workspace, workspace_bytes, bad, chunk, stream);
}
extern "C" int recon_graph_hubs_read(const recon_graph* g, void* workspace, int32_t* counts, const int32_t* bad, int32_t* bad_host, recon_stream_t stream) {
    return recon_graph_counts_read(g, workspace, counts, nullptr, bad, bad_host, stream);
}
extern "C" int recon_graph_counts_read(const recon_graph* g, void* workspace, int32_t* counts, int32_t* live_rows, const int32_t* bad, int32_t* bad_host,
                                       recon_stream_t stream) {
    if (!g || !counts || !workspace || g->N < 0 || g->E < 0 || (bad && !bad_host)) return RECON_ERR_INVALID;
    hipStream_t st = as_stream(stream);
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    if (bad_host) *bad_host = 0;
    if (live_rows) *live_rows = 0;
    if (g->E > 0) {                                                      // sizes, flag and the count of rows with edges in ONE copy (the build's last launch put them side by side)
        int32_t six[6] = {0, 0, 0, 0, 0, 0};
        if (hipMemcpyAsync(six, graph_ws_counts(workspace, g->N, g->E), 6 * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess) return RECON_ERR_LAUNCH;
        if (hipStreamSynchronize(st) != hipSuccess) return RECON_ERR_LAUNCH;
        for (int i = 0; i < 4; ++i) counts[i] = six[i];
        if (bad) *bad_host = six[4];
        if (live_rows) *live_rows = six[5];
    } else if (hipStreamSynchronize(st) != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}
extern "C" int recon_graph_rows_compact(const recon_graph* g, recon_stream_t stream) {
    if (!g || g->N <= 0 || !g->rowptr_dst || g->n_rows <= 0 || g->n_rows > g->N || !g->row_node || !g->rowptr_rows || !g->node_row) return RECON_ERR_INVALID;
    hipLaunchKernelGGL(k_rows_compact, dim3(1), dim3(1024), 0, as_stream(stream), g->rowptr_dst, g->N, g->row_node, g->rowptr_rows, g->node_row);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
static int graph_build(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace, size_t workspace_bytes, int32_t* bad, int32_t chunk,
                       recon_stream_t stream) {
    if (!g || g->N < 0 || g->E < 0) return RECON_ERR_INVALID;
    if (!g->rowptr_dst) return RECON_ERR_INVALID;
    const bool with_src = g->rowptr_src != nullptr;                 // NULL: destination CSR only (row sums need no source view)
    const int32_t N = g->N, E = g->E;
    hipStream_t st = as_stream(stream);
    if (E == 0) {
        if (hipMemsetAsync(g->rowptr_dst, 0, sizeof(int32_t) * (N + 1), st) != hipSuccess ||
            (with_src && hipMemsetAsync(g->rowptr_src, 0, sizeof(int32_t) * (N + 1), st) != hipSuccess)) return RECON_ERR_LAUNCH;
        return RECON_OK;
    }
    if (!edge_dst || !g->eid || !g->dst || !workspace) return RECON_ERR_INVALID;
    if (with_src && (!edge_src || !g->src || !g->slot_by_src)) return RECON_ERR_INVALID;
    if (workspace_bytes < recon_graph_workspace_bytes(N, E)) return RECON_ERR_WORKSPACE;
    const size_t e = align_up(static_cast<size_t>(E) * sizeof(int32_t), 256);
    char* w = static_cast<char*>(workspace);
    SortWs ws;
    ws.kA = reinterpret_cast<int32_t*>(w);
    ws.kB = reinterpret_cast<int32_t*>(w + e);
    ws.vA = reinterpret_cast<int32_t*>(w + 2 * e);
    ws.vB = reinterpret_cast<int32_t*>(w + 3 * e);
    ws.hist = reinterpret_cast<int32_t*>(w + 4 * e);
    ws.nblocks = static_cast<int32_t>(ceil_div64(E, kTile));
    ws.hist2 = reinterpret_cast<int32_t*>(w + 4 * e + align_up(256 * static_cast<size_t>(ws.nblocks) * sizeof(int32_t), 256));
    const dim3 nb(static_cast<unsigned>(ceil_div64(N + 1, 256)));
    if (E <= kSmallE && N <= kSmallN && recon::cfg_char(recon::CFG_GRAPH_SMALL) != '0') {       // one launch (k_graph_build_small)
        const size_t lds = (4096 + 64 + static_cast<size_t>(N) + 1) * sizeof(int32_t);
        const void* kern = with_src ? reinterpret_cast<const void*>(k_graph_build_small<true>) : reinterpret_cast<const void*>(k_graph_build_small<false>);
        if (lds > 48 * 1024 && hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return RECON_ERR_LAUNCH;
        if (with_src) hipLaunchKernelGGL((k_graph_build_small<true>), dim3(1), dim3(1024), lds, st, edge_dst, edge_src, N, E, g->rowptr_dst, g->eid, g->src, g->dst,
                                         g->rowptr_src, g->slot_by_src, ws.kA, ws.vA, ws.kB, ws.vB, bad);
        else hipLaunchKernelGGL((k_graph_build_small<false>), dim3(1), dim3(1024), lds, st, edge_dst, edge_src, N, E, g->rowptr_dst, g->eid, g->src, g->dst,
                                g->rowptr_src, g->slot_by_src, ws.kA, ws.vA, ws.kB, ws.vB, bad);
        if (chunk > 0)
            hipLaunchKernelGGL((k_hub_scan<false>), dim3(2), dim3(1024), 0, st, g->rowptr_dst, N, chunk, graph_ws_counts(workspace, N, E), bad,
                               graph_ws_counts(workspace, N, E), reinterpret_cast<int4*>(g->rowptr_src));
        RECON_CHECK_LAUNCH();
        return RECON_OK;
    }

    // destination CSR: stable sort of (dst, edge column)
    hipLaunchKernelGGL((k_keys_hist<false>), dim3(ws.nblocks), dim3(kThreads), 0, st, edge_dst, nullptr, E, nullptr, ws.kA, ws.vA, ws.hist, ws.nblocks, N, bad, ws.hist2,
                       nullptr, nullptr, chunk > 0 ? graph_ws_counts(workspace, N, E) : nullptr);
    int32_t *ks, *vs;
    int rc = radix_sort_pairs(ws, E, N, &ks, &vs, st, g->dst, g->eid, true);
    if (rc != RECON_OK) return rc;
    if (!with_src) {
        hipLaunchKernelGGL(k_rowptr_lower_bound, nb, dim3(256), 0, st, g->dst, E, N, g->rowptr_dst);
        RECON_CHECK_LAUNCH();
        return RECON_OK;
    }
    // source CSC over CSR slots: stable sort of (src of slot, slot); the destination row pointers ride in the same launch (trailing blocks)
    static_assert(kThreads == 256, "the row-pointer blocks of k_keys_hist cover 256 rows each");
    hipLaunchKernelGGL((k_keys_hist<true>), dim3(ws.nblocks + nb.x), dim3(kThreads), 0, st, edge_src, g->eid, E, g->src, ws.kA, ws.vA, ws.hist, ws.nblocks, N, bad,
                       ws.hist2, g->dst, g->rowptr_dst);
    rc = radix_sort_pairs(ws, E, N, &ks, &vs, st, nullptr, g->slot_by_src, true);
    if (rc != RECON_OK) return rc;
    if (chunk > 0) hipLaunchKernelGGL(k_rowptr_lower_bound_count, nb, dim3(256), 0, st, ks, E, N, g->rowptr_src, g->rowptr_dst, chunk, graph_ws_counts(workspace, N, E), bad);
    else hipLaunchKernelGGL(k_rowptr_lower_bound, nb, dim3(256), 0, st, ks, E, N, g->rowptr_src);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_slot_index(const int64_t* index, const int32_t* eid, int32_t E, int32_t* out32, int64_t* out64, recon_stream_t stream) {
    if (E < 0 || (E > 0 && (!index || !eid || !out32 || !out64))) return RECON_ERR_INVALID;
    if (E == 0) return RECON_OK;
    hipLaunchKernelGGL(k_slot_index, dim3(static_cast<unsigned>(ceil_div64(E, 256))), dim3(256), 0, as_stream(stream), index, eid, E, out32, out64);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
