// K3 — graph preparation for the GAT hot path: int64 COO edge list -> destination-CSR and
// source-CSC index arrays (int32), via a hand-written stable LSD radix sort (8-bit digits,
// wave-ballot ranking).  Replaces the implicit coalesce / sort of torch.sparse.sum at
// GAT/layers.py:56-58.  Integer work, HBM/latency bound; cached per edge tensor by the caller.
#include "recon_common.h"

namespace {

constexpr int kTile = 1024;      // items per block per pass
constexpr int kThreads = 256;    // 4 waves, each ranks 256 consecutive items

__global__ void k_convert_keys(const int64_t* __restrict__ in, int32_t* __restrict__ keys,
                               int32_t* __restrict__ vals, int32_t n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = static_cast<int32_t>(in[i]); vals[i] = i; }
}

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

// single-block exclusive scan (in place); n is at most a few hundred thousand counters
__global__ void __launch_bounds__(1024) k_scan_exclusive(int32_t* __restrict__ data, int32_t n) {
    __shared__ int32_t sums[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = t * chunk, hi = min(lo + chunk, n);
    int32_t s = 0;
    for (int i = lo; i < hi; ++i) s += data[i];
    sums[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan
        int32_t v = (t >= off) ? sums[t - off] : 0;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    int32_t run = sums[t] - s;
    for (int i = lo; i < hi; ++i) { int32_t v = data[i]; data[i] = run; run += v; }
}

__global__ void __launch_bounds__(kThreads) k_radix_scatter(const int32_t* __restrict__ keys_in,
                                                            const int32_t* __restrict__ vals_in,
                                                            int32_t* __restrict__ keys_out,
                                                            int32_t* __restrict__ vals_out, int32_t n, int shift,
                                                            const int32_t* __restrict__ blockhist_scanned,
                                                            int32_t nblocks) {
    __shared__ int32_t wcount[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += kThreads) (&wcount[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * kTile + w * 256;
    int32_t key[4], val[4], rank[4];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = base + j * 64 + lane;
        const bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0;
        val[j] = valid ? vals_in[i] : 0;
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
        int32_t run = blockhist_scanned[t * nblocks + blockIdx.x];
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

__global__ void k_gather_src(const int64_t* __restrict__ edge_src, const int32_t* __restrict__ eid, int32_t E,
                             int32_t* __restrict__ src_out, int32_t* __restrict__ iota) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < E) { src_out[k] = static_cast<int32_t>(edge_src[eid[k]]); iota[k] = k; }
}

__global__ void k_copy_i32(const int32_t* __restrict__ in, int32_t* __restrict__ out, int32_t n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

int bits_for(int32_t n) { int b = 1; while (b < 31 && (1 << b) < n) ++b; return b; }

struct SortWs { int32_t *kA, *kB, *vA, *vB, *hist; int32_t nblocks; };

// stable sort of (kA, vA); result ends up in (*kout, *vout) which point into the ping-pong buffers
int radix_sort_pairs(SortWs& ws, int32_t n, int32_t key_range, int32_t** kout, int32_t** vout, hipStream_t st) {
    const int passes = (bits_for(key_range) + 7) / 8;
    int32_t *ki = ws.kA, *vi = ws.vA, *ko = ws.kB, *vo = ws.vB;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_radix_hist, dim3(ws.nblocks), dim3(kThreads), 0, st, ki, n, shift, ws.hist, ws.nblocks);
        hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, ws.hist, 256 * ws.nblocks);
        hipLaunchKernelGGL(k_radix_scatter, dim3(ws.nblocks), dim3(kThreads), 0, st, ki, vi, ko, vo, n, shift, ws.hist,
                           ws.nblocks);
        int32_t* t;
        t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    *kout = ki; *vout = vi;
    if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH;
    return RECON_OK;
}

}  // namespace

extern "C" size_t recon_graph_workspace_bytes(int32_t N, int32_t E) {
    (void)N;
    const size_t e = align_up(static_cast<size_t>(E > 0 ? E : 1) * sizeof(int32_t), 256);
    const size_t nblocks = static_cast<size_t>(ceil_div64(E > 0 ? E : 1, kTile));
    return 4 * e + align_up(256 * nblocks * sizeof(int32_t), 256);
}

extern "C" int recon_graph_build(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace,
                                 size_t workspace_bytes, recon_stream_t stream) {
    if (!g || g->N < 0 || g->E < 0) return RECON_ERR_INVALID;
    if (!g->rowptr_dst || !g->rowptr_src) return RECON_ERR_INVALID;
    const int32_t N = g->N, E = g->E;
    hipStream_t st = as_stream(stream);
    if (E == 0) {
        if (hipMemsetAsync(g->rowptr_dst, 0, sizeof(int32_t) * (N + 1), st) != hipSuccess ||
            hipMemsetAsync(g->rowptr_src, 0, sizeof(int32_t) * (N + 1), st) != hipSuccess) return RECON_ERR_LAUNCH;
        return RECON_OK;
    }
    if (!edge_dst || !edge_src || !g->eid || !g->src || !g->dst || !g->slot_by_src || !workspace) return RECON_ERR_INVALID;
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
    const dim3 eb(static_cast<unsigned>(ceil_div64(E, 256))), nb(static_cast<unsigned>(ceil_div64(N + 1, 256)));

    // destination CSR: stable sort of (dst, edge column)
    hipLaunchKernelGGL(k_convert_keys, eb, dim3(256), 0, st, edge_dst, ws.kA, ws.vA, E);
    int32_t *ks, *vs;
    int rc = radix_sort_pairs(ws, E, N, &ks, &vs, st);
    if (rc != RECON_OK) return rc;
    hipLaunchKernelGGL(k_copy_i32, eb, dim3(256), 0, st, ks, g->dst, E);
    hipLaunchKernelGGL(k_copy_i32, eb, dim3(256), 0, st, vs, g->eid, E);
    hipLaunchKernelGGL(k_rowptr_lower_bound, nb, dim3(256), 0, st, g->dst, E, N, g->rowptr_dst);
    // source CSC over CSR slots: stable sort of (src of slot, slot)
    hipLaunchKernelGGL(k_gather_src, eb, dim3(256), 0, st, edge_src, g->eid, E, g->src, ws.vA);
    hipLaunchKernelGGL(k_copy_i32, eb, dim3(256), 0, st, g->src, ws.kA, E);
    rc = radix_sort_pairs(ws, E, N, &ks, &vs, st);
    if (rc != RECON_OK) return rc;
    hipLaunchKernelGGL(k_copy_i32, eb, dim3(256), 0, st, vs, g->slot_by_src, E);
    hipLaunchKernelGGL(k_rowptr_lower_bound, nb, dim3(256), 0, st, ks, E, N, g->rowptr_src);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
