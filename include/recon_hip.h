/*
 * recon_hip.h — C ABI of librecon_hip.so: MI355X (gfx950) kernels for RECON's graph-context
 * aggregation hot path.
 *
 * The reference (ansonb/RECON) is pure Python: it has no FFI of its own.  Its "plugin boundary"
 * for this path is the Python class surface of GAT/layers.py and models/layers.py; the drop-in
 * modules in recon_amd/ keep that surface and bind the entry points below with ctypes
 * (INTEGRATION.md shows the stub).  Each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host";
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library allocates nothing
 *     and keeps no mutable global state;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t); it never synchronises;
 *   - return value: 0 = RECON_OK, negative = error (recon_error_string()); no C++ exception
 *     crosses this boundary;
 *   - float tensors are fp32, row-major, contiguous unless a leading dimension is given;
 *     indices handed over by the reference API are int64 (torch.LongTensor), internal index
 *     arrays are int32.
 */
#ifndef RECON_HIP_H
#define RECON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECON_ABI_VERSION 2

enum {
    RECON_OK = 0,
    RECON_ERR_INVALID = -1,     /* null pointer / negative size / inconsistent dims   */
    RECON_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are instantiated for */
    RECON_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after a launch      */
    RECON_ERR_WORKSPACE = -4    /* workspace smaller than recon_*_workspace_bytes()    */
};

typedef void* recon_stream_t; /* hipStream_t */

int recon_version(void);
const char* recon_error_string(int code);

/* Run-time switches (recon_amd/csrc/config.hip): one table, filled from the environment variables of the same names (RECON_GEMM_CFG,
 * RECON_PROP_FWD ... — INTEGRATION.md lists them) the first time the library needs one; every switch chooses between kernels that
 * compute the same result.  recon_config_set overrides one entry (value NULL: unset), recon_config_get returns the current value or NULL.
 * Call them between launches, from the launching thread.  The reference has no counterpart (its only switch is the module-level CUDA flag,
 * GAT/layers.py:10). */
/* Optional NaN word of a device (int32, device memory, zeroed by the caller; NULL: off): the attention kernels store 1 into it when a row sum
 * of attention weights comes out NaN or infinite — the condition behind the reference's three `assert not torch.isnan(...)` per layer call
 * (GAT/layers.py:147, :167, :172), without their three host round trips: nothing waits for the word, the caller reads it when it likes. */
int recon_set_nan_flag(int32_t device, int32_t* flag);
int recon_config_set(const char* name, const char* value);
const char* recon_config_get(const char* name);

/* --------------------------------------------------------------------------------------------
 * K3  graph preparation: COO edge list -> destination-CSR + source-CSC views.
 * Replaces the implicit coalesce/sort inside torch.sparse.sum at GAT/layers.py:56-58 (and the
 * edge concatenation at GAT/layers.py:124-127, done by the caller on the index tensors).
 * Stable: edges of one destination keep their original relative order; duplicates are kept.
 * ------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t N;                  /* nodes                                                     */
    int32_t E;                  /* edges (1-hop + n-hop)                                     */
    int32_t* rowptr_dst;        /* [N+1] CSR over destinations (edge[0,:], the segment id)   */
    int32_t* eid;               /* [E]   CSR slot -> original edge column                    */
    int32_t* src;               /* [E]   source node (edge[1,:]) of each CSR slot            */
    int32_t* dst;               /* [E]   destination node of each CSR slot                   */
    int32_t* rowptr_src;        /* [N+1] CSC over sources; NULL at build: destination CSR only (src, slot_by_src unused) */
    int32_t* slot_by_src;       /* [E]   CSC position -> CSR slot                            */
    /* Hub rows (optional; all zero = every destination row is walked by one wave, whatever its length).  The aggregate-then-
     * project edge kernels give one wavefront to one destination node; a node with hundreds of in-edges (power-law graphs)
     * is then one long serial chain.  With these tables a row of more than hub_chunk slots is cut into pieces of at most
     * hub_chunk slots, each walked by its own wavefront, and summed in table order by a second small launch: results do not
     * depend on scheduling.  Filled by recon_graph_hubs_count() + recon_graph_hubs_fill().                              */
    int32_t hub_chunk;          /* slots per piece (RECON_HUB_CHUNK), 0: no splitting                                    */
    int32_t n_hub;              /* destination nodes with more than hub_chunk slots                                      */
    int32_t n_piece;            /* their pieces                                                                          */
    int32_t* hub_node;          /* [n_hub]     node id, ascending                                                        */
    int32_t* hub_ptr;           /* [n_hub + 1] first piece of each hub                                                   */
    int32_t* piece;             /* [n_piece][4] (node, first slot, end slot, hub ordinal), 16-byte aligned               */
    /* the same for the source-side walk of the backward (CSC rows: a node that is the neighbour in many edges) */
    int32_t n_hub_src, n_piece_src;
    int32_t* hub_node_src;      /* [n_hub_src]                                                                           */
    int32_t* hub_ptr_src;       /* [n_hub_src + 1]                                                                       */
    int32_t* piece_src;         /* [n_piece_src][4] (node, first CSC position, end position, hub ordinal)                */
    float* hub_ws;              /* scratch of the piece partial sums, reused by every call on this graph (calls on one graph
                                   are stream ordered): at least recon_graph_hub_ws_floats() floats for the widest layer   */
    int64_t hub_ws_floats;
    /* Row compaction (optional; n_rows = 0: every node is a row).  A knowledge-graph batch (GAT/main.py:478-516) aggregates into the
     * batch's ~128 entities of a 14 541-entity table: all other destination rows are empty, their layer output is exactly 0
     * (GAT/layers.py:152-158: 0 / 1e-12, elu(0) = 0) and so is everything they contribute backward.  With these tables the
     * aggregate-then-project kernels run their node-parallel stages (V, the three GEMMs, g_V, the destination walk of the backward)
     * over the n_rows rows WITH edges; `out` / `grad_out` of recon_gat_atp_args then hold n_rows rows / are read through row_node.
     * The hub tables of the destination side name ROWS when these are set.  Filled by recon_graph_rows_compact().            */
    int32_t n_rows;             /* destination nodes with at least one in-edge                                              */
    int32_t* row_node;          /* [n_rows]     node id of each row, ascending                                              */
    int32_t* rowptr_rows;       /* [n_rows + 1] first CSR slot of each row (rowptr_dst without the empty rows)              */
    int32_t* node_row;          /* [N]          row of each node, -1: none                                                  */
} recon_graph;

#define RECON_HUB_CHUNK 64

size_t recon_graph_workspace_bytes(int32_t N, int32_t E);
/* Hub tables of a built graph.  count: one pass over rowptr_dst and one over rowptr_src, synchronises the stream and returns the
 * table sizes counts[4] = (n_hub, n_piece, n_hub_src, n_piece_src); all zero: nothing to do.  fill: the caller has set hub_chunk
 * and the four sizes to what count returned and the six table pointers to device arrays of those sizes (a side without hubs may
 * stay NULL).  hub_ws may be set (or grown) at any time before a layer call. */
int recon_graph_hubs_count(const recon_graph* g, int32_t chunk, void* workspace /* device, 16 bytes */, int32_t* counts /* host [4] */,
                           recon_stream_t stream);
int recon_graph_hubs_fill(const recon_graph* g, recon_stream_t stream);
/* The same two calls with the build's range check riding along: recon_graph_build_checked sets *bad (device int32, zero on entry) when an
 * id lies outside [0, N) — the tables are then garbage and must not be used; recon_graph_hubs_count_checked copies the flag to *bad_host in
 * the round trip it makes anyway (the reference fails on such input with an index error: GAT/layers.py:56, torch.sparse_coo_tensor). */
int recon_graph_build_checked(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace, size_t workspace_bytes,
                              int32_t* bad, recon_stream_t stream);
int recon_graph_hubs_count_checked(const recon_graph* g, int32_t chunk, void* workspace, int32_t* counts, const int32_t* bad, int32_t* bad_host,
                                   recon_stream_t stream);
/* Build + hub-table sizes in one chain: recon_graph_build_counted is recon_graph_build_checked whose last launch also counts, for rows
 * of more than `chunk` slots, the four sizes recon_graph_hubs_count would return (they stay in the workspace: one launch less per build —
 * a build is a chain of small dependent launches); recon_graph_hubs_read copies them (and the range-check flag, which that last launch
 * stores behind the sizes: `bad` given to recon_graph_hubs_read only says that *bad_host is wanted — it must be the build's) to the host in
 * one copy, with the one synchronisation of the pair.  The graph needs both views (rowptr_src set); `workspace` is the build's. */
int recon_graph_build_counted(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g, void* workspace, size_t workspace_bytes,
                              int32_t* bad, int32_t chunk, recon_stream_t stream);
int recon_graph_hubs_read(const recon_graph* g, void* workspace, int32_t* counts /* host [4] */, const int32_t* bad, int32_t* bad_host,
                          recon_stream_t stream);
/* floats of hub_ws one KB-GAT layer call (forward or backward) on this graph needs */
size_t recon_graph_hub_ws_floats(const recon_graph* g, int32_t F, int32_t R, int32_t H);
/* recon_graph_hubs_read that also returns the number of destination rows with edges (*live_rows; counted by recon_graph_build_counted's
 * last launch): the caller decides on row compaction from it.  recon_graph_rows_compact: g->n_rows = that count, row_node / rowptr_rows /
 * node_row point to device arrays of n_rows / n_rows + 1 / N int32; one launch; call it BEFORE recon_graph_hubs_fill (whose destination
 * tables are then built over the rows). */
int recon_graph_counts_read(const recon_graph* g, void* workspace, int32_t* counts /* host [4] */, int32_t* live_rows /* host, may be NULL */,
                            const int32_t* bad, int32_t* bad_host, recon_stream_t stream);
int recon_graph_rows_compact(const recon_graph* g, recon_stream_t stream);
/* out[n, :] = rows[node_row[n], :] or 0 (node_row[n] < 0): the layer output of a row-compacted graph back in node order. */
int recon_rows_expand(const float* rows, int32_t ld_rows, const int32_t* node_row, int32_t N, int32_t width, float* out, int32_t ld_out,
                      recon_stream_t stream);

/* edge_dst / edge_src: the two rows of the reference's int64 [2,E] edge tensor (row 0 =
 * aggregation target, row 1 = neighbour; GAT/create_batch.py:429-433).  Returns RECON_ERR_INVALID
 * if N or E overflow int32.  Node ids outside [0,N) are undefined behaviour (as in the reference). */
int recon_graph_build(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g,
                      void* workspace, size_t workspace_bytes, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * G1-G3  SpecialSpmmFinal: out[r,:] = sum_{e: edge[0,e]==r} edge_w[e,:]   (GAT/layers.py:54-64)
 *        backward: grad_edge_w[e,:] = grad_out[edge[0,e],:]               (GAT/layers.py:67-79)
 * ------------------------------------------------------------------------------------------*/
size_t recon_spmm_rowsum_workspace_floats(int32_t E, int32_t out_features);
int recon_spmm_rowsum_fwd(const recon_graph* g, const float* edge_w /*[E,out_features]*/,
                          int32_t out_features, float* out /*[N,out_features]*/,
                          float* workspace /* recon_spmm_rowsum_workspace_floats() floats */, recon_stream_t stream);
/* the same with `row_mod` > 0: slot e reads row (eid[e] % row_mod) of edge_w — several keys share one value row (the gradient of
 * table[i0] + table[i1]: keys (i0 | i1), E value rows) */
int recon_spmm_rowsum_mod_fwd(const recon_graph* g, const float* edge_w, int32_t out_features, int32_t row_mod, float* out,
                              float* workspace, recon_stream_t stream);
int recon_spmm_rowsum_bwd(const int64_t* edge_dst /*[E]*/, int64_t E, const float* grad_out /*[N,out]*/,
                          int32_t out_features, float* grad_edge_w /*[E,out]*/, recon_stream_t stream);
/* out[k,:] = table[idx2[k,0],:] + table[idx2[k,1],:] — the n-hop relation embedding `relation_embed[edge_type_nhop[:, 0]] +
 * relation_embed[edge_type_nhop[:, 1]]` (GAT/models.py:64-65, 80-81) in one pass; the indices are the caller's to validate
 * (0 <= idx2 < table rows); its gradient is recon_spmm_rowsum_mod_fwd over the keys (idx2[:,0] | idx2[:,1]) */
int recon_gather_rows_pair_fwd(const float* table /*[rows,width]*/, const int64_t* idx2 /*[E,2]*/, int64_t E, int32_t width,
                               float* out /*[E,width]*/, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * G4  SpGraphAttentionLayer.forward for H heads that share inputs (GAT/layers.py:111-178; the
 *     head loop is GAT/models.py:71-72).  H = 1 is the single reference layer.
 *
 *   m_e   = A_dst x[dst_e] + A_src x[src_e] + A_rel r_e        a = [A_dst | A_src | A_rel]
 *   s_e   = a_2 . m_e ;  w_e = exp(-leakyrelu_alpha(s_e))      (no max subtraction)
 *   Z_i   = sum_{e: dst_e = i} w_e ;  Z_i == 0 -> 1e-12
 *   out_i = act( sum_e k_e w_e m_e / Z_i )                      k = dropout factors (after the row sum)
 *
 * Internals (caller-allocated, reused by the backward):
 *   P  [2][H][N][D]  projected node features (dst half, src half), head-major
 *   Q  [H][E][D]     projected edge features in CSR-slot order, head-major
 * ------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t N, E;               /* must equal graph->N, graph->E                             */
    int32_t F, R, D, H;         /* in_features, nrela_dim, out_features per head, heads      */
    int32_t concat;             /* 1: ELU on the output (GAT/layers.py:173-175)              */
    float alpha;                /* LeakyReLU negative slope                                  */
    const float* x;             /* [N,F]                                                     */
    const float* edge_embed;    /* [E,R] rows in ORIGINAL edge order (1-hop rows then n-hop) */
    const float* a;             /* [H,D,2F+R]                                                */
    const float* a_2;           /* [H,D]                                                     */
    const float* keep;          /* [H,E] dropout factors in CSR-slot order, or NULL (eval)   */
    float* P;                   /* [2,H,N,D] workspace / saved                               */
    float* Q;                   /* [H,E,D]   workspace / saved                               */
    float* sigma;               /* [H,E] saved scores s_e (CSR-slot order); NULL iff Z NULL  */
    float* Z;                   /* [H,N] saved clamped row sums; NULL = inference call       */
    float* out;                 /* [N, ld_out] ; head h writes columns [h*D, (h+1)*D)        */
    int32_t ld_out;             /* >= H*D                                                    */
} recon_gat_fwd_args;

int recon_gat_fwd(const recon_graph* g, const recon_gat_fwd_args* args, recon_stream_t stream);

/* the two stages of recon_gat_fwd, exported for profiling / tests */
int recon_gat_project(const recon_graph* g, const recon_gat_fwd_args* args, recon_stream_t stream); /* K4: MFMA */
int recon_gat_edge_fwd(const recon_graph* g, const recon_gat_fwd_args* args, recon_stream_t stream); /* K1: HBM  */

/* --------------------------------------------------------------------------------------------
 * K2  backward of recon_gat_fwd (autograd of GAT/layers.py:111-178 incl. the custom backward at
 *     :67-79; formulas in SURVEY.md appendix B).
 * ------------------------------------------------------------------------------------------*/
typedef struct {
    recon_gat_fwd_args fwd;     /* same tensors as the forward call (P, Q, sigma, Z, out filled) */
    const float* grad_out;      /* [N, ld_gout]                                              */
    int32_t ld_gout;
    float* Gm;                  /* [H,E,D]   workspace: d loss / d m_e, CSR-slot order       */
    float* gP;                  /* [2,H,N,D] workspace: d loss / d P                         */
    float* partial;             /* workspace, recon_gat_bwd_partial_floats() floats          */
    float* g_x;                 /* [N,F]      or NULL                                        */
    float* g_edge_embed;        /* [E,R] original edge order, or NULL                        */
    float* g_a;                 /* [H,D,2F+R] or NULL                                        */
    float* g_a_2;               /* [H,D]      or NULL                                        */
} recon_gat_bwd_args;

size_t recon_gat_bwd_partial_floats(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H);
int recon_gat_bwd(const recon_graph* g, const recon_gat_bwd_args* args, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * G4'  the same layer in "aggregate, then project" order (csrc/gat_atp.hip).  Aggregation and
 *      projection are both linear, so  h_i = a . V_i  with
 *        V_i = [ x_i Zk_i/Z_i ; sum_e k_e w_e x[src_e] / Z_i ; sum_e k_e w_e r_e / Z_i ]   (2F+R per head)
 *      and the scores need only u = a_2^T a:  s_e = u_dst.x[dst_e] + u_src.x[src_e] + u_rel.r_e.
 *      Half the MFMA work of recon_gat_fwd when E = 4N and no [E, H*D] intermediate; results differ
 *      from recon_gat_fwd / the reference in fp32 summation order only.  recon_gat_atp_supported()
 *      tells whether the shape is instantiated (F, R even; 2F+R-wide rows fit the register budget).
 * ------------------------------------------------------------------------------------------*/
enum { RECON_SPLIT_BF16X3 = 0, RECON_SPLIT_F16X2 = 2 };

typedef struct {
    int32_t N, E;               /* must equal graph->N, graph->E                             */
    int32_t F, R, D, H;
    int32_t concat;
    float alpha;
    const float* x;             /* [N,F]                                                     */
    const float* edge_embed;    /* [E,R] original edge order                                 */
    const float* a;             /* [H,D,2F+R]                                                */
    const float* a_2;           /* [H,D]                                                     */
    const float* keep;          /* [E,H] dropout factors, CSR-slot order, or NULL            */
    float* u;                   /* [H,2F+R]  workspace / saved: a_2^T a                      */
    float* c_node;              /* [N,2H]    workspace / saved: x.u_dst | x.u_src            */
    float* c_rel;               /* [E,H]     workspace / saved: r_e.u_rel, CSR-slot order    */
    float* V;                   /* [N,H,2F+R] workspace / saved                              */
    float* sigma;               /* [E,H] saved scores (CSR-slot order); NULL iff Z NULL      */
    float* Z;                   /* [N,H] saved clamped row sums; NULL = inference call       */
    float* Zk;                  /* [N,H] saved sum_e k_e w_e                                 */
    float* out;                 /* [N, ld_out]; with a row-compacted graph (recon_graph.n_rows > 0): [n_rows, ld_out], row r = node row_node[r] */
    int32_t ld_out;
    void* a_split;              /* recon_gat_atp_split_bytes() bytes, workspace / saved: the three   *
                                 * bfloat16 term planes of a and a^T for the split-precision GEMMs   *
                                 * (csrc/gemm_bx3.hip; filled by the scores stage, read by the       *
                                 * projection and by the backward's g_V product).  NULL = use the    *
                                 * fp32-MFMA GEMM for those products.                                */
    int32_t split_mode;         /* RECON_SPLIT_*: which split-precision family a_split is for.  With         *
                                 * RECON_SPLIT_F16X2 (csrc/gemm_hx2.hip; needs (2F+R) % 8 == 0, D % 8 == 0,  *
                                 * else the call falls back to BF16X3) the buffers keep their sizes but hold  *
                                 * HALF planes: a_split the two planes of a and a^T, V the two planes         *
                                 * [2][N*H][2F+R] of s_V * V instead of fp32 V (head-major rows; when keep is  *
                                 * NULL the first F columns exist for head 0 only — they are the same for all  *
                                 * heads — and V is an opaque workspace), gh_split the two planes of g_h        */
    float keep_max;             /* upper bound of the factors in `keep` (1/(1-p)); ignored when keep is NULL  */
    void* aux;                  /* RECON_SPLIT_F16X2: recon_hx2_aux_bytes() bytes, 256-byte aligned, workspace *
                                 * / saved: max-magnitude slots of a, x, edge_embed, grad_out (the per-tensor   *
                                 * power-of-two scales derive from them) and a page of zeros; zeroed by the     *
                                 * scores stage; the backward publishes grad_out's slots and re-zeroes them     *
                                 * in its last kernel, so it must be given the aux of the matching forward      */
    const int32_t* ee_index;    /* NULL: edge_embed holds one row per edge.  Else [E], CSR-slot order: edge_embed is a TABLE and the *
                                 * edge in slot k uses row ee_index[k] of it — `relation_embed[edge_type]` (GAT/models.py:156, :79)  *
                                 * read in place instead of materialised as E x R; n-hop edges index rows appended to the table.      *
                                 * The backward then writes g_edge_embed [E,R] in CSR-SLOT order (row k = gradient of the row slot k    *
                                 * used); summing those rows by ee_index gives the table's gradient (recon_spmm_rowsum_fwd).            */
    int32_t ee_rows;            /* rows of the table when ee_index is set (ignored otherwise).  c_rel then holds max(E, ee_rows) x H    *
                                 * floats: the score terms r.u_rel are computed once per table ROW and looked up per edge               */
    int32_t io_bf16;            /* 1: x [N,F] and edge_embed [E,R] are BFLOAT16 (same shapes, 8-byte aligned) and the forward's kernels   *
                                 * read them in place (BASELINE.json configs[4]: "mixed GAT+Propagation stack, bf16"; everything else     *
                                 * of the call is unchanged: fp32 arithmetic, fp32 out).  Forward calls only (scores, aggregate,          *
                                 * project), shapes of recon_gat_atp_bf16_io_supported(), no ee_index; the backward takes fp32 copies.    */
} recon_gat_atp_args;
int recon_gat_atp_bf16_io_supported(int32_t F, int32_t R, int32_t D, int32_t H);

size_t recon_gat_atp_split_bytes(int32_t F, int32_t R, int32_t D, int32_t H);
/* 1 if the RECON_SPLIT_F16X2 mode takes this shape ((2F+R) % 8 == 0, D % 8 == 0, plane offsets within 32 bits); given that,
 * 16-byte aligned a_split / V, 256-byte aligned aux and ld_out % 4 == 0 the call uses it, otherwise it falls back to BF16X3 */
int recon_gat_atp_f16x2_supported(int32_t F, int32_t R, int32_t D, int32_t H);
int recon_gat_atp_supported(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H);
int recon_gat_atp_fwd(const recon_graph* g, const recon_gat_atp_args* args, recon_stream_t stream);
/* the three stages of recon_gat_atp_fwd, exported for profiling / tests */
int recon_gat_atp_scores(const recon_graph* g, const recon_gat_atp_args* args, recon_stream_t stream);    /* u, c_node, c_rel */
int recon_gat_atp_aggregate(const recon_graph* g, const recon_gat_atp_args* args, recon_stream_t stream); /* K1': HBM       */
int recon_gat_atp_project(const recon_graph* g, const recon_gat_atp_args* args, recon_stream_t stream);   /* K4: MFMA       */

typedef struct {
    recon_gat_atp_args fwd;     /* same tensors as the forward call (u, V, sigma, Z, Zk, out filled)  */
    const float* grad_out;      /* [N, ld_gout] — all nodes, also with a row-compacted graph (rows are read through row_node)        */
    int32_t ld_gout;
    float* g_h;                 /* [N,H*D]    workspace (written when concat != 0 or the graph is row-compacted; unused, may be NULL, in F16X2 mode) */
    float* g_V;                 /* [N,H,2F+R] workspace: d loss / d V                                 */
    float* g_sigma;             /* [E,H]      workspace: d loss / d s_e                               */
    float* Gxs;                 /* [E,F]      workspace: per-edge gradient rows bound for x[src_e]    */
    float* gxd;                 /* [N,F]      workspace: destination-side part of d loss / d x        */
    float* Gs;                  /* [N,2H]     workspace: per-node sums of g_sigma (dst view | src view) */
    float* g_u;                 /* [H,2F+R]   workspace: d loss / d (a_2^T a)                          */
    float* q;                   /* [N,H]      workspace: g_h . h per (node, head)                      */
    float* partial;             /* workspace, recon_gat_atp_bwd_partial_floats() floats (split-K)     */
    float* partial2;            /* workspace, recon_gat_atp_bwd_partial2_floats() floats (skinny)     */
    float* g_x;                 /* [N,F]      or NULL                                                 */
    float* g_edge_embed;        /* [E,R] original edge order, or NULL                                 */
    float* g_a;                 /* [H,D,2F+R] or NULL (then g_a_2 must be NULL too)                    */
    float* g_a_2;               /* [H,D]      or NULL                                                 */
    void* gh_split;             /* recon_gat_atp_bwd_split_bytes() bytes, workspace: bfloat16 term planes of g_h for  *
                                 * the split-precision weight-gradient GEMM; NULL (or fwd.a_split NULL) = fp32 MFMA  */
    int32_t g_ee_bf16;          /* 1: g_edge_embed points to BFLOAT16 rows [E,R] — the edge pass rounds its fp32 sums once, where it stores  *
                                 * them (fwd.io_bf16 callers whose edge embeddings take a bfloat16 gradient: 116 MB written as fp32 and    *
                                 * re-read by a cast at BASELINE.json configs[4]).  Only where recon_gat_atp_bwd_gee_bf16_supported() says 1  */
} recon_gat_atp_bwd_args;
/* whether g_ee_bf16 = 1 is available for these widths: fwd.io_bf16 shapes whose heads are walked in ONE group per wave (a second head group
 * adds to the rows the first one stored: that sum has to stay fp32) */
int recon_gat_atp_bwd_gee_bf16_supported(int32_t F, int32_t R, int32_t D, int32_t H);

size_t recon_gat_atp_bwd_split_bytes(int32_t N, int32_t D, int32_t H);

size_t recon_gat_atp_bwd_partial_floats(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H);
size_t recon_gat_atp_bwd_partial2_floats(int32_t N, int32_t E, int32_t F, int32_t R, int32_t D, int32_t H);
int recon_gat_atp_bwd(const recon_graph* g, const recon_gat_atp_bwd_args* args, recon_stream_t stream);
/* the same backward as independently launchable phases (bit mask): PREPARE -> { INPUTS, WEIGHTS } -> FINISH.
 * INPUTS (g_V GEMM, edge pass, source pass, score-vector gradient: HBM bound) and WEIGHTS (g_a = g_h^T V:
 * MFMA bound) only read what PREPARE wrote and touch disjoint workspaces, so they may run on two streams. */
enum { RECON_ATP_BWD_PREPARE = 1, RECON_ATP_BWD_INPUTS = 2, RECON_ATP_BWD_WEIGHTS = 4, RECON_ATP_BWD_FINISH = 8,
       RECON_ATP_BWD_ALL = 15,
       /* modifier for data-parallel callers that reduce the weight gradient across ranks UNDER the INPUTS phase: with WEIGHTS, the
        * split-K second pass runs at once and leaves g_a = G = V^T g_h (the part of g_a that does not depend on INPUTS); with FINISH,
        * g_a is taken to hold G already and only  g_a += a_2 (x) g_u,  g_a_2 = a . g_u  are applied.  Both are linear in (G, g_u), so
        * mean over ranks of g_a = mean(G) + a_2 (x) mean(g_u): all-reduce G early, g_u (H x (2F+R) floats) after INPUTS, then FINISH. */
       RECON_ATP_BWD_EARLY_SUM = 16 };
int recon_gat_atp_bwd_phase(const recon_graph* g, const recon_gat_atp_bwd_args* args, int32_t phases,
                            recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * P1  block adjacency of the GP-GNN step (models/models.py:240-259; copies :450-469, :660-679,
 *     :898-917):  A[b, i*dd+r, j*dd+c] = T[b, e(i,j), r*dd+c] for i != j (e = row-major over ordered
 *     pairs, diagonal skipped) and identity[r,c] for i == j;  dd = 2*embedding_dim, S = n*dd.
 *     T is the per-pair transition tensor AFTER the non-linearity, [B, n(n-1), dd*dd].
 * ------------------------------------------------------------------------------------------*/
int recon_block_adjacency_fwd(const float* T, const float* identity, int32_t B, int32_t n, int32_t dd,
                              float* A /*[B,S,S]*/, recon_stream_t stream);
/* gT [B,n(n-1),dd*dd] and g_identity [dd,dd] (either may be NULL) from gA [B,S,S]; `workspace`
 * (recon_block_adjacency_bwd_workspace_floats() floats) holds the per-slice partial sums of g_identity, which
 * is reduced in a fixed order (required when g_identity is not NULL). */
size_t recon_block_adjacency_bwd_workspace_floats(int32_t B, int32_t n, int32_t dd);
int recon_block_adjacency_bwd(const float* gA, int32_t B, int32_t n, int32_t dd, float* gT, float* g_identity,
                              float* workspace, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * P2 / K5  L-hop gated propagation + head*tail gather (models/models.py:260-274; copies :470-485,
 *     :680-694, :918-932):  h^0 = h0;  h^l = act(A_l h^l-1)  per channel c;
 *     out[b, c, l*dd + x] = h^l[b, c, head[c,x]] * h^l[b, c, tail[c,x]].
 *     One launch runs all L hops with the channel states resident in LDS (MFMA 16x16x4 fp32).
 * ------------------------------------------------------------------------------------------*/
enum { RECON_ACT_LINEAR = 0, RECON_ACT_RELU = 1, RECON_ACT_TANH = 2 };
typedef struct {
    int32_t B, C, S, L, dd;         /* graphs, channels n(n-1), state size n*dd, hops, gather width      */
    int32_t act;                    /* RECON_ACT_*  (model_params.json "non-linear1")                    */
    const float* const* adj;        /* HOST array of L device pointers, each [B,S,S]                     */
    const float* h0;                /* [C,S] shared (h0_batch_stride = 0) or [B,C,S] (stride = C*S)       */
    int64_t h0_batch_stride;
    const int64_t* head_idx;        /* [C,dd] (idx_batch_stride = 0) or [B,C,dd]; values in [0,S)         */
    const int64_t* tail_idx;
    int64_t idx_batch_stride;
    float* out;                     /* [B,C,L*dd]                                                         */
    float* h_saved;                 /* [L,B,C,S] states after each hop (for the backward) or NULL         */
    const float* const* trans;      /* BLOCK MODE (or NULL): HOST array of L device pointers, each [B, n(n-1), dd*dd] — the  *
                                     * transition matrices AFTER the non-linearity; A_l is then never materialised: the     *
                                     * kernels read A_l[b, i*dd+r, j*dd+c] = trans[l][b, e(i,j), r, c] (identity[r,c] on the   *
                                     * diagonal blocks) in place, recon_block_adjacency_fwd's layout.  `adj` is ignored.    *
                                     * Only where recon_propagate_form() == 1 and dd == 16 (S = 16 n).                      */
    const float* identity;          /* [dd,dd], with trans                                                               */
    float* stats;                   /* [B, 2L+1] or NULL: per graph the max magnitudes of h^0..h^L and of   *
                                     * A_1..A_L.  Written by the forward when h_saved is given and          *
                                     * recon_propagate_form() == 1 (the two-term f16 kernels, whose         *
                                     * per-tensor scales in the backward come from these); the backward     *
                                     * runs its two-term form only when given the same buffer              */
    void* split_ws;                 /* recon_propagate_ws_bytes() bytes or NULL.  Wide states (160 < S <= 512, e.g. 32 nodes *
                                     * at 2d = 16): with it the forward runs the two-term f16 form of csrc/prop_hl.hip (A_l  *
                                     * pre-split into half terms per slice of graphs, 64-channel chunks per workgroup);      *
                                     * without it, the fp32 matrix-core form                                                 */
    int64_t split_ws_bytes;
} recon_prop_args;

int recon_propagate_fwd(const recon_prop_args* args, recon_stream_t stream);
/* Bit 0: the forward of this problem runs on the two-term f16 matrix-core kernels (csrc/prop_h.hip: S % 16 == 0, S <= 160, C <= 96,
 * aligned pointers, RECON_PROP_FWD unset or "h"); bit 1: so does the backward (its LDS image also fits); 0: fp32 matrix-core forms.
 * Block mode (`trans`) needs both bits when gradients are wanted.  The adjacency pointers need not be set for this query. */
int recon_propagate_form(const recon_prop_args* args);
/* Bytes of `split_ws` that let recon_propagate_fwd run the wide-state two-term form for this problem (B, S, L; block mode or not),
 * 0 where that form does not exist (S <= 160: the whole graph fits one workgroup and needs no workspace; S > 512). */
size_t recon_propagate_ws_bytes(const recon_prop_args* args);

typedef struct {
    recon_prop_args fwd;            /* h_saved filled by the forward call                                 */
    const float* grad_out;          /* [B,C,L*dd]                                                         */
    float* const* g_adj;            /* HOST array of L device pointers [B,S,S] (entries may be NULL)      */
    float* g_h;                     /* [B,C,S] workspace; on return holds d loss / d h0 per batch element */
    float* const* g_trans;          /* block mode: HOST array of L device pointers [B, n(n-1), dd*dd] (entries may be NULL): *
                                     * the gradient lands in the transition tensors' own layout, g_adj is ignored           */
    float* g_identity;              /* block mode: [dd,dd] summed over graphs, nodes and hops (fixed order), or NULL         */
    float* identity_ws;             /* block mode: recon_propagate_identity_ws_floats() floats when g_identity is wanted    */
    float* wide_ws;                 /* recon_propagate_bwd_ws_floats() floats or NULL.  Wide states (S > 160): with it both products *
                                     * of a hop run as batched GEMMs over the graphs; without it, the channel-chunked kernel         */
    const int32_t* head_blk;        /* [C] device or NULL: the gather indices are BLOCKS of 16 consecutive columns (head_idx[c, x] =   *
                                     * head_blk[c] + x, likewise tail; dd = 16, block starts multiples of 16, head and tail blocks of  *
                                     * a channel distinct, idx_batch_stride = 0 — what utils/embedding_utils.py:184-202 builds; the    *
                                     * caller has checked it).  With chain_ws and fwd.split_ws the wide backward then runs its chain   *
                                     * d loss / d H^l-1 = A_l^T Y_l on the forward's two-term f16 kernel, all hops in one launch       */
    const int32_t* tail_blk;
    float* chain_ws;                /* recon_propagate_bwd_chain_ws_floats() floats or NULL                                            */
} recon_prop_bwd_args;

size_t recon_propagate_identity_ws_floats(int32_t dd);
size_t recon_propagate_bwd_ws_floats(const recon_prop_args* fwd);   /* B*C*S for S > 160 (not in block mode), else 0 */
size_t recon_propagate_bwd_chain_ws_floats(const recon_prop_args* fwd);   /* L * (graphs per slice) * C * S where the chain form exists (160 < S <= 512, dd = 16, shared indices, fwd.split_ws set), else 0 */

int recon_propagate_bwd(const recon_prop_bwd_args* args, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * P1 / P2 / K5 in bfloat16 (BASELINE.json configs[2] "GP-GNN Propagation 3 hops ... bf16", configs[4] "mixed GAT+Propagation
 *     stack, bf16"): the same step (models/models.py:240-274) on bfloat16 tensors — bf16 storage, fp32 accumulation on
 *     v_mfma_f32_16x16x32_bf16, every state and every gradient tensor rounded to bf16 once where the reference's bf16 tensors
 *     are (csrc/prop_b16.hip).  All data pointers are bf16 (uint16_t) data, 16-byte aligned, S % 8 == 0.
 *     recon_propagate_b16_form(): 1 = all hops of a graph in one workgroup (S % 16 == 0, S <= 160, C <= 96, even dd; h_saved
 *     optional), 3 = wide states (S % 64 == 0, 192 <= S <= 512): the state of 128 channels in LDS for all hops, 2 = one batched GEMM
 *     per hop over the graphs (needs h_saved and `zeros`), 0 = shape not taken (the caller converts to float32 and runs
 *     recon_propagate_fwd).  Every form reads `trans` in place in block mode.
 *     The backward runs both products of a hop as batched GEMMs over the graphs for every shape, in block mode without an adjacency.
 * ------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t B, C, S, L, dd;         /* as recon_prop_args                                                 */
    int32_t act;                    /* RECON_ACT_*                                                        */
    const void* const* adj;         /* HOST array of L device pointers, each [B,S,S] bf16 (ignored with trans) */
    const void* h0;                 /* [C,S] (h0_batch_stride = 0) or [B,C,S] bf16                        */
    int64_t h0_batch_stride;        /* elements; multiple of 8                                            */
    const int64_t* head_idx;        /* [C,dd] (idx_batch_stride = 0) or [B,C,dd]; values in [0,S)         */
    const int64_t* tail_idx;
    int64_t idx_batch_stride;
    void* out;                      /* [B,C,L*dd] bf16                                                    */
    void* h_saved;                  /* [L,B,C,S] bf16 states after each hop: for the backward; required by form 2 */
    const void* const* trans;       /* BLOCK MODE (form 1 only) or NULL: L device pointers [B, n(n-1), 256] bf16 */
    const void* identity;           /* [16,16] bf16, with trans                                           */
    const void* zeros;              /* >= 1 KiB of zero bytes, 16-byte aligned (K tails of the GEMMs): form 2 and the backward */
} recon_prop_b16_args;

int recon_propagate_b16_form(const recon_prop_b16_args* args);
int recon_propagate_b16_fwd(const recon_prop_b16_args* args, recon_stream_t stream);

typedef struct {
    recon_prop_b16_args fwd;        /* adj or trans + identity, h_saved (filled by the forward), zeros    */
    const void* grad_out;           /* [B,C,L*dd] bf16                                                    */
    void* const* g_adj;             /* HOST array of L device pointers [B,S,S] bf16 (entries may be NULL) */
    void* g_h;                      /* [B,C,S] bf16; on return d loss / d h0 per batch element            */
    void* ws;                       /* [B,C,S] bf16 workspace                                             */
    const int32_t* head_blk;        /* [C] device or NULL: the gather indices are BLOCKS of dd consecutive columns (head_idx[c, x] =     *
                                     * head_blk[c] + x, likewise tail; what utils/embedding_utils.py:184-202 builds) — the caller has   *
                                     * checked it, idx_batch_stride == 0, head_blk[c] != tail_blk[c]: Y_l is then formed in the GEMM   *
                                     * epilogues instead of by a pass of its own                                                        */
    const int32_t* tail_blk;
    void* const* g_trans;           /* BLOCK MODE (fwd.trans): HOST array of L device pointers [B, n(n-1), 256] bf16 (entries may be NULL) */
    void* g_identity;               /* block mode: [16,16] bf16 summed over graphs, nodes and hops (fp32, fixed order), or NULL          */
    void* diag_ws;                  /* block mode with g_identity: recon_propagate_b16_bwd_diag_elems() bf16 elements                   */
    float* ident_ws;                /* block mode with g_identity: recon_block_adjacency_b16_bwd_workspace_floats(16) floats            */
} recon_prop_b16_bwd_args;

size_t recon_propagate_b16_bwd_diag_elems(const recon_prop_b16_args* fwd);
int recon_propagate_b16_bwd(const recon_prop_b16_bwd_args* args, recon_stream_t stream);

/* P1 in bf16: recon_block_adjacency_fwd / _bwd on bf16 tensors (the identity gradient is summed in fp32, fixed order, rounded once) */
int recon_block_adjacency_b16_fwd(const void* T, const void* identity, int32_t B, int32_t n, int32_t dd, void* A, recon_stream_t stream);
size_t recon_block_adjacency_b16_bwd_workspace_floats(int32_t dd);
int recon_block_adjacency_b16_bwd(const void* gA, int32_t B, int32_t n, int32_t dd, void* gT, void* g_identity, float* workspace,
                                  recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * The GAT_sep_space variant's entity -> relation-space map (GAT_sep_space/main.py:359-364, :372-377; GAT_sep_space/models.py:316-320):
 *     out[t] = x[t] . W[rel[t]]      x [T, in_dim], W [R, in_dim, out_dim], rel [T] in [0, R)
 * without the [T, in_dim, out_dim] gather the reference feeds to torch.bmm.  `order` [T] int32 = the rows in relation order (a stable
 * argsort of rel), `seg_ptr` [R+1] int32 = where each relation's rows start in that order.  transpose_w: W is [R, out_dim, in_dim] and
 * read transposed (the input gradient g_x[t] = g[t] . W[rel[t]]^T).  _wgrad: g_W[r] = sum over the rows t of relation r of x[t]^T g[t]
 * (dense [R, in_dim, out_dim]: relations without rows get zeros), fixed summation order.  in_dim <= 1024, out_dim <= 512 for _wgrad.
 * ------------------------------------------------------------------------------------------*/
int recon_rel_rows_mm(const float* x, const int32_t* order, const int64_t* rel, const float* W, int32_t T, int32_t in_dim, int32_t out_dim,
                      int32_t transpose_w, float* out, recon_stream_t stream);
int recon_rel_rows_mm_wgrad(const float* x, const float* g, const int32_t* order, const int32_t* seg_ptr, int32_t R, int32_t in_dim,
                            int32_t out_dim, float* gW, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * P4  make_start_entity_embeddings (utils/context_utils.py:387-426): h0[b,c=(i,j),:] has the first
 *     entity's embedding in node i's first half-slot and the second entity's in node j's second
 *     half-slot, zero elsewhere, times the start-embedding template.
 * ------------------------------------------------------------------------------------------*/
int recon_start_entity_embeddings(const float* entity_embeddings /*[U,d]*/, const int64_t* pos /*[B,C,2]*/,
                                  const float* templ /*[C,S]*/, int32_t B, int32_t n, int32_t d,
                                  float* out /*[B,C,S]*/, recon_stream_t stream);
/* Its backward, first half (the reference has none of its own: autograd's index_put_ backward of context_utils.py:413-420, whose
 * float atomics fix no summation order): rows [2][B*C][d] = the d-wide pieces of (grad_out * templ) that belong to pos[..., 0]
 * (node i's first half-slot) and pos[..., 1] (node j's second half-slot), and — when `key` is given — key [2][2*B*C] = those entity
 * ids twice (the [2, E] segment key of recon_spmm_rowsum_fwd over a destination-only recon_graph: the fixed-order second half). */
int recon_start_entity_embeddings_bwd(const float* grad_out /*[B,C,S]*/, const int64_t* pos /*[B,C,2]*/, const float* templ /*[C,S]*/,
                                      int32_t B, int32_t n, int32_t d, float* rows /*[2,B*C,d]*/, int64_t* key /*[2,2*B*C] or NULL*/,
                                      recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * P5 / K6  GraphConvolution (models/layers.py:57-63), batched over B graphs (B = 1 is the
 *     reference's 2-D call):  out = relu(adj @ (x @ W) + bias).
 *     support [B*n,out] is caller-allocated scratch / saved for the backward.
 * ------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t B, n, in_features, out_features;
    const float* x;                 /* [B,n,in]   */
    const float* adj;               /* [B,n,n]    */
    const float* weight;            /* [in,out]   */
    const float* bias;              /* [out] or NULL */
    float* support;                 /* [B,n,out]  x @ W */
    float* out;                     /* [B,n,out]  */
    void* w_split;                  /* recon_gcn_split_bytes() bytes or NULL: bfloat16 term planes of W^T and W *
                                     * (filled by recon_gcn_fwd, read again by recon_gcn_bwd); with it x @ W and *
                                     * g_support @ W^T run on the split-precision GEMM (csrc/gemm_bx3.hip)       */
} recon_gcn_args;

size_t recon_gcn_split_bytes(int32_t in_features, int32_t out_features);
int recon_gcn_fwd(const recon_gcn_args* args, recon_stream_t stream);

typedef struct {
    recon_gcn_args fwd;
    const float* grad_out;          /* [B,n,out] */
    float* g_support;               /* [B,n,out] workspace */
    float* partial;                 /* workspace: recon_gcn_bwd_partial_floats() floats */
    float* g_x;                     /* [B,n,in]  or NULL */
    float* g_adj;                   /* [B,n,n]   or NULL */
    float* g_weight;                /* [in,out]  or NULL */
    float* g_bias;                  /* [out]     or NULL */
    void* gs_split;                 /* recon_gcn_bwd_split_bytes() bytes or NULL: bfloat16 term planes of g_support for *
                                     * the split-precision weight-gradient product (needs fwd.w_split as well)         */
} recon_gcn_bwd_args;

size_t recon_gcn_bwd_split_bytes(int32_t B, int32_t n, int32_t out_features);

size_t recon_gcn_bwd_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t out_features);
int recon_gcn_bwd(const recon_gcn_bwd_args* args, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * P5 / K6 in bfloat16 (BASELINE.json configs[2]: "bf16 + MFMA on W-projection"): the same layer on bfloat16 tensors — bf16
 *     storage, fp32 accumulation, x @ W / g_support @ W^T / x^T @ g_support on v_mfma_f32_16x16x32_bf16 (csrc/gemm_b16.hip).
 *     Activations carry a row stride that is a multiple of 8 elements (16 bytes) and at least the feature count rounded up
 *     to 8, so that feature counts like 300 need no repacking between layers; columns past the feature count of `support`,
 *     `out` and `g_support` are written as zeros, those of `x` must be finite.  All pointers are bf16 (uint16_t) data.
 * ------------------------------------------------------------------------------------------*/
typedef struct {
    int32_t B, n, in_features, out_features;
    const void* x; int64_t ldx;     /* [B*n, ldx] bf16, 16-byte aligned.  Fused forward (support == NULL): any even   *
                                     * ldx >= in_features — the kernel masks the K tail, so rows need no padding and *
                                     * whatever lies behind a row's last feature is never multiplied                 */
    const void* adj;                /* [B, n, n] bf16                                                            */
    const void* weight;             /* [in, out] bf16, contiguous                                                */
    const void* bias;               /* [out] bf16 or NULL                                                        */
    void* support; int64_t lds;     /* [B*n, lds] bf16: x @ W, scratch / saved                                    */
    void* out; int64_t ldo;         /* [B*n, ldo] bf16                                                           */
    void* w_planes;                 /* recon_gcn_b16_planes_bytes() bytes, scratch / saved: W^T and W zero padded along k */
    int32_t w_planes_valid;         /* != 0: w_planes already holds the planes of `weight` (a caller that keeps them across      *
                                     * calls while the weight is unchanged — inference — saves the two repacking launches)      */
    /* RAGGED batch (BASELINE.json configs[4]: graphs of up to 256 nodes each, every one its own size), or NULL / 0: graph b owns node  *
     * rows node_ptr[b] .. node_ptr[b+1] of x / support / out (total_rows = node_ptr[B] rows in all) and a dense n_b x n_b adjacency   *
     * at adj + adj_ptr[b] (elements); `n` is then the LARGEST n_b.  The layer of models/layers.py:57-63 applied to every graph:        *
     * x @ W over all rows at once, the aggregate per graph.  `support` is required.                                                    */
    const int32_t* node_ptr;        /* [B+1] device, ascending, node_ptr[0] = 0                                  */
    const int64_t* adj_ptr;         /* [B+1] device: adj_ptr[b] = sum over b' < b of n_b'^2                      */
    int64_t total_rows;
} recon_gcn_b16_args;

typedef struct {
    recon_gcn_b16_args fwd;
    const void* grad_out; int64_t ldg;   /* [B*n, ldg] bf16                                                      */
    void* g_support;                /* [B*n, fwd.lds] bf16 workspace                                             */
    float* partial;                 /* recon_gcn_b16_bwd_partial_floats() floats; ragged batch:                  *
                                     * recon_gcn_b16_bwd_partial_floats(1, total_rows, in, out) + B * out floats  */
    void* g_x; int64_t ldgx;        /* [B*n, ldgx] bf16 or NULL                                                  */
    void* g_adj;                    /* [B, n, n] bf16 or NULL                                                    */
    void* g_weight;                 /* [in, out] bf16 or NULL                                                    */
    void* g_bias;                   /* [out] bf16 or NULL                                                        */
    const void* zeros;              /* >= 1 KiB of zero bytes, 16-byte aligned (K tails of the k-major GEMM)     */
} recon_gcn_b16_bwd_args;

size_t recon_gcn_b16_planes_bytes(int32_t in_features, int32_t out_features);
int recon_gcn_b16_fwd(const recon_gcn_b16_args* args, recon_stream_t stream);
size_t recon_gcn_b16_bwd_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t out_features);
int recon_gcn_b16_bwd(const recon_gcn_b16_bwd_args* args, recon_stream_t stream);

/* L GraphConvolutions that share one adjacency (models/layers.py:57-63 applied L times, as the reference's stacks do), INFERENCE, in one
 * launch: the activations of a graph stay in LDS between the layers ("fused 3-hop, H resident in LDS").  Results are bit-equal
 * to L calls of recon_gcn_b16_fwd's fused form.  Layer 0: in_features -> hidden; layers 1 .. L-1: hidden -> hidden.  Shapes: n <= 32,
 * n % 4 == 0, hidden % 4 == 0, hidden <= 320; everything else answers RECON_ERR_UNSUPPORTED (the caller then runs layer by layer). */
typedef struct {
    int32_t B, n, in_features, hidden, L;
    const void* x; int64_t ldx;     /* [B*n, ldx] bf16, any even ldx >= in_features (the K tail is masked)                      */
    const void* adj;                /* [B, n, n] bf16, 8-byte aligned                                                          */
    const void* const* w_planes;    /* HOST array of L device pointers: W_l^T zero padded along k, [hidden][kp(in_l)] bf16,     *
                                     * written by recon_gcn_b16_transposed_planes() (kp = rounded up to 32)                     */
    const void* const* bias;        /* HOST array of L device pointers [hidden] bf16 (entries, or the array, may be NULL)       */
    void* out; int64_t ldo;         /* [B*n, ldo] bf16, ldo % 8 == 0; columns hidden .. ldo are written as zeros                */
} recon_gcn_b16_stack_args;
int recon_gcn_b16_stack_fwd(const recon_gcn_b16_stack_args* args, recon_stream_t stream);
/* The same stack WITH gradients (training; `gcn_layers.gcn_stack` under autograd): models/layers.py:57-63 applied L times + autograd.
 *   forward : one launch — the kernel above, which also keeps every layer's result (`acts`: the backward's ReLU masks and the operands of the
 *             weight gradients) — behind one repacking launch per layer (`planes`);
 *   backward: one launch for the whole chain g_support_l = adj^T (g_l . [act_l > 0]),  g_{l-1} = g_support_l W_l^T  (the gradient of a graph
 *             stays in LDS between the layers), then ONE k-major split-K launch for every layer's g_W_l = x_l^T g_support_l and one second
 *             pass that also finishes the g_bias_l.
 * Output, g_x and g_bias are bit-equal to L calls of recon_gcn_b16_fwd / recon_gcn_b16_bwd; g_W differs from theirs by the summation
 * order of the split-K partials only (fixed, run-to-run reproducible).  Shapes as for recon_gcn_b16_stack_fwd, and in_features <= 320. */
typedef struct {
    int32_t B, n, in_features, hidden, L;
    const void* x; int64_t ldx;         /* [B*n, ldx] bf16; ldx % 8 == 0 — or, with x_rows set, any even ldx >= in_features (read in place)     */
    void* x_rows; int64_t ldxr;         /* optional [B*n, ldxr] bf16, ldxr % 8 == 0: the forward leaves x here with 16-byte aligned rows for     *
                                         * layer 0's weight gradient (feature counts like 300: no repacking pass in front of the call)          */
    const void* adj;                    /* [B, n, n] bf16, 8-byte aligned                                                                      */
    const void* const* weight;          /* HOST array of L device pointers W_l [in_l][hidden] bf16, contiguous                                 */
    const void* const* bias;            /* HOST array of L device pointers [hidden] bf16 (entries, or the array, may be NULL)                  */
    void* const* planes;                /* HOST array of L workspaces of recon_gcn_b16_planes_bytes(in_l, hidden) bytes: written by the        *
                                         * forward, read by the backward                                                                      */
    void* const* acts; int64_t ldo;     /* HOST array of L buffers [B*n, ldo] bf16 (ldo % 8 == 0): the result of every layer, acts[L-1] = the  *
                                         * stack's result; columns hidden .. ldo are written as zeros                                         */
    /* ---- backward only ---- */
    const void* grad_out; int64_t ldg;  /* [B*n, ldg] bf16: gradient of acts[L-1]; any even ldg >= hidden, 4-byte aligned (read in place)      */
    void* const* g_support;             /* HOST array of L workspaces [B*n, ldo] bf16                                                          */
    float* partial;                     /* recon_gcn_b16_stack_bwd_partial_floats() floats                                                     */
    void* g_x; int64_t ldgx;            /* [B*n, ldgx] bf16 or NULL; ldgx % 4 == 0, >= in_features rounded up to 4, 8-byte aligned              */
    void* const* g_weight;              /* HOST array of L device pointers [in_l][hidden] bf16 (entries, or the array, may be NULL)            */
    void* const* g_bias;                /* HOST array of L device pointers [hidden] bf16 (entries, or the array, may be NULL)                  */
    const void* zeros;                  /* >= 1 KiB of zero bytes, 16-byte aligned                                                             */
} recon_gcn_b16_stack_train_args;
int recon_gcn_b16_stack_train_fwd(const recon_gcn_b16_stack_train_args* args, recon_stream_t stream);
size_t recon_gcn_b16_stack_bwd_partial_floats(int32_t B, int32_t n, int32_t in_features, int32_t hidden, int32_t L);
int recon_gcn_b16_stack_train_bwd(const recon_gcn_b16_stack_train_args* args, recon_stream_t stream);
/* planes [out_features][kp(in_features)] (bf16, (out_features * kp(in_features) * 2 bytes) of a weight [in_features][out_features] */
int recon_gcn_b16_transposed_planes(const void* weight, int32_t in_features, int32_t out_features, void* planes, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K4  fp32 MFMA GEMM used by the projections, exported for tests:
 *     C[M,N] = A[M,K] * B (B given as [N,K] when b_is_nk != 0, else [K,N]); plain row-major.
 * ------------------------------------------------------------------------------------------*/
int recon_sgemm(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                int32_t b_is_nk, float* C, int32_t ldc, recon_stream_t stream);

/* The same kernels with either operand in either orientation — A is [M,K] (a_is_km == 0) or [K,M]; B is [K,N] (b_is_nk == 0) or [N,K] —
 * and split-K (fixed-order second pass) when `workspace` (recon_sgemm_ex_workspace_floats() floats, may be 0) is given: what the
 * models' dense products outside the attention layer run on (`entity_embeddings.mm(self.W_entities)`, GAT/models.py:177, and the
 * gradients of it), in place of the library GEMM behind torch.mm. */
size_t recon_sgemm_ex_workspace_floats(int32_t M, int32_t N, int32_t K);
int recon_sgemm_ex(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, int32_t a_is_km, const float* B, int32_t ldb,
                   int32_t b_is_nk, float* C, int32_t ldc, float* workspace, recon_stream_t stream);

/* Small dense products (tens of MFLOP), plain fp32 FMAs with K split 16 ways inside a workgroup and a fixed-order combine:
 *     C[M,N] = op(A) * op(B);  A is [M,K] (a_is_km == 0) or [K,M]; B is [K,N] (b_is_nk == 0) or [N,K].
 * Replaces `relation_embed.mm(self.W)` (GAT/models.py:75) and its two gradient products, which are launch / latency bound on tile GEMMs. */
size_t recon_sgemm_small_workspace_floats(int32_t M, int32_t N, int32_t K);  /* 0: no workspace needed */
/* workspace: recon_sgemm_small_workspace_floats() floats, or NULL.  With it, products with few output tiles and a long K (weight
 * gradients: 50 x 200 over K = 14 541 rows) also cut K over workgroups and add the parts in fixed order. */
int recon_sgemm_small(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, int32_t a_is_km, const float* B, int32_t ldb,
                      int32_t b_is_nk, float* C, int32_t ldc, float* workspace, recon_stream_t stream);

/* K4'  the same product C[M,N] = A[M,K] * B[N,K]^T at fp32 accuracy on the bf16 matrix cores: every fp32
 * operand is split into three bfloat16 terms and six term products are accumulated in fp32
 * (csrc/gemm_bx3.hip).  B is split into `workspace` (recon_sgemm_bx3_workspace_bytes) first; A on the fly.
 * Requires K % 4 == 0 and 16-byte aligned A rows (RECON_ERR_UNSUPPORTED otherwise).  Exported for tests;
 * the GAT layer uses it for the projection `self.a.mm(...)` (GAT/layers.py:129-137) and its input gradient. */
size_t recon_sgemm_bx3_workspace_bytes(int32_t N, int32_t K);
int recon_sgemm_bx3(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                    float* C, int32_t ldc, void* workspace, recon_stream_t stream);
/* C[M,N] = A^T * B for k-major operands A[K,M], B[K,N] (the weight-gradient form g_a^T = V^T g_h; split-K with a
 * fixed-order second pass; A is split on the fly, B into `workspace` first).  Requires M and lda to be multiples
 * of 4 (RECON_ERR_UNSUPPORTED otherwise). */
size_t recon_sgemm_bx3_tn_workspace_bytes(int32_t M, int32_t N, int32_t K);
int recon_sgemm_bx3_tn(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                       float* C, int32_t ldc, void* workspace, recon_stream_t stream);

/* K4''  the same two products on the fp16 matrix cores with TWO half terms per operand and three term products
 * (csrc/gemm_hx2.hip): every operand is scaled by a per-tensor power of two taken from its max magnitude, so that
 * half's 5 exponent bits suffice, and pre-split into two half planes; fp32-class accuracy at half the MFMA work of
 * the bf16 x 3 form.  These stand-alone entries measure (amax) and split both operands into `workspace` (256-byte
 * aligned) first; the *_presplit forms re-run only the GEMM on the planes a previous call left there (benchmarks).
 * Requires K % 8 == 0 (k-contiguous form); any M, N for the k-major form.  In the GAT layer the planes are written by
 * the kernels that produce the operands (edge aggregation: V; ELU-gradient pass: g_h). */
size_t recon_hx2_aux_bytes(void);      /* size of recon_gat_atp_args.aux */
size_t recon_sgemm_hx2_workspace_bytes(int32_t M, int32_t N, int32_t K);
int recon_sgemm_hx2(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                    float* C, int32_t ldc, void* workspace, recon_stream_t stream);
int recon_sgemm_hx2_presplit(int32_t M, int32_t N, int32_t K, float* C, int32_t ldc, void* workspace, recon_stream_t stream);
size_t recon_sgemm_hx2_tn_workspace_bytes(int32_t M, int32_t N, int32_t K);
int recon_sgemm_hx2_tn(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                       float* C, int32_t ldc, void* workspace, recon_stream_t stream);
int recon_sgemm_hx2_tn_presplit(int32_t M, int32_t N, int32_t K, float* C, int32_t ldc, void* workspace, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * N1  the stage-A loop's loss: batch_gat_loss, GAT/main.py:344-376, with gat_loss_func = nn.MarginRankingLoss(margin) (mean reduction):
 *     triples int64 [n_pos (1 + reps)][3] = (head, relation, tail): n_pos positives, then reps corrupted copies of the block
 *     (reps = 2 * valid_invalid_ratio_gat); pair j compares positive j % n_pos with negative j:
 *     loss = mean_j max(0, |e[h] + r[rel] - e[t]|_1 (positive) - |...|_1 (negative) + margin).
 * fwd (two launches): terms [pairs], loss [1]; ent_key int64 [2][4 pairs] / rel_key int64 [2][2 pairs] (optional) receive the table row of every
 * gradient row the backward writes — the segment keys of recon_spmm_rowsum_fwd (row 0 = key, row 1 = copy).  `counter`: unused (may be NULL);
 * one workgroup adds the terms in index order.
 * bwd: g_ent_rows [4 pairs][D] (positive heads | positive tails | negative heads | negative tails), g_rel_rows [2 pairs][D]
 * (positive | negative); the tables' gradients are their sums by key.  Ids are not range-checked (the caller validates, as for edges). */
int recon_transe_margin_fwd(const float* entity, const float* relation, const int64_t* triples, int64_t n_pos, int32_t reps, int32_t D,
                            float margin, float* terms, float* loss, int64_t* ent_key, int64_t* rel_key, uint32_t* counter,
                            recon_stream_t stream);
/* recon_transe_margin_fwd with ONE key tensor for both tables: keys int64 [2][6 P] (P = n_pos * reps) — the four entity ids of pair j at
 * [q P + j], q < 4, its two relation ids PLUS n_entities at [4 P + q P + j], q < 2 (second row: a copy).  The backward's gradient rows, written
 * as one [6 P, D] block (g_ent_rows = block, g_rel_rows = block + 4 P D), are then summed into a [n_entities + n_relations, D] table by ONE
 * destination-only graph build and ONE recon_spmm_rowsum_fwd instead of two of each. */
int recon_transe_margin_fwd_keys(const float* entity, const float* relation, const int64_t* triples, int64_t n_pos, int32_t reps, int32_t D,
                                 float margin, float* terms, float* loss, int64_t* keys, int64_t n_entities, recon_stream_t stream);
int recon_transe_margin_bwd(const float* entity, const float* relation, const int64_t* triples, int64_t n_pos, int32_t reps, int32_t D,
                            const float* terms, const float* g_loss, float* g_ent_rows, float* g_rel_rows, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * N1  the stage-A batch builders on the device: Corpus.get_batch_adj_data (GAT/create_batch.py:391-436) and
 * Corpus.get_batch_nhop_neighbors_all (:871-895) over the 2-hop neighbourhoods that Corpus.bfs / get_further_neighbors (:788-869) precompute
 * on the host.  The knowledge graph is the grouped CSR recon_amd.sampler.KGNeighbourSampler builds once: (source, target) pairs grouped
 * by source in order of first appearance, the relations of a pair contiguous in insertion order.  All ids int64, device memory.
 *
 * recon_kg_adj_count: per batch position its number of edges nrel [B]; the LAST workgroup scans them (rel_off [B], totals[0] = E) and
 *   compacts the marks into the sorted unique entity / target id lists (uniq_ent, uniq_tgt: room for num_entities ids; totals[1], totals[2]
 *   = their lengths).  ent_mark / tgt_mark: num_entities zeroed bytes each, handed back zeroed; counter: one zeroed uint32, handed back zero.
 * recon_kg_adj_fill: edge int64 [2][E] (row 0 targets, row 1 sources) and edge_type [E], after the caller has read E.
 * recon_kg_nhop: write = 0 counts the quadruples of every source (qcount [S], quad_off [S], total[0]), write = 1 writes them
 *   (quads int64 [total][4] = source, first relation source -> parent, first relation parent -> target, target) in BFS discovery order.
 *   One wave per source, its visited set as a bitmap in LDS (recon_kg_nhop_lds_bytes; RECON_ERR_UNSUPPORTED beyond a CU's LDS). */
typedef struct recon_kg {
    int64_t num_entities;
    const int64_t* pair_ptr;        /* [num_entities + 1] pairs of source s: [pair_ptr[s], pair_ptr[s + 1]) */
    const int64_t* pair_tgt;        /* [P] */
    const int64_t* pair_first_rel;  /* [P] first relation of the pair */
    const uint8_t* not_loop;        /* [P] 0 for (s, s) pairs */
    const int64_t* rel_ptr;         /* [P + 1] relations of pair p: [rel_ptr[p], rel_ptr[p + 1]) */
    const int64_t* rel_sorted;      /* [T] */
} recon_kg;
size_t recon_kg_nhop_lds_bytes(int64_t num_entities);
int recon_kg_adj_count(const recon_kg* kg, const int64_t* entities, int32_t B, int64_t* nrel, uint8_t* ent_mark, uint8_t* tgt_mark,
                       int64_t* rel_off, int64_t* uniq_ent, int64_t* uniq_tgt, int64_t* totals, uint32_t* counter, recon_stream_t stream);
int recon_kg_adj_fill(const recon_kg* kg, const int64_t* entities, int32_t B, const int64_t* rel_off, int64_t E, int64_t* edge,
                      int64_t* edge_type, recon_stream_t stream);
int recon_kg_nhop(const recon_kg* kg, const int64_t* sources, int32_t S, int32_t partial_2hop, int32_t write, int64_t* qcount, int64_t* quad_off,
                  int64_t* quads, int64_t* total, uint32_t* counter, recon_stream_t stream);
/* recon_kg_nhop's count pass (write = 0) over a source list whose LENGTH is still on the device: sources has room for S_max ids, *s_dev of them are
 * valid (the unique-entity list and its count as recon_kg_adj_count leaves them).  Queued behind recon_kg_adj_count, it lets the caller read the
 * 1-hop sizes and the 2-hop total in one host round trip; the write pass is recon_kg_nhop(write = 1) with the count the host then knows. */
int recon_kg_nhop_count_early(const recon_kg* kg, const int64_t* sources, int32_t S_max, const int64_t* s_dev, int32_t partial_2hop, int64_t* qcount,
                              int64_t* quad_off, int64_t* total, recon_stream_t stream);
/* Dead-row pruning of a batch graph for SpKBGATModified (GAT/models.py:167-178: the model keeps `mask * out_entity_1`, so an edge matters only
 * if its destination lies in need = mask U { src(e) : mask[dst(e)] != 0 }).  edge int64 [2][E1] (row 0 destinations) with type [E1], edge_nhop
 * [2][E2] with type_nhop [E2][2] (E2 = 0: none); mask float [N]; need: N bytes of scratch.  The surviving edges are written in their
 * input order: out_edge (room for [2][E1]) holds the destinations at [0, counts[0]) and the sources at [E1, E1 + counts[0]) — the caller's
 * [2, counts[0]] tensor is a view with row stride E1 —, out_type [counts[0]]; likewise out_edge_nhop (row stride E2), out_type_nhop
 * [counts[1]][2].  pos (optional, E1 + E2): the survivors' positions in the concatenated input, the 1-hop ones at [0, counts[0]), the n-hop ones
 * (values from E1) at [E1, E1 + counts[1]).  counts: device int64 [2], read by the caller (one synchronisation).  One launch of one workgroup:
 * batch graphs are tens of thousands of edges. */
int recon_edges_prune(const int64_t* edge, const int64_t* type, int64_t E1, const int64_t* edge_nhop, const int64_t* type_nhop, int64_t E2,
                      const float* mask, int32_t N, uint8_t* need, int64_t* out_edge, int64_t* out_type, int64_t* out_edge_nhop,
                      int64_t* out_type_nhop, int64_t* pos, int64_t* counts, recon_stream_t stream);
/* The same for a whole SpKBGATModified call (GAT/models.py:136-178), one launch: mask [N] is MADE here (zeros, 1.0 at batch_entities [B]; an id
 * outside [0, N) sets counts[2]: the reference's indexing raises), the n-hop list arrives as the quadruples [E2][4] = (source, rel_1, rel_2,
 * target) it is derived from (:145-148), and the survivors of both lists go into ONE buffer out_edge [2][E1 + E2] (row stride E1 + E2), the
 * 1-hop ones at [0, counts[0]), the n-hop ones at [counts[0], counts[0] + counts[1]): the caller's 1-hop, n-hop and concatenated edge tensors are
 * views of it.  out_type [E1], out_type_nhop [E2][2], pos as above; counts: device int64 [3]. */
int recon_edges_prune_batch(const int64_t* batch_entities, int32_t B, const int64_t* edge, const int64_t* type, int64_t E1, const int64_t* quads,
                            int64_t E2, int32_t N, float* mask, uint8_t* need, int64_t* out_edge, int64_t* out_type, int64_t* out_type_nhop,
                            int64_t* pos, int64_t* counts, int64_t* ext_index /* optional, E1 + E2 */, int64_t ext_base, recon_stream_t stream);
/* ext_index: the row of the layers' edge-embedding table each surviving edge reads (GAT/layers.py:126-127 with `relation_embed[edge_type]` read in
 * place): a 1-hop edge its relation id, the j-th surviving n-hop edge row ext_base + j (ext_base = rows of the relation table). */
/* CSR-slot order of an index tensor in one launch: out32[k] = out64[k] = index[eid[k]] (recon_gat_atp_args.ee_index and the segment key of the
 * table gradient). */
int recon_slot_index(const int64_t* index, const int32_t* eid, int32_t E, int32_t* out32, int64_t* out64, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * N1  the tail of SpKBGATModified (GAT/models.py:167-180) and the row normalisation of the entity table (:160):
 *     y[r] = t / max(|t|_2, eps),  t = ew[r] + mask[r] * skip[r]        (skip, mask may be NULL: plain F.normalize(ew, p=2, dim=1); y may alias ew)
 * norm [N] (optional) receives |t|_2 for the backward:  g_t = (g_y - y (y . g_y)) / |t|  (g_y / eps below the clamp), g_skip = mask * g_t. */
int recon_rows_normalize_fwd(const float* ew, const float* skip, const float* mask, int64_t N, int32_t C, float eps, float* y, float* norm,
                             recon_stream_t stream);
int recon_rows_normalize_bwd(const float* g_y, const float* y, const float* norm, const float* mask, int64_t N, int32_t C, float eps, float* g_t,
                             float* g_skip, recon_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECON_HIP_H */
