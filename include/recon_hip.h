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

#define RECON_ABI_VERSION 1

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
    int32_t* rowptr_src;        /* [N+1] CSC over sources                                    */
    int32_t* slot_by_src;       /* [E]   CSC position -> CSR slot                            */
} recon_graph;

size_t recon_graph_workspace_bytes(int32_t N, int32_t E);

/* edge_dst / edge_src: the two rows of the reference's int64 [2,E] edge tensor (row 0 =
 * aggregation target, row 1 = neighbour; GAT/create_batch.py:429-433).  Returns RECON_ERR_INVALID
 * if N or E overflow int32.  Node ids outside [0,N) are undefined behaviour (as in the reference). */
int recon_graph_build(const int64_t* edge_dst, const int64_t* edge_src, recon_graph* g,
                      void* workspace, size_t workspace_bytes, recon_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * G1-G3  SpecialSpmmFinal: out[r,:] = sum_{e: edge[0,e]==r} edge_w[e,:]   (GAT/layers.py:54-64)
 *        backward: grad_edge_w[e,:] = grad_out[edge[0,e],:]               (GAT/layers.py:67-79)
 * ------------------------------------------------------------------------------------------*/
int recon_spmm_rowsum_fwd(const recon_graph* g, const float* edge_w /*[E,out_features]*/,
                          int32_t out_features, float* out /*[N,out_features]*/, recon_stream_t stream);
int recon_spmm_rowsum_bwd(const int64_t* edge_dst /*[E]*/, int64_t E, const float* grad_out /*[N,out]*/,
                          int32_t out_features, float* grad_edge_w /*[E,out]*/, recon_stream_t stream);

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
 * K4  fp32 MFMA GEMM used by the projections, exported for tests:
 *     C[M,N] = A[M,K] * B (B given as [N,K] when b_is_nk != 0, else [K,N]); plain row-major.
 * ------------------------------------------------------------------------------------------*/
int recon_sgemm(int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                int32_t b_is_nk, float* C, int32_t ldc, recon_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECON_HIP_H */
