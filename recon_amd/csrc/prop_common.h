// Shared between prop.hip (fp32-MFMA forms) and prop_h.hip (f16 x 2 split-precision forms) of the GP-GNN propagation.
#pragma once
#include "recon_common.h"

namespace recon {

constexpr int kMaxHops = 8;

struct PropK {
    const float* adj[kMaxHops];
    const float* h0; int64_t h0_bs;
    const int64_t* head; const int64_t* tail; int64_t idx_bs;
    float* out; float* hsave;
    int32_t B, C, S, L, dd, act, CC, Sp, pitch;
    float* stats;             // [B][2L+1] max magnitudes of h^0..h^L, A_1..A_L per graph (two-term forms, with hsave) or null
    const float* trans[kMaxHops];     // block mode (identity != null): transition tensors [B][C][dd*dd]; adj unused
    const float* identity;
    void* ws; int64_t ws_bytes;       // wide-state two-term form (prop_hl.hip): split workspace or null
};

// backward of the two-term form (prop_h.hip): all L hops of a graph in one persistent workgroup
struct PropBwdH {
    const float* adj[kMaxHops];
    float* gadj[kMaxHops];    // [B,S,S] per hop or null
    const float* h0; int64_t h0_bs;
    const float* hsave;       // [L][B][C][S]
    const int64_t* head; const int64_t* tail; int64_t idx_bs;
    const float* gout;        // [B][C][L*dd]
    float* gH;                // [B][C][S] out: gradient wrt h^0 per batch element
    const float* stats;       // [B][2L+1] from the forward
    int32_t B, C, S, L, dd, act;
    const float* trans[kMaxHops];     // block mode (identity != null)
    float* gtrans[kMaxHops];
    const float* identity;
    float* gident_ws;         // [grid][dd*dd] per-workgroup partial sums of the identity gradient, or null
};

struct PropBwdK {
    const float* A;           // adj of this hop [B,S,S]
    const float* Hl;          // state after this hop  [B,C,S]
    const float* Hprev;       // state before this hop [B,C,S] or h0
    int64_t hprev_bs;         // batch stride of Hprev (0 for a shared h0)
    const int64_t* head; const int64_t* tail; int64_t idx_bs;
    const float* gout;        // [B,C,L*dd]
    float* gH;                // [B,C,S] in: grad wrt H^l (ignored when first), out: grad wrt H^l-1
    float* gA;                // [B,S,S] or null
    int32_t B, C, S, L, dd, act, CC, Sp, pitch, hop, first, chunks;
};

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == RECON_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RECON_ACT_TANH) return tanhf(v);
    return v;
}
// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_bwd(float y, int act) {
    if (act == RECON_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == RECON_ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// prop_h.hip: forward on the f16 matrix cores with two-term operands (fp32-class accuracy); false / UNSUPPORTED for shapes it
// does not take (the caller then runs a fp32-MFMA form of prop.hip)
bool prop_fwd_h_supported(const PropK& p);
int prop_fwd_h(const PropK& p, hipStream_t st);
// prop_hl.hip: the same arithmetic for wide states (160 < S <= 512) in 64-channel chunks, A_l pre-split into a caller-owned workspace
size_t prop_hl_ws_bytes(int B, int S, int L);
bool prop_fwd_hl_supported(const PropK& p);
int prop_fwd_hl(const PropK& p, hipStream_t st);
// prop_hl.hip: the backward's chain d loss / d H^{l-1} = A_l^T Y_l, Y_{l-1} = (. + relation gradient) act'(H^{l-1}) for a SLICE of G graphs
// (all pointers slice-local) on the forward's kernel; step k = 0 .. L-1 is hop l = L - k
struct PropBwdHL {
    const float* adj_step[kMaxHops];  // A_{L-k} [G][S][S]; block mode (identity != null): the transition tensor T_{L-k} [G][C][256]
    const float* identity;            // block mode: [16][16]
    const float* y_in;                // Y_L [G][C][S], or null with hlast
    const float* hlast;               // H^L [G][C][S]: the chain kernel forms Y_L in its prologue (no row kernel, no [G, C, S] round trip)
    const float* hmask[kMaxHops];     // H^{l-1} [G][C][S], null for the last step (l = 1)
    float* ysave[kMaxHops];           // Y_{l-1} [G][C][S]; the last step's is d loss / d h^0
    const float* gout;                // [G][C][L dd]
    const int32_t* hblk; const int32_t* tblk;   // [C] first columns of the head / tail blocks (multiples of 16; dd = 16)
    int32_t gout_off[kMaxHops];       // (l - 2) dd
    int32_t G, C, S, L, dd, act;
    void* ws; int64_t ws_bytes;       // split workspace (recon_propagate_ws_bytes)
    unsigned char* yplanes; float* yisg;   // plane sets / scale sets of Y_L, Y_{L-1}, .., Y_1 for the d A products (prop_bwd_hl_ws_layout), or null
};
int64_t prop_bwd_hl_slice(int C, int S, int L, int64_t ws_bytes, int B);   // graphs per slice the workspace allows (0: form not available)
int prop_bwd_hl_chain(const PropBwdHL& a, hipStream_t st);
size_t prop_bwd_hl_ws_floats(int C, int S, int L, int64_t G);
void prop_bwd_hl_ws_layout(int C, int S, int L, int64_t G, float* ws, float** y_in, unsigned char** planes, float** isg, size_t* plane_set_bytes, size_t* isg_set_floats);
// d A_l[g] = Y_l[g]^T H^{l-1}[g] from one plane set of the chain kernel
// block mode (gdiag != null): the off-diagonal blocks go to gtrans [G][C][256] (may be null), the diagonal ones to gdiag [G][S / 16][256]; out unused
int prop_bwd_hl_gadj(const unsigned char* yplanes, const float* yisg, const float* Hprev, int64_t h_bs, float* out, float* gtrans, float* gdiag, int G,
                     int C, int S, hipStream_t st);
bool prop_bwd_h_shape_ok(int C, int S);   // LDS budget of the backward's two-term form
int prop_h_grid(int B);               // workgroups the two-term kernels launch for B graphs (one per CU, persistent)
bool prop_bwd_h_supported(const PropBwdH& p);
int prop_bwd_h(const PropBwdH& p, hipStream_t st);

}  // namespace recon
