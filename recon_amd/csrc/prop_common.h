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

}  // namespace recon
