// The tail of SpKBGATModified._encode — GAT/models.py:167-180:
//     mask = zeros(N); mask[batch entities] = 1;  out = entity_embeddings.mm(W_entities) + mask.unsqueeze(-1) * out_entity;  out = F.normalize(out, p=2, dim=1)
// — and the in-place row normalisation of the entity table in front of the model (:160).  As torch ops: eight launches forward and as
// many backward for the tail, four for the table; here one each (a wave per row, fp32, the row sums in fixed order).
#include "recon_common.h"

namespace recon {
namespace {

// y = t / max(|t|_2, eps) with t = ew + (mask ? skip : 0); `norm` keeps the un-clamped |t|_2 for the backward.  skip / mask may be null (plain
// normalisation); y may alias ew.
__global__ void __launch_bounds__(256) k_rows_normalize(const float* __restrict__ ew, const float* __restrict__ skip, const float* __restrict__ mask,
                                                         int64_t N, int32_t C, float eps, float* __restrict__ y, float* __restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const bool on = skip && (!mask || mask[r] != 0.f);
    const float mv = (skip && mask) ? mask[r] : 1.f;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float t = ew[r * C + c] + (on ? mv * skip[r * C + c] : 0.f);
        ss = fmaf(t, t, ss);
    }
    ss = group_sum<64>(ss);
    const float nr = sqrtf(ss), inv = 1.f / fmaxf(nr, eps);
    for (int c = lane; c < C; c += 64) {
        const float t = ew[r * C + c] + (on ? mv * skip[r * C + c] : 0.f);
        y[r * C + c] = t * inv;
    }
    if (norm && lane == 0) norm[r] = nr;
}

// g_t = (g_y - y (y . g_y)) / |t|   (|t| >= eps; below it the clamp's derivative is zero: g_t = g_y / eps);  g_skip = mask g_t
__global__ void __launch_bounds__(256) k_rows_normalize_bwd(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ norm,
                                                             const float* __restrict__ mask, int64_t N, int32_t C, float eps, float* __restrict__ gt,
                                                             float* __restrict__ gskip) {
    const int lane = threadIdx.x & 63;
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot = fmaf(gy[r * C + c], y[r * C + c], dot);
    dot = group_sum<64>(dot);
    const float nr = norm[r];
    const bool clamped = nr < eps;
    const float inv = 1.f / fmaxf(nr, eps), mv = mask ? mask[r] : 1.f;
    for (int c = lane; c < C; c += 64) {
        const float g = clamped ? gy[r * C + c] * inv : (gy[r * C + c] - y[r * C + c] * dot) * inv;
        gt[r * C + c] = g;
        if (gskip) gskip[r * C + c] = mv * g;
    }
}
}  // namespace
}  // namespace recon

extern "C" int recon_rows_normalize_fwd(const float* ew, const float* skip, const float* mask, int64_t N, int32_t C, float eps, float* y, float* norm,
                                        recon_stream_t stream) {
    if (N < 0 || C <= 0 || eps <= 0.f) return RECON_ERR_INVALID;
    if (N == 0) return RECON_OK;
    if (!ew || !y) return RECON_ERR_INVALID;
    if (ceil_div64(N, 4) >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(recon::k_rows_normalize, dim3(static_cast<unsigned>(ceil_div64(N, 4))), dim3(256), 0, as_stream(stream), ew, skip, mask, N, C, eps, y, norm);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}

extern "C" int recon_rows_normalize_bwd(const float* g_y, const float* y, const float* norm, const float* mask, int64_t N, int32_t C, float eps, float* g_t,
                                        float* g_skip, recon_stream_t stream) {
    if (N < 0 || C <= 0 || eps <= 0.f) return RECON_ERR_INVALID;
    if (N == 0) return RECON_OK;
    if (!g_y || !y || !norm || !g_t) return RECON_ERR_INVALID;
    if (ceil_div64(N, 4) >= (1LL << 31)) return RECON_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(recon::k_rows_normalize_bwd, dim3(static_cast<unsigned>(ceil_div64(N, 4))), dim3(256), 0, as_stream(stream), g_y, y, norm, mask, N, C, eps, g_t,
                       g_skip);
    RECON_CHECK_LAUNCH();
    return RECON_OK;
}
