// K4' — fp32-accurate GEMM on the bf16 matrix cores by operand splitting ("bf16 x 3").
//
// Every fp32 operand element is split on the fly into three bfloat16 terms
//     x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)   (24 mantissa bits)
// and the product a*b is accumulated in fp32 from the six term pairs of weight >= 2^-16:
//     a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0        (dropped: a1b2, a2b1, a2b2 <= 2^-24 relative)
// with v_mfma_f32_32x32x16_bf16 (2.5 PF dense): 6 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64
// cycles per 32x32x16 block => 2.67x the fp32-MFMA ceiling (~419 TF fp32-equivalent), at fp32-class
// accuracy (bfloat16 keeps fp32's exponent range, so no scaling is needed).  NPROD = 3 keeps only
// a0b0 + a0b1 + a1b0 (~2^-16 relative per product) for gradients that tolerate it.
//
// Block tile 128x128x32, 4 waves (2x2), each wave 2x2 MFMA tiles.  Global fp32 -> registers (prefetch
// of the next K tile) -> split -> LDS [3 terms][128 rows][32 k (+8 pad)] bf16, K-contiguous for both
// operands so every fragment is one ds_read_b128.  Same operand descriptors as gemm_f32.hip.
#include <stdlib.h>
#include "gemm_common.h"

namespace recon {
namespace {

constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK + 8, NT = 256;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Split3 { bf16x4 t[3]; };

__device__ __forceinline__ Split3 split4(const float (&v)[4]) {
    Split3 s;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const __bf16 h = static_cast<__bf16>(v[j]); s.t[0][j] = h; r[j] = v[j] - static_cast<float>(h); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const __bf16 h = static_cast<__bf16>(r[j]); s.t[1][j] = h; r[j] = r[j] - static_cast<float>(h); }
#pragma unroll
    for (int j = 0; j < 4; ++j) s.t[2][j] = static_cast<__bf16>(r[j]);
    return s;
}

// One operand tile: 128 (m or n) x 32 (k) fp32 -> LDS T[3][128][PITCH] bf16.
//   K_MINOR : global contiguous along k: item = (row, k quad); 4 items per thread
//   !K_MINOR: global contiguous along m/n: item = (4 consecutive k rows, mn quad); 1 item (4 float4) per thread
template <bool K_MINOR>
struct TileLoader3 {
    float r[4][4];
    int64_t fix[K_MINOR ? 4 : 1];

    __device__ __forceinline__ void init(const OperandDesc& d, int32_t mn0, int32_t mn_ext) {
        const int t = threadIdx.x;
        if constexpr (K_MINOR) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mn = mn0 + (t + NT * p) / 8;
                fix[p] = mn < mn_ext ? major_off(d, mn) : -1;
            }
        } else {
            const int mn = mn0 + (t & 31) * 4;
            fix[0] = mn < mn_ext ? minor_off(d.Dseg, d.Sseg, mn) : -1;
        }
    }
    __device__ __forceinline__ void load(const OperandDesc& d, int32_t k0, int32_t k_end) {
        const int t = threadIdx.x;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            r[p][0] = r[p][1] = r[p][2] = r[p][3] = 0.f;
            if constexpr (K_MINOR) {
                const int kq = k0 + ((t + NT * p) & 7) * 4;
                if (fix[p] >= 0 && kq < k_end) {
                    const float4 v = *reinterpret_cast<const float4*>(d.base + fix[p] + minor_off(d.Dseg, d.Sseg, kq));
                    r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
                }
            } else {
                const int k = k0 + (t >> 5) * 4 + p;
                if (fix[0] >= 0 && k < k_end) {
                    const float4 v = *reinterpret_cast<const float4*>(d.base + major_off(d, k) + fix[0]);
                    r[p][0] = v.x; r[p][1] = v.y; r[p][2] = v.z; r[p][3] = v.w;
                }
            }
        }
    }
    __device__ __forceinline__ void store(__bf16 (*T)[128][PITCH]) const {
        const int t = threadIdx.x;
        if constexpr (K_MINOR) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = (t + NT * p) / 8, kq = ((t + NT * p) & 7) * 4;
                const Split3 s = split4(r[p]);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&T[q][row][kq]) = s.t[q];
            }
        } else {
            const int kq = (t >> 5) * 4, mq = (t & 31) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // one m/n row: its 4 consecutive k values
                const float v[4] = {r[0][j], r[1][j], r[2][j], r[3][j]};
                const Split3 s = split4(v);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&T[q][mq + j][kq]) = s.t[q];
            }
        }
    }
};

template <bool A_KMINOR, bool B_KMINOR, int NPROD>
__global__ void __launch_bounds__(NT) k_gemm_bf16x3(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM][PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[3][BN][PITCH];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);

    TileLoader3<A_KMINOR> la;
    TileLoader3<B_KMINOR> lb;
    la.init(p.A, m0, p.M);
    lb.init(p.B, n0, p.N);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int mb = (wid >> 1) * 64, nb = (wid & 1) * 64;
    const int lr = lane & 31, lk = lane >> 5;

    if (k_begin < k_end) {
        la.load(p.A, k_begin, k_end);
        lb.load(p.B, k_begin, k_end);
        la.store(As);
        lb.store(Bs);
    }
    __syncthreads();
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) { la.load(p.A, k0 + BK, k_end); lb.load(p.B, k0 + BK, k_end); }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kk = ks * 16 + lk * 8;
            bf16x8 a[2][3], b[2][3];
            constexpr int NS = NPROD == 6 ? 3 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    a[i][q] = *reinterpret_cast<const bf16x8*>(&As[q][mb + i * 32 + lr][kk]);
                    b[i][q] = *reinterpret_cast<const bf16x8*>(&Bs[q][nb + i * 32 + lr][kk]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (NPROD == 6) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) { la.store(As); lb.store(Bs); }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + nb + j * 32 + lr;
        if (col >= p.N) continue;
        int64_t coff;
        float* base;
        if (p.partial) { base = p.partial + static_cast<int64_t>(blockIdx.z) * p.M * p.N; coff = col; }
        else { base = p.C.base; coff = minor_off(p.C.Dseg, p.C.Sseg, col); }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + mb + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row >= p.M) continue;
                const int64_t roff = p.partial ? static_cast<int64_t>(row) * p.N : out_row_off(p.C, row);
                base[roff + coff] = acc[i][j][r];
            }
        }
    }
}

}  // namespace

// EXPERIMENTAL, off by default: measured on MI355X (round 1) this kernel is correct (all parity tests
// pass) but not yet faster than the exact-fp32 MFMA kernel (123 vs 120 TF at 4096^3, slower at K = 200):
// MfmaUtil 32 %, the on-the-fly operand split costs ~355 VALU instructions per 48 MFMAs per wave and the
// 204-register footprint allows only 2 waves/SIMD.  RECON_GEMM=bf16x3 switches it on;
// RECON_GEMM_NPROD=3 keeps 3 of the 6 term products (tuning only).
bool gemm_bf16x3_enabled() {
    static const bool on = getenv("RECON_GEMM") && getenv("RECON_GEMM")[0] == 'b';
    return on;
}

int gemm_bf16x3_launch(const GemmArgs& a, bool a_k_minor, bool b_k_minor, int32_t split_k, hipStream_t st) {
    static const int nprod = (getenv("RECON_GEMM_NPROD") && atoi(getenv("RECON_GEMM_NPROD")) == 3) ? 3 : 6;
    dim3 grid(static_cast<unsigned>(ceil_div64(a.N, BN)), static_cast<unsigned>(ceil_div64(a.M, BM)), static_cast<unsigned>(split_k));
#define LAUNCH(AK, BK_)                                                                                   \
    do {                                                                                                  \
        if (nprod == 6) hipLaunchKernelGGL((k_gemm_bf16x3<AK, BK_, 6>), grid, dim3(NT), 0, st, a);        \
        else hipLaunchKernelGGL((k_gemm_bf16x3<AK, BK_, 3>), grid, dim3(NT), 0, st, a);                   \
    } while (0)
    if (a_k_minor && b_k_minor) LAUNCH(true, true);
    else if (a_k_minor && !b_k_minor) LAUNCH(true, false);
    else if (!a_k_minor && !b_k_minor) LAUNCH(false, false);
    else return RECON_ERR_UNSUPPORTED;
#undef LAUNCH
    return RECON_OK;
}

}  // namespace recon
