// Internal helpers shared by the HIP translation units of librecon_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "recon_hip.h"

#define RECON_CHECK_LAUNCH()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return RECON_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t as_stream(recon_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

namespace recon {

constexpr int kWave = 64;

// ---- run-time switches (config.hip): read from the environment once, at first use; nullptr = unset, like getenv ----------------------
enum CfgKey { CFG_BGEMM_CFG, CFG_GCN_FUSED, CFG_GCN_FUSED_BWD, CFG_GCN_FUSED_PARTS, CFG_GCN_STACK_PARTS, CFG_GEMM_CFG, CFG_GEMM_LIN, CFG_GEMM_SPLITK,
              CFG_GEMM_XCD, CFG_PROP_B16, CFG_PROP_B16_YPOST, CFG_PROP_BWD, CFG_PROP_BWD_CHAIN, CFG_PROP_BWD_WIDE, CFG_PROP_FWD, CFG_PROP_LDS_KB,
              CFG_ATP_ROW_SCALE, CFG_GRAPH_SMALL, CFG_GCN_STACK_GPW, CFG_KG_NHOP, CFG_HX2_RING, CFG_K2_LDS_RING, CFG_K2_PERSIST, CFG_COUNT };
const char* cfg(CfgKey k);
int cfg_int(CfgKey k, int dflt);
char cfg_char(CfgKey k);                  // first character, '\0' when unset
int32_t* nan_flag();                      // the calling device's NaN word (recon_set_nan_flag) or nullptr

// ---- DPP cross-lane adds (VALU, no LDS traffic) -------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float swap_add32(float a, float b) {          // lanes < 32: a[l] + a[l + 32];  lanes >= 32: b[l - 32] + b[l]
    // (elements copied to scalars first: __builtin_bit_cast applied to an element of an ext_vector reads element 0 — clang 20 / ROCm 7.2)
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float swap_add16(float a, float b) {          // even 16-lane rows: a[l] + a[l + 16];  odd rows: b[l - 16] + b[l]
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
// Sum over aligned groups of G consecutive lanes (G = 4..64, power of two); every lane of the
// group receives the total.  Quad and row steps are DPP; the 32- and 64-lane steps are v_permlane16_swap / v_permlane32_swap
// (round 6; they went through ds_bpermute: same operands per add, bit-identical sums).  Summation order is fixed => deterministic.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group width");
    v += dpp_mov<0xB1>(v);                      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                      // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
    if constexpr (G >= 16) v += dpp_mov<0x140>(v);  // row_mirror
    if constexpr (G >= 32) v = swap_add16(v, v);
    if constexpr (G >= 64) v = swap_add32(v, v);
    return v;
}

// Reduce NV per-lane values over the 64 lanes of a wave at once: each halving step exchanges HALF of the
// remaining values with the lane whose bit (32, 16, 8) differs, so NV values cost NV-1 exchanges + log2(64/NV)
// DPP steps instead of NV full reductions.  Lane l ends up with the total of value index l >> (6 - log2 NV).
// The exchanges across lane bits 32 and 16 are v_permlane32_swap / v_permlane16_swap (gfx950): ONE VALU instruction swaps the upper
// 32-lane (odd 16-lane) rows of one register with the lower (even) rows of another — both halves of a butterfly step — where
// __shfl_xor went through the LDS crossbar (ds_bpermute + two selects per value, and an lgkmcnt wait in front of every add: 31
// instructions and seven exposed LDS round trips per 8 values in k_gat_atp_bwd's edge loop, now 18 and none).  Same operands per
// add as before: results are bit-identical.
template <int NV>
__device__ __forceinline__ float multi_sum(float (&v)[NV], int lane) {
    static_assert(NV == 1 || NV == 2 || NV == 4 || NV == 8, "values per wave");
    if constexpr (NV == 1) {
        return group_sum<64>(v[0]);
    } else {
        float w[NV / 2];
#pragma unroll
        for (int j = 0; j < NV / 2; ++j) w[j] = swap_add32(v[j], v[j + NV / 2]);
        if constexpr (NV == 2) {
            return group_sum<16>(swap_add16(w[0], w[0]));
        } else {
            float x[NV / 4];
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) x[j] = swap_add16(w[j], w[j + NV / 4]);
            if constexpr (NV == 4) {
                return group_sum<16>(x[0]);
            } else {
                const bool up3 = lane & 8;
                const float mine = up3 ? x[1] : x[0], other = up3 ? x[0] : x[1];
                const float r = mine + dpp_mov<0x128>(other);             // row_ror:8 — lane l reads lane l ^ 8 of its 16-lane row
                return group_sum<8>(r);
            }
        }
    }
}

// ---- global -> LDS copy (16 bytes per lane: LDS[dst + 16 lane] = *src) the compiler does not see -----------------------------
// Through __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) knows that an LDS write is in flight and, unable to tell the stages of a ring
// apart, drains the request counter (s_waitcnt vmcnt(0)) in front of the next ds_read of ANY LDS address: behind the loop-head barrier
// and behind every copy issued between the MFMA rows of a step — a three-stage ring runs as a serial copy / compute loop (k_bgemm_b16,
// round 4: four exposed L2 -> LDS latencies per 32 MFMAs).  Issued as inline asm the copy is invisible to that bookkeeping; the waits
// that order it are the caller's counted `s_waitcnt vmcnt(N)` + s_barrier (dma_wait<N>()).  Unknown requests in flight only make the
// compiler's own counted waits for ITS loads conservative (requests of a kind return in order), never wrong.  m0 (the LDS base of the
// instruction) is saved and restored inside the block: the register is reserved, naming it as a clobber is not allowed.
// `lds_dst` must be wave-uniform.
__device__ __forceinline__ void dma16_to_lds(const void* src, const void* lds_dst) {
    const uint32_t a = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)(lds_dst))));
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "v"(src), "s"(a) : "memory");
}
// The same through a buffer descriptor (range-checked: a lane whose offset lies past `bytes` copies zeros), for rings whose tails and dead
// pieces rely on that.  `base`, `bytes` and `lds_dst` must be wave-uniform.
typedef uint32_t dma_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_u32x4 dma_descriptor(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    auto uni = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v)); };
    return dma_u32x4{uni(static_cast<uint32_t>(a)), uni(static_cast<uint32_t>(a >> 32) & 0xffffu), uni(bytes), 0x00020000u};
}
// (`scalar_offset` is added to every lane's address AFTER the range check of `byte_offset`: a K-step advance that keeps masked lanes masked)
__device__ __forceinline__ void dma16_buffer_to_lds(dma_u32x4 desc, uint32_t byte_offset, const void* lds_dst, uint32_t scalar_offset = 0) {
    const uint32_t a = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)(lds_dst))));
    const uint32_t so = __builtin_amdgcn_readfirstlane(scalar_offset);
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "v"(byte_offset), "s"(desc), "s"(a), "s"(so) : "memory");
}
template <int N>
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// max over aligned groups of 16 lanes (DPP only), every lane of the group receives it
__device__ __forceinline__ float group_max16(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
// FOUR per-lane values -> their maxima over the wave, value j's in lanes [16 j, 16 j + 16): two halving exchanges (3 permutes) + the DPP steps
__device__ __forceinline__ float multi_max4(const float (&v)[4], int lane) {
    const bool up = lane & 32, up2 = lane & 16;
    const float w0 = fmaxf(up ? v[2] : v[0], __shfl_xor(up ? v[0] : v[2], 32, 64));
    const float w1 = fmaxf(up ? v[3] : v[1], __shfl_xor(up ? v[1] : v[3], 32, 64));
    return group_max16(fmaxf(up2 ? w1 : w0, __shfl_xor(up2 ? w0 : w1, 16, 64)));
}

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void load_vec(float (&r)[VEC], const float* p) {
    if constexpr (VEC == 4) { float4 t = *reinterpret_cast<const float4*>(p); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
    else if constexpr (VEC == 2) { float2 t = *reinterpret_cast<const float2*>(p); r[0] = t.x; r[1] = t.y; }
    else { r[0] = *p; }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&r)[VEC]) {
    if constexpr (VEC == 4) { *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]); }
    else if constexpr (VEC == 2) { *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]); }
    else { *p = r[0]; }
}

// ---- f16 x 2 split-precision operands (gemm_hx2.hip): per-tensor power-of-two scale from the tensor's max magnitude ----
// amax values live in device memory as fp32 BIT PATTERNS (uint32): non-negative floats order like unsigned integers, so
// producers publish them with atomicMax (exact, order independent).  A scale descriptor names up to two slots and a host
// multiplier: amax = max(*p0, *p1) * mul  (p0 == nullptr: unscaled, s = 1).
// Each amax QUANTITY is kHx2Slots words spread 256 bytes apart (device-scope atomics on ONE address execute one after the
// other at the memory side, ~5-12 ns each: 12 k of them cost 60-90 us; hashed over 32 lines in different channels they
// overlap); producers hash their wave into a slot, consumers take the max over the slots with 32 scalar loads.
constexpr int kHx2Slots = 32, kHx2SlotStride = 64;                       // stride in uint32 words
constexpr int kHx2QuantityWords = kHx2Slots * kHx2SlotStride;            // 2048 words = 8 KiB per quantity
constexpr size_t kHx2AuxQuantities = 4;
constexpr size_t kHx2ZeroPageOffset = kHx2AuxQuantities * kHx2QuantityWords * 4;       // byte offset of the 1 KiB page of zeros
constexpr size_t kHx2AuxBytes = kHx2ZeroPageOffset + 1024;
// Per-block maxima published by PLAIN stores (no zeroing needed): block b of the producer owns one word of its quantity that no
// slot uses — word 1 + b / 32 behind slot b % 32; the consumer that needs the scale first gathers the words (hx2_blkmax_wave) and
// one of its blocks writes all 32 slots for the kernels that read the quantity the usual way.
constexpr int kHx2BlkMaxWords = kHx2Slots * (kHx2SlotStride - 1);
__host__ __device__ inline int hx2_blkmax_word(int b) { return (b % kHx2Slots) * kHx2SlotStride + 1 + b / kHx2Slots; }
struct Hx2Scale { const uint32_t* p0; const uint32_t* p1; float mul; };

// s = 2^(14 - floor(log2 amax)): s * amax in [2^14, 2^15) — half overflows at 65504 = 2^16 - 32
__device__ __forceinline__ float hx2_scale_of(float amax) {
    const uint32_t e = (__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu;
    if (e == 0u || e == 255u) return 1.f;                              // zero / denormal / inf / nan: leave the data alone
    int se = 268 - static_cast<int>(e);                                // 127 + 14 - (e - 127)
    se = se < 1 ? 1 : (se > 253 ? 253 : se);
    return __builtin_bit_cast(float, static_cast<uint32_t>(se) << 23);
}
__device__ __forceinline__ uint32_t hx2_amax_bits(const uint32_t* p) {
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < kHx2Slots; ++i) { const uint32_t v = p[i * kHx2SlotStride]; b = v > b ? v : b; }
    return b;
}
__device__ __forceinline__ float hx2_scale(const Hx2Scale& q) {
    if (!q.p0) return 1.f;
    uint32_t b = hx2_amax_bits(q.p0);
    if (q.p1) { const uint32_t b1 = hx2_amax_bits(q.p1); b = b1 > b ? b1 : b; }
    return hx2_scale_of(__builtin_bit_cast(float, b) * q.mul);
}
// the same for a whole wave at once (all 64 lanes must call): ONE vector load (lane l reads slot l & 31 of quantity l >> 5) and
// a butterfly max instead of 64 scalar loads and as many s_max — K1' pays 1.7 us for the scalar form at 8 192 short-lived waves
__device__ __forceinline__ float hx2_scale_wave(const Hx2Scale& q) {
    if (!q.p0) return 1.f;
    const int lane = threadIdx.x & 63;
    const uint32_t* p = (lane < 32 || !q.p1) ? q.p0 : q.p1;
    uint32_t b = p[(lane & 31) * kHx2SlotStride];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(b, off, 64); b = o > b ? o : b; }
    return hx2_scale_of(__builtin_bit_cast(float, b) * q.mul);
}
// max over nblk block words of a quantity (all 64 lanes must call); returns the fp32 bit pattern
__device__ __forceinline__ uint32_t hx2_blkmax_wave(const uint32_t* quantity, int nblk) {
    const int lane = threadIdx.x & 63;
    uint32_t b = 0;
    for (int k = lane; k < nblk; k += 64) { const uint32_t v = quantity[hx2_blkmax_word(k)]; b = v > b ? v : b; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(b, off, 64); b = o > b ? o : b; }
    return b;
}
// exact reciprocal of a power of two with exponent field in [1, 253]
__device__ __forceinline__ float hx2_inv(float s) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(uint32_t, s)); }

// two ALREADY SCALED fp32 values -> packed high terms / packed low terms (round to nearest even; the residual is exact).
// Written on 2-vectors so that the compiler selects v_cvt_pk_f16_f32 (one instruction converts and packs both): 5
// instructions per pair instead of 9 — the producers run this on every element of V and g_h.
__device__ __forceinline__ void hx2_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {x0, x1};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
    const f16x2_t l = __builtin_convertvector(r, f16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// wave-wide max of a non-negative per-lane value, then at most one atomicMax per wave into the wave's hashed slot of the
// (zeroed) quantity.  A wave first READS its slot (device scope, relaxed) and only sends the atomic when it would raise the
// value: the value is monotone, so a stale read can only cause a redundant atomic, never a lost maximum.
__device__ __forceinline__ void hx2_amax_commit(float m, uint32_t* quantity) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = (blockIdx.x + blockIdx.y * gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        uint32_t* slot = quantity + ((w * 2654435761u) >> 27) * kHx2SlotStride;          // top 5 bits of a multiplicative hash
        const uint32_t bits = __builtin_bit_cast(uint32_t, m);
        if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
}

// the same for a maximum that is already wave-uniform (no butterfly)
__device__ __forceinline__ void hx2_amax_commit_uniform(float m, uint32_t* quantity) {
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = (blockIdx.x + blockIdx.y * gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        uint32_t* slot = quantity + ((w * 2654435761u) >> 27) * kHx2SlotStride;
        const uint32_t bits = __builtin_bit_cast(uint32_t, m);
        if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
}

}  // namespace recon

// ---- generic fp32 MFMA GEMM with strided / gathered / head-major operand addressing -------
namespace recon {

// element (major i, minor j) of an operand lives at
//   base + major_off(i) + minor_off(j)
//   major_off(i) = gather ? gather[i]*S1 : (i % P)*S1 + (i / P)*S2
//   minor_off(j) = (j % Dseg) + (j / Dseg)*Sseg          (contiguous inside segments of Dseg)
struct OperandDesc {
    const float* base;
    const int32_t* gather;
    int64_t S1, S2, Sseg;
    int32_t P, Dseg;
};
inline OperandDesc plain_operand(const float* base, int64_t ld) {
    OperandDesc d; d.base = base; d.gather = nullptr; d.S1 = ld; d.S2 = 0; d.Sseg = 0; d.P = 0x7fffffff; d.Dseg = 0x7fffffff; return d;
}
struct OutputDesc {
    float* base;
    const int32_t* scatter;   // row index map (permutation) or null
    int64_t S1, S2, Sseg;
    int32_t P, Dseg;
};
inline OutputDesc plain_output(float* base, int64_t ld) {
    OutputDesc d; d.base = base; d.scatter = nullptr; d.S1 = ld; d.S2 = 0; d.Sseg = 0; d.P = 0x7fffffff; d.Dseg = 0x7fffffff; return d;
}

// C[M,N] = A[M,K] * B[K,N].  a_k_minor: A's contiguous index is k (else m); b_k_minor: B's contiguous
// index is k (else n).  split_k > 1 needs `partial` (split_k*M*N floats) and runs a deterministic
// second-pass reduction.
int gemm_f32(int32_t M, int32_t N, int32_t K, const OperandDesc& A, bool a_k_minor, const OperandDesc& B,
             bool b_k_minor, const OutputDesc& C, int32_t split_k, float* partial, hipStream_t stream);
int gemm_pick_split_k(int32_t M, int32_t N, int32_t K, int32_t batch = 1);

// batched variant: `batch` independent problems, operand/output bases advanced by *_bs elements per batch;
// epilogue 0 = none, 1 = ELU (applied after the split-K reduction when split_k > 1).
// `partial` must hold batch*split_k*M*N floats when split_k > 1.
// c_transpose = 1: C describes the TRANSPOSE of the product (element (m, n) is stored at C row n, column m), so a
// caller can put the 129..208-wide dimension of a product on N, where the 128 x 208 tile has 4 % padding.  The
// transposition happens in the split-K second pass: `partial` is then required even for split_k = 1.
struct GemmBatch { int32_t batch; int64_t a_bs, b_bs, c_bs; int32_t epilogue; int32_t c_transpose = 0; };
int gemm_f32_batched(int32_t M, int32_t N, int32_t K, const OperandDesc& A, bool a_k_minor, const OperandDesc& B,
                     bool b_k_minor, const OutputDesc& C, const GemmBatch& bt, int32_t split_k, float* partial,
                     hipStream_t stream);

// second pass of a split-K product with bf16 output (gemm_b16.hip): out[M][N] (row stride ldo) = sum of `splits` fp32 partials [z][M][N]
struct B16ReduceJob { const float* partial; uint16_t* out; int64_t ldo; int32_t splits, M, N; };
struct B16KmProduct { const void* A; int64_t lda; const void* B; int64_t ldb; void* out; int64_t ldo; float* partial; int32_t M, N; };

// split-precision (bf16 x 3) MFMA GEMM, gemm_bx3.hip: C = act(A . B^T), A fp32 k-contiguous, B pre-split bf16 planes
// [3][batch][N][bx3_kp(K)] written by bx3_split_planes (transposed = true reads src as [K][rows]).
int32_t bx3_kp(int32_t K);
int bx3_split_planes(const float* src, int64_t ld, int64_t src_bs, bool transposed, int32_t rows, int32_t K, int32_t batch, void* dst,
                     hipStream_t st);
bool bx3_supported(const OperandDesc& A, int32_t K, const GemmBatch& bt);
int gemm_bx3_batched(int32_t M, int32_t N, int32_t K, const OperandDesc& A, const void* planes, const OutputDesc& C,
                     const GemmBatch& bt, hipStream_t st);
// k-major form (A = fp32 [K][M] split on the fly, B = bf16 planes [3][K][ldb], n contiguous): partial[batch][split][M][N];
// reduce with splitk_reduce (gemm_f32.hip).  split_k must be bx3_kmajor_splits(K, requested) (K tiles of 32 per split).
bool bx3_kmajor_supported(const float* A, int64_t lda, int64_t a_bs, int64_t ldb, int64_t b_bs, int32_t M, int32_t N);
int bx3_kmajor_split_k(int32_t M, int32_t N, int32_t K, int32_t batch);
int bx3_kmajor_splits(int32_t K, int32_t split_k);
int gemm_bx3_kmajor_batched(int32_t M, int32_t N, int32_t K, const float* A, int64_t lda, int64_t a_bs, const void* Bplanes, int64_t ldb,
                            int64_t b_plane, int64_t b_bs, int32_t batch, int32_t split_k, float* partial, hipStream_t st);
// split-precision (f16 x 2) MFMA GEMMs on PRE-SPLIT operands, gemm_hx2.hip
int32_t hx2_kp(int32_t K);
int hx2_amax(const float* src, int64_t rows, int32_t cols, int64_t ld, uint32_t* slot, hipStream_t st);
int hx2_split_planes(const float* src, int64_t ld, int64_t src_bs, bool transposed, int32_t rows, int32_t K, int32_t batch, void* dst,
                     const Hx2Scale& sc, hipStream_t st);
// Both orientations of ONE batched matrix src[b][R][C] as half planes — planes[q][b][r][c] (rows R, k = C, zero padded to Cp) and the
// transposed planes[q][b][c][r] (rows C, k = R, padded to Rp) — as the work of block (bx, by = batch entry, bz = orientation) of a
// gx x batch x 2 grid of 256 threads.  A device function so that it can ride in another kernel's launch (k_row_dots_x): these
// few-MB passes cost a kernel turn-around (~5 us) more than their traffic.  The transposed half walks the source row-wise
// (consecutive threads = consecutive c) so that its reads are coalesced; each thread gathers 8 rows of one column.
// blkmax_quantity: the source's max magnitude arrives as nblk per-block words of that quantity (plain stores of the kernel in
// front, hx2_blkmax_word) instead of in its 32 slots; every wave gathers them, and block (0,0,0) fills the slots for later readers.
struct Hx2SplitBoth {
    const float* src; int64_t src_bs; int32_t R, C, Cp, Rp, batch, gx;
    _Float16* dst_n; _Float16* dst_t; Hx2Scale sc; uint32_t* blkmax_quantity; int32_t nblk;
};
__device__ __forceinline__ void hx2_split_both_block(const Hx2SplitBoth& p, int bx, int by, int bz) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    float s;
    if (p.blkmax_quantity) {
        const uint32_t bits = hx2_blkmax_wave(p.blkmax_quantity, p.nblk);
        if (bx == 0 && by == 0 && bz == 0 && threadIdx.x < kHx2Slots) p.blkmax_quantity[threadIdx.x * kHx2SlotStride] = bits;
        s = hx2_scale_of(__builtin_bit_cast(float, bits) * p.sc.mul);
    } else {
        s = hx2_scale(p.sc);
    }
    const float* sp = p.src + by * p.src_bs;
    const int64_t idx = static_cast<int64_t>(bx) * 256 + threadIdx.x;
    const int R = p.R, C_ = p.C, Cp = p.Cp, Rp = p.Rp;
    float v[8];
    _Float16* d;
    int64_t plane;
    if (bz == 0) {
        const int kq = static_cast<int>(idx % (Cp / 8)), r = static_cast<int>(idx / (Cp / 8));
        if (r >= R) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = 8 * kq + j; v[j] = k < C_ ? sp[static_cast<int64_t>(r) * C_ + k] : 0.f; }
        plane = static_cast<int64_t>(p.batch) * R * Cp;
        d = p.dst_n + (static_cast<int64_t>(by) * R + r) * Cp + 8 * kq;
    } else {
        const int c = static_cast<int>(idx % C_), kq = static_cast<int>(idx / C_);      // consecutive threads: consecutive source columns
        if (kq >= Rp / 8) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = 8 * kq + j; v[j] = k < R ? sp[static_cast<int64_t>(k) * C_ + c] : 0.f; }
        plane = static_cast<int64_t>(p.batch) * C_ * Rp;
        d = p.dst_t + (static_cast<int64_t>(by) * C_ + c) * Rp + 8 * kq;
    }
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hx2_split2(v[2 * j] * s, v[2 * j + 1] * s, hi[j], lo[j]);
    *reinterpret_cast<u32x4_t*>(d) = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u32x4_t*>(d + plane) = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
}
// fills everything of `p` but the scale source; false when the arguments do not fit (alignment)
bool hx2_split_both_args(const float* src, int64_t src_bs, int32_t R, int32_t C_, int32_t batch, void* dst_n, void* dst_t, Hx2SplitBoth* p);
int hx2_split_planes_both(const float* src, int64_t src_bs, int32_t R, int32_t C_, int32_t batch, void* dst_n, void* dst_t, const Hx2Scale& sc,
                          hipStream_t st, uint32_t* blkmax_quantity = nullptr, int32_t nblk = 0);
bool hx2_supported(const void* Ap, int64_t a_plane, int64_t a_row, int64_t a_bs, int32_t K);
// row_inv (optional): A's rows carry PER-ROW power-of-two scales instead of one per tensor — row m of batch entry z was scaled by
// 1 / row_inv[z * row_inv_bs + m] when its planes were written (k_elu_grad_q: no pass over the tensor for a global maximum first); the
// epilogue multiplies the row by it.  `sa` must then describe no scale.
int gemm_hx2_batched(int32_t M, int32_t N, int32_t K, const void* Ap, int64_t a_plane, int64_t a_row, int64_t a_bs, const void* Bplanes,
                     const OutputDesc& C, const GemmBatch& bt, const Hx2Scale& sa, const Hx2Scale& sb, hipStream_t st, int32_t a_shared_k = 0,
                     const float* row_inv = nullptr, int64_t row_inv_bs = 0);
bool hx2_kmajor_supported(const void* Ap, int64_t lda, int64_t a_plane, int64_t a_bs, const void* Bp, int64_t ldb, int64_t b_plane,
                          int64_t b_bs, int32_t M, int32_t N);
int gemm_hx2_kmajor_batched(int32_t M, int32_t N, int32_t K, const void* Ap, int64_t lda, int64_t a_plane, int64_t a_bs, const void* Bp,
                            int64_t ldb, int64_t b_plane, int64_t b_bs, int32_t batch, int32_t split_k, float* partial, const void* zeros,
                            const Hx2Scale& sa, const Hx2Scale& sb, hipStream_t st, int32_t a_shared_m = 0, const float* k_inv = nullptr,
                            int64_t k_inv_bs = 0);
// k_inv (optional): B's rows (the contraction index k) carry per-row power-of-two scales s_k = 1 / k_inv[z * k_inv_bs + k] while `sb`
// names the tensor's global maximum (scale s_g <= every s_k).  The kernel brings the rows to the common scale on the OTHER operand: A's
// fragments are multiplied by s_g / s_k (a power of two <= 1, exact in half precision down to its subnormals) as they leave LDS.
// C = epilogue(sum over splits of partial[batch][split][M][N]) through C's addressing; transpose: element (m, n) -> C(n, m)
int splitk_reduce(const float* partial, int32_t splits, int32_t M, int32_t N, const OutputDesc& C, int64_t c_bs, int32_t batch,
                  int32_t epilogue, bool transpose, hipStream_t st);

}  // namespace recon
