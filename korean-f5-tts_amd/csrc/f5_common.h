// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the F5-TTS engine.
// Wave = 64 lanes everywhere; MFMA tiles are 16x16 (bf16 / f16: 16x16x32, f32: 16x16x4).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define F5_WAVE 64

// (operand precision ids F5_PREC_F32 / F5_PREC_BF16 / F5_PREC_F16 live in include/f5_hip.h)
// activation ids used by GEMM epilogues
enum { F5_ACT_NONE = 0, F5_ACT_GELU_TANH = 1, F5_ACT_GELU_ERF = 2, F5_ACT_SILU = 3, F5_ACT_MISH = 4, F5_ACT_LOGCLAMP = 5 };

namespace f5 {

__device__ __forceinline__ float gelu_tanh(float x) {
    // torch: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
    // 0.5 (1 + tanh u) == sigmoid(2u) == 1 / (1 + 2^(-2 u log2 e)): ONE bare v_exp_f32 and one v_rcp_f32 (1 ulp each).  __expf() is
    // v_exp_f32 wrapped in a range reduction with an exec-masked branch: ~18 instructions per element, and at 128 elements per lane
    // the GELU epilogue of the FF1 projection was VALU-bound (csrc/gemm.h EpiStore::staged).  An overflowing 2^(..) gives x * 0.
    const float c0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, c1 = c0 * 0.044715f;
    const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(c1, x * x, c0));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float mish(float x) {
    // x * tanh(softplus(x)), torch softplus threshold 20.  With n = e^x: tanh(log(1 + n)) = ((1+n)^2 - 1) / ((1+n)^2 + 1)
    // = t / (t + 2), t = n (n + 2) -- no cancellation for either sign of x, one exp and one rcp instead of the
    // log1pf + tanhf expansions (which made every kernel that inlines it ~1 KB larger per element)
    const float n = expf(fminf(x, 20.0f));
    const float t = n * (n + 2.0f);
    return x > 20.0f ? x : x * t * __frcp_rn(t + 2.0f);
}
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case F5_ACT_GELU_TANH: return gelu_tanh(v);
        case F5_ACT_GELU_ERF: return gelu_erf(v);
        case F5_ACT_SILU: return silu(v);
        case F5_ACT_MISH: return mish(v);
        case F5_ACT_LOGCLAMP: return logf(fmaxf(v, 1e-5f));  // mel.clamp(min=1e-5).log(), modules.py:103
        default: return v;
    }
}

// f32 -> f16 must be a conversion of the ROUNDED f32 value.  Left to itself hipcc folds a multiply (or multiply-add) that feeds the
// conversion into v_fma_mixlo_f16 -- the product rounded ONCE, straight to f16 -- in some instantiations of an epilogue and not in
// others: two tile shapes then disagree in the last f16 bit of a few elements (tools/cfg_dep.py f16: 6e-4 after two layers), the
// same defect -ffp-contract=off removes for f32.  Vector stores convert PAIRS (v_cvt_pk_f16_f32: a plain conversion, nothing to
// fold into); the scalar form passes the value through an empty asm (opaque at the conversion, emits nothing).  An asm barrier
// on every element of the vector stores made hipcc wait for each group of three loads in the LayerNorm kernel (different register
// allocation -> write-after-write waits): 34 -> 51 us per launch at 32,768 rows.
__device__ __forceinline__ float f16_src(float v) {
    asm("" : "+v"(v));
    return v;
}
typedef __attribute__((ext_vector_type(2))) float f32x2_cv;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_cv;
__device__ __forceinline__ unsigned cvt2_f16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cv{a, b}, f16x2_cv));
}
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)f16_src(v); }

__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
    bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ __forceinline__ void store4(f16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<u32x2*>(p) = u32x2{cvt2_f16(a, b), cvt2_f16(c, d)};
}

// four f32 -> 8 bytes of 16-bit type TO (the value store4 would write)
template <typename TO> __device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d);
template <> __device__ __forceinline__ u32x2 pack4<bf16_t>(float a, float b, float c, float d) {
    bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    return __builtin_bit_cast(u32x2, v);
}
template <> __device__ __forceinline__ u32x2 pack4<f16_t>(float a, float b, float c, float d) {
    return u32x2{cvt2_f16(a, b), cvt2_f16(c, d)};
}

// ---- F5_PREC_F16X3 helpers: x = hi + lo with hi = f16(x), lo = f16(x - hi) (gemm2.h MODE 3, attn.h) ----
__device__ __forceinline__ void split4_f16(const u32x4& c, u32x2& hi, u32x2& lo) {
    typedef __attribute__((ext_vector_type(2))) float v2f;
    typedef __attribute__((ext_vector_type(2))) _Float16 v2h;
    const f32x4 x = __builtin_bit_cast(f32x4, c);
    const v2f a{x[0], x[1]}, b{x[2], x[3]};
    const v2h ah = __builtin_convertvector(a, v2h), bh = __builtin_convertvector(b, v2h);
    const v2h al = __builtin_convertvector(a - __builtin_convertvector(ah, v2f), v2h);
    const v2h bl = __builtin_convertvector(b - __builtin_convertvector(bh, v2f), v2h);
    hi = u32x2{__builtin_bit_cast(unsigned, ah), __builtin_bit_cast(unsigned, bh)};
    lo = u32x2{__builtin_bit_cast(unsigned, al), __builtin_bit_cast(unsigned, bl)};
}
// Elements col .. col+3 (col % 4 == 0) of an f32 row stored ALREADY SPLIT, in the layout the split-operand GEMM reads (the same
// 128 bytes per 32 elements as split_planar_kernel gives the weights: chunk g = hi of k = 4g..4g+3, 16+4g..16+4g+3, chunk 4+g = lo):
// the producer of a GEMM A operand (LayerNorm, attention, the GELU epilogue) pays the split once instead of every wave that
// reads the fragment.  `row` must be 128-byte aligned (ld % 32 == 0).
__device__ __forceinline__ void store4_planar(float* row, int col, float a, float b, float c, float d) {
    u32x2 hi, lo;
    split4_f16(__builtin_bit_cast(u32x4, f32x4{a, b, c, d}), hi, lo);
    const int kk = col & 31;
    char* p = reinterpret_cast<char*>(row + (col & ~31)) + ((kk & 15) >> 2) * 16 + (kk >> 4) * 8;
    *reinterpret_cast<u32x2*>(p) = hi;
    *reinterpret_cast<u32x2*>(p + 64) = lo;
}
template <typename TO> __device__ __forceinline__ void store4_at(TO* row, int col, int planar, float a, float b, float c, float d) {
    if constexpr (std::is_same_v<TO, float>) {
        if (planar) { store4_planar(row, col, a, b, c, d); return; }
    }
    store4(row + col, a, b, c, d);
}

// eight f32 -> one 16-byte MFMA operand fragment of 16-bit type T
template <typename T> __device__ __forceinline__ u32x4 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7);
template <> __device__ __forceinline__ u32x4 pack8<bf16_t>(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    bf16x8 v = {(bf16_t)a0, (bf16_t)a1, (bf16_t)a2, (bf16_t)a3, (bf16_t)a4, (bf16_t)a5, (bf16_t)a6, (bf16_t)a7};
    return __builtin_bit_cast(u32x4, v);
}
template <> __device__ __forceinline__ u32x4 pack8<f16_t>(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    return u32x4{cvt2_f16(a0, a1), cvt2_f16(a2, a3), cvt2_f16(a4, a5), cvt2_f16(a6, a7)};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace f5
