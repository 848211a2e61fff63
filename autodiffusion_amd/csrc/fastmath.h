// Single-instruction transcendental forms (v_rcp_f32 / v_exp_f32 / v_rsq_f32, 1 ulp).  The library spellings (1.0f / x,
// exp2f, 1 / sqrtf) expand to correctly rounded / denormal-safe sequences of 5-10 instructions, which dominate the
// inner loops of the softmax and GELU code.  Results that would be denormal flush to zero.
#pragma once
#include <hip/hip_runtime.h>

namespace adx {

__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rsqrt_fast(float x) { return __builtin_amdgcn_rsqf(x); }

// x + (value of x in lane ^ 16) and x + (value in lane ^ 32) with the gfx950 row / half swaps: vector-ALU instructions,
// where __shfl_xor goes through the LDS crossbar (ds_bpermute) and costs an LDS round trip inside a dependent chain.
__device__ __forceinline__ float xor16_add(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// fp32 -> fp16 with saturation: finite values beyond the fp16 range become +-65504 instead of infinity.
__device__ __forceinline__ _Float16 f16_sat(float v) { return (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f); }

// Four consecutive values as 16-bit operand halves of a matrix product (8 bytes each): hi = the value rounded to bf16 (or, with
// f16, to fp16, saturating), lo = (value - hi) rounded the same way.  Scalars in, packed words out: an array parameter here made
// the callers keep their value arrays in scratch memory.
struct Halves4 { uint2 hi, lo; };
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 p = {f16_sat(a), f16_sat(b)};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ Halves4 split_halves4(float v0, float v1, float v2, float v3, bool f16) {
    Halves4 r;
    if (f16) {
        const _Float16 h0 = f16_sat(v0), h1 = f16_sat(v1), h2 = f16_sat(v2), h3 = f16_sat(v3);
        r.hi = make_uint2(pack_f16x2(v0, v1), pack_f16x2(v2, v3));
        r.lo = make_uint2(pack_f16x2(v0 - (float)h0, v1 - (float)h1), pack_f16x2(v2 - (float)h2, v3 - (float)h3));
    } else {
        const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1, h2 = (__bf16)v2, h3 = (__bf16)v3;
        r.hi = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        r.lo = make_uint2(pack_bf16x2(v0 - (float)h0, v1 - (float)h1), pack_bf16x2(v2 - (float)h2, v3 - (float)h3));
    }
    return r;
}

}  // namespace adx
