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
// f16, to fp16), lo = (value - hi) rounded the same way.
struct Halves4 { uint2 hi, lo; };
__device__ __forceinline__ Halves4 split_halves4(const float (&v)[4], bool f16) {
    Halves4 r;
    if (f16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = f16_sat(v[e]); lo[e] = f16_sat(v[e] - (float)hi[e]); }
        r.hi = __builtin_bit_cast(uint2, hi); r.lo = __builtin_bit_cast(uint2, lo);
    } else {
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        b4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
        r.hi = __builtin_bit_cast(uint2, hi); r.lo = __builtin_bit_cast(uint2, lo);
    }
    return r;
}

}  // namespace adx
