// Single-instruction transcendental forms (v_rcp_f32 / v_exp_f32 / v_rsq_f32, 1 ulp).  The library spellings (1.0f / x,
// exp2f, 1 / sqrtf) expand to correctly rounded / denormal-safe sequences of 5-10 instructions, which dominate the
// inner loops of the softmax and GELU code.  Results that would be denormal flush to zero.
#pragma once
#include <hip/hip_runtime.h>

namespace adx {

__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rsqrt_fast(float x) { return __builtin_amdgcn_rsqf(x); }

}  // namespace adx
