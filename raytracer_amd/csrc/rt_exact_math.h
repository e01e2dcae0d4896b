// rt_exact_math.h -- correctly rounded fp32 division and square root for "tame" operands, at
// roughly half the instruction cost of the sequences hipcc emits for `a / b` and `sqrtf(a)`.
//
// hipcc's IEEE division is  v_div_scale x2, v_rcp_f32, six fma-class steps, v_div_fmas,
// v_div_fixup; its sqrtf is a range scale, v_sqrt_f32, a +-1 ulp residual test, unscale and a
// class test.  The scale / fixup / class steps only act on operands near the ends of the
// exponent range (or zero, infinity, NaN).  For operands inside the ranges checked by
// tame_for_div / tame_for_sqrt they are the identity, and what remains is the arithmetic below,
// step for step, so the results are the same bits.  Callers test tameness and fall back to the
// plain operators otherwise; tools/check_exact_math.hip compares both against the compiler's
// operators on the GPU over billions of operands.
//
// The gain comes from sharing: normalising a vector divides three numerators by one length, and
// the refined reciprocal (v_rcp_f32 + one Newton step) is computed once.
#pragma once

#include <hip/hip_runtime.h>

namespace rt_exact {

constexpr float kTameLo = 0x1p-40f;   // well inside every threshold of v_div_scale_f32 / sqrt scaling
constexpr float kTameHi = 0x1p+40f;

// |v| in [2^-40, 2^40]: no scaling in v_div_scale_f32 for any pair of such operands (exponent
// difference <= 80 < 96, numerator exponent > 23, quotient and reciprocal normal), no range scaling
// in sqrtf (threshold 2^-96), no zero / infinity / NaN.
__device__ __forceinline__ bool tame(float v) {
    const float a = __builtin_fabsf(v);
    return a >= kTameLo && a <= kTameHi;
}

__device__ __forceinline__ bool tame3(float x, float y, float z) {
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    return lo >= kTameLo && hi <= kTameHi;
}

// The same test for callers that already know |x|,|y|,|z| <= 2^40 (the renderer bounds every
// scene coordinate by 1e11 < 2^37, so every vector it normalises is): only the small side is
// left to check.
__device__ __forceinline__ bool tame3_bounded(float x, float y, float z) {
    return __builtin_fminf(__builtin_fminf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z)) >= kTameLo;
}

// v_rcp_f32 (1 ulp) + one Newton step: the reciprocal the division sequence multiplies by
__device__ __forceinline__ float refined_rcp(float y) {
    const float r = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// x / y given r = refined_rcp(y); x and y tame
__device__ __forceinline__ float div_with_rcp(float x, float y, float r) {
    float q = x * r;
    float rem = __builtin_fmaf(-y, q, x);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-y, q, x);
    return __builtin_fmaf(rem, r, q);
}

__device__ __forceinline__ float div_tame(float x, float y) { return div_with_rcp(x, y, refined_rcp(y)); }

// sqrtf(m) for tame m > 0: v_sqrt_f32 is within 1 ulp; pick among {s-1ulp, s, s+1ulp} by the
// sign of the exact residuals m - s_dn*s and m - s_up*s
// sqrt_tame is valid for any finite m >= 2^-96 (below that hipcc's sequence rescales); callers
// with possibly tiny or zero m test this first.
constexpr float kSqrtTameLo = 0x1p-90f;

__device__ __forceinline__ float sqrt_tame(float m) {
    float s = __builtin_amdgcn_sqrtf(m);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, m);
    const float r_up = __builtin_fmaf(-s_up, s, m);
    s = (r_dn <= 0.0f) ? s_dn : s;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
}

}  // namespace rt_exact
