// check_exact_math: compares rt_exact::div_tame / div_with_rcp / sqrt_tame with the compiler's
// IEEE `a / b` and `sqrtf(a)` on the GPU, bit for bit, over pseudo-random and adversarial
// operands inside the tame range.  Prints a JSON line; exit status 1 on any mismatch.
//
//   check_exact_math [rounds]      each round = 2^26 operand pairs per test
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../raytracer_amd/csrc/rt_exact_math.h"

__device__ __forceinline__ unsigned mix(unsigned v) {
    v = v * 747796405u + 2891336453u;
    const unsigned w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}

// float with random sign and mantissa and a biased exponent drawn from [lo, hi]
__device__ __forceinline__ float make_float(unsigned bits, unsigned lo, unsigned hi) {
    const unsigned e = lo + (bits >> 9) % (hi - lo + 1);
    unsigned mant = bits & 0x7fffffu;
    const unsigned sel = (bits >> 23) & 7u;            // adversarial mantissas now and then
    if(sel == 0) mant = 0x7fffffu;
    if(sel == 1) mant = 0u;
    if(sel == 2) mant = 0x7ffffeu | (bits & 1u);
    if(sel == 3) mant = bits & 3u;
    return __uint_as_float((bits & 0x80000000u) | (e << 23) | mant);
}

__global__ __launch_bounds__(256) void check(unsigned round, unsigned long long *mismatch) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned local[3] = {0, 0, 0};
    for(unsigned k = 0; k < 64; k++) {
        const unsigned h0 = mix(gid * 64u + k + round * 0x9e3779b9u);
        const unsigned h1 = mix(h0 ^ 0x85ebca6bu);
        // tame range is 2^-40..2^40: biased exponents 87..167
        const float x = make_float(h0, 87, 167);
        const float y = make_float(h1, 87, 167);
        // narrow range around 1 as well (what the renderer mostly feeds it)
        const float xn = make_float(h0, 120, 130);
        const float yn = __builtin_fabsf(make_float(h1, 125, 129));

        if(rt_exact::tame(x) && rt_exact::tame(y)) {
            local[0] += __float_as_uint(rt_exact::div_tame(x, y)) != __float_as_uint(x / y);
        }
        local[0] += __float_as_uint(rt_exact::div_tame(xn, yn)) != __float_as_uint(xn / yn);

        // shared reciprocal: three numerators over one denominator
        const float r = rt_exact::refined_rcp(yn);
        const float x2 = make_float(mix(h1), 100, 128);
        local[1] += __float_as_uint(rt_exact::div_with_rcp(xn, yn, r)) != __float_as_uint(xn / yn);
        local[1] += __float_as_uint(rt_exact::div_with_rcp(x2, yn, r)) != __float_as_uint(x2 / yn);

        const float m = __builtin_fabsf(x);
        local[2] += __float_as_uint(rt_exact::sqrt_tame(m)) != __float_as_uint(__builtin_sqrtf(m));
        const float mn = __builtin_fabsf(xn);
        local[2] += __float_as_uint(rt_exact::sqrt_tame(mn)) != __float_as_uint(__builtin_sqrtf(mn));
        // the whole range sqrt_tame is used on: 2^-90 up to the largest finite floats
        const float mw = __builtin_fabsf(make_float(h1, 37, 254));
        if(mw >= rt_exact::kSqrtTameLo) local[2] += __float_as_uint(rt_exact::sqrt_tame(mw)) != __float_as_uint(__builtin_sqrtf(mw));
    }
    for(int t = 0; t < 3; t++)
        if(local[t]) atomicAdd(&mismatch[t], static_cast<unsigned long long>(local[t]));
}

int main(int argc, char **argv) {
    const unsigned rounds = argc > 1 ? static_cast<unsigned>(std::atoi(argv[1])) : 16u;
    unsigned long long *d = nullptr, h[3] = {0, 0, 0};
    if(hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)) != hipSuccess) return 2;
    if(hipMemset(d, 0, sizeof(h)) != hipSuccess) return 2;
    for(unsigned r = 0; r < rounds; r++) hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, r, d);
    if(hipDeviceSynchronize() != hipSuccess) return 2;
    if(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const double pairs = static_cast<double>(rounds) * 4096 * 256 * 64;
    std::printf("{\"pairs_per_test\": %.0f, \"div_mismatch\": %llu, \"div_shared_rcp_mismatch\": %llu, \"sqrt_mismatch\": %llu}\n",
                pairs * 2, h[0], h[1], h[2]);
    (void)hipFree(d);
    return (h[0] | h[1] | h[2]) ? 1 : 0;
}
