// valu_peak: measures what one MI355X sustains for the instruction kinds the renderer is made
// of, so roofline fractions are priced against measured ceilings as well as the datasheet's.
//   fma      v_fma_f32            (2 flop / lane)
//   pkfma    v_pk_fma_f32         (4 flop / lane)
//   mul+add  v_mul_f32, v_add_f32 (the exact build's unfused pairs)
//   mullo    v_mul_lo_u32         (RNG hash)
//   rcp/rsq/sqrt                  transcendental-rate ops
//   div      IEEE-correct a/b     (v_div_scale / v_rcp / v_fma chain / v_div_fmas / v_div_fixup)
// Prints one line per kind: instructions/s per lane-op and the implied TFLOP/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kIters = 4096;
constexpr int kAcc = 16;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, float seed, unsigned useed) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[kAcc];
    v2f p[kAcc];
    unsigned u[kAcc];
#pragma unroll
    for(int i = 0; i < kAcc; i++) {
        a[i] = seed + 0.001f * i + 1e-6f * tid;
        p[i] = v2f{a[i], a[i] * 0.5f};
        u[i] = useed + i * 2654435761u + tid;
    }
    const float m = 0.9999f + seed * 1e-7f, c = 1e-4f + seed * 1e-8f;
    const v2f pm = {m, m}, pc = {c, c};
    for(int it = 0; it < kIters; it++) {
#pragma unroll
        for(int i = 0; i < kAcc; i++) {
            if constexpr(KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);
            if constexpr(KIND == 1) p[i] = __builtin_elementwise_fma(p[i], pm, pc);
            if constexpr(KIND == 2) a[i] = a[i] * m + c;
            if constexpr(KIND == 3) u[i] = u[i] * 747796405u + 2891336453u;
            if constexpr(KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if constexpr(KIND == 5) a[i] = __builtin_amdgcn_rsqf(a[i]);
            if constexpr(KIND == 6) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            if constexpr(KIND == 7) a[i] = m / a[i];
            if constexpr(KIND == 8) a[i] = __builtin_sqrtf(a[i] + m);
        }
    }
    float s = 0.0f;
#pragma unroll
    for(int i = 0; i < kAcc; i++) s += a[i] + p[i].x + p[i].y + static_cast<float>(u[i]);
    if(s == 123.456f) out[tid] = s;
}

template <int KIND>
void run(const char *name, double flop_per_op, float *d_out) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for(int r = 0; r < reps; r++) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = static_cast<double>(blocks) * threads * kIters * kAcc * reps;
    const double per_s = lane_ops / (ms * 1e-3);
    std::printf("%-8s %8.3f ms  %8.2f T lane-ops/s  %8.2f TFLOP/s\n", name, ms / reps, per_s * 1e-12, per_s * flop_per_op * 1e-12);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    float *d_out = nullptr;
    if(hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(float) * 256 * 8 * 256) != hipSuccess) return 1;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    std::printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    run<0>("fma", 2, d_out);
    run<1>("pkfma", 4, d_out);
    run<2>("mul+add", 2, d_out);
    run<3>("mullo", 0, d_out);
    run<4>("rcp", 1, d_out);
    run<5>("rsq", 1, d_out);
    run<6>("sqrt_hw", 1, d_out);
    run<7>("div_ieee", 1, d_out);
    run<8>("sqrt_ieee", 1, d_out);
    hipFree(d_out);
    return 0;
}
