// valu_peak: measures what one MI355X sustains for the instruction kinds the renderer is made
// of, so roofline fractions can be priced against measured ceilings as well as the datasheet's.
// Every probe is a loop of inline-asm instructions on 16 independent registers (the compiler
// cannot pack, fuse or fold them), 8 waves per SIMD resident.
// Prints one line per kind: wave-instructions issued per second per lane and implied TFLOP/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kIters = 2048;
constexpr int kAcc = 16;

#define HIPCHK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if(e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                             \
        }                                                                             \
    } while(0)

struct scalar_bank {
    float v[16];
};

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, float seed, unsigned useed, scalar_bank bank) {
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
    float m = 0.9999f + seed * 1e-7f, c = 1e-4f + seed * 1e-8f;
    v2f pm = {m, m}, pc = {c, c};
    unsigned um = 747796405u + useed;
    for(int it = 0; it < kIters; it++) {
#pragma unroll
        for(int i = 0; i < kAcc; i++) {
            if constexpr(KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if constexpr(KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if constexpr(KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if constexpr(KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr(KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if constexpr(KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if constexpr(KIND == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if constexpr(KIND == 8) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if constexpr(KIND == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if constexpr(KIND == 10) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 11) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i]));
            if constexpr(KIND == 12) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u[i]));
            if constexpr(KIND == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 16) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(u[i]));
            if constexpr(KIND == 17) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(u[i]));
            if constexpr(KIND == 18) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 19) asm volatile("v_cmp_le_f32 vcc, 0, %0" : : "v"(a[i]) : "vcc");
            if constexpr(KIND == 20) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 21) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(um), "v"(um));
            if constexpr(KIND == 22) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 23) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 24) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(m));
            if constexpr(KIND == 25) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(um));
            if constexpr(KIND == 26) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(bank.v[0]), "v"(c));  // one scalar-register operand, the same every time
            if constexpr(KIND == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(bank.v[0]), "v"(c));     // VOP2 form
            if constexpr(KIND == 28) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if constexpr(KIND == 29) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(bank.v[0]));
            if constexpr(KIND == 30) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(bank.v[i % 16]), "v"(c));  // a different scalar register each time
            if constexpr(KIND == 31) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(bank.v[i % 16]), "v"(c));
            if constexpr(KIND == 32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(bank.v[i / 4]), "v"(c));   // each scalar register four times in a row
            if constexpr(KIND == 33) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(bank.v[i / 2]), "v"(c));   // twice in a row
            if constexpr(KIND == 34) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "s"(bank.v[i % 16]));
            if constexpr(KIND == 35) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "s"(bank.v[i % 16]));
            if constexpr(KIND == 36) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(um));
            if constexpr(KIND == 14) a[i] = m / a[i];                    // IEEE-correct division
            if constexpr(KIND == 15) a[i] = __builtin_sqrtf(a[i] + m);   // IEEE-correct sqrt (+1 add)
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
    scalar_bank bank;
    for(int i = 0; i < 16; i++) bank.v[i] = 0.9999f + 1e-6f * i;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f, 12345u, bank);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0));
    const int reps = 5;
    for(int r = 0; r < reps; r++) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0f, 12345u, bank);
    HIPCHK(hipEventRecord(e1));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_ops = static_cast<double>(blocks) * threads * kIters * kAcc * reps;
    const double per_s = lane_ops / (ms * 1e-3);
    // cycles per wave64 instruction per SIMD at 2.4 GHz: 1024 SIMDs, 64 lanes per instruction
    const double cyc = 2.4e9 * 1024.0 * 64.0 / per_s;
    std::printf("%-10s %8.3f ms  %7.2f T lane-ops/s  %7.2f TFLOP/s  %5.2f cyc/wave-instr/SIMD@2.4GHz\n", name, ms / reps,
                per_s * 1e-12, per_s * flop_per_op * 1e-12, cyc);
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
}

int main() {
    float *d_out = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(float) * 256 * 8 * 256));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    std::printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    run<0>("fma", 2, d_out);
    run<1>("pk_fma", 4, d_out);
    run<2>("mul", 1, d_out);
    run<3>("add", 1, d_out);
    run<4>("pk_mul", 2, d_out);
    run<5>("pk_add", 2, d_out);
    run<6>("mul_lo_u32", 0, d_out);
    run<7>("rcp", 1, d_out);
    run<8>("rsq", 1, d_out);
    run<9>("sqrt", 1, d_out);
    run<10>("xor", 0, d_out);
    run<11>("lshr", 0, d_out);
    run<12>("cvt_f32_u32", 0, d_out);
    run<13>("mad_u32_u24", 0, d_out);
    run<16>("cvt_sdwa_w1", 0, d_out);
    run<17>("cvt_ubyte1", 0, d_out);
    run<18>("alignbit", 0, d_out);
    run<19>("cmp_f32", 0, d_out);
    run<20>("add_u32", 0, d_out);
    run<21>("pk_fma_f16", 4, d_out);
    run<22>("pk_add_f16", 2, d_out);
    run<23>("pk_mul_f16", 2, d_out);
    run<24>("cvt_pkrtz", 0, d_out);
    run<25>("pk_max_f16", 0, d_out);
    run<26>("fma_sgpr_op", 2, d_out);
    run<27>("fmac_sgpr_op", 2, d_out);
    run<28>("max3", 0, d_out);
    run<30>("fma_sgpr_rot", 2, d_out);
    run<31>("fmac_sgpr_rot", 2, d_out);
    run<32>("fma_sgpr_x4", 2, d_out);
    run<33>("fma_sgpr_x2", 2, d_out);
    run<34>("add_u32_sgpr", 0, d_out);
    run<35>("mov_from_sgpr", 0, d_out);
    run<36>("mov_vgpr", 0, d_out);
    run<29>("sub_sgpr_op", 1, d_out);
    run<14>("div_ieee", 1, d_out);
    run<15>("sqrt_ieee", 1, d_out);
    HIPCHK(hipFree(d_out));
    return 0;
}
