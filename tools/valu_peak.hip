// valu_peak: what one MI355X SIMD sustains for the vector instructions the renderer is made of,
// in shader clocks per wave-instruction (clock-independent) and as a chip-wide rate at the clock
// the chip actually held.
//
// Method.  The probe loop is one inline-asm block on hard-coded vector registers, so operand
// banks are controlled (a register's bank is its number mod 4): 16 accumulators v16..v31, second
// operands v32..v47, third operands v48..v63; instruction i reads v(16+i), v(32+(i+1)%16),
// v(48+(i+2)%16) -- three different banks -- and the "same bank" variants read v(32+i), v(48+i).
// 64 instructions per loop iteration, a scalar counter.  Every wave brackets its loop with
// s_memtime (shader clock) and s_memrealtime (constant-rate clock), so a run yields
//   sustained shader clock               = d(memtime) / d(memrealtime) * realtime rate
//   cycles per wave-instruction per SIMD = that clock * SIMDs * 64 / (lane-operations per second)
// without assuming any clock (the rate is the whole launch's work over its HIP-event time).  The
// grid is CUs * 4 * W waves, W = 1, 2, 4, 8 per SIMD.  Each measurement runs >= 50 ms.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HIPCHK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if(e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                             \
        }                                                                             \
    } while(0)

// one instruction per accumulator, operand registers rotated so the three sources sit in three banks
#define ROT16(OP)                                                                                                      \
    OP(16, 33, 50) OP(17, 34, 51) OP(18, 35, 52) OP(19, 36, 53) OP(20, 37, 54) OP(21, 38, 55) OP(22, 39, 56) OP(23, 40, 57) \
    OP(24, 41, 58) OP(25, 42, 59) OP(26, 43, 60) OP(27, 44, 61) OP(28, 45, 62) OP(29, 46, 63) OP(30, 47, 48) OP(31, 32, 49)
// all three sources in the accumulator's bank
#define SAME16(OP)                                                                                                     \
    OP(16, 32, 48) OP(17, 33, 49) OP(18, 34, 50) OP(19, 35, 51) OP(20, 36, 52) OP(21, 37, 53) OP(22, 38, 54) OP(23, 39, 55) \
    OP(24, 40, 56) OP(25, 41, 57) OP(26, 42, 58) OP(27, 43, 59) OP(28, 44, 60) OP(29, 45, 61) OP(30, 46, 62) OP(31, 47, 63)

#define I_FMA(a, b, c) "v_fma_f32 v" #a ", v" #a ", v" #b ", v" #c "\n"
#define I_FMA_S(a, b, c) "v_fma_f32 v" #a ", v" #a ", s22, v" #c "\n"              /* a scalar-register operand */
#define I_FMAC_S(a, b, c) "v_fmac_f32 v" #a ", s22, v" #c "\n"
#define I_MUL(a, b, c) "v_mul_f32 v" #a ", v" #a ", v" #b "\n"
#define I_ADD(a, b, c) "v_add_f32 v" #a ", v" #a ", v" #c "\n"
#define I_ADD_S(a, b, c) "v_add_f32 v" #a ", s22, v" #a "\n"
#define I_PKFMA(a, b, c) "v_pk_fma_f32 v[" #a ":" #a "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #c ":" #c "+1]\n"
/* packed FMA with a scalar-register pair as one source (both halves read the pair's low word): two scalar-operand FMAs an instruction */
#define I_PKFMA_S(a, b, c) "v_pk_fma_f32 v[" #a ":" #a "+1], v[" #a ":" #a "+1], s[22:23], v[" #c ":" #c "+1] op_sel_hi:[1,0,1]\n"
#define I_MULLO(a, b, c) "v_mul_lo_u32 v" #a ", v" #a ", v" #b "\n"
#define I_ADDU(a, b, c) "v_add_u32 v" #a ", v" #a ", v" #b "\n"
#define I_ADDU_S(a, b, c) "v_add_u32 v" #a ", s22, v" #a "\n"
#define I_XOR(a, b, c) "v_xor_b32 v" #a ", v" #a ", v" #b "\n"
#define I_CVT(a, b, c) "v_cvt_f32_u32 v" #a ", v" #b "\n"
#define I_CVT_SDWA(a, b, c) "v_cvt_f32_u32_sdwa v" #a ", v" #b " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define I_CMP(a, b, c) "v_cmp_le_f32 vcc, v" #a ", v" #b "\n"
#define I_RCP(a, b, c) "v_rcp_f32 v" #a ", v" #b "\n"
#define I_SQRT(a, b, c) "v_sqrt_f32 v" #a ", v" #b "\n"
#define I_MAX3(a, b, c) "v_max3_f32 v" #a ", v" #a ", v" #b ", v" #c "\n"
#define I_MOV_S(a, b, c) "v_mov_b32 v" #a ", s22\n"
// the fast camera pass's missing sample: key add (scalar operand), two conversions, four fmas, one compare
#define I_SAMPLE(a, b, c)                                                                                              \
    "v_add_u32 v" #a ", s22, v" #b "\n"                                                                               \
    "v_cvt_f32_u32_sdwa v" #b ", v" #a " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"                        \
    "v_cvt_f32_u32_sdwa v" #c ", v" #a " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"                        \
    "v_fma_f32 v0, v" #b ", v1, v2\n"                                                                                  \
    "v_fma_f32 v3, v" #c ", v4, v5\n"                                                                                  \
    "v_fma_f32 v3, v" #c ", v3, v6\n"                                                                                  \
    "v_fma_f32 v0, v" #b ", v0, v3\n"                                                                                  \
    "v_cmp_le_f32 vcc, 0, v0\n"

// one quarter-rate instruction among three full-rate ones: do they share the issue slots?
#define I_RCP_FMA3(a, b, c)                                                                                            \
    "v_rcp_f32 v" #a ", v" #a "\n"                                                                                     \
    "v_fma_f32 v" #b ", v" #b ", v" #c ", v" #c "\n"                                                                   \
    "v_fma_f32 v" #b ", v" #b ", v" #c ", v" #c "\n"                                                                   \
    "v_fma_f32 v" #b ", v" #b ", v" #c ", v" #c "\n"
// the fast camera pass's hitting sample: step, three fmas, square root, add, reciprocal, fma, add
#define I_HIT(a, b, c)                                                                                                 \
    "v_add_f32 v" #b ", v" #c ", v" #b "\n"                                                                            \
    "v_fma_f32 v0, v1, v" #b ", v2\n"                                                                                  \
    "v_fma_f32 v0, v" #b ", v0, v5\n"                                                                                  \
    "v_fma_f32 v3, v" #b ", v4, v5\n"                                                                                  \
    "v_sqrt_f32 v0, v0\n"                                                                                              \
    "v_add_f32 v0, v3, v0\n"                                                                                           \
    "v_rcp_f32 v0, v0\n"                                                                                               \
    "v_fma_f32 v" #a ", v" #b ", v0, v" #a "\n"                                                                        \
    "v_add_f32 v" #a ", v" #a ", v0\n"

// the hitting sample with the reciprocal refined from the previous sample's (two fmas) instead of v_rcp_f32
#define I_HIT_NEWTON(a, b, c)                                                                                          \
    "v_add_f32 v" #b ", v" #c ", v" #b "\n"                                                                            \
    "v_fma_f32 v0, v1, v" #b ", v2\n"                                                                                  \
    "v_fma_f32 v0, v" #b ", v0, v5\n"                                                                                  \
    "v_fma_f32 v3, v" #b ", v4, v5\n"                                                                                  \
    "v_sqrt_f32 v0, v0\n"                                                                                              \
    "v_add_f32 v0, v3, v0\n"                                                                                           \
    "v_fma_f32 v3, -v0, v" #c ", 1.0\n"                                                                                \
    "v_fma_f32 v0, v3, v" #c ", v" #c "\n"                                                                             \
    "v_fma_f32 v" #a ", v" #b ", v0, v" #a "\n"                                                                        \
    "v_add_f32 v" #a ", v" #a ", v0\n"

// the hitting sample with no quarter-rate instruction at all: the hit distance refined from the
// previous sample's by one Newton step on dd t^2 - 2 h t + c = 0 (T = t / 2, rho ~ 1 / sqrt(disc)
// carried along and refined beside it); T and rho are carried from sample to sample (v7, v8)
#define I_HIT_NEWTON_T(a, b, c)                                                                                        \
    "v_add_f32 v" #b ", v" #c ", v" #b "\n"                                                                            \
    "v_fma_f32 v0, v" #b ", v" #b ", v2\n"                                                                             \
    "v_fma_f32 v3, v" #b ", v4, v5\n"                                                                                  \
    "v_fma_f32 v6, v0, v7, -v3\n"                                                                                      \
    "v_fma_f32 v0, v7, v6, v1\n"                                                                                       \
    "v_fma_f32 v6, 2.0, v6, v3\n"                                                                                      \
    "v_fma_f32 v6, v6, v8, 1.0\n"                                                                                      \
    "v_fma_f32 v8, v8, v6, v8\n"                                                                                       \
    "v_fma_f32 v7, v0, v8, v7\n"                                                                                       \
    "v_fma_f32 v" #a ", v" #b ", v7, v" #a "\n"                                                                        \
    "v_add_f32 v9, v9, v7\n"

// what makes an instruction mix slower than its parts: one fully dependent chain; a destination that
// is none of the sources; the hit mix with its two quarter-rate instructions replaced by adds
#define I_FMA_CHAIN(a, b, c) "v_fma_f32 v16, v16, v" #b ", v" #c "\n"
#define I_FMA_DST(a, b, c) "v_fma_f32 v" #a ", v" #b ", v" #c ", v1\n"
#define I_HIT_NO_TRANS(a, b, c)                                                                                        \
    "v_add_f32 v" #b ", v" #c ", v" #b "\n"                                                                            \
    "v_fma_f32 v0, v1, v" #b ", v2\n"                                                                                  \
    "v_fma_f32 v0, v" #b ", v0, v5\n"                                                                                  \
    "v_fma_f32 v3, v" #b ", v4, v5\n"                                                                                  \
    "v_add_f32 v0, v0, v0\n"                                                                                           \
    "v_add_f32 v0, v3, v0\n"                                                                                           \
    "v_add_f32 v0, v0, v0\n"                                                                                           \
    "v_fma_f32 v" #a ", v" #b ", v0, v" #a "\n"                                                                        \
    "v_add_f32 v" #a ", v" #a ", v0\n"

// one reciprocal among N independent fmas
#define FMA_B(b, c) "v_fma_f32 v" #b ", v" #b ", v" #c ", v" #c "\n"
#define I_RCP_FMA1(a, b, c) "v_rcp_f32 v" #a ", v" #a "\n" FMA_B(b, c)
#define I_RCP_FMA7(a, b, c) "v_rcp_f32 v" #a ", v" #a "\n" FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c)
#define I_RCP_FMA15(a, b, c) I_RCP_FMA7(a, b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c) FMA_B(b, c)
// two reciprocals back to back among 6 fmas
#define I_RCP2_FMA6(a, b, c) "v_rcp_f32 v" #a ", v" #a "\n v_rcp_f32 v" #c ", v" #c "\n" FMA_B(b, a) FMA_B(b, a) FMA_B(b, a) FMA_B(b, a) FMA_B(b, a) FMA_B(b, a)

// the same nine instructions for four samples at a time, the quarter-rate ones back to back
#define HIT_HEAD(u, d, h) "v_add_f32 v" #u ", v48, v" #u "\n v_fma_f32 v" #d ", v1, v" #u ", v2\n v_fma_f32 v" #d ", v" #u ", v" #d ", v5\n v_fma_f32 v" #h ", v" #u ", v4, v5\n"
#define HIT_TAIL(u, d, a, b) "v_fma_f32 v" #a ", v" #u ", v" #d ", v" #a "\n v_add_f32 v" #b ", v" #b ", v" #d "\n"
#define HIT_GROUPED                                                                                                    \
    HIT_HEAD(32, 20, 24) HIT_HEAD(33, 21, 25) HIT_HEAD(34, 22, 26) HIT_HEAD(35, 23, 27)                                \
    "v_sqrt_f32 v20, v20\n v_sqrt_f32 v21, v21\n v_sqrt_f32 v22, v22\n v_sqrt_f32 v23, v23\n"                        \
    "v_add_f32 v20, v24, v20\n v_add_f32 v21, v25, v21\n v_add_f32 v22, v26, v22\n v_add_f32 v23, v27, v23\n"        \
    "v_rcp_f32 v20, v20\n v_rcp_f32 v21, v21\n v_rcp_f32 v22, v22\n v_rcp_f32 v23, v23\n"                            \
    HIT_TAIL(32, 20, 16, 28) HIT_TAIL(33, 21, 17, 29) HIT_TAIL(34, 22, 18, 30) HIT_TAIL(35, 23, 19, 31)

// packed instructions use register pairs: 8 pair-accumulators
#define PAIR8(OP) OP(16, 34, 52) OP(18, 36, 54) OP(20, 38, 56) OP(22, 40, 58) OP(24, 42, 60) OP(26, 44, 62) OP(28, 46, 48) OP(30, 32, 50)

struct stamp {
    unsigned long long cycles, ticks;
};

#define PROBE_KERNEL(NAME, BODY64)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(stamp *out, int iters, float operand) {                              \
        unsigned long long t0, t1, r0, r1;                                                                             \
        asm volatile("v_mov_b32 v0, 1.0\n v_mov_b32 v1, 0x3a83126f\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 1.0\n"         \
                     "v_mov_b32 v4, 0x3a83126f\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, -1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, -1.0\n v_mov_b32 v9, 0\n" ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9"); \
        asm volatile(                                                                                                  \
            "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n"                     \
            "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n"                     \
            "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n"                     \
            "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n"                     \
            "v_mov_b32 v32, 0x3f7ff972\n v_mov_b32 v33, 0x3f7ff972\n v_mov_b32 v34, 0x3f7ff972\n v_mov_b32 v35, 0x3f7ff972\n" \
            "v_mov_b32 v36, 0x3f7ff972\n v_mov_b32 v37, 0x3f7ff972\n v_mov_b32 v38, 0x3f7ff972\n v_mov_b32 v39, 0x3f7ff972\n" \
            "v_mov_b32 v40, 0x3f7ff972\n v_mov_b32 v41, 0x3f7ff972\n v_mov_b32 v42, 0x3f7ff972\n v_mov_b32 v43, 0x3f7ff972\n" \
            "v_mov_b32 v44, 0x3f7ff972\n v_mov_b32 v45, 0x3f7ff972\n v_mov_b32 v46, 0x3f7ff972\n v_mov_b32 v47, 0x3f7ff972\n" \
            "v_mov_b32 v48, 0x38d1b717\n v_mov_b32 v49, 0x38d1b717\n v_mov_b32 v50, 0x38d1b717\n v_mov_b32 v51, 0x38d1b717\n" \
            "v_mov_b32 v52, 0x38d1b717\n v_mov_b32 v53, 0x38d1b717\n v_mov_b32 v54, 0x38d1b717\n v_mov_b32 v55, 0x38d1b717\n" \
            "v_mov_b32 v56, 0x38d1b717\n v_mov_b32 v57, 0x38d1b717\n v_mov_b32 v58, 0x38d1b717\n v_mov_b32 v59, 0x38d1b717\n" \
            "v_mov_b32 v60, 0x38d1b717\n v_mov_b32 v61, 0x38d1b717\n v_mov_b32 v62, 0x38d1b717\n v_mov_b32 v63, 0x38d1b717\n" \
            ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33",   \
                "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",   \
                "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");                       \
        t0 = __builtin_amdgcn_s_memtime();                                                                             \
        r0 = __builtin_amdgcn_s_memrealtime();                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(t0), "s"(r0));                                                      \
        asm volatile("s_mov_b32 s20, %0\n s_mov_b32 s22, %1\n s_mov_b32 s23, %1\n"                                                       \
                     "1:\n" BODY64 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                 \
                     :: "s"(iters), "s"(operand)                                                                       \
                     : "s20", "s22", "s23", "scc", "vcc", "v0", "v3", "v6", "v7", "v8", "v9", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
                       "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",      \
                       "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",      \
                       "v60", "v61", "v62", "v63");                                                                    \
        t1 = __builtin_amdgcn_s_memtime();                                                                             \
        r1 = __builtin_amdgcn_s_memrealtime();                                                                         \
        if((threadIdx.x & 63u) == 0u) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = stamp{t1 - t0, r1 - r0};     \
    }

#define X4(B) B B B B
PROBE_KERNEL(k_fma, X4(ROT16(I_FMA)))
PROBE_KERNEL(k_fma_same_bank, X4(SAME16(I_FMA)))
PROBE_KERNEL(k_fma_sgpr, X4(ROT16(I_FMA_S)))
PROBE_KERNEL(k_fmac_sgpr, X4(ROT16(I_FMAC_S)))
PROBE_KERNEL(k_mul, X4(ROT16(I_MUL)))
PROBE_KERNEL(k_add, X4(ROT16(I_ADD)))
PROBE_KERNEL(k_add_sgpr, X4(ROT16(I_ADD_S)))
PROBE_KERNEL(k_pk_fma, X4(PAIR8(I_PKFMA)) X4(PAIR8(I_PKFMA)))
PROBE_KERNEL(k_pk_fma_sgpr, X4(PAIR8(I_PKFMA_S)) X4(PAIR8(I_PKFMA_S)))
PROBE_KERNEL(k_mul_lo_u32, X4(ROT16(I_MULLO)))
PROBE_KERNEL(k_add_u32, X4(ROT16(I_ADDU)))
PROBE_KERNEL(k_add_u32_sgpr, X4(ROT16(I_ADDU_S)))
PROBE_KERNEL(k_xor, X4(ROT16(I_XOR)))
PROBE_KERNEL(k_cvt_f32_u32, X4(ROT16(I_CVT)))
PROBE_KERNEL(k_cvt_sdwa, X4(ROT16(I_CVT_SDWA)))
PROBE_KERNEL(k_cmp_f32, X4(ROT16(I_CMP)))
PROBE_KERNEL(k_rcp, X4(ROT16(I_RCP)))
PROBE_KERNEL(k_sqrt, X4(ROT16(I_SQRT)))
PROBE_KERNEL(k_max3, X4(ROT16(I_MAX3)))
PROBE_KERNEL(k_mov_sgpr, X4(ROT16(I_MOV_S)))
PROBE_KERNEL(k_camera_sample, ROT16(I_SAMPLE))      // 16 samples x 8 instructions = 128 per iteration
PROBE_KERNEL(k_rcp_fma3, X4(ROT16(I_RCP_FMA3)))     // 64 groups of 4 instructions
PROBE_KERNEL(k_camera_hit, ROT16(I_HIT))            // 16 samples x 9 instructions = 144 per iteration
PROBE_KERNEL(k_camera_hit_grouped, X4(HIT_GROUPED))
PROBE_KERNEL(k_camera_hit_newton, ROT16(I_HIT_NEWTON))      // 16 samples x 10 instructions
PROBE_KERNEL(k_camera_hit_newton_t, ROT16(I_HIT_NEWTON_T))  // 16 samples x 11 instructions
PROBE_KERNEL(k_rcp_fma1, X4(ROT16(I_RCP_FMA1)))
PROBE_KERNEL(k_rcp_fma7, ROT16(I_RCP_FMA7))
PROBE_KERNEL(k_rcp_fma15, ROT16(I_RCP_FMA15))
PROBE_KERNEL(k_rcp2_fma6, ROT16(I_RCP2_FMA6))
PROBE_KERNEL(k_fma_chain, X4(ROT16(I_FMA_CHAIN)))
PROBE_KERNEL(k_fma_dst, X4(ROT16(I_FMA_DST)))
PROBE_KERNEL(k_hit_no_trans, ROT16(I_HIT_NO_TRANS))

// FP32 MFMA beside the vector unit: v_mfma_f32_32x32x2_f32 on four independent accumulator tiles
// (a[0:63]), alone and with FILL v_fma_f32 issued after each -- does the vector unit keep its rate
// while the matrix pipe runs?  One "instruction" of the report is one MFMA (4,096 flop a wave) plus
// its fillers.
#define MFMA4 "v_mfma_f32_32x32x2_f32 a[0:15], v32, v33, a[0:15]\n" FILL "v_mfma_f32_32x32x2_f32 a[16:31], v34, v35, a[16:31]\n" FILL \
              "v_mfma_f32_32x32x2_f32 a[32:47], v36, v37, a[32:47]\n" FILL "v_mfma_f32_32x32x2_f32 a[48:63], v38, v39, a[48:63]\n" FILL
#define MFMA_KERNEL(NAME)                                                                                              \
    __global__ __launch_bounds__(256) void NAME(stamp *out, int iters, float operand) {                              \
        unsigned long long t0, t1, r0, r1;                                                                             \
        asm volatile("v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n"           \
                     "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n"           \
                     "v_mov_b32 v32, 0x3a83126f\n v_mov_b32 v33, 0x3a83126f\n v_mov_b32 v34, 0x3a83126f\n v_mov_b32 v35, 0x3a83126f\n" \
                     "v_mov_b32 v36, 0x3a83126f\n v_mov_b32 v37, 0x3a83126f\n v_mov_b32 v38, 0x3a83126f\n v_mov_b32 v39, 0x3a83126f\n" \
                     "v_mov_b32 v48, 0x38d1b717\n v_mov_b32 v49, 0x3f7ff972\n"                                         \
                     ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v48", "v49"); \
        t0 = __builtin_amdgcn_s_memtime();                                                                             \
        r0 = __builtin_amdgcn_s_memrealtime();                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(t0), "s"(r0));                                                      \
        asm volatile("s_mov_b32 s20, %0\n"                                                                             \
                     "1:\n" MFMA4 MFMA4 MFMA4 MFMA4 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"   \
                     :: "s"(iters), "s"(operand)                                                                       \
                     : "s20", "scc", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8",  \
                       "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25",      \
                       "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42",     \
                       "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59",     \
                       "a60", "a61", "a62", "a63");                                                                    \
        t1 = __builtin_amdgcn_s_memtime();                                                                             \
        r1 = __builtin_amdgcn_s_memrealtime();                                                                         \
        if((threadIdx.x & 63u) == 0u) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = stamp{t1 - t0, r1 - r0};     \
    }
#define F8 "v_fma_f32 v16, v16, v49, v48\n v_fma_f32 v17, v17, v49, v48\n v_fma_f32 v18, v18, v49, v48\n v_fma_f32 v19, v19, v49, v48\n" \
           "v_fma_f32 v20, v20, v49, v48\n v_fma_f32 v21, v21, v49, v48\n v_fma_f32 v22, v22, v49, v48\n v_fma_f32 v23, v23, v49, v48\n"
#define FILL ""
MFMA_KERNEL(k_mfma_f32)
#undef FILL
#define FILL F8
MFMA_KERNEL(k_mfma_f32_8fma)
#undef FILL
#define FILL F8 F8
MFMA_KERNEL(k_mfma_f32_16fma)
#undef FILL
#define FILL F8 F8 F8
MFMA_KERNEL(k_mfma_f32_24fma)
#undef FILL
#define FILL F8 F8 F8 F8
MFMA_KERNEL(k_mfma_f32_32fma)
#undef FILL

typedef void (*probe_fn)(stamp *, int, float);

const char *g_only = nullptr;       // with an argument: only the probes whose name contains it

void run(const char *name, probe_fn fn, int per_iteration, double flop_per_lane_op, int cus, double realtime_hz, stamp *d_out) {
    if(g_only && !std::strstr(name, g_only)) return;
    for(int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = cus * waves_per_simd;         // 256 threads = 4 waves, one per SIMD of a CU
        const int n_waves = blocks * 4;
        // size the loop for >= 50 ms: calibrate on a short run
        int iters = 2000;
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        float ms = 0.0f;
        for(int pass = 0; pass < 2; pass++) {
            HIPCHK(hipEventRecord(e0));
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.9999f);
            HIPCHK(hipEventRecord(e1));
            HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            if(pass == 0) iters = static_cast<int>(iters * 60.0 / (ms > 0.01f ? ms : 0.01f));
        }
        std::vector<stamp> h(static_cast<size_t>(n_waves));
        HIPCHK(hipMemcpy(h.data(), d_out, sizeof(stamp) * h.size(), hipMemcpyDeviceToHost));
        double cycles = 0.0, ticks = 0.0;
        for(const stamp &s : h) {
            cycles += static_cast<double>(s.cycles);
            ticks += static_cast<double>(s.ticks);
        }
        const double instr_per_wave = static_cast<double>(iters) * per_iteration;
        const double clock_ghz = cycles / ticks * realtime_hz * 1e-9;
        const double lane_ops = instr_per_wave * n_waves * 64.0 / (ms * 1e-3);
        const double cyc_per_instr = clock_ghz * 1e9 * (cus * 4.0) * 64.0 / lane_ops;
        std::printf("%-16s waves/SIMD %d  %6.3f cyc/wave-instr/SIMD  clock %5.3f GHz  %7.2f T lane-ops/s  %7.2f TFLOP/s  (%.1f ms)\n", name,
                    waves_per_simd, cyc_per_instr, clock_ghz, lane_ops * 1e-12, lane_ops * flop_per_lane_op * 1e-12, ms);
        HIPCHK(hipEventDestroy(e0));
        HIPCHK(hipEventDestroy(e1));
    }
}

int main(int argc, char **argv) {
    if(argc > 1) g_only = argv[1];
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    int wall_khz = 0;
    HIPCHK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s CUs %d max clock %d kHz realtime clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate, wall_khz);
    stamp *d_out = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(stamp) * static_cast<size_t>(cus) * 8 * 4));
    const double hz = wall_khz * 1e3;
    run("fma", k_fma, 64, 2, cus, hz, d_out);
    run("fma_same_bank", k_fma_same_bank, 64, 2, cus, hz, d_out);
    run("fma_sgpr_operand", k_fma_sgpr, 64, 2, cus, hz, d_out);
    run("fmac_sgpr_operand", k_fmac_sgpr, 64, 2, cus, hz, d_out);
    run("mul", k_mul, 64, 1, cus, hz, d_out);
    run("add", k_add, 64, 1, cus, hz, d_out);
    run("add_sgpr_operand", k_add_sgpr, 64, 1, cus, hz, d_out);
    run("pk_fma", k_pk_fma, 64, 4, cus, hz, d_out);
    run("pk_fma_sgpr_operand", k_pk_fma_sgpr, 64, 4, cus, hz, d_out);
    run("mul_lo_u32", k_mul_lo_u32, 64, 0, cus, hz, d_out);
    run("add_u32", k_add_u32, 64, 0, cus, hz, d_out);
    run("add_u32_sgpr_op", k_add_u32_sgpr, 64, 0, cus, hz, d_out);
    run("xor", k_xor, 64, 0, cus, hz, d_out);
    run("cvt_f32_u32", k_cvt_f32_u32, 64, 0, cus, hz, d_out);
    run("cvt_f32_u32_sdwa", k_cvt_sdwa, 64, 0, cus, hz, d_out);
    run("cmp_f32", k_cmp_f32, 64, 0, cus, hz, d_out);
    run("rcp", k_rcp, 64, 1, cus, hz, d_out);
    run("sqrt", k_sqrt, 64, 1, cus, hz, d_out);
    run("max3", k_max3, 64, 0, cus, hz, d_out);
    run("mov_from_sgpr", k_mov_sgpr, 64, 0, cus, hz, d_out);
    run("camera_sample_mix", k_camera_sample, 128, 1, cus, hz, d_out);
    run("rcp+3fma", k_rcp_fma3, 256, 7.0 / 4.0, cus, hz, d_out);
    run("camera_hit_mix", k_camera_hit, 144, 13.0 / 9.0, cus, hz, d_out);
    run("camera_hit_by_4", k_camera_hit_grouped, 144, 13.0 / 9.0, cus, hz, d_out);
    run("camera_hit_newton", k_camera_hit_newton, 160, 16.0 / 10.0, cus, hz, d_out);
    run("camera_hit_newton_t", k_camera_hit_newton_t, 176, 20.0 / 11.0, cus, hz, d_out);
    run("rcp+1fma", k_rcp_fma1, 128, 1.5, cus, hz, d_out);
    run("rcp+7fma", k_rcp_fma7, 128, 15.0 / 8.0, cus, hz, d_out);
    run("rcp+15fma", k_rcp_fma15, 256, 31.0 / 16.0, cus, hz, d_out);
    run("2rcp+6fma", k_rcp2_fma6, 128, 14.0 / 8.0, cus, hz, d_out);
    run("fma_one_chain", k_fma_chain, 64, 2, cus, hz, d_out);
    run("fma_dst_not_src", k_fma_dst, 64, 2, cus, hz, d_out);
    run("hit_mix_no_trans", k_hit_no_trans, 144, 13.0 / 9.0, cus, hz, d_out);
    // one unit = one v_mfma_f32_32x32x2_f32 (64 flop a lane) and the v_fma_f32 issued after it (2 flop a lane each)
    run("mfma_f32_32x32x2", k_mfma_f32, 16, 64, cus, hz, d_out);
    run("mfma+8fma", k_mfma_f32_8fma, 16, 64 + 16, cus, hz, d_out);
    run("mfma+16fma", k_mfma_f32_16fma, 16, 64 + 32, cus, hz, d_out);
    run("mfma+24fma", k_mfma_f32_24fma, 16, 64 + 48, cus, hz, d_out);
    run("mfma+32fma", k_mfma_f32_32fma, 16, 64 + 64, cus, hz, d_out);
    HIPCHK(hipFree(d_out));
    return 0;
}
