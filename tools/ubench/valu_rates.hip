// tools/ubench/valu_rates.hip -- issue-rate microbenchmark for the VALU / LDS instructions the FSR
// kernels are made of (gfx950).  Prints cycles per wave-instruction per SIMD at full occupancy.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

// shader-clock ticks (s_memtime) and constant 100 MHz ticks (s_memrealtime) around the measured loop, one wave: the ratio is the
// clock the loop actually ran at, so that the table can be stated in true cycles (the chip is power-managed: 2.1-2.4 GHz)
__device__ unsigned long long g_ticks[2];
#define OVR_T0 const unsigned long long ovr_c0 = clock64(), ovr_w0 = wall_clock64();
#define OVR_T1 if (blockIdx.x == 0 && threadIdx.x == 0) { g_ticks[0] = clock64() - ovr_c0; g_ticks[1] = wall_clock64() - ovr_w0; }
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(NAME, BODY, ...)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                 \
    {                                                                                             \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;        \
        __shared__ float lds[4096];                                                               \
        lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;                                        \
        __syncthreads();                                                                          \
        unsigned addr = (threadIdx.x * 16) & 8191;                                                \
        (void)addr;                                                                               \
        OVR_T0 for (int i = 0; i < iters; ++i) { REP8(BODY) }                                             \
        OVR_T1 out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7; \
    }

// 8 independent instructions per BODY -> 64 per loop iteration
#define V8(INS)                                                     \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a0) : "v"(b0));        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a1) : "v"(b1));        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a2) : "v"(b2));        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a3) : "v"(b3));        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a4) : "v"(b4));        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a5) : "v"(b5));        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a6) : "v"(b6));        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(a7) : "v"(b7));
#define V8_2(INS)                                                   \
    asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(b0));            \
    asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(b1));            \
    asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(b2));            \
    asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(b3));            \
    asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(b4));            \
    asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(b5));            \
    asm volatile(INS " %0, %0, %1" : "+v"(a6) : "v"(b6));            \
    asm volatile(INS " %0, %0, %1" : "+v"(a7) : "v"(b7));
#define V8_1(INS)                                                   \
    asm volatile(INS " %0, %1" : "+v"(a0) : "v"(b0));                \
    asm volatile(INS " %0, %1" : "+v"(a1) : "v"(b1));                \
    asm volatile(INS " %0, %1" : "+v"(a2) : "v"(b2));                \
    asm volatile(INS " %0, %1" : "+v"(a3) : "v"(b3));                \
    asm volatile(INS " %0, %1" : "+v"(a4) : "v"(b4));                \
    asm volatile(INS " %0, %1" : "+v"(a5) : "v"(b5));                \
    asm volatile(INS " %0, %1" : "+v"(a6) : "v"(b6));                \
    asm volatile(INS " %0, %1" : "+v"(a7) : "v"(b7));
// packed f32: 64-bit register pairs
#define P4(INS)                                                                                    \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p0) : "v"(q0));                                        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p1) : "v"(q1));                                        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p2) : "v"(q2));                                        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p3) : "v"(q3));                                        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p4) : "v"(q4));                                        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p5) : "v"(q5));                                        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p6) : "v"(q6));                                        \
    asm volatile(INS " %0, %0, %1, %0" : "+v"(p7) : "v"(q7));
#define P4_2(INS)                                                                                  \
    asm volatile(INS " %0, %0, %1" : "+v"(p0) : "v"(q0));                                            \
    asm volatile(INS " %0, %0, %1" : "+v"(p1) : "v"(q1));                                            \
    asm volatile(INS " %0, %0, %1" : "+v"(p2) : "v"(q2));                                            \
    asm volatile(INS " %0, %0, %1" : "+v"(p3) : "v"(q3));                                            \
    asm volatile(INS " %0, %0, %1" : "+v"(p4) : "v"(q4));                                            \
    asm volatile(INS " %0, %0, %1" : "+v"(p5) : "v"(q5));                                            \
    asm volatile(INS " %0, %0, %1" : "+v"(p6) : "v"(q6));                                            \
    asm volatile(INS " %0, %0, %1" : "+v"(p7) : "v"(q7));

typedef float float2v __attribute__((ext_vector_type(2)));
#define PKERNEL(NAME, BODY)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                 \
    {                                                                                             \
        float s = seed + threadIdx.x;                                                             \
        float2v p0 = {s, s + 1}, p1 = {s + 2, s + 3}, p2 = {s + 4, s + 5}, p3 = {s + 6, s + 7}, p4 = {s + 8, s + 9}, p5 = {s + 10, s + 11}, p6 = {s + 12, s + 13}, p7 = {s + 14, s + 15}; \
        float2v q0 = p0 * .5f, q1 = p1 * .5f, q2 = p2 * .5f, q3 = p3 * .5f, q4 = p4 * .5f, q5 = p5 * .5f, q6 = p6 * .5f, q7 = p7 * .5f; \
        OVR_T0 for (int i = 0; i < iters; ++i) { REP8(BODY) }                                             \
        float2v r = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7 + q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;  \
        OVR_T1 out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;                                           \
    }

KERNEL(k_fma_f32, V8("v_fma_f32"))
KERNEL(k_mul_f32, V8_2("v_mul_f32"))
KERNEL(k_add_f32, V8_2("v_add_f32"))
KERNEL(k_min_f32, V8_2("v_min_f32"))
KERNEL(k_min3_f32, V8("v_min3_f32"))
KERNEL(k_med3_f32, V8("v_med3_f32"))
KERNEL(k_pk_fma_f16, V8("v_pk_fma_f16"))
KERNEL(k_pk_mul_f16, V8_2("v_pk_mul_f16"))
KERNEL(k_pk_min_f16, V8_2("v_pk_min_f16"))
KERNEL(k_fma_mix_f32, V8("v_fma_mix_f32"))
KERNEL(k_fma_mixlo_f16, V8("v_fma_mixlo_f16"))
KERNEL(k_cvt_f32_f16, V8_1("v_cvt_f32_f16"))
KERNEL(k_cvt_f16_f32, V8_1("v_cvt_f16_f32"))
KERNEL(k_cvt_f32_ubyte0, V8_1("v_cvt_f32_ubyte0"))
KERNEL(k_rcp_f32, V8_1("v_rcp_f32"))

// Are the class costs additive when the classes are interleaved, as they are in real kernels?  Eight independent instructions per
// BODY as above, mixed: if a mix costs less than the sum of its parts the classes overlap (a transcendental among ordinary VALU
// work issues in ~4 cycles; only back-to-back transcendentals pay 8), and tools/isa_costs.py must price them accordingly.
#define F1(R, B) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(R) : "v"(B));
#define T1(R, B) asm volatile("v_rcp_f32 %0, %1" : "+v"(R) : "v"(B));
#define S1(R, B) asm volatile("v_min3_f32 %0, %0, %1, %0" : "+v"(R) : "v"(B));
#define MIX_1T_7F T1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) F1(a4, b4) F1(a5, b5) F1(a6, b6) F1(a7, b7)
#define MIX_2T_6F T1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) T1(a4, b4) F1(a5, b5) F1(a6, b6) F1(a7, b7)
#define MIX_1T_7S T1(a0, b0) S1(a1, b1) S1(a2, b2) S1(a3, b3) S1(a4, b4) S1(a5, b5) S1(a6, b6) S1(a7, b7)
#define MIX_4S_4F S1(a0, b0) F1(a1, b1) S1(a2, b2) F1(a3, b3) S1(a4, b4) F1(a5, b5) S1(a6, b6) F1(a7, b7)
KERNEL(k_mix_1t_7f, MIX_1T_7F)
KERNEL(k_mix_2t_6f, MIX_2T_6F)
KERNEL(k_mix_1t_7s, MIX_1T_7S)
KERNEL(k_mix_4s_4f, MIX_4S_4F)

// The full pair matrix: strict alternation X Y X Y ... of two classes (4 + 4 per BODY), scalar and packed registers in one kernel.
// tools/isa_costs.py prices an instruction by its class AND the class of the VALU instruction in front of it from these.
#define P1(R, Q) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(R) : "v"(Q));
#define MKERNEL(NAME, BODY)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                 \
    {                                                                                             \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;        \
        float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};                        \
        float2v q0 = p0 * .5f, q1 = p1 * .5f, q2 = p2 * .5f, q3 = p3 * .5f;                        \
        OVR_T0 for (int i = 0; i < iters; ++i) { REP8(BODY) }                                      \
        float2v r = p0 + p1 + p2 + p3 + q0 + q1 + q2 + q3;                                         \
        OVR_T1 out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7; \
    }
MKERNEL(k_alt_fs, F1(a0, b0) S1(a1, b1) F1(a2, b2) S1(a3, b3) F1(a4, b4) S1(a5, b5) F1(a6, b6) S1(a7, b7))
MKERNEL(k_alt_fp, F1(a0, b0) P1(p0, q0) F1(a2, b2) P1(p1, q1) F1(a4, b4) P1(p2, q2) F1(a6, b6) P1(p3, q3))
MKERNEL(k_alt_ft, F1(a0, b0) T1(a1, b1) F1(a2, b2) T1(a3, b3) F1(a4, b4) T1(a5, b5) F1(a6, b6) T1(a7, b7))
MKERNEL(k_alt_sp, S1(a0, b0) P1(p0, q0) S1(a2, b2) P1(p1, q1) S1(a4, b4) P1(p2, q2) S1(a6, b6) P1(p3, q3))
MKERNEL(k_alt_st, S1(a0, b0) T1(a1, b1) S1(a2, b2) T1(a3, b3) S1(a4, b4) T1(a5, b5) S1(a6, b6) T1(a7, b7))
MKERNEL(k_alt_pt, P1(p0, q0) T1(a1, b1) P1(p1, q1) T1(a3, b3) P1(p2, q2) T1(a5, b5) P1(p3, q3) T1(a7, b7))
// runs of two: X X Y Y X X Y Y (does the pairing depend on adjacency only?)
MKERNEL(k_run2_fs, F1(a0, b0) F1(a1, b1) S1(a2, b2) S1(a3, b3) F1(a4, b4) F1(a5, b5) S1(a6, b6) S1(a7, b7))
MKERNEL(k_run2_fp, F1(a0, b0) F1(a1, b1) P1(p0, q0) P1(p1, q1) F1(a4, b4) F1(a5, b5) P1(p2, q2) P1(p3, q3))
// the mix of the EASU pair block: 3 packed, 3 fast, 1 slow per 7 (+1 fast)
MKERNEL(k_easu_like, P1(p0, q0) F1(a0, b0) P1(p1, q1) F1(a1, b1) S1(a2, b2) P1(p2, q2) F1(a3, b3) F1(a4, b4))

// ---- round 5 (Next #1): longer runs, the s_nop the hazard recogniser inserts, and v_cndmask on VCC as compiled code uses it
// runs of four and eight: is the F<->P (F<->S) interaction paid per class TRANSITION or per adjacent pair?
MKERNEL(k_run4_fs, F1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) S1(a4, b4) S1(a5, b5) S1(a6, b6) S1(a7, b7))
MKERNEL(k_run4_fp, F1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) P1(p0, q0) P1(p1, q1) P1(p2, q2) P1(p3, q3))
#define F8BLK F1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) F1(a4, b4) F1(a5, b5) F1(a6, b6) F1(a7, b7)
#define P8BLK P1(p0, q0) P1(p1, q1) P1(p2, q2) P1(p3, q3) P1(p0, q1) P1(p1, q2) P1(p2, q3) P1(p3, q0)
#define S8BLK S1(a0, b0) S1(a1, b1) S1(a2, b2) S1(a3, b3) S1(a4, b4) S1(a5, b5) S1(a6, b6) S1(a7, b7)
MKERNEL(k_run8_fp, F8BLK P8BLK)            // 16 per BODY
MKERNEL(k_run16_fp, F8BLK F8BLK P8BLK P8BLK) // 32 per BODY
MKERNEL(k_run8_fs, F8BLK S8BLK)
// F S P cycles: does an S between F and P buy the F<->P transition back?
MKERNEL(k_cyc_fsp, F1(a0, b0) S1(a1, b1) P1(p0, q0) F1(a2, b2) S1(a3, b3) P1(p1, q1) F1(a4, b4) S1(a5, b5) P1(p2, q2))
// EASU's cell shape: 4 F (the two squared distances), 9 P; and the same with one S behind each of the first F's
MKERNEL(k_cell_4f9p, F1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) P8BLK P1(p0, q2))
MKERNEL(k_cell_4fs9p, F1(a0, b0) S1(a4, b4) F1(a1, b1) S1(a5, b5) F1(a2, b2) F1(a3, b3) P8BLK P1(p0, q2))
MKERNEL(k_cell_4f2s9p, F1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) P8BLK P1(p0, q2) S1(a4, b4) S1(a5, b5))
// s_nop 0 behind every VALU instruction (what the compiler puts between a v_rcp / v_pk_mul and its consumer): cost of the nop itself
#define FN1(R, B) asm volatile("v_fma_f32 %0, %0, %1, %0\n\ts_nop 0" : "+v"(R) : "v"(B));
#define PN1(R, Q) asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n\ts_nop 0" : "+v"(R) : "v"(Q));
MKERNEL(k_fma_snop, FN1(a0, b0) FN1(a1, b1) FN1(a2, b2) FN1(a3, b3) FN1(a4, b4) FN1(a5, b5) FN1(a6, b6) FN1(a7, b7))
MKERNEL(k_fma_snop_1in8, FN1(a0, b0) F1(a1, b1) F1(a2, b2) F1(a3, b3) F1(a4, b4) F1(a5, b5) F1(a6, b6) F1(a7, b7))
MKERNEL(k_pk_snop, PN1(p0, q0) PN1(p1, q1) PN1(p2, q2) PN1(p3, q3) PN1(p0, q1) PN1(p1, q2) PN1(p2, q3) PN1(p3, q0))
// dependent chains (each instruction consumes the previous result): the hardware interlocks? cost of dependence vs 8 independent streams
#define FD1(R, B) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(R) : "v"(B));
MKERNEL(k_fma_dep, FD1(a0, b0) FD1(a0, b1) FD1(a0, b2) FD1(a0, b3) FD1(a0, b4) FD1(a0, b5) FD1(a0, b6) FD1(a0, b7))
MKERNEL(k_pk_dep, P1(p0, q0) P1(p0, q1) P1(p0, q2) P1(p0, q3) P1(p0, q0) P1(p0, q1) P1(p0, q2) P1(p0, q3))
// v_cndmask on VCC the way compiled code has it: a compare writes VCC, two independent fast ops sit in the hazard slots, the select
// reads VCC (e32); the same with an SGPR-pair mask (e64).  4 VALU instructions per statement.
#define CSEL_VCC(R, B, X, Y) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_fma_f32 %2, %2, %1, %2\n\tv_fma_f32 %3, %3, %1, %3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(R), "+v"(B), "+v"(X), "+v"(Y) :: "vcc");
#define CSEL_SGPR(R, B, X, Y) asm volatile("v_cmp_lt_f32 %4, %0, %1\n\tv_fma_f32 %2, %2, %1, %2\n\tv_fma_f32 %3, %3, %1, %3\n\tv_cndmask_b32 %0, %0, %1, %4" : "+v"(R), "+v"(B), "+v"(X), "+v"(Y), "=&s"(msk));
MKERNEL(k_csel_vcc, CSEL_VCC(a0, b0, a4, a5) CSEL_VCC(a1, b1, a6, a7) CSEL_VCC(a2, b2, a4, a5) CSEL_VCC(a3, b3, a6, a7))
__global__ __launch_bounds__(256) void k_csel_sgpr(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f;
    unsigned long long msk;
    OVR_T0 for (int i = 0; i < iters; ++i) { REP8(CSEL_SGPR(a0, b0, a4, a5) CSEL_SGPR(a1, b1, a6, a7) CSEL_SGPR(a2, b2, a4, a5) CSEL_SGPR(a3, b3, a6, a7)) }
    OVR_T1 out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3;
}
// v_cndmask reading a VCC nobody writes in the loop, and no clobber (no compiler-inserted s_nop): the instruction alone
#define VCNDIN8                                                        \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b0)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b1)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b2)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b3)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(b4)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a5) : "v"(b5)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(b6)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(b7));
KERNEL(k_cndmask_vcc_in, VCNDIN8)

// v_cndmask on VCC, continued: 23 cycles for back-to-back e32 selects on VCC above, 4.1 with an SGPR-pair mask.  Is it the encoding
// (VOP2 e32 reads VCC implicitly) or VCC itself, and does it survive other instructions in between?
#define VCNDE64 \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a0) : "v"(b0)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a1) : "v"(b1)); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a2) : "v"(b2)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a3) : "v"(b3)); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a4) : "v"(b4)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a5) : "v"(b5)); \
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a6) : "v"(b6)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a7) : "v"(b7));
KERNEL(k_cnd_e64_vcc, VCNDE64)
// one compare, five selects on it (NVScaler's `sh ? t1 : t0` groups), two fast ops: 8 VALU per statement
#define CND5_VCC asm volatile("v_cmp_lt_f32 vcc, %0, %5\n\tv_fma_f32 %6, %6, %5, %6\n\tv_fma_f32 %7, %7, %5, %7\n\t" \
    "v_cndmask_b32 %0, %0, %5, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc\n\tv_cndmask_b32 %2, %2, %5, vcc\n\tv_cndmask_b32 %3, %3, %5, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(b0), "+v"(a6), "+v"(a7) :: "vcc");
#define CND5_SGPR asm volatile("v_cmp_lt_f32 %8, %0, %5\n\tv_fma_f32 %6, %6, %5, %6\n\tv_fma_f32 %7, %7, %5, %7\n\t" \
    "v_cndmask_b32 %0, %0, %5, %8\n\tv_cndmask_b32 %1, %1, %5, %8\n\tv_cndmask_b32 %2, %2, %5, %8\n\tv_cndmask_b32 %3, %3, %5, %8\n\tv_cndmask_b32 %4, %4, %5, %8" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(b0), "+v"(a6), "+v"(a7), "=&s"(msk));
KERNEL(k_cnd5_vcc, CND5_VCC)
__global__ __launch_bounds__(256) void k_cnd5_sgpr(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b0 = a0 * 0.5f;
    unsigned long long msk;
    OVR_T0 for (int i = 0; i < iters; ++i) { REP8(CND5_SGPR) }
    OVR_T1 out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a6 + a7 + b0;
}
// selects on VCC separated by one / three fast ops
#define CNDF(R, B, X) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_fma_f32 %2, %2, %1, %2" : "+v"(R), "+v"(B), "+v"(X));
#define CNDFFF(R, B, X, Y, Z) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_fma_f32 %2, %2, %1, %2\n\tv_fma_f32 %3, %3, %1, %3\n\tv_fma_f32 %4, %4, %1, %4" : "+v"(R), "+v"(B), "+v"(X), "+v"(Y), "+v"(Z));
KERNEL(k_cnd_alt_fma, CNDF(a0, b0, a4) CNDF(a1, b1, a5) CNDF(a2, b2, a6) CNDF(a3, b3, a7))
KERNEL(k_cnd_3fma, CNDFFF(a0, b0, a4, a5, a6) CNDFFF(a1, b1, a7, a4, a5))
KERNEL(k_rcp_f16, V8_1("v_rcp_f16"))
KERNEL(k_sub_u32, V8_2("v_sub_u32"))
KERNEL(k_lshrrev, V8_2("v_lshrrev_b32"))
KERNEL(k_cvt_pkrtz, V8_2("v_cvt_pkrtz_f16_f32"))
KERNEL(k_pk_add_f16, V8_2("v_pk_add_f16"))
KERNEL(k_dot2_f32_f16, V8("v_dot2_f32_f16"))
KERNEL(k_min_u32, V8_2("v_min_u32"))
KERNEL(k_max_i32, V8_2("v_max_i32"))
KERNEL(k_min3_u32, V8("v_min3_u32"))
KERNEL(k_and_b32, V8_2("v_and_b32"))
KERNEL(k_or_b32, V8_2("v_or_b32"))
KERNEL(k_lshl_or_b32, V8("v_lshl_or_b32"))
KERNEL(k_lshl_add_u32, V8("v_lshl_add_u32"))
KERNEL(k_add3_u32, V8("v_add3_u32"))
KERNEL(k_perm_b32, V8("v_perm_b32"))
KERNEL(k_bfe_u32, V8("v_bfe_u32"))
KERNEL(k_cvt_pk_u8_f32, V8("v_cvt_pk_u8_f32"))
KERNEL(k_cvt_u32_f32, V8_1("v_cvt_u32_f32"))
KERNEL(k_cvt_f32_u32, V8_1("v_cvt_f32_u32"))
KERNEL(k_floor_f32, V8_1("v_floor_f32"))
KERNEL(k_mov_b32, V8_1("v_mov_b32"))
KERNEL(k_fmac_f32, V8_2("v_fmac_f32"))
KERNEL(k_mul_u32_u24, V8_2("v_mul_u32_u24"))
KERNEL(k_mad_u32_u24, V8("v_mad_u32_u24"))
KERNEL(k_max_f32, V8_2("v_max_f32"))
KERNEL(k_max3_f32, V8("v_max3_f32"))
#define VCND8                                                                  \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b0) : "vcc"); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b1) : "vcc"); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b2) : "vcc"); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b3) : "vcc"); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(b4) : "vcc"); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a5) : "v"(b5) : "vcc"); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(b6) : "vcc"); \
    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(b7) : "vcc");
KERNEL(k_cndmask, VCND8)
#define VCMP8                                                        \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a0), "v"(b0) : "vcc"); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a1), "v"(b1) : "vcc"); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a2), "v"(b2) : "vcc"); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a3), "v"(b3) : "vcc"); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a4), "v"(b4) : "vcc"); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a5), "v"(b5) : "vcc"); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a6), "v"(b6) : "vcc"); \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a7), "v"(b7) : "vcc");
KERNEL(k_cmp_lt_f32, VCMP8)
// round 2: integer multiplies (address arithmetic of the staging loops), selects on an SGPR mask, DPP moves, 64-bit adds
KERNEL(k_mul_lo_u32, V8_2("v_mul_lo_u32"))
KERNEL(k_mul_hi_u32, V8_2("v_mul_hi_u32"))
KERNEL(k_med3_i32, V8("v_med3_i32"))
KERNEL(k_sub_f32, V8_2("v_sub_f32"))
#define VFMACL8                                                      \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a0) : "v"(b0)); \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a1) : "v"(b1)); \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a2) : "v"(b2)); \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a3) : "v"(b3)); \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a4) : "v"(b4)); \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a5) : "v"(b5)); \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a6) : "v"(b6)); \
    asm volatile("v_fma_f32 %0, %0, %1, %0 clamp" : "+v"(a7) : "v"(b7));
KERNEL(k_fma_f32_clamp, VFMACL8)
#define VFMAABS8                                                      \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a0) : "v"(b0)); \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a1) : "v"(b1)); \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a2) : "v"(b2)); \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a3) : "v"(b3)); \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a4) : "v"(b4)); \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a5) : "v"(b5)); \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a6) : "v"(b6)); \
    asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a7) : "v"(b7));
KERNEL(k_fma_f32_mods, VFMAABS8)
#define VCNDS8                                                                  \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(b0), "s"(m)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(b1), "s"(m)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(b2), "s"(m)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(b3), "s"(m)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a4) : "v"(b4), "s"(m)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a5) : "v"(b5), "s"(m)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(b6), "s"(m)); \
    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a7) : "v"(b7), "s"(m));
__global__ __launch_bounds__(256) void k_cndmask_sgpr(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;
    unsigned long long m = __builtin_amdgcn_ballot_w64(seed + threadIdx.x > 40.f);
    OVR_T0 for (int i = 0; i < iters; ++i) { REP8(VCNDS8) }
    OVR_T1 out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}
#define VDPP8(CTRL)                                                            \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b0)); \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(b1)); \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(b2)); \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(b3)); \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a4) : "v"(b4)); \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a5) : "v"(b5)); \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a6) : "v"(b6)); \
    asm volatile("v_mov_b32_dpp %0, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a7) : "v"(b7));
KERNEL(k_mov_dpp_row_shr, VDPP8("row_shr:1"))
KERNEL(k_mov_dpp_wave_shr, VDPP8("wave_shr:1"))
#define VADDDPP8                                                            \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(b0)); \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(b1)); \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(b2)); \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(b3)); \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a4) : "v"(b4)); \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a5) : "v"(b5)); \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a6) : "v"(b6)); \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a7) : "v"(b7));
KERNEL(k_add_f32_dpp, VADDDPP8)
#define VBPERM8                                                            \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(addr)); \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a1) : "v"(addr)); \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a2) : "v"(addr)); \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a3) : "v"(addr)); \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a4) : "v"(addr)); \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a5) : "v"(addr)); \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a6) : "v"(addr)); \
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a7) : "v"(addr));
KERNEL(k_ds_bpermute, VBPERM8)
#define VPKCL4                                                                                    \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p0) : "v"(q0));                                        \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p1) : "v"(q1));                                        \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p2) : "v"(q2));                                        \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p3) : "v"(q3));                                        \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p4) : "v"(q4));                                        \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p5) : "v"(q5));                                        \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p6) : "v"(q6));                                        \
    asm volatile("v_pk_fma_f32 %0, %0, %1, %0 clamp" : "+v"(p7) : "v"(q7));
PKERNEL(k_pk_fma_f32_clamp, VPKCL4)
PKERNEL(k_pk_fma_f32, P4("v_pk_fma_f32"))
PKERNEL(k_pk_mul_f32, P4_2("v_pk_mul_f32"))
PKERNEL(k_pk_add_f32, P4_2("v_pk_add_f32"))

// LDS reads: 8 independent reads then one wait
#define L8(INS, W)                                                                                 \
    asm volatile(INS " %0, %1" : "=v"(r0) : "v"(addr));                                              \
    asm volatile(INS " %0, %1 offset:256" : "=v"(r1) : "v"(addr));                                   \
    asm volatile(INS " %0, %1 offset:512" : "=v"(r2) : "v"(addr));                                   \
    asm volatile(INS " %0, %1 offset:768" : "=v"(r3) : "v"(addr));                                   \
    asm volatile(INS " %0, %1 offset:1024" : "=v"(r4) : "v"(addr));                                  \
    asm volatile(INS " %0, %1 offset:1280" : "=v"(r5) : "v"(addr));                                  \
    asm volatile(INS " %0, %1 offset:1536" : "=v"(r6) : "v"(addr));                                  \
    asm volatile(INS " %0, %1 offset:1792" : "=v"(r7) : "v"(addr));                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define LKERNEL(NAME, TYPE, INS, STRIDE)                                                           \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                 \
    {                                                                                             \
        __shared__ __attribute__((aligned(16))) float lds[8192];                                   \
        for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed + i;                           \
        __syncthreads();                                                                          \
        unsigned addr = (threadIdx.x & 63) * STRIDE + (threadIdx.x >> 6) * 4096;                    \
        TYPE r0, r1, r2, r3, r4, r5, r6, r7;                                                       \
        float acc = 0;                                                                            \
        OVR_T0 for (int i = 0; i < iters; ++i) { REP8(L8(INS, 0)) acc += ((float *)&r0)[0] + ((float *)&r7)[0]; } \
        OVR_T1 out[blockIdx.x * 256 + threadIdx.x] = acc + ((float *)&r1)[0] + ((float *)&r2)[0] + ((float *)&r3)[0] + ((float *)&r4)[0] + ((float *)&r5)[0] + ((float *)&r6)[0]; \
    }
typedef float float4v __attribute__((ext_vector_type(4)));
LKERNEL(k_ds_read_b32, float, "ds_read_b32", 4)
LKERNEL(k_ds_read_b64, float2v, "ds_read_b64", 8)
LKERNEL(k_ds_read_b128, float4v, "ds_read_b128", 16)
LKERNEL(k_ds_read_b64_s6, float2v, "ds_read_b64", 6 * 8 / 8 * 8)   // stride 0.75-ish pattern approximated below


__global__ void k_probe_cvt(float *out)
{
    const float vals[8] = {0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 254.5f, 255.7f, -0.7f};
    for (int i = 0; i < 8; ++i) { unsigned r = 0; asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(vals[i])); out[i] = (float)r; }
}
typedef void (*kern_t)(float *, int, float);

// Which clock do the figures below refer to?  One wave of a sustained v_fma loop reads s_memtime (shader-clock counter) and
// s_memrealtime (constant 100 MHz) around the loop: their ratio x 100 MHz is the shader clock this load actually runs at.
__global__ __launch_bounds__(256) void k_clock_probe(float *out, unsigned long long *ticks, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) { REP8(V8("v_fma_f32")) }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

static double run(kern_t k, float *d, int iters, int blocks)
{
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 8, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    return ms;
}

int main(int argc, char **argv)
{
    const char *only = argc > 1 ? argv[1] : nullptr; // run only the rows whose name contains this string
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz\n", p.gcnArchName, cus, ghz);
    const int blocks = cus * 8; // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * blocks);
    struct { const char *n; kern_t k; int per_iter; } ks[] = {
        {"v_fma_f32", k_fma_f32, 64}, {"v_mul_f32", k_mul_f32, 64}, {"v_add_f32", k_add_f32, 64}, {"v_min_f32", k_min_f32, 64},
        {"v_min3_f32", k_min3_f32, 64}, {"v_med3_f32", k_med3_f32, 64},
        {"v_pk_fma_f32", k_pk_fma_f32, 64}, {"v_pk_mul_f32", k_pk_mul_f32, 64}, {"v_pk_add_f32", k_pk_add_f32, 64},
        {"v_pk_fma_f16", k_pk_fma_f16, 64}, {"v_pk_mul_f16", k_pk_mul_f16, 64}, {"v_pk_add_f16", k_pk_add_f16, 64}, {"v_pk_min_f16", k_pk_min_f16, 64},
        {"v_fma_mix_f32", k_fma_mix_f32, 64}, {"v_fma_mixlo_f16", k_fma_mixlo_f16, 64}, {"v_dot2_f32_f16", k_dot2_f32_f16, 64},
        {"v_cvt_f32_f16", k_cvt_f32_f16, 64}, {"v_cvt_f16_f32", k_cvt_f16_f32, 64}, {"v_cvt_pkrtz_f16_f32", k_cvt_pkrtz, 64},
        {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte0, 64},
        {"v_rcp_f32", k_rcp_f32, 64}, {"mix 1 rcp + 7 fma (per 8)", k_mix_1t_7f, 64}, {"mix 2 rcp + 6 fma (per 8)", k_mix_2t_6f, 64},
        {"mix 1 rcp + 7 min3 (per 8)", k_mix_1t_7s, 64}, {"mix 4 min3 + 4 fma (per 8)", k_mix_4s_4f, 64},
        {"alt F S  (fma, min3)", k_alt_fs, 64}, {"alt F P  (fma, pk_fma)", k_alt_fp, 64}, {"alt F T  (fma, rcp)", k_alt_ft, 64},
        {"alt S P  (min3, pk_fma)", k_alt_sp, 64}, {"alt S T  (min3, rcp)", k_alt_st, 64}, {"alt P T  (pk_fma, rcp)", k_alt_pt, 64},
        {"runs FFSS", k_run2_fs, 64}, {"runs FFPP", k_run2_fp, 64}, {"easu-like 3P 4F 1S", k_easu_like, 64}, {"v_rcp_f16", k_rcp_f16, 64}, {"v_sub_u32", k_sub_u32, 64}, {"v_lshrrev_b32", k_lshrrev, 64},
        {"v_min_u32", k_min_u32, 64}, {"v_max_i32", k_max_i32, 64}, {"v_min3_u32", k_min3_u32, 64}, {"v_max_f32", k_max_f32, 64}, {"v_max3_f32", k_max3_f32, 64},
        {"v_and_b32", k_and_b32, 64}, {"v_or_b32", k_or_b32, 64}, {"v_lshl_or_b32", k_lshl_or_b32, 64}, {"v_lshl_add_u32", k_lshl_add_u32, 64},
        {"v_add3_u32", k_add3_u32, 64}, {"v_perm_b32", k_perm_b32, 64}, {"v_bfe_u32", k_bfe_u32, 64}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8_f32, 64},
        {"v_cvt_u32_f32", k_cvt_u32_f32, 64}, {"v_cvt_f32_u32", k_cvt_f32_u32, 64}, {"v_floor_f32", k_floor_f32, 64}, {"v_mov_b32", k_mov_b32, 64},
        {"v_fmac_f32", k_fmac_f32, 64}, {"v_mul_u32_u24", k_mul_u32_u24, 64}, {"v_mad_u32_u24", k_mad_u32_u24, 64},
        {"v_cndmask_b32", k_cndmask, 64}, {"v_cmp_lt_f32", k_cmp_lt_f32, 64},
        {"v_mul_lo_u32", k_mul_lo_u32, 64}, {"v_mul_hi_u32", k_mul_hi_u32, 64}, {"v_med3_i32", k_med3_i32, 64}, {"v_sub_f32", k_sub_f32, 64},
        {"v_fma_f32 clamp", k_fma_f32_clamp, 64}, {"v_fma_f32 |a|,-c mods", k_fma_f32_mods, 64}, {"v_pk_fma_f32 clamp", k_pk_fma_f32_clamp, 64},
        {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr, 64}, {"v_mov_b32_dpp row_shr:1", k_mov_dpp_row_shr, 64},
        {"v_mov_b32_dpp wave_shr:1", k_mov_dpp_wave_shr, 64}, {"v_add_f32_dpp row_shr:1", k_add_f32_dpp, 64}, {"ds_bpermute_b32 (+wait)", k_ds_bpermute, 64},
        {"ds_read_b32", k_ds_read_b32, 64}, {"ds_read_b64", k_ds_read_b64, 64}, {"ds_read_b128", k_ds_read_b128, 64},
        // round 5
        {"runs FFFFSSSS", k_run4_fs, 64}, {"runs FFFFPPPP", k_run4_fp, 64}, {"runs F8 P8", k_run8_fp, 128}, {"runs F16 P16", k_run16_fp, 256}, {"runs F8 S8", k_run8_fs, 128},
        {"cycle F S P", k_cyc_fsp, 72}, {"cell 4F 9P", k_cell_4f9p, 104}, {"cell F S F S F F 9P", k_cell_4fs9p, 120}, {"cell 4F 9P 2S", k_cell_4f2s9p, 120},
        {"v_fma_f32 + s_nop 0 (each)", k_fma_snop, 64}, {"v_fma_f32, s_nop 0 behind 1 of 8", k_fma_snop_1in8, 64}, {"v_pk_fma_f32 + s_nop 0 (each)", k_pk_snop, 64},
        {"v_fma_f32 dependent chain", k_fma_dep, 64}, {"v_pk_fma_f32 dependent chain", k_pk_dep, 64},
        {"cmp->vcc, 2 fma, cndmask vcc (per instr)", k_csel_vcc, 128}, {"cmp->sgpr, 2 fma, cndmask sgpr (per instr)", k_csel_sgpr, 128},
        {"v_cndmask_b32 vcc (no writer, no nop)", k_cndmask_vcc_in, 64},
        {"v_cndmask_b32_e64 .., vcc (8 in a row)", k_cnd_e64_vcc, 64}, {"cmp->vcc, 2 fma, 5 cndmask vcc (per instr)", k_cnd5_vcc, 64}, {"cmp->sgpr, 2 fma, 5 cndmask sgpr (per instr)", k_cnd5_sgpr, 64},
        {"alt cndmask vcc, fma", k_cnd_alt_fma, 64}, {"cndmask vcc + 3 fma", k_cnd_3fma, 64},
    };
    {
        unsigned long long *dt, ht[2] = {0, 0};
        hipMalloc(&dt, 16);
        int wall_khz = 0;
        hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
        for (int rep = 0; rep < 3; ++rep) { // ~25 ms each: long enough for the power controller to settle
            hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, 0, d, dt, 40000, 1.0f);
            hipDeviceSynchronize();
            hipMemcpy(ht, dt, 16, hipMemcpyDeviceToHost);
            printf("clock probe (sustained v_fma_f32, 8 waves/SIMD): s_memtime %llu ticks, s_memrealtime %llu ticks (wall clock rate %d kHz) -> ratio %.4f = %.1f MHz if s_memtime is the shader clock\n",
                   ht[0], ht[1], wall_khz, (double)ht[0] / (double)ht[1], (double)ht[0] / (double)ht[1] * wall_khz * 1e-3);
        }
        hipFree(dt);
    }
    const int iters = 4000;
    for (auto &k : ks) {
        if (only && !strstr(k.n, only)) continue;
        double ms = run(k.k, d, iters, blocks);
        // per SIMD: 8 waves each issuing iters*per_iter instructions
        double instr_per_simd = 8.0 * iters * k.per_iter;
        double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
        unsigned long long ht[2] = {0, 1};
        hipMemcpyFromSymbol(ht, HIP_SYMBOL(g_ticks), sizeof(ht));
        const double mhz = (double)ht[0] / (double)ht[1] * 100.0; // s_memrealtime: 100 MHz
        printf("%-22s %8.3f ms  %6.2f cycles / wave-instruction / SIMD (at %.2f GHz nominal)  | ran at %6.1f MHz -> %5.2f true cycles\n", k.n, ms, cyc, ghz, mhz,
               cyc * mhz / (ghz * 1e3));
    }
    hipLaunchKernelGGL(k_probe_cvt, dim3(1), dim3(1), 0, 0, d);
    float h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32 of {0.49,0.5,0.51,1.5,2.5,254.5,255.7,-0.7} = %g %g %g %g %g %g %g %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    hipFree(d);
    return 0;
}
