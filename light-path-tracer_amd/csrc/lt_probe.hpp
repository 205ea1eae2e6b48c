// lt_probe.hpp -- VALU issue-cost microbenchmarks (diagnostics behind lt_valu_issue_probe).
//
// Each kernel runs the same instruction on 8 independent register chains per lane, 8 waves per
// SIMD on every CU, and the host turns the time into SIMD cycles per wave-instruction.  The roofline
// of the integrate kernel is an issue-rate roofline: these numbers say what one instruction of
// each class costs on gfx950, so the kernel's instruction mix can be priced.
#pragma once
#include <hip/hip_runtime.h>

namespace lt {

#define LT_PROBE_BODY8(INS)                                                                         \
    asm volatile(INS(0) "\n\t" INS(1) "\n\t" INS(2) "\n\t" INS(3) "\n\t" INS(4) "\n\t" INS(5) "\n\t" \
                 INS(6) "\n\t" INS(7)                                                                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                 : "v"(m), "v"(b), "s"(sc)                                                           \
                 : "vcc");

#define LT_PROBE_KERNEL(NAME, INS)                                                        \
    __global__ void __launch_bounds__(256) NAME(int iters, float sc, float *sink)          \
    {                                                                                      \
        float x = threadIdx.x * 1e-3f + 1.0f;                                              \
        float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7; \
        float m = 0.999f + x * 1e-6f, b = 1e-3f + x * 1e-7f;                               \
        if (sc == 0.0f) { m = 0.999f; b = 1e-3f; a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = 1.0f; } /* constant data */ \
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < iters; ++i) {                                                  \
            LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) \
            LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) \
        }                                                                                  \
        unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                   \
        if (s == 12345.678f) sink[0] = s;                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                         \
            ((unsigned long long *)sink)[2] = c1 - c0;                                     \
            ((unsigned long long *)sink)[3] = r1 - r0;                                     \
        }                                                                                  \
    }

// operand map: %N (N = 0..7) the chain register, %8 = m (VGPR), %9 = b (VGPR), %10 = sc (SGPR)
#define I_FMA(N) "v_fma_f32 %" #N ", %" #N ", %8, %9"
#define I_FMAC(N) "v_fmac_f32 %" #N ", %8, %9"
#define I_MUL(N) "v_mul_f32 %" #N ", %" #N ", %8"
#define I_ADD(N) "v_add_f32 %" #N ", %" #N ", %9"
#define I_FMA_S(N) "v_fma_f32 %" #N ", %" #N ", %10, %9"
#define I_FMAAK(N) "v_fmaak_f32 %" #N ", %" #N ", %8, 0x3a83126f"
#define I_MOV(N) "v_mov_b32 %" #N ", %8"
#define I_MAX(N) "v_max_f32 %" #N ", %" #N ", %9"
#define I_AND(N) "v_and_b32 %" #N ", %" #N ", %8"
#define I_CNDMASK(N) "v_cndmask_b32 %" #N ", %" #N ", %8, vcc"
#define I_CMP(N) "v_cmp_lt_f32 vcc, %" #N ", %8"
#define I_CMP_CND(N) "v_cmp_lt_f32 vcc, %" #N ", %8\n\tv_cndmask_b32 %" #N ", %" #N ", %9, vcc"
#define I_RCP(N) "v_rcp_f32 %" #N ", %" #N
#define I_SQRT(N) "v_sqrt_f32 %" #N ", %" #N
#define I_RNDNE(N) "v_rndne_f32 %" #N ", %" #N
#define I_CVT(N) "v_cvt_i32_f32 %" #N ", %" #N
#define I_XOR(N) "v_xor_b32 %" #N ", %" #N ", %8"
#define I_SIN(N) "v_sin_f32 %" #N ", %" #N
#define I_FMA_NEG(N) "v_fma_f32 %" #N ", -%" #N ", %8, %9"
#define I_MUL_E64(N) "v_mul_f32_e64 %" #N ", %" #N ", %8"
#define I_FMA_RCP(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_rcp_f32 %" #N ", %" #N

#define I_FMA_INL(N) "v_fma_f32 %" #N ", %" #N ", 2.0, %9"
#define I_MUL_INL(N) "v_mul_f32 %" #N ", 2.0, %" #N
#define I_MUL_S(N) "v_mul_f32 %" #N ", %10, %" #N
#define I_FMAC_S(N) "v_fmac_f32 %" #N ", %10, %9"
#define I_ADD_S(N) "v_add_f32 %" #N ", %10, %" #N
#define I_MIN(N) "v_min_f32 %" #N ", %" #N ", %9"
#define I_CLASS(N) "v_cmp_class_f32 vcc, %" #N ", %8"
#define I_LSHL(N) "v_lshlrev_b32 %" #N ", 1, %" #N
#define I_ADDU(N) "v_add_u32 %" #N ", %" #N ", %8"
#define I_BFI(N) "v_bfi_b32 %" #N ", %8, %" #N ", %9"
#define I_FMAMK(N) "v_fmamk_f32 %" #N ", %" #N ", 0x3a83126f, %9"
#define I_MED3(N) "v_med3_f32 %" #N ", %" #N ", %8, %9"
#define I_RCP_FMA2(N) "v_rcp_f32 %" #N ", %" #N "\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9"
#define I_FMA7_RCP(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_rcp_f32 %" #N ", %" #N
#define I_FMA_CMP(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_cmp_lt_f32 vcc, %" #N ", %8"

#define I_P1(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_max_f32 %" #N ", %" #N ", %9"
#define I_P2(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_max_f32 %" #N ", %" #N ", %9"
#define I_P3(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_mul_f32 %" #N ", %10, %" #N
#define I_P4(N) "v_cmp_lt_f32 vcc, %" #N ", %8\n\tv_cndmask_b32 %" #N ", %" #N ", %9, vcc\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9"
#define I_P5(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_cvt_i32_f32 %" #N ", %" #N
#define I_P6(N) "v_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_rcp_f32 %" #N ", %" #N "\n\tv_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_max_f32 %" #N ", %" #N ", %9"
#define I_P7(N) "v_max_f32 %" #N ", %" #N ", %9\n\tv_rcp_f32 %" #N ", %" #N
// dependent chains: every instruction reads the previous one's result (N ignored)
#define I_DEP_FMA(N) "v_fma_f32 %0, %0, %8, %9"
#define I_DEP_FMA_MAX(N) "v_fma_f32 %0, %0, %8, %9\n\tv_max_f32 %0, %0, %9"
#define I_DEP_MAX(N) "v_max_f32 %0, %0, %9"
#define I_DEP_FMA_RCP(N) "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %0, %0, %8, %9\n\tv_rcp_f32 %0, %0"
#define I_DEP2_FMA(N) "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9"
#define I_DEP_MUL(N) "v_mul_f32 %0, %0, %8"
LT_PROBE_KERNEL(k_probe_dep_fma, I_DEP_FMA)
LT_PROBE_KERNEL(k_probe_dep_fma_max, I_DEP_FMA_MAX)
LT_PROBE_KERNEL(k_probe_dep_max, I_DEP_MAX)
LT_PROBE_KERNEL(k_probe_dep_fma_rcp, I_DEP_FMA_RCP)
LT_PROBE_KERNEL(k_probe_dep2_fma, I_DEP2_FMA)
LT_PROBE_KERNEL(k_probe_dep_mul, I_DEP_MUL)
#define I_RCPD0(N) "v_rcp_f32 %0, %0\n\tv_fma_f32 %0, %0, %8, %9"
LT_PROBE_KERNEL(k_probe_rcpd0, I_RCPD0)
#define I_RCPD3(N) "v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %0, %0, %8, %9"
LT_PROBE_KERNEL(k_probe_rcpd3, I_RCPD3)
#define I_RCPD7(N) "v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %0, %0, %8, %9"
LT_PROBE_KERNEL(k_probe_rcpd7, I_RCPD7)
#define I_RCPD14(N) "v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %0, %0, %8, %9"
LT_PROBE_KERNEL(k_probe_rcpd14, I_RCPD14)
#define I_RCPD28(N) "v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\tv_fma_f32 %0, %0, %8, %9"
LT_PROBE_KERNEL(k_probe_rcpd28, I_RCPD28)
LT_PROBE_KERNEL(k_probe_fma, I_FMA)
LT_PROBE_KERNEL(k_probe_p1, I_P1)
LT_PROBE_KERNEL(k_probe_p2, I_P2)
LT_PROBE_KERNEL(k_probe_p3, I_P3)
LT_PROBE_KERNEL(k_probe_p4, I_P4)
LT_PROBE_KERNEL(k_probe_p5, I_P5)
LT_PROBE_KERNEL(k_probe_p6, I_P6)
LT_PROBE_KERNEL(k_probe_p7, I_P7)
LT_PROBE_KERNEL(k_probe_fma_inl, I_FMA_INL)
LT_PROBE_KERNEL(k_probe_mul_inl, I_MUL_INL)
LT_PROBE_KERNEL(k_probe_mul_s, I_MUL_S)
LT_PROBE_KERNEL(k_probe_fmac_s, I_FMAC_S)
LT_PROBE_KERNEL(k_probe_add_s, I_ADD_S)
LT_PROBE_KERNEL(k_probe_min, I_MIN)
LT_PROBE_KERNEL(k_probe_class, I_CLASS)
LT_PROBE_KERNEL(k_probe_lshl, I_LSHL)
LT_PROBE_KERNEL(k_probe_addu, I_ADDU)
LT_PROBE_KERNEL(k_probe_bfi, I_BFI)
LT_PROBE_KERNEL(k_probe_fmamk, I_FMAMK)
LT_PROBE_KERNEL(k_probe_med3, I_MED3)
LT_PROBE_KERNEL(k_probe_rcp_fma2, I_RCP_FMA2)
LT_PROBE_KERNEL(k_probe_fma7_rcp, I_FMA7_RCP)
LT_PROBE_KERNEL(k_probe_fma_cmp, I_FMA_CMP)
LT_PROBE_KERNEL(k_probe_fmac, I_FMAC)
LT_PROBE_KERNEL(k_probe_mul, I_MUL)
LT_PROBE_KERNEL(k_probe_add, I_ADD)
LT_PROBE_KERNEL(k_probe_fma_s, I_FMA_S)
LT_PROBE_KERNEL(k_probe_fmaak, I_FMAAK)
LT_PROBE_KERNEL(k_probe_mov, I_MOV)
LT_PROBE_KERNEL(k_probe_max, I_MAX)
LT_PROBE_KERNEL(k_probe_and, I_AND)
LT_PROBE_KERNEL(k_probe_cndmask, I_CNDMASK)
LT_PROBE_KERNEL(k_probe_cmp, I_CMP)
LT_PROBE_KERNEL(k_probe_cmp_cnd, I_CMP_CND)
LT_PROBE_KERNEL(k_probe_rcp, I_RCP)
LT_PROBE_KERNEL(k_probe_sqrt, I_SQRT)
LT_PROBE_KERNEL(k_probe_rndne, I_RNDNE)
LT_PROBE_KERNEL(k_probe_cvt, I_CVT)
LT_PROBE_KERNEL(k_probe_xor, I_XOR)
LT_PROBE_KERNEL(k_probe_sin, I_SIN)
LT_PROBE_KERNEL(k_probe_fma_neg, I_FMA_NEG)
LT_PROBE_KERNEL(k_probe_mul_e64, I_MUL_E64)
LT_PROBE_KERNEL(k_probe_fma3_rcp, I_FMA_RCP)


// packed (two floats per lane per instruction) chains: 8 independent register PAIRS per lane
#define LT_PROBE_KERNEL_PK(NAME, INS)                                                     \
    __global__ void __launch_bounds__(256) NAME(int iters, float sc, float *sink)          \
    {                                                                                      \
        typedef float v2f __attribute__((ext_vector_type(2)));                             \
        float x = threadIdx.x * 1e-3f + 1.0f;                                              \
        v2f a0 = {x, x + 1}, a1 = {x + 2, x + 3}, a2 = {x + 4, x + 5}, a3 = {x + 6, x + 7}; \
        v2f a4 = {x + 8, x + 9}, a5 = {x + 10, x + 11}, a6 = {x + 12, x + 13}, a7 = {x + 14, x + 15}; \
        v2f m = {0.999f + x * 1e-6f, 0.998f}, b = {1e-3f + x * 1e-7f, 2e-3f};                \
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < iters; ++i) {                                                  \
            LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) \
            LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) LT_PROBE_BODY8(INS) \
        }                                                                                  \
        unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                     \
        if (s.x + s.y == 12345.678f) sink[0] = s.x;                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                         \
            ((unsigned long long *)sink)[2] = c1 - c0;                                     \
            ((unsigned long long *)sink)[3] = r1 - r0;                                     \
        }                                                                                  \
    }
#define I_PK_FMA(N) "v_pk_fma_f32 %" #N ", %" #N ", %8, %9"
#define I_PK_MUL(N) "v_pk_mul_f32 %" #N ", %" #N ", %8"
#define I_PK_ADD(N) "v_pk_add_f32 %" #N ", %" #N ", %9"
#define I_PK_DEP_FMA(N) "v_pk_fma_f32 %0, %0, %8, %9"
#define I_PK_FMA_FMA(N) "v_pk_fma_f32 %" #N ", %" #N ", %8, %9\n\tv_fma_f32 %" #N ", %" #N ", %8, %9"
LT_PROBE_KERNEL_PK(k_probe_pk_fma, I_PK_FMA)
LT_PROBE_KERNEL_PK(k_probe_pk_mul, I_PK_MUL)
LT_PROBE_KERNEL_PK(k_probe_pk_add, I_PK_ADD)
LT_PROBE_KERNEL_PK(k_probe_pk_dep_fma, I_PK_DEP_FMA)

struct ProbeEntry {
    const char *name;
    void (*kernel)(int, float, float *);
    int instr_per_body; // wave-instructions per INS(N)
};

static const ProbeEntry g_probes[] = {
    {"v_fma_f32", k_probe_fma, 1},       {"v_fmac_f32", k_probe_fmac, 1},     {"v_mul_f32", k_probe_mul, 1},
    {"v_add_f32", k_probe_add, 1},       {"v_fma_f32(sgpr)", k_probe_fma_s, 1}, {"v_fmaak_f32", k_probe_fmaak, 1},
    {"v_mov_b32", k_probe_mov, 1},       {"v_max_f32", k_probe_max, 1},       {"v_and_b32", k_probe_and, 1},
    {"v_cndmask_b32", k_probe_cndmask, 1}, {"v_cmp_lt_f32", k_probe_cmp, 1},  {"v_cmp+v_cndmask", k_probe_cmp_cnd, 2},
    {"v_rcp_f32", k_probe_rcp, 1},       {"v_sqrt_f32", k_probe_sqrt, 1},     {"v_rndne_f32", k_probe_rndne, 1},
    {"v_cvt_i32_f32", k_probe_cvt, 1},   {"v_xor_b32", k_probe_xor, 1},       {"v_sin_f32", k_probe_sin, 1},
    {"v_fma_f32(neg)", k_probe_fma_neg, 1}, {"v_mul_f32_e64", k_probe_mul_e64, 1}, {"3xfma+rcp", k_probe_fma3_rcp, 4},
    {"v_fma_f32(inline 2.0)", k_probe_fma_inl, 1}, {"v_mul_f32(inline)", k_probe_mul_inl, 1},
    {"v_mul_f32(sgpr)", k_probe_mul_s, 1}, {"v_fmac_f32(sgpr)", k_probe_fmac_s, 1}, {"v_add_f32(sgpr)", k_probe_add_s, 1},
    {"v_min_f32", k_probe_min, 1}, {"v_cmp_class_f32", k_probe_class, 1}, {"v_lshlrev_b32", k_probe_lshl, 1},
    {"v_add_u32", k_probe_addu, 1}, {"v_bfi_b32", k_probe_bfi, 1}, {"v_fmamk_f32", k_probe_fmamk, 1},
    {"v_med3_f32", k_probe_med3, 1}, {"rcp+2fma", k_probe_rcp_fma2, 3}, {"7xfma+rcp", k_probe_fma7_rcp, 8},
    {"fma+cmp", k_probe_fma_cmp, 2},
    {"dep fma", k_probe_dep_fma, 1}, {"dep fma,max", k_probe_dep_fma_max, 2}, {"dep max", k_probe_dep_max, 1},
    {"dep 3fma,rcp", k_probe_dep_fma_rcp, 4}, {"2 chains fma", k_probe_dep2_fma, 2}, {"dep mul", k_probe_dep_mul, 1},
    {"rcp,0fma,dep-fma", k_probe_rcpd0, 2},
    {"rcp,3fma,dep-fma", k_probe_rcpd3, 5},
    {"rcp,7fma,dep-fma", k_probe_rcpd7, 9},
    {"rcp,14fma,dep-fma", k_probe_rcpd14, 16},
    {"rcp,28fma,dep-fma", k_probe_rcpd28, 30},
    {"v_pk_fma_f32", k_probe_pk_fma, 1}, {"v_pk_mul_f32", k_probe_pk_mul, 1}, {"v_pk_add_f32", k_probe_pk_add, 1},
    {"dep v_pk_fma_f32", k_probe_pk_dep_fma, 1},
    {"fma+max", k_probe_p1, 2}, {"2fma+max", k_probe_p2, 3}, {"fma+mul(sgpr)", k_probe_p3, 2},
    {"cmp+cnd+2fma", k_probe_p4, 4}, {"3fma+cvt", k_probe_p5, 4}, {"fma+rcp+fma+max", k_probe_p6, 4}, {"max+rcp", k_probe_p7, 2},
};
static const int g_n_probes = sizeof(g_probes) / sizeof(g_probes[0]);

} // namespace lt
