// Lone wavefront: time per instruction of ONE dependent chain, by instruction form and by the operand that carries the dependence
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, ASM) \
__global__ void NAME(float *sink, unsigned long long *cyc, int iters) { \
    float x = threadIdx.x * 1e-3f + 1.0f, a0 = x, m = 0.999f, b = 1e-3f; \
    unsigned long long c0 = __builtin_amdgcn_s_memtime(); \
    for (int i = 0; i < iters; ++i) { REP16(asm volatile(ASM : "+v"(a0) : "v"(m), "v"(b));) } \
    unsigned long long c1 = __builtin_amdgcn_s_memtime(); \
    if (a0 == 12345.678f) sink[0] = a0; if (threadIdx.x == 0) cyc[0] = c1 - c0; }
KERNEL(k_fma_src0, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fma_src2, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_mul_src0, "v_mul_f32 %0, %0, %1")
KERNEL(k_mul_src1, "v_mul_f32 %0, %1, %0")
KERNEL(k_add_src0, "v_add_f32 %0, %0, %2")
KERNEL(k_mov, "v_mov_b32 %0, %0")
KERNEL(k_fma_nop1, "v_fma_f32 %0, %0, %1, %2\n\ts_nop 0")
KERNEL(k_fma_indep1, "v_fma_f32 %0, %0, %1, %2\n\tv_mul_f32 v100, %1, %2")
KERNEL(k_fma_indep2, "v_fma_f32 %0, %0, %1, %2\n\tv_mul_f32 v100, %1, %2\n\tv_mul_f32 v101, %1, %2")
KERNEL(k_fma_indep3, "v_fma_f32 %0, %0, %1, %2\n\tv_mul_f32 v100, %1, %2\n\tv_mul_f32 v101, %1, %2\n\tv_mul_f32 v102, %1, %2")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
template <typename K> void run(K kern, const char *name, int per)
{
    float *sink; unsigned long long *cyc, h = 0;
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&cyc, 8);
    const int iters = 100000;
    kern<<<1, 64>>>(sink, cyc, iters); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %.2f cycles per group of %d = %.2f per instruction\n", name, (double)h / ((double)iters * 16), per, (double)h / ((double)iters * 16 * per));
}
int main()
{
    run(k_fma_src0, "v_fma_f32 (VOP3), dependence through src0", 1);
    run(k_fma_src2, "v_fma_f32 (VOP3), dependence through src2 (addend)", 1);
    run(k_fmac, "v_fmac_f32 (VOP2), dependence through the accumulator", 1);
    run(k_mul_src0, "v_mul_f32 (VOP2), dependence through src0", 1);
    run(k_mul_src1, "v_mul_f32 (VOP2), dependence through src1", 1);
    run(k_add_src0, "v_add_f32 (VOP2), dependence through src0", 1);
    run(k_mov, "v_mov_b32, dependent", 1);
    run(k_rcp, "v_rcp_f32, dependent", 1);
    run(k_fma_nop1, "v_fma_f32 dependent + s_nop 0", 2);
    run(k_fma_indep1, "v_fma_f32 dependent + 1 independent v_mul", 2);
    run(k_fma_indep2, "v_fma_f32 dependent + 2 independent v_mul", 3);
    run(k_fma_indep3, "v_fma_f32 dependent + 3 independent v_mul", 4);
    return 0;
}
