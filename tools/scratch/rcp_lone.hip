// Lone wavefront: a reciprocal inside a dependent FMA stream -- v_rcp_f32 + one Newton step against the six-instruction
// software reciprocal (M<float>::rcp_pos), each followed by K dependent FMAs
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int K> __global__ void k(float *sink, unsigned long long *cyc, int iters)
{
    float x = threadIdx.x * 1e-3f + 1.5f, m = 0.999f, b = 1.0e-3f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float y, e, e2;
            if (MODE == 0) {
                asm volatile("v_rcp_f32 %0, %1" : "=v"(y) : "v"(x));
                asm volatile("v_fma_f32 %0, -%1, %2, 1.0" : "=v"(e) : "v"(x), "v"(y));
                asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(x) : "v"(y), "v"(e));
            } else {
                asm volatile("v_sub_u32 %0, 0x7EF311C0, %1" : "=v"(y) : "v"(x));
                asm volatile("v_fma_f32 %0, -%1, %2, 1.0" : "=v"(e) : "v"(x), "v"(y));
                asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(e2) : "v"(e));
                asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(y) : "v"(y), "v"(e2));
                asm volatile("v_fma_f32 %0, -%1, %2, 1.0" : "=v"(e) : "v"(x), "v"(y));
                asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(x) : "v"(y), "v"(e));
            }
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(b));
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (x == 12345.678f) sink[0] = x;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
template <int MODE, int K> double run()
{
    float *sink; unsigned long long *cyc, h = 0;
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&cyc, 8);
    const int iters = 50000;
    k<MODE, K><<<1, 64>>>(sink, cyc, iters); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    return (double)h / ((double)iters * 8);
}
int main()
{
    printf("dependent FMAs after the reciprocal:            0        4        16\n");
    printf("v_rcp_f32 + Newton step (3 instr):       %7.1f  %7.1f  %7.1f  cycles per group\n", run<0, 0>(), run<0, 4>(), run<0, 16>());
    printf("software reciprocal rcp_pos (6 instr):   %7.1f  %7.1f  %7.1f  cycles per group\n", run<1, 0>(), run<1, 4>(), run<1, 16>());
    return 0;
}
