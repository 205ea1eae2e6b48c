// Lone wavefront: cycles per instruction of dependent chains of v_fma_f32 / v_pk_fma_f32 / v_pk_mul_f32, with 1, 2, 4
// independent chains interleaved (is a lone wave bound by issue or by the latency of the previous result?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE, int CHAINS> __global__ void k(float *sink, unsigned long long *cyc, int iters)
{
    float x = threadIdx.x * 1e-3f + 1.0f;
    float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    f2 p0 = {x, x + 1}, p1 = {x + 2, x + 3}, p2 = {x + 4, x + 5}, p3 = {x + 6, x + 7};
    float m = 0.999f, b = 1e-3f;
    f2 mm = {0.999f, 0.998f}, bb = {1e-3f, 2e-3f};
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(b));
                  if (CHAINS > 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(b));
                  if (CHAINS > 2) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(b)); })
        } else if (MODE == 1) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(mm), "v"(bb));
                  if (CHAINS > 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(mm), "v"(bb));
                  if (CHAINS > 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(mm), "v"(bb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(mm), "v"(bb)); })
        } else if (MODE == 2) { // pk then a scalar op that depends on the pk result's low half, then pk depending on the scalar
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(mm), "v"(bb));
                  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(p0.x), "v"(m));
                  if (CHAINS > 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(mm), "v"(bb)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(p1.x), "v"(m)); })
        } else { // MODE 3: pk with op_sel swap on the dependent operand
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(p0) : "v"(mm), "v"(bb));
                  if (CHAINS > 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(p1) : "v"(mm), "v"(bb));)
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = r1 - r0; }
}
template <int MODE, int CHAINS> void run(const char *name, int per_rep)
{
    float *sink; unsigned long long *cyc, h[2] = {0, 0};
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&cyc, 16);
    const int iters = 100000;
    k<MODE, CHAINS><<<1, 64>>>(sink, cyc, iters); (void)hipDeviceSynchronize();
    k<MODE, CHAINS><<<1, 64>>>(sink, cyc, iters); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-46s chains %d: %.2f s_memtime ticks per instruction, %.2f ns per instruction, s_memtime at %.0f MHz\n", name, CHAINS, (double)h[0] / ((double)iters * 16 * per_rep), (double)h[1] * 10.0 / ((double)iters * 16 * per_rep), (double)h[0] / ((double)h[1] * 0.01));
}
int main()
{
    run<0, 1>("v_fma_f32 dependent", 1); run<0, 2>("v_fma_f32", 2); run<0, 4>("v_fma_f32", 4);
    run<1, 1>("v_pk_fma_f32 dependent", 1); run<1, 2>("v_pk_fma_f32", 2); run<1, 4>("v_pk_fma_f32", 4);
    run<3, 1>("v_pk_fma_f32 dependent, op_sel swap", 1); run<3, 2>("v_pk_fma_f32 op_sel swap", 2);
    run<2, 1>("v_pk_fma_f32 -> v_fma_f32 on its low half -> ...", 2); run<2, 2>("the same", 4);
    return 0;
}
