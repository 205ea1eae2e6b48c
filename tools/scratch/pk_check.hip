// Do gfx950's packed float32 operations round like the scalar ones?  (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 against
// v_fma_f32 / v_mul_f32 / v_add_f32 on the same operands, random bit patterns of assorted magnitudes.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *a, const float *b, const float *c, uint32_t *out, int n)
{
#pragma clang fp contract(off)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    f2 A = {a[2 * i], a[2 * i + 1]}, B = {b[2 * i], b[2 * i + 1]}, C = {c[2 * i], c[2 * i + 1]};
    f2 F = __builtin_elementwise_fma(A, B, C), Mu = A * B, Ad = A + C;
    f2 Fz = __builtin_elementwise_fma(A, B, (f2){-0.0f, -0.0f});
    float f0, f1, m0, m1, d0, d1;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(A.x), "v"(B.x), "v"(C.x));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(A.y), "v"(B.y), "v"(C.y));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(A.x), "v"(B.x));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(A.y), "v"(B.y));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(d0) : "v"(A.x), "v"(C.x));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(d1) : "v"(A.y), "v"(C.y));
    auto bits = [](float x) { return __float_as_uint(x); };
    uint32_t bad = 0;
    bad |= (bits(F.x) != bits(f0) || bits(F.y) != bits(f1)) ? 1u : 0u;
    bad |= (bits(Mu.x) != bits(m0) || bits(Mu.y) != bits(m1)) ? 2u : 0u;
    bad |= (bits(Ad.x) != bits(d0) || bits(Ad.y) != bits(d1)) ? 4u : 0u;
    bad |= (bits(Fz.x) != bits(m0) || bits(Fz.y) != bits(m1)) ? 8u : 0u;
    out[i] = bad;
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> a(n), b(n), c(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) {
        auto mk = [&](int spread) { uint32_t m = (uint32_t)rnd() & 0x807fffffu; int e = 127 - spread + (int)(rnd() % (2 * spread + 1)); if (e < 0) e = 0; if (e > 254) e = 254;
                                     uint32_t u = m | ((uint32_t)e << 23); float f; memcpy(&f, &u, 4); return f; };
        int spread = (i % 4 == 0) ? 127 : (i % 4 == 1 ? 3 : 20);
        a[i] = mk(spread); b[i] = mk(spread); c[i] = mk(spread);
    }
    float *da, *db, *dc; uint32_t *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 2);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(da, db, dc, dout, n);
    std::vector<uint32_t> out(n / 2);
    hipMemcpy(out.data(), dout, n * 2, hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0};
    int shown = 0;
    for (int i = 0; i < n / 2; ++i) for (int j = 0; j < 4; ++j) if (out[i] & (1u << j)) { ++bad[j]; if (shown < 6) { printf("  pair %d flags %u: a %a %a b %a %a c %a %a\n", i, out[i], a[2*i], a[2*i+1], b[2*i], b[2*i+1], c[2*i], c[2*i+1]); ++shown; } }
    printf("pairs %d: pk_fma != fma: %ld   pk_mul != mul: %ld   pk_add != add: %ld   pk_fma(a,b,-0) != mul: %ld\n", n / 2, bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
