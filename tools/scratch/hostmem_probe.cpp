// Diagnostic: how fast does the CPU read host memory the GPU has just written by DMA, by allocation kind?
// hipcc -O2 -o hostmem_probe hostmem_probe.cpp && ./hostmem_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 64u << 20;
    void *d;
    hipMalloc(&d, n);
    hipMemset(d, 7, n);
    char *dst = (char *)malloc(n);
    memset(dst, 1, n);
    struct K { const char *name; unsigned flags; int reg; } kinds[] = {
        {"hipHostMalloc Default", hipHostMallocDefault, 0}, {"hipHostMalloc Portable", hipHostMallocPortable, 0},
        {"hipHostMalloc NonCoherent", hipHostMallocNonCoherent, 0}, {"hipHostMalloc Coherent", hipHostMallocCoherent, 0},
        {"hipHostMalloc Mapped|Portable", hipHostMallocMapped | hipHostMallocPortable, 0},
        {"malloc + hipHostRegister", 0, 1}};
    for (auto &k : kinds) {
        void *h = nullptr;
        if (k.reg) { h = aligned_alloc(4096, n); memset(h, 0, n); if (hipHostRegister(h, n, hipHostRegisterDefault) != hipSuccess) { printf("%s: register failed\n", k.name); continue; } }
        else if (hipHostMalloc(&h, n, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); (void)hipGetLastError(); continue; }
        double t0 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
        double t1 = now();
        memcpy(dst, h, n);
        double t2 = now();
        std::thread([&] { memcpy(dst, h, n); }).join();
        double t3 = now();
        std::vector<std::thread> th;
        for (int i = 0; i < 6; ++i) th.emplace_back([&, i] { memcpy(dst + i * (n / 6), (char *)h + i * (n / 6), n / 6); });
        for (auto &t : th) t.join();
        double t4 = now();
        memset(h, 3, n);
        double t5 = now();
        printf("%-32s D2H %6.2f ms (%5.1f GB/s) | memcpy to malloc: main %7.2f ms  1 thread %7.2f ms  6 threads %7.2f ms | memset %7.2f ms\n", k.name,
               (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
        if (k.reg) { hipHostUnregister(h); free(h); } else hipHostFree(h);
    }
    double t0 = now();
    hipMemcpy(dst, d, n, hipMemcpyDeviceToHost);
    printf("hipMemcpy D2H into pageable malloc memory: %.2f ms\n", (now() - t0) * 1e3);
    t0 = now();
    hipMemcpy(dst, d, n, hipMemcpyDeviceToHost);
    printf("   again: %.2f ms\n", (now() - t0) * 1e3);
    return 0;
}
