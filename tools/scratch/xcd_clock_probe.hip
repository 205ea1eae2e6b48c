// Does an XCD that hosts ONE busy wavefront hold a higher clock than its seven fully loaded neighbours?
// (The serial chain of DESIGN.md 5.1 runs at the ~1.9-2.1 GHz the chip holds under the bulk of the frame; if the clock is
// per XCD, a launch that keeps the chain's XCD free of bulk work would run the chain at 2.4 GHz from its first step.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_clock_probe tools/scratch/xcd_clock_probe.hip && /tmp/xcd_clock_probe
// One persistent grid, `per_cu` workgroups of 256 per CU (workgroup b runs on XCD b % 8; checked with XCC_ID).
// mode 0: every wavefront of every XCD runs the bulk loop (8 independent FMA chains per lane), and wave 0 of the first
//         workgroup of XCD `cool` additionally times a dependent FMA chain first;
// mode 1: on XCD `cool` only that one wavefront runs (the chain), every other wavefront there exits at once;
// mode 2: the chain alone on an otherwise idle chip.
// mode 3: as mode 0, but the bulk wavefronts of XCD `cool` stop after a quarter of the bulk loop: how fast does the clock of
//         an XCD answer when its load goes away?  (the chain is timed in 32 segments; printed as MHz per segment)
// Reported: the chain's clock (s_memtime cycles / s_memrealtime 100 MHz ticks) and time, and the bulk's clocks by XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Rec { unsigned long long cyc, ticks, xcc, kind; };
constexpr int SEGS = 32;
struct Seg { unsigned long long cyc[SEGS], ticks[SEGS], t_end[SEGS]; }; // kind 1 = chain, 2 = bulk (wave 0 of a workgroup), 0 = idle

__device__ __forceinline__ float bulk_loop(float x, int iters)
{
    float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    const float m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        }
    }
    return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__device__ __forceinline__ float chain_loop(float x, int iters)
{
    const float m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) x = fmaf(x, m, c);
    }
    return x;
}

__global__ void __launch_bounds__(256) k_probe(int mode, int cool, int chain_iters, int bulk_iters, Rec *recs, float *sink, Seg *seg, unsigned long long *t_start)
{
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; // HW_REG_XCC_ID, bits 3:0
    const int wave = threadIdx.x >> 6;
    const bool first_wg_of_xcd = (int)blockIdx.x < 8; // workgroups 0..7 are the first ones of XCDs 0..7 (if b % 8 holds)
    const bool is_chain = xcc == cool && first_wg_of_xcd && wave == 0;
    float x = threadIdx.x * 1e-3f + 1.0f, acc = 0.f;
    Rec r{0, 0, (unsigned long long)xcc, 0};
    if (is_chain) {
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) t_start[0] = t0;
        unsigned long long cs = c0, ts = t0;
        for (int sgm = 0; sgm < SEGS; ++sgm) {
            acc += chain_loop(x + acc * 1e-30f, chain_iters / SEGS);
            __builtin_amdgcn_s_waitcnt(0);
            unsigned long long c = __builtin_amdgcn_s_memtime(), t = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) { seg->cyc[sgm] = c - cs; seg->ticks[sgm] = t - ts; seg->t_end[sgm] = t - t0; }
            cs = c; ts = t;
        }
        unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
        r.cyc = c1 - c0; r.ticks = t1 - t0; r.kind = 1;
    } else if (mode == 2 || (mode == 1 && xcc == cool)) {
        // idle
    } else {
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
        acc += bulk_loop(x, (mode == 3 && xcc == cool) ? bulk_iters / 4 : bulk_iters);
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
        r.cyc = c1 - c0; r.ticks = t1 - t0; r.kind = 2;
    }
    if ((threadIdx.x & 63) == 0 && (wave == 0)) recs[blockIdx.x] = r;
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char **argv)
{
    int per_cu = argc > 1 ? atoi(argv[1]) : 2;          // workgroups of 256 per CU: 2 -> 2 waves per SIMD
    int chain_iters = argc > 2 ? atoi(argv[2]) : 12000; // x 64 dependent FMAs (~10 cycles each alone): ~3 ms
    int bulk_iters = argc > 3 ? atoi(argv[3]) : 60000;  // x 64 FMAs per lane
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * per_cu;
    Rec *d_recs; float *d_sink; Seg *d_seg; unsigned long long *d_t0;
    CHECK(hipMalloc(&d_recs, sizeof(Rec) * grid));
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMalloc(&d_seg, sizeof(Seg)));
    CHECK(hipMalloc(&d_t0, 8));
    Seg seg;
    std::vector<Rec> recs(grid);
    printf("# %s, %d CUs, grid %d x 256 (%d workgroups per CU), chain %d x 64 dependent FMAs, bulk %d x 64 FMAs per lane\n", prop.name,
           prop.multiProcessorCount, grid, per_cu, chain_iters, bulk_iters);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode)
            for (int cool : {0, 3}) {
                CHECK(hipMemset(d_recs, 0, sizeof(Rec) * grid));
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipEventRecord(e0, 0));
                k_probe<<<grid, 256>>>(mode, cool, chain_iters, bulk_iters, d_recs, d_sink, d_seg, d_t0);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(recs.data(), d_recs, sizeof(Rec) * grid, hipMemcpyDeviceToHost));
                double bc[8] = {0}, bt[8] = {0}; int bn[8] = {0}, mism = 0;
                double chain_mhz = 0, chain_ms = 0; int chain_xcc = -1;
                for (int b = 0; b < grid; ++b) {
                    const Rec &r = recs[b];
                    if ((int)r.xcc != b % 8) ++mism;
                    if (r.kind == 1) { chain_mhz = (double)r.cyc / r.ticks * 100.0; chain_ms = r.ticks / 1e5; chain_xcc = (int)r.xcc; }
                    if (r.kind == 2) { bc[r.xcc] += r.cyc; bt[r.xcc] += r.ticks; ++bn[r.xcc]; }
                }
                printf("mode %d cool XCD %d: kernel %.3f ms; chain on XCD %d: %.1f MHz, %.3f ms; bulk MHz by XCD:", mode, cool, ms, chain_xcc,
                       chain_mhz, chain_ms);
                for (int x = 0; x < 8; ++x) printf(" %s", bn[x] ? (std::to_string((int)(bc[x] / bt[x] * 100.0))).c_str() : "-");
                printf("  (workgroups whose XCC_ID != b %% 8: %d)\n", mism);
                if (mode == 3 || (mode == 0 && rep == 1)) {
                    CHECK(hipMemcpy(&seg, d_seg, sizeof(Seg), hipMemcpyDeviceToHost));
                    printf("    chain MHz by segment (segment end, ms):");
                    for (int g = 0; g < SEGS; ++g) printf(" %d(%.2f)", (int)((double)seg.cyc[g] / seg.ticks[g] * 100.0), seg.t_end[g] / 1e5);
                    printf("\n");
                }
            }
    return 0;
}
