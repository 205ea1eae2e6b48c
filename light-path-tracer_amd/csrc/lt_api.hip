// lt_api.hip -- host side of libltrace_hip.so: the C-ABI declared in include/ltrace.h.
//
// Everything here is plumbing around the three kernels of lt_kernels.hpp: build the
// wave-uniform constant blocks, size the grids, launch on the caller's stream, time with
// HIP events.  No CPU compute path exists: without a GPU the entry points fail.
#include "../../include/ltrace.h"
#include "lt_kernels.hpp"
#ifdef LT_PROBES
#include "lt_probe.hpp"
#endif
#include "lt_dense.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <vector>

#ifndef LT_BUILD_ID
#define LT_BUILD_ID "unversioned"
#endif

using namespace lt;

// compiled in lt_k2_lone.hip (its own translation unit: another instruction scheduler, see there)
namespace lt {
extern template __global__ void k_kerr_direct<float, Rk4<float>>(KerrConsts<float>, const typename Vec4<float>::type *__restrict__,
                                                                 typename Vec4<float>::type *__restrict__, typename Vec4<float>::type *__restrict__,
                                                                 int64_t, uint32_t, uint4 *__restrict__, uint64_t *__restrict__, unsigned long long *__restrict__);
}

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(LT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" int lt_version(void) { return LT_VERSION; }
extern "C" const char *lt_build_id(void) { return LT_BUILD_ID; }
extern "C" const char *lt_last_error(void) { return g_err; }

extern "C" int lt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int lt_set_device(int device)
{
    if (lt_device_count() <= 0) return fail(LT_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    return LT_OK;
}

static int require_device()
{
    if (lt_device_count() <= 0)
        return fail(LT_ERR_NO_DEVICE, "libltrace_hip: no HIP device visible (this library has no CPU path)");
    return LT_OK;
}

// ---------------------------------------------------------------------------------------------
// per-device context.  Everything a call needs beyond the caller's own buffers is owned per
// (device, stream): the grow-only workspace of ray records, and for the host-pointer entry points
// the device-side inputs and outputs.  Two calls on different streams of one device
// therefore never share memory and may run concurrently; calls on the same stream are ordered by
// the stream.  Nothing is allocated per call once the buffers have grown to the frame size.
// ---------------------------------------------------------------------------------------------
struct EventQuad { hipEvent_t e[4]; };

struct Grow { // grow-only device allocation
    void *p = nullptr;
    size_t bytes = 0;
};

struct StreamSlot {
    hipStream_t stream = nullptr;
    Grow ws;     // ray records of lt_render_dev / the batch twins
    Grow dev;    // lt_render / batch twins: device-side inputs and outputs
    Grow dense;  // lt_integrate_dense_dev, length-binned launch: histogram, cursors, keys, permutation
    Grow blocks; // block-owner table mode: this partition's block list on the device
    std::vector<int32_t> blocks_host; // what `blocks` holds (skip the upload when unchanged)
    EventQuad own{}; // lt_render's private timing events (created on first use)
    bool own_ok = false;
};

struct Ctx {
    std::vector<StreamSlot *> slots;
    std::vector<EventQuad> pending; // recorded by lt_render_dev(timing=1), not yet collected
    std::vector<EventQuad> pool;
};

// lt_render's private timing events of a slot, all four or none (a failed creation leaves nothing behind)
static int slot_events(StreamSlot *sl)
{
    if (sl->own_ok) return LT_OK;
    int made = 0;
    for (auto &e : sl->own.e) {
        if (hipEventCreate(&e) != hipSuccess) break;
        ++made;
    }
    if (made != (int)(sizeof(sl->own.e) / sizeof(sl->own.e[0]))) {
        for (int j = 0; j < made; ++j) (void)hipEventDestroy(sl->own.e[j]);
        return fail(LT_ERR_HIP, "hipEventCreate failed");
    }
    sl->own_ok = true;
    return LT_OK;
}

static std::mutex g_mu;
static Ctx g_ctx[64];
struct MultiStream { int dev, idx; hipStream_t s; }; // lt_render_multi: one stream per (device, partition)
static std::vector<MultiStream> g_multi_streams;

static int cur_ctx(Ctx **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(LT_ERR_INVALID_ARG, "device index %d out of range", dev);
    *out = &g_ctx[dev];
    return LT_OK;
}

static int get_slot(hipStream_t s, StreamSlot **out)
{
    Ctx *c;
    int rc = cur_ctx(&c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    for (StreamSlot *sl : c->slots)
        if (sl->stream == s) { *out = sl; return LT_OK; }
    StreamSlot *sl = new StreamSlot;
    sl->stream = s;
    c->slots.push_back(sl);
    *out = sl;
    return LT_OK;
}

// Only `stream` ever touches the buffer, so draining that stream is enough before it is replaced.
static int grow(Grow &g, size_t need, hipStream_t stream)
{
    if (need <= g.bytes) return LT_OK;
    if (g.p) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipFree(g.p));
        g.p = nullptr;
        g.bytes = 0;
    }
    need = (need + 4095) & ~(size_t)4095;
    HIP_TRY(hipMalloc(&g.p, need));
    g.bytes = need;
    return LT_OK;
}

static void release(Grow &g)
{
    if (g.p) (void)hipFree(g.p);
    g.p = nullptr;
    g.bytes = 0;
}

// Workspace layout: 256 B of control words (queue head) | STAT_SLOTS x 8 partial counters of the epilogue | ic[n_q] |
// fin0[n_q] | fin1[n_q], each a 4-vector of T.  The partial counters are zero between frames (zeroed when the buffer
// is allocated, and again by k_stats_reduce after it has read them).
constexpr size_t WS_CTRL_BYTES = 256 + (size_t)STAT_SLOTS * 8 * sizeof(unsigned long long);
struct Workspace { unsigned long long *head; unsigned long long *partials; void *ic, *fin0, *fin1; };

static int get_workspace(hipStream_t stream, size_t n_q, size_t elem, Workspace *w)
{
    StreamSlot *sl;
    int rc = get_slot(stream, &sl);
    if (rc) return rc;
    size_t vec = 4 * elem;
    const void *before = sl->ws.p;
    if ((rc = grow(sl->ws, WS_CTRL_BYTES + 3 * n_q * vec, stream))) return rc;
    if (sl->ws.p != before) HIP_TRY(hipMemsetAsync(sl->ws.p, 0, WS_CTRL_BYTES, stream)); // a new buffer: control words and partial counters start at zero
    char *base = (char *)sl->ws.p;
    w->head = (unsigned long long *)base;
    w->partials = (unsigned long long *)(base + 256);
    w->ic = base + WS_CTRL_BYTES;
    w->fin0 = base + WS_CTRL_BYTES + n_q * vec;
    w->fin1 = base + WS_CTRL_BYTES + 2 * n_q * vec;
    return LT_OK;
}

// Carves 256-byte aligned pieces out of one grow-only buffer.
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int cu_count(int *out)
{
    static int cached[64];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (!cached[dev & 63]) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        cached[dev & 63] = prop.multiProcessorCount;
    }
    *out = cached[dev & 63];
    return LT_OK;
}

// Frees what the library holds for (current device, stream): call it before destroying a stream that was used
// with the library, or the buffers stay until lt_shutdown().
extern "C" int lt_release_stream(void *stream)
{
    int rc = require_device();
    if (rc) return rc;
    Ctx *c;
    if ((rc = cur_ctx(&c))) return rc;
    StreamSlot *found = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < c->slots.size(); ++i)
            if (c->slots[i]->stream == (hipStream_t)stream) {
                found = c->slots[i];
                c->slots.erase(c->slots.begin() + (long)i);
                break;
            }
    }
    if (!found) return LT_OK;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    release(found->ws);
    release(found->dev);
    release(found->dense);
    release(found->blocks);
    if (found->own_ok) for (auto &e : found->own.e) (void)hipEventDestroy(e);
    delete found;
    return LT_OK;
}

extern "C" int lt_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = lt_device_count();
    int keep = 0;
    if (n > 0) (void)hipGetDevice(&keep);
    for (int d = 0; d < n && d < 64; ++d) {
        Ctx &c = g_ctx[d];
        if (c.slots.empty() && c.pool.empty() && c.pending.empty()) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        for (StreamSlot *sl : c.slots) {
            release(sl->ws); release(sl->dev); release(sl->dense); release(sl->blocks);
            if (sl->own_ok) for (auto &e : sl->own.e) (void)hipEventDestroy(e);
            delete sl;
        }
        c.slots.clear();
        for (auto &q : c.pending) for (auto &e : q.e) (void)hipEventDestroy(e);
        for (auto &q : c.pool) for (auto &e : q.e) (void)hipEventDestroy(e);
        c.pending.clear();
        c.pool.clear();
    }
    for (auto &m : g_multi_streams) {
        (void)hipSetDevice(m.dev);
        (void)hipStreamDestroy(m.s);
    }
    g_multi_streams.clear();
    if (n > 0) (void)hipSetDevice(keep);
    return LT_OK;
}

extern "C" void lt_default_opts(lt_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->bg_sampling = LT_BG_GLOBAL; // measured (profiles/r02_background_sampling.txt): the gather itself is 0.06 ms of a 4096^2 frame
    o->integrator = LT_INTEGRATOR_RK4;
    o->precision = 32;
    o->schedule = LT_SCHED_DIRECT;
    o->row_block = 16;
    o->n_parts = 1;
    o->axis_refine_frac = 0.07;
    o->phi_max = 50.0;
    o->h_max = 0.0; // 0 -> metric default (0.05 Schwarzschild, 1.0 Kerr)
}

// ---------------------------------------------------------------------------------------------
// constant blocks
// ---------------------------------------------------------------------------------------------
// _psi_frame, image_lens.py:21-61.
static void psi_frame(double psi_y, double psi_x, double *d, double *ex, double *ey, bool *in_front)
{
    d[0] = sin(psi_x) * cos(psi_y);
    d[1] = -sin(psi_y);
    d[2] = cos(psi_x) * cos(psi_y);
    *in_front = d[2] > 1e-12;
    auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto norm = [&](const double *a) { return sqrt(dot(a, a)); };
    const double cx[3] = {1, 0, 0}, cy[3] = {0, 1, 0};
    double t = dot(cx, d);
    for (int i = 0; i < 3; ++i) ex[i] = cx[i] - t * d[i];
    double n = norm(ex);
    if (n < 1e-12) {
        t = dot(cy, d);
        for (int i = 0; i < 3; ++i) ex[i] = cy[i] - t * d[i];
        n = norm(ex);
    }
    n = n > 1e-12 ? n : 1e-12;
    for (int i = 0; i < 3; ++i) ex[i] /= n;
    double t1 = dot(cy, d), t2 = dot(cy, ex);
    for (int i = 0; i < 3; ++i) ey[i] = cy[i] - t1 * d[i] - t2 * ex[i];
    n = norm(ey);
    if (n < 1e-12) {
        ey[0] = d[1] * ex[2] - d[2] * ex[1];
        ey[1] = d[2] * ex[0] - d[0] * ex[2];
        ey[2] = d[0] * ex[1] - d[1] * ex[0];
        n = norm(ey);
    }
    n = n > 1e-12 ? n : 1e-12;
    for (int i = 0; i < 3; ++i) ey[i] /= n;
}

extern "C" int64_t lt_local_rows(int32_t height, int32_t row_block, int32_t n_parts, int32_t part)
{
    if (height <= 0 || row_block <= 0 || n_parts <= 0 || part < 0 || part >= n_parts) return -1;
    int64_t rows = 0;
    for (int64_t b = part; b * row_block < height; b += n_parts) {
        int64_t r0 = b * row_block, r1 = r0 + row_block;
        rows += (r1 < height ? r1 : height) - r0;
    }
    return rows;
}

extern "C" int64_t lt_global_row(int64_t local_row, int32_t row_block, int32_t n_parts, int32_t part)
{
    int64_t b = local_row / row_block, o = local_row - b * row_block;
    return (b * n_parts + part) * row_block + o;
}

// Row blocks partition `o.part` owns (ascending) and how many rows that is.  Block-cyclic unless a table is given.
static int partition_blocks(int32_t height, const lt_opts &o, std::vector<int32_t> *blocks, int64_t *rows)
{
    const int64_t nb = (height + (int64_t)o.row_block - 1) / o.row_block;
    if (o.block_owner && o.n_blocks != nb)
        return fail(LT_ERR_INVALID_ARG, "block_owner has %d entries, the frame has %lld row blocks of %d rows", o.n_blocks,
                    (long long)nb, o.row_block);
    *rows = 0;
    if (blocks) blocks->clear();
    for (int64_t b = 0; b < nb; ++b) {
        int owner = o.block_owner ? (int)o.block_owner[b] : (int)(b % o.n_parts);
        if (o.block_owner && owner >= o.n_parts) return fail(LT_ERR_INVALID_ARG, "block_owner[%lld] = %d, n_parts = %d", (long long)b, owner, o.n_parts);
        if (owner != o.part) continue;
        int64_t r0 = b * o.row_block, r1 = r0 + o.row_block;
        *rows += (r1 < height ? r1 : height) - r0;
        if (blocks) blocks->push_back((int32_t)b);
    }
    return LT_OK;
}

static int make_metric(const lt_metric *m, double r_obs, double theta_obs, double h_schw, MetricConsts *mc)
{
    if (!(m->M > 0.0)) return fail(LT_ERR_INVALID_ARG, "M must be positive");
    mc->kind = m->kind;
    mc->M = m->M;
    mc->a = m->kind == LT_METRIC_KERR ? m->a : 0.0;
    if (m->kind == LT_METRIC_KERR && fabs(m->a) > m->M)
        return fail(LT_ERR_INVALID_ARG, "|a|=%g exceeds M=%g", fabs(m->a), m->M); // metrics.py:849-850
    if (m->kind != LT_METRIC_KERR && m->kind != LT_METRIC_SCHWARZSCHILD)
        return fail(LT_ERR_INVALID_ARG, "unknown metric kind %d", m->kind);
    mc->r_obs = r_obs;
    mc->theta_obs = theta_obs;
    mc->r_plus = mc->M + sqrt(mc->M * mc->M - mc->a * mc->a); // metrics.py:853
    mc->r_capture = mc->r_plus * 1.01;                        // metrics.py:428, :579
    mc->R_S = 2.0 * mc->M;                                    // metrics.py:742
    mc->phi_h = h_schw;
    mc->evals_fixed = 0;
    mc->evals_per_step = 4;
    { // the observer-only part of metrics.py:148-218, in its operation order (K1 used to redo this per ray)
        double r = r_obs, a = mc->a, M_ = mc->M;
        double sin_th = sin(theta_obs), cos_th = cos(theta_obs);
        double sin_th_sq = sin_th * sin_th;
        if (sin_th_sq < 1e-15) sin_th_sq = 1e-15;
        double Sigma = r * r + a * a * cos_th * cos_th;
        double Delta = r * r - 2.0 * M_ * r + a * a;
        mc->obs_ok = Delta > 0.0 && Sigma > 0.0;
        mc->obs_sin_th = sin_th; mc->obs_cos2 = cos_th * cos_th; mc->obs_sin2 = sin_th_sq;
        mc->obs_sqrt_Sigma = mc->obs_ok ? sqrt(Sigma) : 0.0;
        mc->obs_sqrt_Delta = mc->obs_ok ? sqrt(Delta) : 1.0;
        double A_val = (r * r + a * a) * (r * r + a * a) - a * a * Delta * sin_th_sq;
        double SD = mc->obs_ok ? Sigma * Delta : 1.0;
        mc->obs_g_tt = -A_val / SD;
        mc->obs_g_tphi = -2.0 * M_ * a * r / SD;
        mc->obs_g_rr = mc->obs_ok ? Delta / Sigma : 1.0;
        mc->obs_g_thth = mc->obs_ok ? 1.0 / Sigma : 0.0;
        mc->obs_g_phiphi = (Delta - a * a * sin_th_sq) / (SD * sin_th_sq);
    }
    return LT_OK;
}

template <typename T> static KerrConsts<T> make_kerr(const MetricConsts &mc, double lambda_max, double h_max)
{
    KerrConsts<T> k;
    k.M = (T)mc.M; k.a = (T)mc.a; k.a2 = (T)(mc.a * mc.a); k.two_M = (T)(2.0 * mc.M);
    k.r_cut = (T)(mc.r_plus * 1.001);
    k.r_capture = (T)mc.r_capture;
    k.r_escape = (T)(mc.r_obs * 2.0);
    k.r_obs = (T)mc.r_obs; k.theta_obs = (T)mc.theta_obs;
    k.lambda_max = (T)lambda_max;
    k.h_max = (T)h_max;
    k.rc4 = (T)(mc.r_capture * 4.0); k.rc2 = (T)(mc.r_capture * 2.0); k.rc12 = (T)(mc.r_capture * 1.2);
    return k;
}

template <typename T> static SchwConsts<T> make_schw(const MetricConsts &mc, double phi_max, double h_max)
{
    SchwConsts<T> k;
    k.M = (T)mc.M; k.three_M = (T)(3.0 * mc.M);
    k.u0 = (T)(1.0 / mc.r_obs);
    k.u_capture = (T)(1.0 / (mc.R_S * 1.01)); // metrics.py:66
    k.u_escape = (T)(1.0 / (2.0 * mc.r_obs)); // metrics.py:67
    k.h_max = (T)h_max; k.phi_max = (T)phi_max;
    double nf = floor(phi_max / h_max + 1e-9);
    double rest = phi_max - nf * h_max;
    if (rest < 1e-9 * h_max) rest = 0.0;
    k.n_full = (uint32_t)nf;
    k.h_last = (T)rest;
    return k;
}

// ---------------------------------------------------------------------------------------------
// stage launchers
// ---------------------------------------------------------------------------------------------
// Brackets the three kernels of one frame with HIP events.  Two modes: a quad from the device's pool that
// ends up on the `pending` list for lt_timing_collect (lt_render_dev with opts->timing), or a caller-owned
// quad (lt_render's private one, read directly).  A quad taken from the pool goes back to it on every
// path that does not finish().
struct Timer {
    bool on = false, pooled = false, done = false;
    EventQuad q{};
    Ctx *ctx = nullptr;
    ~Timer()
    {
        if (on && pooled && !done) {
            std::lock_guard<std::mutex> lk(g_mu);
            ctx->pool.push_back(q);
        }
    }
    int begin(bool enable, const EventQuad *own)
    {
        if (own) { on = true; q = *own; return LT_OK; }
        if (!enable) return LT_OK;
        int rc = cur_ctx(&ctx);
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(g_mu);
        if (!ctx->pool.empty()) { q = ctx->pool.back(); ctx->pool.pop_back(); }
        else {
            for (int i = 0; i < 4; ++i) {
                hipError_t e = hipEventCreate(&q.e[i]);
                if (e != hipSuccess) {
                    for (int j = 0; j < i; ++j) (void)hipEventDestroy(q.e[j]);
                    return fail(LT_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
                }
            }
        }
        on = pooled = true;
        return LT_OK;
    }
    int mark(int i, hipStream_t s)
    {
        if (on) HIP_TRY(hipEventRecord(q.e[i], s));
        return LT_OK;
    }
    void finish()
    {
        if (!on || !pooled) return;
        std::lock_guard<std::mutex> lk(g_mu);
        ctx->pending.push_back(q);
        done = true;
    }
};

extern "C" int lt_timing_collect(double *prologue_ms, double *integrate_ms, double *epilogue_ms, int32_t *calls)
{
    Ctx *c;
    int rc = cur_ctx(&c);
    if (rc) return rc;
    double t[3] = {0, 0, 0};
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (auto &q : c->pending) {
        HIP_TRY(hipEventSynchronize(q.e[3]));
        for (int i = 0; i < 3; ++i) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, q.e[i], q.e[i + 1]));
            t[i] += ms;
        }
        c->pool.push_back(q);
        ++n;
    }
    c->pending.clear();
    if (prologue_ms) *prologue_ms = t[0];
    if (integrate_ms) *integrate_ms = t[1];
    if (epilogue_ms) *epilogue_ms = t[2];
    if (calls) *calls = n;
    return LT_OK;
}

// Diagnostic wave stamps (LT_STAMPS_FILE=path): one uint4 per wavefront, dumped raw after the launch.
struct StampDump {
    const char *path = getenv("LT_STAMPS_FILE");
    uint4 *dev = nullptr;
    size_t n = 0;
    int begin(size_t waves)
    {
        if (!path) return LT_OK;
        n = waves;
        HIP_TRY(hipMalloc((void **)&dev, n * sizeof(uint4)));
        HIP_TRY(hipMemset(dev, 0, n * sizeof(uint4)));
        return LT_OK;
    }
    int end()
    {
        if (!path) return LT_OK;
        std::vector<uint4> h(n);
        HIP_TRY(hipMemcpy(h.data(), dev, n * sizeof(uint4), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(dev));
        if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), sizeof(uint4), n, f); fclose(f); }
        return LT_OK;
    }
};

// Wavefront slots of the device for a kernel launched with 64-thread workgroups (0 if the query fails: the caller then
// launches one workgroup per tile).  Asked once per kernel.
template <auto Kernel> static int resident_slots()
{
    static const int slots = [] {
        int per_cu = 0, cus = 0;
        if (cu_count(&cus) != LT_OK) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)Kernel, 64, 0) != hipSuccess || per_cu <= 0) return 0;
        return per_cu * cus;
    }();
    return slots;
}

template <typename T>
static int launch_integrate(const MetricConsts &mc, const lt_opts &o, double lambda_max, const Workspace &w,
                            int64_t n_q, hipStream_t s, uint64_t *kstats)
{
    using V = typename Vec4<T>::type;
    int rc;
    unsigned grid = (unsigned)((n_q + 255) / 256);
    if (mc.kind == LT_METRIC_SCHWARZSCHILD) {
        SchwConsts<T> k = make_schw<T>(mc, o.phi_max, o.h_max);
        k_schw_rk4_direct<T><<<grid, 256, 0, s>>>(k, (const V *)w.ic, (V *)w.fin0, (V *)w.fin1, n_q);
    } else {
        KerrConsts<T> k = make_kerr<T>(mc, lambda_max, o.h_max);
        const bool exact = o.integrator == LT_INTEGRATOR_DP45_EXACT;
        const bool dp45 = o.integrator == LT_INTEGRATOR_DP45 || exact;
        if (dp45 && sizeof(T) != 8) return fail(LT_ERR_UNSUPPORTED, "DP45 needs precision 64");
        StampDump sd;
        if (o.schedule == LT_SCHED_DIRECT) {
            static const int k2_block = [] { int b = env_int("LT_K2_BLOCK", 64);
                                             return (b == 64 || b == 128 || b == 256) ? b : 64; }();
            unsigned kgrid = (unsigned)((n_q + k2_block - 1) / k2_block);
            // steps after which a wavefront is "long" (ghost lanes, lone-wave step form, issue priority).  384: with the packed
            // lone-wave step of round 2 an earlier switch pays on chain-bound launches -- 2048^2 4.19 -> 4.10 ms, one rank of 8
            // 4.00 -> 3.95, of 4 4.18 -> 4.12 -- and leaves the 4096^2 frame where it was (10.75 / 10.73 ms); 128 costs that
            // frame 1.5 % (profiles/r03_xcd_clock.txt, tools/scratch/d_long_sweep.sh)
            static const int long_iters = env_int("LT_D_LONG", 384);
            // Tiles are handed out from a queue head to a grid that fills the chip once (k_kerr_direct); LT_D_PERSIST=0
            // or wider workgroups: one workgroup per tile, as in round 1.
            static const int persist = env_int("LT_D_PERSIST", 1);
            unsigned long long *head = nullptr;
            auto resident_grid = [&](int slots) { // (a launch that fits the chip needs no queue)
                if (slots > 0 && (unsigned)slots < kgrid) { kgrid = (unsigned)slots; head = w.head; }
            };
            const bool want_queue_head = persist && k2_block == 64;
            if ((rc = sd.begin((size_t)(n_q / 64)))) return rc;
            if constexpr (sizeof(T) == 8) {
                if (dp45 && !exact) {
                    if (want_queue_head) resident_grid(resident_slots<k_kerr_direct<T, Dp45<T>>>());
                    if (head) HIP_TRY(hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
                    k_kerr_direct<T, Dp45<T>><<<kgrid, k2_block, 0, s>>>(k, (const V *)w.ic, (V *)w.fin0, (V *)w.fin1, n_q, (uint32_t)(long_iters / 3), sd.dev, kstats, head);
                }
                if (exact) {
                    if (want_queue_head) resident_grid(resident_slots<k_kerr_direct<T, Dp45<T, true>>>());
                    if (head) HIP_TRY(hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
                    k_kerr_direct<T, Dp45<T, true>><<<kgrid, k2_block, 0, s>>>(k, (const V *)w.ic, (V *)w.fin0, (V *)w.fin1, n_q, (uint32_t)(long_iters / 3), sd.dev, kstats, head);
                }
            }
            if (!dp45) {
                if (want_queue_head) resident_grid(resident_slots<k_kerr_direct<T, Rk4<T>>>());
                if (head) HIP_TRY(hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
                k_kerr_direct<T, Rk4<T>><<<kgrid, k2_block, 0, s>>>(k, (const V *)w.ic, (V *)w.fin0, (V *)w.fin1, n_q, (uint32_t)long_iters, sd.dev, kstats, head);
            }
        } else {
            int cus;
            if ((rc = cu_count(&cus))) return rc;
            // persistent grid: blocks-per-CU x CUs, never more blocks than there are 256-ray pieces
            static const int bpc = env_int("LT_Q_BPC", sizeof(T) == 4 ? 8 : 3);
            static const int chunk = env_int("LT_Q_CHUNK", 64);
            static const int refill_min = env_int("LT_Q_REFILL", 4);
            static const int long_steps = env_int("LT_Q_LONG", 600);
            unsigned qgrid = (unsigned)(cus * bpc);
            if (qgrid > grid) qgrid = grid;
            HIP_TRY(hipMemsetAsync(w.head, 0, sizeof(unsigned long long), s));
            if ((rc = sd.begin((size_t)qgrid * 4))) return rc;
            if constexpr (sizeof(T) == 8) {
                if (dp45 && !exact) // "long" is measured in step attempts: DP45 rays take ~50, not ~150
                    k_kerr_queue<T, Dp45<T>><<<qgrid, 256, 0, s>>>(k, (const V *)w.ic, (V *)w.fin0, (V *)w.fin1, (uint64_t)n_q,
                                                                 w.head, (uint32_t)chunk, (uint32_t)refill_min,
                                                                 (uint32_t)(long_steps / 3), sd.dev, kstats);
                if (exact)
                    k_kerr_queue<T, Dp45<T, true>><<<qgrid, 256, 0, s>>>(k, (const V *)w.ic, (V *)w.fin0, (V *)w.fin1, (uint64_t)n_q,
                                                                       w.head, (uint32_t)chunk, (uint32_t)refill_min,
                                                                       (uint32_t)(long_steps / 3), sd.dev, kstats);
            }
            if (!dp45)
                k_kerr_queue<T, Rk4<T>><<<qgrid, 256, 0, s>>>(k, (const V *)w.ic, (V *)w.fin0, (V *)w.fin1, (uint64_t)n_q, w.head,
                                                            (uint32_t)chunk, (uint32_t)refill_min, (uint32_t)long_steps, sd.dev,
                                                            kstats);
        }
        HIP_TRY(hipGetLastError());
        if ((rc = sd.end())) return rc;
    }
    HIP_TRY(hipGetLastError());
    return LT_OK;
}

static int check_opts(const lt_metric *metric, lt_opts *o)
{
    if (o->precision != 32 && o->precision != 64) return fail(LT_ERR_INVALID_ARG, "precision must be 32 or 64");
    if (metric->kind == LT_METRIC_KERR && (o->integrator == LT_INTEGRATOR_DP45 || o->integrator == LT_INTEGRATOR_DP45_EXACT) &&
        o->precision != 64)
        return fail(LT_ERR_UNSUPPORTED,
                    "DP45 at the reference tolerances needs float64 (rtol 1e-8 is below float32 epsilon)");
    if (o->integrator != LT_INTEGRATOR_DP45 && o->integrator != LT_INTEGRATOR_RK4 && o->integrator != LT_INTEGRATOR_DP45_EXACT)
        return fail(LT_ERR_INVALID_ARG, "unknown integrator %d", o->integrator);
    if (o->schedule != LT_SCHED_DIRECT && o->schedule != LT_SCHED_QUEUE)
        return fail(LT_ERR_INVALID_ARG, "unknown schedule %d", o->schedule);
    if (o->bg_sampling != LT_BG_LDS_TILES && o->bg_sampling != LT_BG_GLOBAL)
        return fail(LT_ERR_INVALID_ARG, "unknown bg_sampling %d", o->bg_sampling);
    if (o->h_max <= 0.0) o->h_max = metric->kind == LT_METRIC_KERR ? 1.0 : 0.05;
    if (o->phi_max <= 0.0) o->phi_max = 50.0;
    if (o->row_block <= 0) o->row_block = 16;
    if (o->n_parts <= 0) o->n_parts = 1;
    if (o->part < 0 || o->part >= o->n_parts) return fail(LT_ERR_INVALID_ARG, "part %d not in [0, %d)", o->part, o->n_parts);
    if (o->block_owner && o->n_blocks <= 0) return fail(LT_ERR_INVALID_ARG, "block_owner given with n_blocks = %d", o->n_blocks);
    return LT_OK;
}

static int render_dev_impl(const lt_camera *cam, const lt_metric *metric, const lt_opts *opts, const float *d_bg,
                           int32_t bg_channels, float *d_fa, uint16_t *d_w, int8_t *d_status, uint32_t *d_steps,
                           float *d_rgb, uint8_t *d_rgba, uint64_t *d_stats, const EventQuad *own_events)
{
    int rc = require_device();
    if (rc) return rc;
    if (!cam || !metric || !opts) return fail(LT_ERR_INVALID_ARG, "null camera / metric / opts");
    if (cam->width <= 0 || cam->height <= 0) return fail(LT_ERR_INVALID_ARG, "empty frame %dx%d", cam->width, cam->height);
    if (d_bg && bg_channels != 1 && bg_channels != 3) return fail(LT_ERR_INVALID_ARG, "bg_channels must be 1 or 3");
    lt_opts o = *opts;
    if ((rc = check_opts(metric, &o))) return rc;
    MetricConsts mc;
    if ((rc = make_metric(metric, cam->r_obs, cam->theta_obs, o.h_max, &mc))) return rc;
    if (metric->kind == LT_METRIC_KERR && o.integrator != LT_INTEGRATOR_RK4) { mc.evals_fixed = 1; mc.evals_per_step = 6; }

    CamConsts c;
    memset(&c, 0, sizeof(c));
    c.W = cam->width; c.H = cam->height;
    c.row_block = o.row_block; c.n_parts = o.n_parts; c.part = o.part;
    hipStream_t s = (hipStream_t)o.stream;
    std::vector<int32_t> owned;
    int64_t rows_owned = 0;
    if ((rc = partition_blocks(c.H, o, &owned, &rows_owned))) return rc;
    c.rows_local = (int)rows_owned;
    c.block_list = nullptr;
    if (o.block_owner && !owned.empty()) { // the partition's block list goes to the device (once: re-uploaded only when it changes)
        StreamSlot *sl;
        if ((rc = get_slot(s, &sl))) return rc;
        if (sl->blocks_host != owned || !sl->blocks.p) {
            if ((rc = grow(sl->blocks, owned.size() * sizeof(int32_t), s))) return rc;
            HIP_TRY(hipStreamSynchronize(s)); // frames in flight on this stream still read the old list
            HIP_TRY(hipMemcpy(sl->blocks.p, owned.data(), owned.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            sl->blocks_host = owned;
        }
        c.block_list = (const int32_t *)sl->blocks.p;
    }
    c.loop_around = o.loop_around;
    c.half_W = c.W / 2.0; c.half_H = c.H / 2.0;
    c.fx = (c.W / 2.0) / tan(cam->hfov / 2); // image_lens.py:138-139
    c.fy = (c.H / 2.0) / tan(cam->vfov / 2);
    bool front;
    psi_frame(cam->psi_y, cam->psi_x, c.d, c.ex, c.ey, &front);
    c.refine_on = front && metric->kind == LT_METRIC_KERR;
    if (front) { // image_lens.py:210-214
        c.bh_x_cam = c.d[0] / c.d[2];
        double x_lo = fabs((0 - c.half_W) / c.fx - c.bh_x_cam), x_hi = fabs((c.W - 1 - c.half_W) / c.fx - c.bh_x_cam);
        double m = x_lo > x_hi ? x_lo : x_hi;
        c.refine_thresh = o.axis_refine_frac * (m > 1e-12 ? m : 1e-12);
    }
    // top/bottom symmetry exactly when the reference applies it (image_lens.py:218-220)
    c.use_tb = o.tb_symmetry && metric->kind == LT_METRIC_KERR &&
               fabs(cam->theta_obs - M_PI / 2) <= 1e-8 + 1e-5 * (M_PI / 2) && fabs(cam->psi_y) <= 1e-8;
    if (c.use_tb && (o.n_parts != 1 || o.block_owner)) return fail(LT_ERR_UNSUPPORTED, "tb_symmetry needs n_parts == 1 and no block_owner table");
    c.trace_rows = c.use_tb ? (c.H + 1) / 2 : c.rows_local;
    c.tiles_x = (c.W + 7) / 8;
    if (c.rows_local <= 0) return LT_OK; // a partition may own no rows
    if (c.rows_local > 65535) // (the epilogue's launch grid carries the row in its y dimension; checked before anything is launched)
        return fail(LT_ERR_UNSUPPORTED, "a partition of %d rows exceeds the epilogue's launch grid (65535 rows): split it (n_parts)", c.rows_local);
    int tiles_y = (c.trace_rows + 7) / 8;
    c.tiles_y = tiles_y;
    { // "hot" tile rectangle, queued first: bounds the critical curve (largest impact parameter of a
      // spherical photon orbit: the retrograde equatorial one for Kerr, 3 sqrt(3) M for a = 0), + margin
        c.hot_x0 = c.hot_x1 = c.hot_y0 = c.hot_y1 = 0;
        c.strip_x0 = c.strip_x1 = 0;
        if (front && metric->kind == LT_METRIC_KERR) {
            double a = mc.a, M_ = mc.M;
            double b_max = 3.0 * sqrt(3.0) * M_;
            if (a != 0.0) { // Bardeen: r_ret = 2M (1 + cos(2/3 acos(|a|/M))), xi(r) as in metrics.py:886-887
                double aa = fabs(a);
                double r_ph = 2.0 * M_ * (1.0 + cos(2.0 / 3.0 * acos(aa / M_)));
                double Dl = r_ph * r_ph - 2.0 * M_ * r_ph + aa * aa;
                double xi = (r_ph * r_ph + aa * aa) / aa - 2.0 * r_ph * Dl / (aa * (r_ph - M_));
                if (fabs(xi) > b_max) b_max = fabs(xi);
            }
            b_max = 1.05 * b_max + 0.3 * M_;
            double f0 = 1.0 - 2.0 * M_ / cam->r_obs;
            double sin_al = f0 > 0 ? b_max * sqrt(f0) / cam->r_obs : 1.0;
            if (sin_al < 0.98) {
                double tan_al = sin_al / sqrt(1.0 - sin_al * sin_al);
                double bx = c.d[0] / c.d[2] * c.fx + c.half_W, by = c.d[1] / c.d[2] * c.fy + c.half_H; // BH pixel
                double rx = tan_al * c.fx + 8.0, ry = tan_al * c.fy + 8.0;                             // pixels
                auto clampi = [](double v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); };
                c.hot_x0 = clampi(floor((bx - rx) / 8.0), 0, c.tiles_x);
                c.hot_x1 = clampi(ceil((bx + rx) / 8.0), 0, c.tiles_x);
                // rows: global pixel rows -> this partition's local rows (block-cyclic: about 1/n_parts of them)
                double ly0 = (by - ry) / o.n_parts - o.row_block, ly1 = (by + ry) / o.n_parts + o.row_block;
                if (o.block_owner) { // any assignment: local rows of the first / last owned block that touches the band
                    ly0 = 1e18; ly1 = -1.0;
                    for (size_t i = 0; i < owned.size(); ++i) {
                        double g0 = (double)owned[i] * o.row_block, g1 = g0 + o.row_block;
                        if (g1 < by - ry || g0 > by + ry) continue;
                        if ((double)i * o.row_block < ly0) ly0 = (double)i * o.row_block;
                        ly1 = (double)(i + 1) * o.row_block;
                    }
                    if (ly1 < 0) ly0 = ly1 = 0.0;
                }
                c.hot_y0 = clampi(floor(ly0 / 8.0), 0, tiles_y);
                c.hot_y1 = clampi(ceil(ly1 / 8.0), 0, tiles_y);
                if (c.hot_x1 <= c.hot_x0 || c.hot_y1 <= c.hot_y0) c.hot_x0 = c.hot_x1 = c.hot_y0 = c.hot_y1 = 0;
                // spin-axis strip: +-16 px around the column the BH projects to; then move the rectangle's
                // column range onto the grid with the strip removed
                c.strip_x0 = clampi(floor((bx - 16.0) / 8.0), 0, c.tiles_x);
                c.strip_x1 = clampi(ceil((bx + 16.0) / 8.0), 0, c.tiles_x);
                if (c.strip_x1 < c.strip_x0) c.strip_x1 = c.strip_x0;
                int sw = c.strip_x1 - c.strip_x0;
                auto compact = [&](int x) { return x <= c.strip_x0 ? x : (x - sw > c.strip_x0 ? x - sw : c.strip_x0); };
                c.hot_x0 = compact(c.hot_x0);
                c.hot_x1 = compact(c.hot_x1);
                if (c.hot_x1 <= c.hot_x0) c.hot_x0 = c.hot_x1 = c.hot_y0 = c.hot_y1 = 0;
            }
        }
    }
    int64_t n_q = (int64_t)c.tiles_x * tiles_y * 64;
    if (n_q / 64 >= (int64_t)1 << 31) // (the prologue decodes a tile's queue position in 32 bits; 2^31 tiles of records would be 6 TB)
        return fail(LT_ERR_INVALID_ARG, "frame of %d x %d tiles: more than 2^31 - 1", c.tiles_x, tiles_y);
    double lambda_max = fmax(5000.0, 6.0 * cam->r_obs); // metrics.py:1132

    size_t elem = o.precision == 32 ? sizeof(float) : sizeof(double);
    Workspace w;
    if ((rc = get_workspace(s, (size_t)n_q, elem, &w))) return rc;
    void *ic = w.ic, *fin0 = w.fin0, *fin1 = w.fin1;
    Timer tm;
    if ((rc = tm.begin(o.timing != 0, own_events))) return rc;
    unsigned gq = (unsigned)((n_q + 255) / 256);
    unsigned gp = 0;
    // the epilogue adds its counters into STAT_SLOTS partial sets (workgroup index mod STAT_SLOTS) of the workspace; one
    // small launch behind it folds them into the caller's counters
    FrameOut fo{d_bg, bg_channels, d_fa, d_w, d_status, d_steps, d_rgb, d_rgba, d_stats ? (uint64_t *)w.partials : nullptr};

    if ((rc = tm.mark(0, s))) return rc;
    if (o.precision == 32) k_prologue_camera<float><<<gq, 256, 0, s>>>(c, mc, (float4 *)ic, n_q);
    else k_prologue_camera<double><<<gq, 256, 0, s>>>(c, mc, (double4 *)ic, n_q);
    HIP_TRY(hipGetLastError());
    if ((rc = tm.mark(1, s))) return rc;
    rc = o.precision == 32 ? launch_integrate<float>(mc, o, lambda_max, w, n_q, s, d_stats)
                           : launch_integrate<double>(mc, o, lambda_max, w, n_q, s, d_stats);
    if (rc) return rc;
    if ((rc = tm.mark(2, s))) return rc;
    // background tiles staged in LDS when a background is lensed (opts->bg_sampling)
    const bool lds_path = o.bg_sampling == LT_BG_LDS_TILES && d_bg && (d_rgb || d_rgba);
    if (lds_path) {
        int64_t n16 = (int64_t)((c.W + 15) / 16) * ((c.rows_local + 15) / 16);
        gp = (unsigned)n16; // one 16x16 tile per workgroup
        if (o.precision == 32) k_epilogue_frame_lds<float><<<gp, 256, 0, s>>>(c, mc, (const float4 *)fin0, (const float4 *)fin1, fo);
        else k_epilogue_frame_lds<double><<<gp, 256, 0, s>>>(c, mc, (const double4 *)fin0, (const double4 *)fin1, fo);
    } else {
        const bool has_bg = d_bg != nullptr && (d_rgb || d_rgba);
        const dim3 ge((unsigned)((c.W + EPILOGUE_BLOCK - 1) / EPILOGUE_BLOCK), (unsigned)c.rows_local);
        if (o.precision == 32) {
            if (has_bg) k_epilogue_frame<float, true><<<ge, EPILOGUE_BLOCK, 0, s>>>(c, mc, (const float4 *)fin0, (const float4 *)fin1, fo);
            else k_epilogue_frame<float, false><<<ge, EPILOGUE_BLOCK, 0, s>>>(c, mc, (const float4 *)fin0, (const float4 *)fin1, fo);
        } else {
            if (has_bg) k_epilogue_frame<double, true><<<ge, EPILOGUE_BLOCK, 0, s>>>(c, mc, (const double4 *)fin0, (const double4 *)fin1, fo);
            else k_epilogue_frame<double, false><<<ge, EPILOGUE_BLOCK, 0, s>>>(c, mc, (const double4 *)fin0, (const double4 *)fin1, fo);
        }
    }
    if (d_stats) k_stats_reduce<<<1, STAT_SLOTS, 0, s>>>(w.partials, (unsigned long long *)d_stats);
    HIP_TRY(hipGetLastError());
    if ((rc = tm.mark(3, s))) return rc;
    tm.finish();
    return LT_OK;
}

extern "C" int lt_render_dev(const lt_camera *cam, const lt_metric *metric, const lt_opts *opts, const float *d_bg,
                             int32_t bg_channels, float *d_fa, uint16_t *d_w, int8_t *d_status, uint32_t *d_steps,
                             float *d_rgb, uint8_t *d_rgba, uint64_t *d_stats)
{
    return render_dev_impl(cam, metric, opts, d_bg, bg_channels, d_fa, d_w, d_status, d_steps, d_rgb, d_rgba, d_stats,
                           nullptr);
}

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { HIP_TRY(hipMalloc(&p, n ? n : 1)); return LT_OK; }
};

// ---- pinned host memory for callers of the host-pointer entry points ---------------------------
// A destination inside such a block is written by DMA straight from the device (PCIe rate); any other
// destination is pageable memory, which the HIP runtime has to pin page by page during the copy.
extern "C" void *lt_host_alloc(size_t bytes)
{
    if (lt_device_count() <= 0) { (void)fail(LT_ERR_NO_DEVICE, "lt_host_alloc: no HIP device visible"); return nullptr; }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { (void)fail(LT_ERR_HIP, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

extern "C" int lt_host_free(void *p)
{
    if (!p) return LT_OK;
    HIP_TRY(hipHostFree(p));
    return LT_OK;
}

// Device -> host for a list of outputs: one asynchronous copy each, straight into the caller's memory.  Into
// pinned memory (lt_host_alloc) that is a DMA at PCIe rate (measured 55 GB/s: 1.2 ms for a 4096^2 RGBA8 frame).
// Into pageable memory the HIP runtime pins the destination pages on the fly: 14 ms for the same 64 MiB the first
// time a buffer is used, 1.2 ms when the same buffer is passed again (tools/scratch/hostmem_probe.cpp).  A staging
// scheme of our own (pinned bounce buffer + host copy threads) was built and measured slower than that.
struct OutCopy { void *dst; const void *src; size_t bytes; };

static int copy_out(hipStream_t s, const std::vector<OutCopy> &outs)
{
    for (const OutCopy &o : outs)
        if (o.bytes) HIP_TRY(hipMemcpyAsync(o.dst, o.src, o.bytes, hipMemcpyDeviceToHost, s));
    return LT_OK;
}

extern "C" int lt_render(const lt_camera *cam, const lt_metric *metric, const lt_opts *opts, const float *bg,
                         int32_t bg_channels, float *out_fa, uint16_t *out_w, int8_t *out_status, uint32_t *out_steps,
                         float *out_rgb, uint8_t *out_rgba, lt_stats *stats)
{
    int rc = require_device();
    if (rc) return rc;
    if (!cam || !metric || !opts) return fail(LT_ERR_INVALID_ARG, "null camera / metric / opts");
    lt_opts o = *opts;
    if ((rc = check_opts(metric, &o))) return rc;
    if (cam->height <= 0 || cam->width <= 0) return fail(LT_ERR_INVALID_ARG, "bad frame / partition");
    int64_t rows = 0;
    if ((rc = partition_blocks(cam->height, o, nullptr, &rows))) return rc;
    if (bg && bg_channels != 1 && bg_channels != 3) return fail(LT_ERR_INVALID_ARG, "bg_channels must be 1 or 3");
    size_t n = (size_t)rows * cam->width;
    size_t n_full = (size_t)cam->height * cam->width;
    int nch = bg ? bg_channels : 3;
    hipStream_t s = (hipStream_t)o.stream;
    StreamSlot *sl;
    if ((rc = get_slot(s, &sl))) return rc;
    // device-side inputs / outputs: pieces of ONE grow-only buffer of this (device, stream) -- nothing is
    // allocated per call once it has reached the frame size
    Carver cv;
    size_t o_stats = cv.take(LT_STAT_WORDS * 8);
    size_t o_bg = bg ? cv.take(n_full * bg_channels * sizeof(float)) : 0;
    size_t o_fa = out_fa ? cv.take(n * 4) : 0, o_w = out_w ? cv.take(n * 2) : 0, o_st = out_status ? cv.take(n) : 0;
    size_t o_steps = out_steps ? cv.take(n * 4) : 0, o_rgb = out_rgb ? cv.take(n * nch * 4) : 0;
    size_t o_rgba = out_rgba ? cv.take(n * 4) : 0;
    if ((rc = grow(sl->dev, cv.off, s))) return rc;
    char *base = (char *)sl->dev.p;
    auto at = [&](bool want, size_t off) -> void * { return want ? (void *)(base + off) : nullptr; };
    if (bg) HIP_TRY(hipMemcpyAsync(base + o_bg, bg, n_full * bg_channels * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(base + o_stats, 0, LT_STAT_WORDS * 8, s));
    if ((rc = slot_events(sl))) return rc;
    o.timing = 0; // private events: concurrent lt_render_dev(timing = 1) callers keep theirs
    rc = render_dev_impl(cam, metric, &o, (const float *)at(bg != nullptr, o_bg), bg_channels, (float *)at(out_fa, o_fa),
                         (uint16_t *)at(out_w, o_w), (int8_t *)at(out_status, o_st), (uint32_t *)at(out_steps, o_steps),
                         (float *)at(out_rgb, o_rgb), (uint8_t *)at(out_rgba, o_rgba), (uint64_t *)(base + o_stats),
                         &sl->own);
    if (rc) return rc;
    lt_stats st;
    memset(&st, 0, sizeof(st));
    std::vector<OutCopy> outs;
    // small and first to be consumed go first; the framebuffer(s) follow
    if (out_rgba) outs.push_back({out_rgba, base + o_rgba, n * 4});
    if (out_fa) outs.push_back({out_fa, base + o_fa, n * 4});
    if (out_w) outs.push_back({out_w, base + o_w, n * 2});
    if (out_status) outs.push_back({out_status, base + o_st, n});
    if (out_steps) outs.push_back({out_steps, base + o_steps, n * 4});
    if (out_rgb) outs.push_back({out_rgb, base + o_rgb, n * nch * 4});
    if ((rc = copy_out(s, outs))) return rc;
    HIP_TRY(hipMemcpyAsync(st.counters, base + o_stats, LT_STAT_WORDS * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (rows > 0) {
        float ms[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], sl->own.e[i], sl->own.e[i + 1]));
        st.prologue_ms = ms[0]; st.integrate_ms = ms[1]; st.epilogue_ms = ms[2];
    }
    if (stats) *stats = st;
    return LT_OK;
}

// ---- one frame on several devices from one process ----------------------------------------------
static int multi_stream(int dev, int idx, hipStream_t *out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &m : g_multi_streams)
        if (m.dev == dev && m.idx == idx) { *out = m.s; return LT_OK; }
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    g_multi_streams.push_back({dev, idx, s});
    *out = s;
    return LT_OK;
}

extern "C" int lt_render_multi(const lt_camera *cam, const lt_metric *metric, const lt_opts *opts, int32_t n_gpus,
                               const int32_t *devices, const float *bg, int32_t bg_channels, float *out_fa,
                               uint16_t *out_w, int8_t *out_status, uint32_t *out_steps, float *out_rgb,
                               uint8_t *out_rgba, lt_stats *stats)
{
    int rc = require_device();
    if (rc) return rc;
    if (!cam || !metric || !opts) return fail(LT_ERR_INVALID_ARG, "null camera / metric / opts");
    if (n_gpus < 1 || n_gpus > 64) return fail(LT_ERR_INVALID_ARG, "n_gpus = %d", n_gpus);
    if (cam->width <= 0 || cam->height <= 0) return fail(LT_ERR_INVALID_ARG, "empty frame %dx%d", cam->width, cam->height);
    if (bg && bg_channels != 1 && bg_channels != 3) return fail(LT_ERR_INVALID_ARG, "bg_channels must be 1 or 3");
    lt_opts o = *opts;
    if ((rc = check_opts(metric, &o))) return rc;
    const int n_dev = lt_device_count();
    std::vector<int> dev((size_t)n_gpus);
    for (int p = 0; p < n_gpus; ++p) {
        dev[(size_t)p] = devices ? devices[p] : p;
        if (dev[(size_t)p] < 0 || dev[(size_t)p] >= n_dev)
            return fail(LT_ERR_INVALID_ARG, "partition %d wants device %d but %d device(s) are visible", p, dev[(size_t)p], n_dev);
    }
    int keep = 0;
    HIP_TRY(hipGetDevice(&keep));
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{keep};

    const int W = cam->width, H = cam->height, nch = bg ? bg_channels : 3;
    const size_t n_full = (size_t)W * H;
    struct Part { hipStream_t s; StreamSlot *sl; char *base; int64_t rows; size_t o_stats, o_fa, o_w, o_st, o_steps, o_rgb, o_rgba; };
    std::vector<Part> parts((size_t)n_gpus);
    std::vector<lt_stats> each((size_t)n_gpus); // destination of asynchronous copies: must outlive the drain below
    // Any return before the end leaves kernels and device-to-host copies of other partitions in flight, writing the
    // caller's arrays and `each`: drain every stream that was handed work before the frame (or the error) is returned.
    struct Drain {
        std::vector<std::pair<int, hipStream_t>> used;
        bool done = false;
        ~Drain()
        {
            if (done) return;
            for (auto &u : used) { (void)hipSetDevice(u.first); (void)hipStreamSynchronize(u.second); }
        }
    } drain;
    // 1. launch every partition (asynchronous): all devices compute at the same time
    for (int p = 0; p < n_gpus; ++p) {
        Part &P = parts[(size_t)p];
        HIP_TRY(hipSetDevice(dev[(size_t)p]));
        if ((rc = multi_stream(dev[(size_t)p], p, &P.s))) return rc;
        drain.used.push_back({dev[(size_t)p], P.s});
        if ((rc = get_slot(P.s, &P.sl))) return rc;
        P.rows = lt_local_rows(H, o.row_block, n_gpus, p);
        size_t n = (size_t)P.rows * W;
        Carver cv;
        P.o_stats = cv.take(LT_STAT_WORDS * 8);
        size_t o_bg = bg ? cv.take(n_full * bg_channels * sizeof(float)) : 0;
        P.o_fa = out_fa ? cv.take(n * 4) : 0; P.o_w = out_w ? cv.take(n * 2) : 0; P.o_st = out_status ? cv.take(n) : 0;
        P.o_steps = out_steps ? cv.take(n * 4) : 0; P.o_rgb = out_rgb ? cv.take(n * nch * 4) : 0;
        P.o_rgba = out_rgba ? cv.take(n * 4) : 0;
        if ((rc = grow(P.sl->dev, cv.off, P.s))) return rc;
        P.base = (char *)P.sl->dev.p;
        char *base = P.base;
        auto at = [&](bool want, size_t off) -> void * { return want ? (void *)(base + off) : nullptr; };
        if (bg) HIP_TRY(hipMemcpyAsync(base + o_bg, bg, n_full * bg_channels * sizeof(float), hipMemcpyHostToDevice, P.s));
        HIP_TRY(hipMemsetAsync(base + P.o_stats, 0, LT_STAT_WORDS * 8, P.s));
        if ((rc = slot_events(P.sl))) return rc;
        lt_opts op = o;
        op.n_parts = n_gpus; op.part = p; op.stream = (void *)P.s; op.timing = 0;
        op.block_owner = nullptr; op.n_blocks = 0; // lt_render_multi partitions block-cyclically
        rc = render_dev_impl(cam, metric, &op, (const float *)at(bg != nullptr, o_bg), bg_channels, (float *)at(out_fa, P.o_fa),
                             (uint16_t *)at(out_w, P.o_w), (int8_t *)at(out_status, P.o_st), (uint32_t *)at(out_steps, P.o_steps),
                             (float *)at(out_rgb, P.o_rgb), (uint8_t *)at(out_rgba, P.o_rgba), (uint64_t *)(base + P.o_stats),
                             &P.sl->own);
        if (rc) return rc;
    }
    // 2. every device copies its row blocks to their place in the caller's full-frame arrays
    lt_stats total;
    memset(&total, 0, sizeof(total));
    for (int p = 0; p < n_gpus; ++p) {
        Part &P = parts[(size_t)p];
        HIP_TRY(hipSetDevice(dev[(size_t)p]));
        std::vector<OutCopy> outs;
        auto rows_of = [&](void *dst, size_t off, size_t px_bytes) {
            if (!dst) return;
            size_t row_bytes = (size_t)W * px_bytes;
            for (int64_t l0 = 0; l0 < P.rows; l0 += o.row_block) {
                int64_t g0 = lt_global_row(l0, o.row_block, n_gpus, p);
                int64_t nr = P.rows - l0 < o.row_block ? P.rows - l0 : o.row_block;
                outs.push_back({(char *)dst + (size_t)g0 * row_bytes, P.base + off + (size_t)l0 * row_bytes, (size_t)nr * row_bytes});
            }
        };
        rows_of(out_rgba, P.o_rgba, 4);
        rows_of(out_fa, P.o_fa, 4);
        rows_of(out_w, P.o_w, 2);
        rows_of(out_status, P.o_st, 1);
        rows_of(out_steps, P.o_steps, 4);
        rows_of(out_rgb, P.o_rgb, (size_t)nch * 4);
        if ((rc = copy_out(P.s, outs))) return rc;
        memset(&each[(size_t)p], 0, sizeof(lt_stats));
        HIP_TRY(hipMemcpyAsync(each[(size_t)p].counters, P.base + P.o_stats, LT_STAT_WORDS * 8, hipMemcpyDeviceToHost, P.s));
    }
    for (int p = 0; p < n_gpus; ++p) {
        Part &P = parts[(size_t)p];
        HIP_TRY(hipSetDevice(dev[(size_t)p]));
        HIP_TRY(hipStreamSynchronize(P.s));
        for (int i = 0; i < LT_STAT_WORDS; ++i) total.counters[i] += each[(size_t)p].counters[i];
        if (P.rows > 0) {
            float ms[3] = {0, 0, 0};
            for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], P.sl->own.e[i], P.sl->own.e[i + 1]));
            if (ms[0] > total.prologue_ms) total.prologue_ms = ms[0];
            if (ms[1] > total.integrate_ms) total.integrate_ms = ms[1];
            if (ms[2] > total.epilogue_ms) total.epilogue_ms = ms[2];
        }
    }
    drain.done = true; // every stream was synchronised above
    if (stats) *stats = total;
    return LT_OK;
}

// ---------------------------------------------------------------------------------------------
// array-in / array-out twins
// ---------------------------------------------------------------------------------------------
static int trace_batch(const MetricConsts &mc, lt_opts &o, double lambda_max, const double *alphas, const double *thetas,
                       const uint8_t *refines, int64_t n, double *out_fa, int64_t *out_w, int8_t *out_status,
                       uint32_t *out_evals)
{
    int rc;
    if (n < 0) return fail(LT_ERR_INVALID_ARG, "negative ray count");
    if (n == 0) return LT_OK; // image_lens.py:163-166: empty input is legal
    if (!alphas || !out_fa || !out_w) return fail(LT_ERR_INVALID_ARG, "null alphas / out_fa / out_w");
    if (mc.kind == LT_METRIC_KERR && !thetas) return fail(LT_ERR_INVALID_ARG, "Kerr needs thetas");
    int64_t n_q = (n + 63) / 64 * 64;
    size_t elem = o.precision == 32 ? sizeof(float) : sizeof(double);
    // The twins are synchronous calls on the default stream, like the reference's ("no async").  Records
    // and staging buffers belong to the (device, default stream) slot, so they never alias what an lt_render_dev
    // call on another stream is using, and nothing is allocated per call once they have grown.
    hipStream_t s = nullptr;
    Workspace w;
    if ((rc = get_workspace(s, (size_t)n_q, elem, &w))) return rc;
    void *ic = w.ic, *fin0 = w.fin0, *fin1 = w.fin1;
    StreamSlot *sl;
    if ((rc = get_slot(s, &sl))) return rc;
    Carver cv;
    size_t o_al = cv.take(n * 8), o_th = thetas ? cv.take(n * 8) : 0, o_ref = refines ? cv.take(n) : 0;
    size_t o_fa = cv.take(n * 8), o_w = cv.take(n * 8), o_st = out_status ? cv.take(n) : 0, o_ev = out_evals ? cv.take(n * 4) : 0;
    if ((rc = grow(sl->dev, cv.off, s))) return rc;
    char *base = (char *)sl->dev.p;
    const double *d_al = (const double *)(base + o_al);
    const double *d_th = thetas ? (const double *)(base + o_th) : nullptr;
    const uint8_t *d_ref = refines ? (const uint8_t *)(base + o_ref) : nullptr;
    double *d_fa = (double *)(base + o_fa);
    int64_t *d_w = (int64_t *)(base + o_w);
    int8_t *d_st = out_status ? (int8_t *)(base + o_st) : nullptr;
    uint32_t *d_ev = out_evals ? (uint32_t *)(base + o_ev) : nullptr;
    HIP_TRY(hipMemcpyAsync((void *)d_al, alphas, n * 8, hipMemcpyHostToDevice, s));
    if (thetas) HIP_TRY(hipMemcpyAsync((void *)d_th, thetas, n * 8, hipMemcpyHostToDevice, s));
    if (refines) HIP_TRY(hipMemcpyAsync((void *)d_ref, refines, n, hipMemcpyHostToDevice, s));
    unsigned gq = (unsigned)((n_q + 255) / 256);
    if (o.precision == 32) k_prologue_arrays<float><<<gq, 256, 0, s>>>(mc, d_al, d_th, d_ref, n, (float4 *)ic, n_q);
    else k_prologue_arrays<double><<<gq, 256, 0, s>>>(mc, d_al, d_th, d_ref, n, (double4 *)ic, n_q);
    HIP_TRY(hipGetLastError());
    rc = o.precision == 32 ? launch_integrate<float>(mc, o, lambda_max, w, n_q, s, nullptr)
                           : launch_integrate<double>(mc, o, lambda_max, w, n_q, s, nullptr);
    if (rc) return rc;
    unsigned gn = (unsigned)((n + 255) / 256);
    if (o.precision == 32)
        k_epilogue_arrays<float><<<gn, 256, 0, s>>>(mc, (const float4 *)fin0, (const float4 *)fin1, n, d_fa, d_w, d_st, d_ev);
    else
        k_epilogue_arrays<double><<<gn, 256, 0, s>>>(mc, (const double4 *)fin0, (const double4 *)fin1, n, d_fa, d_w, d_st, d_ev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_fa, d_fa, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_w, d_w, n * 8, hipMemcpyDeviceToHost, s));
    if (out_status) HIP_TRY(hipMemcpyAsync(out_status, d_st, n, hipMemcpyDeviceToHost, s));
    if (out_evals) HIP_TRY(hipMemcpyAsync(out_evals, d_ev, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LT_OK;
}

extern "C" int lt_trace_batch_schw(double M, double r_obs, const double *alphas, int64_t n, double phi_max, double h_max,
                                   int precision, double *out_fa, int64_t *out_w, int8_t *out_status, uint32_t *out_rhs_evals)
{
    int rc = require_device();
    if (rc) return rc;
    lt_metric m{LT_METRIC_SCHWARZSCHILD, 0, M, 0.0};
    lt_opts o;
    lt_default_opts(&o);
    o.precision = precision; o.phi_max = phi_max; o.h_max = h_max;
    if ((rc = check_opts(&m, &o))) return rc;
    MetricConsts mc;
    if ((rc = make_metric(&m, r_obs, M_PI / 2, o.h_max, &mc))) return rc;
    return trace_batch(mc, o, 0.0, alphas, nullptr, nullptr, n, out_fa, out_w, out_status, out_rhs_evals);
}

extern "C" int lt_trace_batch_kerr(double M, double a, double r_obs, const double *alphas, const double *thetas,
                                   double theta_obs, double lambda_max, const uint8_t *axis_refines, int integrator,
                                   int precision, int schedule, int64_t n, double *out_fa, int64_t *out_w,
                                   int8_t *out_status, uint32_t *out_rhs_evals)
{
    int rc = require_device();
    if (rc) return rc;
    lt_metric m{LT_METRIC_KERR, 0, M, a};
    lt_opts o;
    lt_default_opts(&o);
    o.integrator = integrator; o.precision = precision; o.schedule = schedule;
    if ((rc = check_opts(&m, &o))) return rc;
    MetricConsts mc;
    if ((rc = make_metric(&m, r_obs, theta_obs, 0.0, &mc))) return rc;
    if (integrator != LT_INTEGRATOR_RK4) { mc.evals_fixed = 1; mc.evals_per_step = 6; }
    return trace_batch(mc, o, lambda_max, alphas, thetas, axis_refines, n, out_fa, out_w, out_status, out_rhs_evals);
}

extern "C" int lt_kerr_rhs_probe(double M, double a, const double *states, const double *p_phi, int64_t n, int precision,
                                 double *out)
{
    int rc = require_device();
    if (rc) return rc;
    if (n <= 0) return LT_OK;
    lt_metric m{LT_METRIC_KERR, 0, M, a};
    MetricConsts mc;
    if ((rc = make_metric(&m, 50.0, M_PI / 2, 0.0, &mc))) return rc;
    DevBuf ds, dp, dout;
    if ((rc = ds.alloc(n * 40)) || (rc = dp.alloc(n * 8)) || (rc = dout.alloc(n * 40))) return rc;
    HIP_TRY(hipMemcpy(ds.p, states, n * 40, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dp.p, p_phi, n * 8, hipMemcpyHostToDevice));
    unsigned g = (unsigned)((n + 63) / 64);
    if (precision == 32)
        k_kerr_rhs_probe<float><<<g, 64>>>(make_kerr<float>(mc, 5000.0, 1.0), (const double *)ds.p, (const double *)dp.p, n,
                                           (double *)dout.p);
    else
        k_kerr_rhs_probe<double><<<g, 64>>>(make_kerr<double>(mc, 5000.0, 1.0), (const double *)ds.p, (const double *)dp.p,
                                            n, (double *)dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, n * 40, hipMemcpyDeviceToHost));
    return LT_OK;
}

extern "C" int lt_scatter_rows_dev(const void *d_part, void *d_full, int32_t height, int32_t width, int32_t elem_bytes,
                                   int32_t row_block, int32_t n_parts, int32_t part, void *stream)
{
    int rc = require_device();
    if (rc) return rc;
    int64_t rows = lt_local_rows(height, row_block, n_parts, part);
    if (rows < 0 || width <= 0 || elem_bytes <= 0) return fail(LT_ERR_INVALID_ARG, "bad scatter arguments");
    if (rows > 65535) return fail(LT_ERR_INVALID_ARG, "a partition of %lld rows (at most 65535: grid.y carries the row)", (long long)rows);
    if (rows == 0) return LT_OK;
    if (!d_part || !d_full) return fail(LT_ERR_INVALID_ARG, "null pointer");
    int64_t row_bytes = (int64_t)width * elem_bytes;
    dim3 grid((unsigned)((row_bytes + 16 * 256 - 1) / (16 * 256)), (unsigned)rows);
    k_scatter_rows<<<grid, 256, 0, (hipStream_t)stream>>>((const uint8_t *)d_part, (uint8_t *)d_full, (int)rows, row_bytes,
                                                          row_block, n_parts, part);
    HIP_TRY(hipGetLastError());
    return LT_OK;
}

extern "C" int lt_scatter_rows_indexed_dev(const void *d_rows, void *d_full, const int64_t *d_row_index, int64_t n_rows,
                                           int64_t height, int64_t row_bytes, void *stream)
{
    int rc = require_device();
    if (rc) return rc;
    if (n_rows < 0 || height <= 0 || row_bytes <= 0 || n_rows > 65535 * (int64_t)65535)
        return fail(LT_ERR_INVALID_ARG, "bad scatter arguments (n_rows %lld, height %lld, row_bytes %lld)", (long long)n_rows,
                    (long long)height, (long long)row_bytes);
    if (n_rows == 0) return LT_OK;
    if (!d_rows || !d_full || !d_row_index) return fail(LT_ERR_INVALID_ARG, "null pointer");
    // grid.y carries the row (at most 65535 per launch): a frame of more rows goes in slices
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
        int64_t nr = n_rows - r0 < 65535 ? n_rows - r0 : 65535;
        dim3 grid((unsigned)((row_bytes + 16 * 256 - 1) / (16 * 256)), (unsigned)nr);
        k_scatter_rows_indexed<<<grid, 256, 0, (hipStream_t)stream>>>((const uint8_t *)d_rows + r0 * row_bytes, (uint8_t *)d_full,
                                                                      d_row_index + r0, nr, height, row_bytes);
    }
    HIP_TRY(hipGetLastError());
    return LT_OK;
}

#ifdef LT_PROBES
// Diagnostic microbenchmarks (tools/issue_probe.py, lone_step.py ...): compiled only into the probe build
// (`python __graft_entry__.py --probes` -> lib/libltrace_probes.so), never into the product library.
extern "C" int lt_valu_peak_probe(int mode, int iters, double *tflops)
{
    int rc = require_device();
    if (rc) return rc;
    DevBuf sink;
    if ((rc = sink.alloc(64))) return rc;
    hipDeviceProp_t prop;
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    unsigned grid = (unsigned)prop.multiProcessorCount * 8; // 8 blocks of 256 = 32 waves per CU
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    k_valu_probe<<<grid, 256>>>(mode, 16, (float *)sink.p); // warm-up
    HIP_TRY(hipEventRecord(e0, 0));
    k_valu_probe<<<grid, 256>>>(mode, iters, (float *)sink.p);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    double fma_per_lane = (double)iters * 64.0 * (mode == 1 ? 2.0 : 1.0);
    double flops = fma_per_lane * 2.0 * 256.0 * grid;
    if (tflops) *tflops = flops / (ms * 1e-3) / 1e12;
    return LT_OK;
}

// VALU issue-cost probe: instruction class `index` (see lt_probe.hpp), `waves_per_simd` resident waves
// per SIMD (1..8) on every CU.  Reports the kernel time and the number of wave-instructions each SIMD
// issued, i.e. ns per wave-instruction per SIMD (multiply by the shader clock for cycles).
extern "C" int lt_valu_issue_probe(int index, int waves_per_simd, int iters, int constant_data, char *name_out,
                                   int name_len, double *ns_per_instr, double *clock_mhz)
{
    int rc = require_device();
    if (rc) return rc;
    if (index < 0 || index >= g_n_probes) return fail(LT_ERR_INVALID_ARG, "probe index %d out of range [0,%d)", index, g_n_probes);
    if (waves_per_simd < 1 || waves_per_simd > 8) return fail(LT_ERR_INVALID_ARG, "waves_per_simd must be 1..8");
    const ProbeEntry &pe = g_probes[index];
    if (name_out && name_len > 0) snprintf(name_out, (size_t)name_len, "%s", pe.name);
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    DevBuf sink;
    if ((rc = sink.alloc(64))) return rc;
    unsigned grid = (unsigned)(cus * waves_per_simd);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float sc = constant_data ? 0.0f : 0.999f;
    pe.kernel<<<grid, 256>>>(8, sc, (float *)sink.p);
    HIP_TRY(hipEventRecord(e0, 0));
    pe.kernel<<<grid, 256>>>(iters, sc, (float *)sink.p);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    double instr_per_simd = (double)iters * 64.0 * pe.instr_per_body * waves_per_simd; // 8 bodies x 8 chains
    if (ns_per_instr) *ns_per_instr = ms * 1e6 / instr_per_simd;
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(h, sink.p, sizeof(h), hipMemcpyDeviceToHost));
    if (clock_mhz) *clock_mhz = h[3] ? (double)h[2] / (double)h[3] * 100.0 : 0.0; // s_memtime / s_memrealtime(100 MHz)
    return LT_OK;
}

extern "C" int lt_valu_issue_probe_count(void) { return g_n_probes; }

// RK4-step issue probe: `iters` steps of the Kerr RK4 step per lane at `waves_per_simd` resident waves
// per SIMD.  Returns shader cycles per step per wave-slot-on-a-SIMD (i.e. elapsed cycles x
// waves_per_simd / iters ... divided back out: cycles one SIMD spends per wave-step) and the clock.
extern "C" int lt_rk4_step_probe(int precision, int waves_per_simd, int iters, double *cycles_per_wave_step,
                                 double *clock_mhz)
{
    int rc = require_device();
    if (rc) return rc;
    if (waves_per_simd < 1 || waves_per_simd > 8) return fail(LT_ERR_INVALID_ARG, "waves_per_simd must be 1..8");
    lt_metric m{LT_METRIC_KERR, 0, 1.0, 0.9};
    MetricConsts mc;
    if ((rc = make_metric(&m, 50.0, M_PI / 2, 0.0, &mc))) return rc;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    DevBuf out;
    if ((rc = out.alloc(256))) return rc;
    unsigned grid = (unsigned)(cus * waves_per_simd);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        if (rep) HIP_TRY(hipEventRecord(e0, 0));
        if (precision == 32)
            k_probe_rk4_step<float><<<grid, 256>>>(make_kerr<float>(mc, 5000.0, 1.0), rep ? iters : 16, 0.01f, (float *)out.p);
        else
            k_probe_rk4_step<double><<<grid, 256>>>(make_kerr<double>(mc, 5000.0, 1.0), rep ? iters : 16, 0.01, (double *)out.p);
    }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // the in-kernel stamps come from the oldest wave, which wins issue arbitration: use them for the
    // clock only, and the launch's wall time for the throughput
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
    double mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    if (clock_mhz) *clock_mhz = mhz;
    if (cycles_per_wave_step) *cycles_per_wave_step = (double)ms * 1e-3 * mhz * 1e6 / ((double)iters * waves_per_simd);
    return LT_OK;
}

// Piece probe (diagnostic): PIECE 0 sincos, 1 right-hand side without sincos, 2 the same without the
// reciprocal, 3 the two polynomials alone; 4 evaluations per loop iteration.  Returns SIMD cycles per
// evaluation per wave.
#include "lt_probe_pieces.hpp"
extern "C" int lt_piece_probe(int piece, int waves_per_simd, int iters, double *cycles_per_eval, double *clock_mhz)
{
    int rc = require_device();
    if (rc) return rc;
    lt_metric m{LT_METRIC_KERR, 0, 1.0, 0.9};
    MetricConsts mc;
    if ((rc = make_metric(&m, 50.0, M_PI / 2, 0.0, &mc))) return rc;
    int cus;
    if ((rc = cu_count(&cus))) return rc;
    DevBuf out;
    if ((rc = out.alloc(256))) return rc;
    unsigned grid = (unsigned)(cus * waves_per_simd);
    KerrConsts<float> k = make_kerr<float>(mc, 5000.0, 1.0);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        int it = rep ? iters : 16;
        if (rep) HIP_TRY(hipEventRecord(e0, 0));
        if (piece == 0) k_probe_piece<0><<<grid, 256>>>(k, it, (float *)out.p);
        else if (piece == 1) k_probe_piece<1><<<grid, 256>>>(k, it, (float *)out.p);
        else if (piece == 2) k_probe_piece<2><<<grid, 256>>>(k, it, (float *)out.p);
        else if (piece == 3) k_probe_piece<3><<<grid, 256>>>(k, it, (float *)out.p);
        else if (piece == 4) k_probe_piece<4><<<grid, 256>>>(k, it, (float *)out.p);
        else if (piece == 5) k_probe_piece<5><<<grid, 256>>>(k, it, (float *)out.p);
        else if (piece == 6) k_probe_piece<6><<<grid, 256>>>(k, it, (float *)out.p);
        else k_probe_piece<7><<<grid, 256>>>(k, it, (float *)out.p);
    }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
    double mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    if (clock_mhz) *clock_mhz = mhz;
    if (cycles_per_eval) *cycles_per_eval = (double)ms * 1e-3 * mhz * 1e6 / ((double)iters * 4.0 * waves_per_simd);
    return LT_OK;
}

#endif // LT_PROBES

#include "lt_api_stages.inc"
#include "lt_api_dense.inc"
