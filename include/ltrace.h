/*
 * include/ltrace.h -- C-ABI of libltrace_hip.so, the MI355X (gfx950) null-geodesic
 * ray-tracing library.
 *
 * This is the drop-in boundary for the per-pixel backward light-ray integrator of
 * dhg14n9/Light-path-tracer.  Plain pointers and sizes only; no torch / numpy types.
 * Every entry point names the reference interface it replaces (file:line in the
 * reference tree).  The library is HIP-only: with no usable GPU every compute entry
 * point returns LT_ERR_NO_DEVICE -- there is no CPU fallback.
 *
 * Return convention: 0 on success, a negative LT_ERR_* code otherwise;
 * lt_last_error() returns a thread-local message for the most recent failure.
 */
#ifndef LTRACE_H
#define LTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_VERSION 200 /* 0.2.0: lt_stats grew to 16 counters; per-stream workspaces; lt_render_multi */

#define LT_OK 0
#define LT_ERR_INVALID_ARG (-1)
#define LT_ERR_HIP (-2)
#define LT_ERR_NO_DEVICE (-3)
#define LT_ERR_UNSUPPORTED (-4)

/* metric kinds: metrics.py:735 (Schwarzschild), :840 (Kerr) */
#define LT_METRIC_SCHWARZSCHILD 0
#define LT_METRIC_KERR 1

/* integrators: metrics.py:419-567 (DP45, the reference's production Kerr path, float64 only)
 *              metrics.py:570-658 (radius-banded fixed-step RK4; float32 or float64)
 * Schwarzschild always uses its orbit-equation RK4, metrics.py:49-117. */
#define LT_INTEGRATOR_DP45 0
#define LT_INTEGRATOR_RK4 1
#define LT_INTEGRATOR_DP45_EXACT 2 /* DP45 with the step-size controller evaluated in float64, operation by operation as
                                      metrics.py:506-522, :560-564 write it (LT_INTEGRATOR_DP45 evaluates it in float32, free
                                      of divisions and pow).  Divisions and the power are ~1-ulp reciprocal / Newton forms, not
                                      IEEE library calls: an accept / reject decision can differ from the reference's only
                                      when err_norm is within a few ulp of 1 (measured: no golden ray) */

/* ray -> lane scheduling of the integrate kernel */
#define LT_SCHED_DIRECT 0 /* one work-item per ray, one 8x8 pixel tile per wavefront        */
#define LT_SCHED_QUEUE 1  /* persistent wavefronts, ray queue, ballot/prefix-sum lane refill */

/* per-ray status, as the reference's integrators return it (metrics.py:69, :125) */
#define LT_STATUS_ESCAPED 1
#define LT_STATUS_CAPTURED (-1)
#define LT_STATUS_INVALID 0

/* Pinhole camera of image_lens.py:133-152 / :193-208 (pixel corners, no +0.5).
 * psi = (pitch_up, yaw_right) BH offset in radians, image_lens.py:21-61. */
typedef struct lt_camera {
    int32_t width, height;
    double hfov, vfov;
    double psi_y, psi_x;
    double r_obs;     /* observer radius (in the same units as M) */
    double theta_obs; /* observer inclination; the reference only ever uses pi/2 */
} lt_camera;

typedef struct lt_metric {
    int32_t kind; /* LT_METRIC_* */
    int32_t reserved;
    double M;
    double a; /* spin, |a| <= M; ignored for Schwarzschild */
} lt_metric;

typedef struct lt_opts {
    int32_t integrator;      /* LT_INTEGRATOR_*; Kerr only */
    int32_t precision;       /* 32 or 64: arithmetic of the integrate kernel */
    int32_t schedule;        /* LT_SCHED_* */
    int32_t tb_symmetry;     /* 0 = trace every row; 1 = reference behaviour incl. its
                                off-by-one mirror (image_lens.py:218-220, :272-276) */
    int32_t loop_around;     /* render_loop_around of image_lens.py:296-298 */
    int32_t row_block;       /* rows per block of the block-cyclic row partition (>0) */
    int32_t n_parts;         /* number of partitions (GPUs); 1 = whole frame */
    int32_t part;            /* this call renders blocks b with b % n_parts == part */
    double axis_refine_frac; /* Y_AXIS_REFINE_FRAC = 0.07, image_lens.py:14 */
    double phi_max;          /* Schwarzschild: 50.0 (metrics.py:833) */
    double h_max;            /* Schwarzschild: 0.05 (metrics.py:833); Kerr RK4: 1.0 (metrics.py:677) */
    void *stream;            /* hipStream_t to launch on; NULL = the default stream */
    int32_t timing;          /* !=0: bracket each kernel with HIP events (lt_timing_collect) */
    int32_t bg_sampling;     /* LT_BG_*: how the epilogue reads the background image */
    const uint16_t *block_owner; /* NULL: row block b belongs to partition b % n_parts (block-cyclic).  Else a HOST
                                array of n_blocks = ceil(height / row_block) entries, block_owner[b] in [0, n_parts):
                                any assignment of row blocks to partitions (e.g. cost-weighted from the previous
                                frame's step counts, sharding.balance_blocks).  A partition's local rows are its
                                blocks in ascending order.  Read during the call only. */
    int32_t n_blocks;        /* entries of block_owner (checked against the frame) */
    int32_t reserved;
} lt_opts;

/* background sampling of the epilogue kernel (same texels, bit-identical images) */
#define LT_BG_LDS_TILES 0 /* source bounding box of each 256-pixel group staged in LDS by coalesced row reads;
                             groups whose box does not fit (strong lensing) use the global gather */
#define LT_BG_GLOBAL 1    /* one global nearest-neighbour read per pixel */

/* Counters produced by the epilogue kernel (one 64-bit word each, device or host). */
#define LT_STAT_RAYS 0      /* rays integrated */
#define LT_STAT_STEPS 1     /* sum over rays of integrator steps (RK4 steps / DP45 attempts) */
#define LT_STAT_RHS_EVALS 2 /* sum over rays of right-hand-side evaluations */
#define LT_STAT_ESCAPED 3
#define LT_STAT_CAPTURED 4
#define LT_STAT_INVALID 5
/* Produced by the integrate kernel itself (Kerr; zero for Schwarzschild): the work it issued and the
 * clock the chip held while it ran.  bench.py prices its executed-instruction roofline in these. */
#define LT_STAT_WAVE_ITERS 6 /* sum over wavefronts of integrator loop iterations the wavefront issued */
#define LT_STAT_WAVES 7      /* wavefronts of the integrate kernel                                     */
#define LT_STAT_CLK_CYCLES 8 /* shader-clock cycles (s_memtime) over the lifetime of every 64th wave   */
#define LT_STAT_CLK_TICKS 9  /* 100 MHz real-time ticks (s_memrealtime) over the same lifetimes        */
#define LT_STAT_BG_TILES_LDS 10    /* epilogue, lensed background: 256-pixel groups whose source texels were staged in LDS */
#define LT_STAT_BG_TILES_GLOBAL 11 /* ... and groups that fell back to the per-pixel global gather                  */
#define LT_STAT_WORDS 16

typedef struct lt_stats {
    uint64_t counters[LT_STAT_WORDS];
    double prologue_ms, integrate_ms, epilogue_ms; /* HIP-event times of the three kernels */
} lt_stats;

/* ---- library / device ------------------------------------------------------------- */
int lt_version(void);
/* Hash of the kernel sources and compiler flags this binary was built from (set by the build script);
 * profiles under profiles/ carry it so that a figure measured on another build is never reused. */
const char *lt_build_id(void);
const char *lt_last_error(void);
int lt_device_count(void);
int lt_set_device(int device);
/* Frees workspaces, staging buffers and events created lazily by the calls below.
 *
 * Concurrency contract.  Everything the library allocates on a caller's behalf -- the ray records of
 * lt_render_dev, the device-side outputs and the pinned staging of lt_render and of the batch twins --
 * is owned per (device, stream): calls on DIFFERENT streams of a device never share memory and may be
 * in flight at the same time; calls on the SAME stream are ordered by the stream.  The host-pointer
 * entry points (lt_render, lt_trace_batch_*, lt_integrate_dense ...) are synchronous and must not be
 * entered from two host threads with the same stream at once.  The batch twins always use the default
 * (NULL) stream, like the reference's synchronous calls.  Buffers grow to the largest frame seen and are
 * then reused: nothing is allocated per call. */
int lt_shutdown(void);
/* Frees what the library holds for (current device, stream) after draining the stream: call it before destroying
 * a stream that was used with the library (otherwise its buffers stay until lt_shutdown). */
int lt_release_stream(void *stream);
void lt_default_opts(lt_opts *o);

/* ---- array-in / array-out twins of the reference batch drivers ---------------------- *
 * HOST pointers; the library stages H2D / D2H itself.  Same meaning as the reference:   *
 * out_fa[i] = final_alpha if the ray escaped else NaN; out_w[i] = number of half orbits. */

/* Replaces Schwarzschild.trace_rays_batch (metrics.py:831-833) ->
 * _trace_rays_batch_schwarzschild (metrics.py:661-668).  precision 32 or 64.
 * out_status / out_rhs_evals (4 per RK4 step) may be NULL. */
int lt_trace_batch_schw(double M, double r_obs, const double *alphas, int64_t n, double phi_max,
                        double h_max, int precision, double *out_fa, int64_t *out_w,
                        int8_t *out_status, uint32_t *out_rhs_evals);

/* Replaces Kerr.trace_rays_batch (metrics.py:1128-1132) -> _trace_rays_batch_kerr
 * (metrics.py:671-679).  integrator LT_INTEGRATOR_*, precision 32 or 64 (DP45: 64 only).
 * axis_refines: one byte per ray (numpy bool), may be NULL (= all false).
 * out_status / out_rhs_evals may be NULL. */
int lt_trace_batch_kerr(double M, double a, double r_obs, const double *alphas, const double *thetas,
                        double theta_obs, double lambda_max, const uint8_t *axis_refines,
                        int integrator, int precision, int schedule, int64_t n, double *out_fa,
                        int64_t *out_w, int8_t *out_status, uint32_t *out_rhs_evals);

/* Device probe of the inlined Kerr right-hand side (metrics.py:221-303) for parity tests:
 * states (n,5) [r, theta, phi, p_r, p_theta], p_phi (n), out (n,5); host pointers, float64 I/O,
 * evaluated in float32 or float64 on the GPU. */
int lt_kerr_rhs_probe(double M, double a, const double *states, const double *p_phi, int64_t n,
                      int precision, double *out);

/* ---- fused frame path ---------------------------------------------------------------- *
 * Replaces, in one call, build_alpha_lookup (image_lens.py:133-152),                     *
 * precompute_final_alpha_lookup / _2d (image_lens.py:155-178 / :185-280) and             *
 * render_lensed_image (image_lens.py:296-397) + the float->RGBA8 step of                 *
 * mpimg.imsave (image_lens.py:510).  Nothing per-ray crosses PCIe on the way in.         */

/* Number of image rows partition `part` of `n_parts` owns (block-cyclic, blocks of row_block). */
int64_t lt_local_rows(int32_t height, int32_t row_block, int32_t n_parts, int32_t part);
/* Global row index of local row `local_row` of that partition. */
int64_t lt_global_row(int64_t local_row, int32_t row_block, int32_t n_parts, int32_t part);

/* DEVICE pointers (any may be NULL = not wanted), sized for R = lt_local_rows(...) rows:
 *   d_bg     (H, W, bg_channels) float32 background, the FULL frame on every partition
 *            (NULL: shadow render, escaped pixels white);  bg_channels 1 or 3
 *   d_fa     (R, W) float32 final_alpha lookup (NaN unless escaped)
 *   d_w      (R, W) uint16 winding lookup
 *   d_status (R, W) int8
 *   d_steps  (R, W) uint32 integrator steps of the ray
 *   d_rgb    (R, W, bg_channels or 3) float32 lensed image, as render_lensed_image returns it
 *   d_rgba   (R, W, 4) uint8, as imsave writes it
 *   d_stats  LT_STAT_WORDS uint64 counters, ACCUMULATED into (caller zeroes)
 * Asynchronous on opts->stream. */
int lt_render_dev(const lt_camera *cam, const lt_metric *metric, const lt_opts *opts,
                  const float *d_bg, int32_t bg_channels, float *d_fa, uint16_t *d_w,
                  int8_t *d_status, uint32_t *d_steps, float *d_rgb, uint8_t *d_rgba,
                  uint64_t *d_stats);

/* Same with HOST pointers (stages the background H2D, results D2H, synchronises); stats may be NULL.
 * Device-side buffers persist per (device, stream).  Every output is copied straight into the caller's memory:
 * into a block from lt_host_alloc (or other pinned / registered host memory) by DMA at PCIe rate; into pageable
 * memory through the HIP runtime, which pins the pages on the fly (about 10x slower the first time a buffer is
 * used, PCIe rate when the same buffer is passed again). */
int lt_render(const lt_camera *cam, const lt_metric *metric, const lt_opts *opts,
              const float *bg, int32_t bg_channels, float *out_fa, uint16_t *out_w,
              int8_t *out_status, uint32_t *out_steps, float *out_rgb, uint8_t *out_rgba,
              lt_stats *stats);

/* Pinned host memory for the outputs of the host-pointer entry points (NULL on failure, see
 * lt_last_error); ltrace.py hands such blocks out as numpy arrays. */
void *lt_host_alloc(size_t bytes);
int lt_host_free(void *p);

/* One frame on several GPUs of this node from ONE process (SURVEY 8b `lt_render_multi`): partition p of
 * n_gpus (block-cyclic rows, opts->row_block; opts->n_parts / part / stream are ignored) is rendered on
 * device devices[p] (NULL: device p), all partitions concurrently, and every device copies its rows
 * straight into the caller's full-frame HOST arrays -- no device-to-device gather.  `devices` may name
 * one device several times (partitions then queue on it).  Outputs as lt_render, sized for the FULL
 * (H, W) frame; stats sums the counters and takes the slowest device's kernel times.
 * The RCCL gather of the north-star contract is the multi-process path (sharding.FrameGather / bench.py). */
int lt_render_multi(const lt_camera *cam, const lt_metric *metric, const lt_opts *opts, int32_t n_gpus,
                    const int32_t *devices, const float *bg, int32_t bg_channels, float *out_fa, uint16_t *out_w,
                    int8_t *out_status, uint32_t *out_steps, float *out_rgb, uint8_t *out_rgba, lt_stats *stats);

/* ---- the first and the last stage on their own (HOST pointers) ------------------------------ *
 * The reference's pipeline is three calls; lt_render fuses them, these keep each call a GPU twin. */

/* Replaces build_alpha_lookup (image_lens.py:133-152) and the theta / axis-refine-column part of
 * precompute_final_alpha_lookup_2d (image_lens.py:193-216).  Any output may be NULL:
 * out_alpha (H, W) float32, out_theta (H, W) float64, out_axis_cols (W) bytes. */
int lt_pixel_angles(const lt_camera *cam, double axis_refine_frac, float *out_alpha, double *out_theta,
                    uint8_t *out_axis_cols);

/* Replaces render_lensed_image (image_lens.py:296-397) for caller-supplied lookups (+ imsave's RGBA8):
 * bg (H, W, bg_channels) float32, fa (H, W) float32, winding (H, W) uint16 or NULL;
 * out_rgb (H, W, bg_channels) float32 and / or out_rgba (H, W, 4) uint8. */
int lt_shade(const lt_camera *cam, int32_t loop_around, const float *bg, int32_t bg_channels, const float *fa,
             const uint16_t *winding, float *out_rgb, uint8_t *out_rgba);

/* Scatter a partition's (R, W, elem_bytes) rows into the full (H, W, elem_bytes) frame
 * (device pointers, async on `stream`): the un-permute step after the multi-GPU gather. */
int lt_scatter_rows_dev(const void *d_part, void *d_full, int32_t height, int32_t width,
                        int32_t elem_bytes, int32_t row_block, int32_t n_parts, int32_t part,
                        void *stream);

/* The same for ANY assignment of rows (e.g. the frame as rank 0 received it, partition after partition, under a
 * row-block -> rank table): d_rows holds n_rows rows of row_bytes bytes, source row i goes to row d_row_index[i] of
 * d_full (height rows).  d_row_index is a DEVICE array of n_rows int64; entries outside [0, height) are skipped.
 * One launch per 65535 rows, asynchronous on `stream`. */
int lt_scatter_rows_indexed_dev(const void *d_rows, void *d_full, const int64_t *d_row_index, int64_t n_rows,
                                int64_t height, int64_t row_bytes, void *stream);

/* ---- batched dense trajectories ------------------------------------------------------------- *
 * Replaces geodesic_tracer.integrate_geodesic (geodesic_tracer.py:22-71) -- solve_ivp(RK45) on     *
 * metric.geodesic_equations (metrics.py:763-790 / :946-1029) with a capture and an escape radius   *
 * event -- for n 8-D initial states (t, r, theta, phi, p_t, p_r, p_theta, p_phi) at once, as      *
 * metric.initial_conditions returns them (metrics.py:792-808 / :1032-1107).  float64.              */
typedef struct lt_dense_opts {
    double lambda_max;   /* affine range, geodesic_tracer.py:22 (1000)                                  */
    double r_stop_inner; /* < 0: the metric's capture radius 1.01 r_plus (geodesic_tracer.py:43-44)     */
    double r_stop_outer; /* < 0: twice each track's start radius (geodesic_tracer.py:45-46)             */
    double rtol, atol;   /* geodesic_tracer.py:64-65 (1e-8, 1e-10)                                      */
    double max_step;     /* geodesic_tracer.py:63 (1.0)                                                 */
    int64_t max_points;  /* record capacity per track (>= 2)                                            */
    int32_t max_attempts; /* guard against a track that never ends (status -2); solve_ivp has none      */
    int32_t length_binning; /* order of the launch, never of the output: 1 = predict every track's length with a cheap
                               loose-tolerance pass and launch the tracks of every window of 2048 longest first, so that
                               the 64 tracks of a wavefront end together (records are byte-identical either way); -1 = caller order;
                               0 = automatic (binned when the batch is at least twice what the chip holds at once) */
    void *stream;        /* hipStream_t; NULL = the default stream                                      */
} lt_dense_opts;
void lt_default_dense_opts(lt_dense_opts *o);

/* Track status: which of solve_ivp's endings the track took. */
#define LT_TRACK_RANGE_END 0   /* lambda_max reached (solve_ivp status 0)        */
#define LT_TRACK_CAPTURE_EVENT 1 /* r fell through r_stop_inner (status 1, event 0) */
#define LT_TRACK_ESCAPE_EVENT 2  /* r rose through r_stop_outer (status 1, event 1) */
#define LT_TRACK_FAILED (-1)     /* step size underflow (solve_ivp status -1)       */
#define LT_TRACK_ATTEMPT_LIMIT (-2)

/* HOST pointers.  state0 (n, 8).  Records are track-major, the reference's solution.t / solution.y per track:
 *   out_t (n, max_points), out_y (n, max_points, 8): track i's k-th point is out_t[i*max_points + k],
 *   out_y[(i*max_points + k)*8 + c] -- solution.t[k], solution.y[c, k].  (Version 100 wrote them point-major.)
 *   A track that ended on a radius event and whose record has room holds ONE more value, out_t[i*max_points + count]:
 *   the affine parameter at which the step containing the event would have ended (solve_ivp's `solution.sol` interpolates
 *   the last stretch with the dense output of that whole step; geodesic_tracer.Track.sol rebuilds it from this).
 *   out_count (n): points of the complete record.  If it exceeds max_points only the first
 *   max_points - 1 points are kept and the last slot holds the final point.
 *   out_status (n): LT_TRACK_*.   out_nfev (n): right-hand-side evaluations (solution.nfev). */
int lt_integrate_dense(const lt_metric *metric, const lt_dense_opts *opts, const double *state0, int64_t n,
                       double *out_t, double *out_y, int32_t *out_count, int8_t *out_status, int32_t *out_nfev);
/* Same with DEVICE pointers, asynchronous on opts->stream. */
int lt_integrate_dense_dev(const lt_metric *metric, const lt_dense_opts *opts, const double *d_state0, int64_t n,
                           double *d_out_t, double *d_out_y, int32_t *d_out_count, int8_t *d_out_status,
                           int32_t *d_out_nfev);
/* Diagnostic: the keys the length-binned launch orders the tracks by -- predicted step attempts of each track,
 * clamped to 2047 -- from the loose-tolerance predictor pass alone.  HOST pointers, out_key (n) uint16. */
int lt_dense_predict_lengths(const lt_metric *metric, const lt_dense_opts *opts, const double *state0, int64_t n,
                             uint16_t *out_key);
/* Device probe of the 8-D right-hand side for parity tests: states (n, 8) -> out (n, 8), host pointers. */
int lt_rhs8_probe(const lt_metric *metric, const double *states, int64_t n, double *out);

/* Sum of HIP-event times (ms) of the prologue / integrate / epilogue kernels over all
 * lt_render_dev calls made with opts->timing != 0 since the last collect; *calls = how many.
 * Synchronises on the recorded events. */
int lt_timing_collect(double *prologue_ms, double *integrate_ms, double *epilogue_ms, int32_t *calls);

#ifdef __cplusplus
}
#endif
#endif /* LTRACE_H */
