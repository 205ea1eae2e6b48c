"""ctypes binding of libltrace_hip.so (C-ABI: include/ltrace.h).

Thin by design: structs, argument marshalling and error translation only.  The
library is the product; it is HIP-only.  If the shared object is missing, or no
GPU is visible, calls raise -- there is no CPU fallback here or anywhere in this
package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LTRACE_LIB", os.path.join(_HERE, "lib", "libltrace_hip.so"))

OK, ERR_INVALID_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4
METRIC_SCHWARZSCHILD, METRIC_KERR = 0, 1
INTEGRATOR_DP45, INTEGRATOR_RK4, INTEGRATOR_DP45_EXACT = 0, 1, 2
SCHED_DIRECT, SCHED_QUEUE = 0, 1
STAT_RAYS, STAT_STEPS, STAT_RHS_EVALS, STAT_ESCAPED, STAT_CAPTURED, STAT_INVALID = range(6)
STAT_WAVE_ITERS, STAT_WAVES, STAT_CLK_CYCLES, STAT_CLK_TICKS = 6, 7, 8, 9
STAT_BG_TILES_LDS, STAT_BG_TILES_GLOBAL = 10, 11
STAT_WORDS = 16

INTEGRATORS = {"dp45": INTEGRATOR_DP45, "rk4": INTEGRATOR_RK4, "dp45_exact": INTEGRATOR_DP45_EXACT}
SCHEDULES = {"direct": SCHED_DIRECT, "queue": SCHED_QUEUE}


class LtraceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libltrace_hip error {code}: {msg}")
        self.code = code


class Camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("hfov", C.c_double), ("vfov", C.c_double),
                ("psi_y", C.c_double), ("psi_x", C.c_double),
                ("r_obs", C.c_double), ("theta_obs", C.c_double)]


class Metric(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("M", C.c_double), ("a", C.c_double)]


class Opts(C.Structure):
    _fields_ = [("integrator", C.c_int32), ("precision", C.c_int32), ("schedule", C.c_int32),
                ("tb_symmetry", C.c_int32), ("loop_around", C.c_int32), ("row_block", C.c_int32),
                ("n_parts", C.c_int32), ("part", C.c_int32),
                ("axis_refine_frac", C.c_double), ("phi_max", C.c_double), ("h_max", C.c_double),
                ("stream", C.c_void_p), ("timing", C.c_int32), ("bg_sampling", C.c_int32),
                ("block_owner", C.c_void_p), ("n_blocks", C.c_int32), ("reserved", C.c_int32)]


BG_LDS_TILES, BG_GLOBAL = 0, 1


class DenseOpts(C.Structure):
    _fields_ = [("lambda_max", C.c_double), ("r_stop_inner", C.c_double), ("r_stop_outer", C.c_double),
                ("rtol", C.c_double), ("atol", C.c_double), ("max_step", C.c_double),
                ("max_points", C.c_int64), ("max_attempts", C.c_int32), ("length_binning", C.c_int32),
                ("stream", C.c_void_p)]


TRACK_RANGE_END, TRACK_CAPTURE_EVENT, TRACK_ESCAPE_EVENT, TRACK_FAILED, TRACK_ATTEMPT_LIMIT = 0, 1, 2, -1, -2


class Stats(C.Structure):
    _fields_ = [("counters", C.c_uint64 * STAT_WORDS),
                ("prologue_ms", C.c_double), ("integrate_ms", C.c_double), ("epilogue_ms", C.c_double)]


_dp = C.POINTER(C.c_double)
_lib = None

# name -> (restype, argtypes); every symbol include/ltrace.h declares
SIGNATURES = {
    "lt_version": (C.c_int, []),
    "lt_build_id": (C.c_char_p, []),
    "lt_host_alloc": (C.c_void_p, [C.c_size_t]),
    "lt_host_free": (C.c_int, [C.c_void_p]),
    "lt_render_multi": (C.c_int, [C.POINTER(Camera), C.POINTER(Metric), C.POINTER(Opts), C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(Stats)]),
    "lt_last_error": (C.c_char_p, []),
    "lt_device_count": (C.c_int, []),
    "lt_set_device": (C.c_int, [C.c_int]),
    "lt_shutdown": (C.c_int, []),
    "lt_release_stream": (C.c_int, [C.c_void_p]),
    "lt_default_opts": (None, [C.POINTER(Opts)]),
    "lt_trace_batch_schw": (C.c_int, [C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_trace_batch_kerr": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_double,
                                      C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_kerr_rhs_probe": (C.c_int, [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "lt_local_rows": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "lt_global_row": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "lt_render_dev": (C.c_int, [C.POINTER(Camera), C.POINTER(Metric), C.POINTER(Opts), C.c_void_p, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_render": (C.c_int, [C.POINTER(Camera), C.POINTER(Metric), C.POINTER(Opts), C.c_void_p, C.c_int32,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.POINTER(Stats)]),
    "lt_scatter_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p]),
    "lt_scatter_rows_indexed_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "lt_timing_collect": (C.c_int, [_dp, _dp, _dp, C.POINTER(C.c_int32)]),
    "lt_pixel_angles": (C.c_int, [C.POINTER(Camera), C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_shade": (C.c_int, [C.POINTER(Camera), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p]),
    "lt_default_dense_opts": (None, [C.POINTER(DenseOpts)]),
    "lt_integrate_dense": (C.c_int, [C.POINTER(Metric), C.POINTER(DenseOpts), C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_integrate_dense_dev": (C.c_int, [C.POINTER(Metric), C.POINTER(DenseOpts), C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lt_rhs8_probe": (C.c_int, [C.POINTER(Metric), C.c_void_p, C.c_int64, C.c_void_p]),
    "lt_dense_predict_lengths": (C.c_int, [C.POINTER(Metric), C.POINTER(DenseOpts), C.c_void_p, C.c_int64, C.c_void_p]),
}


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != OK:
        raise LtraceError(rc, load().lt_last_error().decode("utf-8", "replace"))


def device_count():
    return int(load().lt_device_count())


def require_gpu():
    if device_count() <= 0:
        raise LtraceError(ERR_NO_DEVICE, "no HIP device visible; this package has no CPU path")


def default_opts(**kw):
    o = Opts()
    load().lt_default_opts(C.byref(o))
    for k, v in kw.items():
        if k == "integrator" and isinstance(v, str):
            v = INTEGRATORS[v]
        if k == "schedule" and isinstance(v, str):
            v = SCHEDULES[v]
        if k == "block_owner":
            set_block_owner(o, v)
            continue
        setattr(o, k, v)
    return o


def set_block_owner(opts, owner):
    """Attach a row-block -> partition table (uint16, one entry per row block) to opts; None = block-cyclic."""
    if owner is None:
        opts.block_owner, opts.n_blocks, opts._owner_keep = None, 0, None
        return
    tab = np.ascontiguousarray(owner, dtype=np.uint16)
    opts._owner_keep = tab                       # the struct only holds the address
    opts.block_owner, opts.n_blocks = tab.ctypes.data, tab.size


def owned_rows(height, row_block, owner, part):
    """Global row index of every local row of partition `part` under a block-owner table (ascending)."""
    owner = np.asarray(owner)
    rows = [np.arange(b * row_block, min((b + 1) * row_block, height)) for b in np.nonzero(owner == part)[0]]
    return np.concatenate(rows).astype(np.int64) if rows else np.zeros(0, dtype=np.int64)


def _np_ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _out(a, dtype, n, name):
    if a is None:
        return None
    if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous and a.size == n
            and a.flags.writeable):
        raise ValueError(f"{name} must be a writable C-contiguous {np.dtype(dtype).name} array of {n} elements")
    return a


def trace_batch_schw(M, r_obs, alphas, out_fa, out_w, phi_max=50.0, h_max=0.05, precision=32,
                     out_status=None, out_rhs_evals=None):
    """In-place twin of _trace_rays_batch_schwarzschild (reference metrics.py:661-668)."""
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    n = al.size
    _out(out_fa, np.float64, n, "out_fa")
    _out(out_w, np.int64, n, "out_w")
    _out(out_status, np.int8, n, "out_status")
    _out(out_rhs_evals, np.uint32, n, "out_rhs_evals")
    _check(load().lt_trace_batch_schw(M, r_obs, _np_ptr(al), n, phi_max, h_max, precision,
                                      _np_ptr(out_fa), _np_ptr(out_w), _np_ptr(out_status),
                                      _np_ptr(out_rhs_evals)))


def trace_batch_kerr(M, a, r_obs, alphas, thetas, theta_obs, lambda_max, axis_refines, out_fa, out_w,
                     integrator=INTEGRATOR_RK4, precision=32, schedule=SCHED_DIRECT,
                     out_status=None, out_rhs_evals=None):
    """In-place twin of _trace_rays_batch_kerr (reference metrics.py:671-679)."""
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    th = np.ascontiguousarray(thetas, dtype=np.float64)
    n = al.size
    if th.size != n:
        raise ValueError("alphas and thetas differ in length")
    ar = None
    if axis_refines is not None:
        ar = np.ascontiguousarray(axis_refines).astype(np.uint8)
        if ar.size != n:
            raise ValueError("axis_refines has the wrong length")
    _out(out_fa, np.float64, n, "out_fa")
    _out(out_w, np.int64, n, "out_w")
    _out(out_status, np.int8, n, "out_status")
    _out(out_rhs_evals, np.uint32, n, "out_rhs_evals")
    if isinstance(integrator, str):
        integrator = INTEGRATORS[integrator]
    if isinstance(schedule, str):
        schedule = SCHEDULES[schedule]
    _check(load().lt_trace_batch_kerr(M, a, r_obs, _np_ptr(al), _np_ptr(th), theta_obs, lambda_max, _np_ptr(ar),
                                      integrator, precision, schedule, n, _np_ptr(out_fa), _np_ptr(out_w),
                                      _np_ptr(out_status), _np_ptr(out_rhs_evals)))


def kerr_rhs_probe(M, a, states, p_phi, precision=32):
    st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 5)
    pp = np.ascontiguousarray(p_phi, dtype=np.float64).ravel()
    out = np.empty_like(st)
    _check(load().lt_kerr_rhs_probe(M, a, _np_ptr(st), _np_ptr(pp), st.shape[0], precision, _np_ptr(out)))
    return out


def local_rows(height, row_block, n_parts, part):
    return int(load().lt_local_rows(height, row_block, n_parts, part))


def global_rows(height, row_block, n_parts, part):
    """Global row index of every local row of a partition (host helper for tests / gathers)."""
    n = local_rows(height, row_block, n_parts, part)
    lib = load()
    return np.array([lib.lt_global_row(i, row_block, n_parts, part) for i in range(n)], dtype=np.int64)


# ---- pinned output arrays -----------------------------------------------------------------------------
# lt_render writes a destination that lies in pinned host memory by DMA straight from the device; a pageable
# numpy array costs an extra pass through the library's staging area.  render() therefore hands out arrays
# backed by lt_host_alloc blocks.  Blocks are recycled by size once the last numpy view of them is gone
# (allocating pinned memory costs milliseconds, a 4096^2 frame every call).
import threading
import weakref

_pinned_free = {}     # nbytes -> [address, ...]
_PINNED_POOL_LIMIT = int(os.environ.get("LTRACE_PINNED_POOL_MB", "2048")) << 20
_PINNED_OUTPUTS = os.environ.get("LTRACE_PINNED_OUTPUTS", "1") != "0"    # 0: render() returns ordinary numpy arrays
_pinned_pooled = 0
_pinned_lock = threading.Lock()      # finalizers run on whatever thread drops the last reference


def _pinned_release(addr, nbytes):
    global _pinned_pooled
    if _lib is None:
        return
    with _pinned_lock:
        if _pinned_pooled + nbytes <= _PINNED_POOL_LIMIT:
            _pinned_free.setdefault(nbytes, []).append(addr)
            _pinned_pooled += nbytes
            return
    _lib.lt_host_free(C.c_void_p(addr))


def pinned_empty(shape, dtype, strict=False):
    """numpy array of `shape` / `dtype` in pinned host memory (lt_host_alloc); freed or recycled when the array
    and all its views are gone.  Ordinary memory for empty arrays, with LTRACE_PINNED_OUTPUTS=0, and -- unless
    `strict` -- when the pinned allocation fails (a caller that keeps many frames alive holds that much
    non-swappable memory; lt_render accepts a pageable destination, it is only slower the first time it sees it)."""
    global _pinned_pooled
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes == 0 or (not _PINNED_OUTPUTS and not strict):
        return np.empty(shape, dtype=dtype)
    nbytes = (nbytes + 4095) & ~4095
    addr = None
    with _pinned_lock:
        lst = _pinned_free.get(nbytes)
        if lst:
            addr = lst.pop()
            _pinned_pooled -= nbytes
    if addr is None:
        addr = load().lt_host_alloc(nbytes)
        if not addr:
            if strict:
                raise LtraceError(ERR_HIP, load().lt_last_error().decode("utf-8", "replace"))
            return np.empty(shape, dtype=dtype)
    buf = (C.c_ubyte * nbytes).from_address(addr)
    weakref.finalize(buf, _pinned_release, addr, nbytes).atexit = False   # at exit the process frees it
    n = int(np.prod(shape, dtype=np.int64))
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def _frame_outputs(rows, W, nch, gray, want):
    out = {}
    if "fa" in want:
        out["fa"] = pinned_empty((rows, W), np.float32)
    if "winding" in want:
        out["winding"] = pinned_empty((rows, W), np.uint16)
    if "status" in want:
        out["status"] = pinned_empty((rows, W), np.int8)
    if "steps" in want:
        out["steps"] = pinned_empty((rows, W), np.uint32)
    if "rgb" in want:
        out["rgb"] = pinned_empty((rows, W) if gray else (rows, W, nch), np.float32)
    if "rgba" in want:
        out["rgba"] = pinned_empty((rows, W, 4), np.uint8)
    return out


def _background(cam, background):
    if background is None:
        return None, 3, False
    bg = np.ascontiguousarray(background, dtype=np.float32)
    if bg.shape[:2] != (cam.height, cam.width):
        raise ValueError("background must have the frame's height and width")
    nch = 1 if bg.ndim == 2 else bg.shape[2]
    if nch not in (1, 3):
        raise ValueError("background must be grayscale or RGB")
    return bg, nch, bg.ndim == 2


def render(cam, metric, opts, background=None, want=("fa", "winding", "status", "steps", "rgb", "rgba")):
    """Host-pointer frame render (lt_render).  Returns dict of numpy arrays (pinned memory) + 'stats'."""
    if opts.block_owner:
        rows = len(owned_rows(cam.height, opts.row_block or 16, opts._owner_keep, opts.part))
    else:
        rows = local_rows(cam.height, opts.row_block or 16, opts.n_parts or 1, opts.part)
    if rows < 0 or cam.width <= 0:
        raise LtraceError(ERR_INVALID_ARG, f"bad frame {cam.width}x{cam.height} or partition {opts.part}/{opts.n_parts}")
    bg, nch, gray = _background(cam, background)
    out = _frame_outputs(rows, cam.width, nch, gray, want)
    st = Stats()
    _check(load().lt_render(C.byref(cam), C.byref(metric), C.byref(opts), _np_ptr(bg), nch,
                            _np_ptr(out.get("fa")), _np_ptr(out.get("winding")), _np_ptr(out.get("status")),
                            _np_ptr(out.get("steps")), _np_ptr(out.get("rgb")), _np_ptr(out.get("rgba")),
                            C.byref(st)))
    out["stats"] = stats_dict(st.counters, st.prologue_ms, st.integrate_ms, st.epilogue_ms)
    return out


def render_multi(cam, metric, opts, n_gpus, devices=None, background=None,
                 want=("fa", "winding", "status", "steps", "rgb", "rgba")):
    """One frame on `n_gpus` devices of this node from this one process (lt_render_multi): full-frame arrays."""
    bg, nch, gray = _background(cam, background)
    out = _frame_outputs(cam.height, cam.width, nch, gray, want)
    dv = None
    if devices is not None:
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        if dv.size != n_gpus:
            raise ValueError("devices must list one device per partition")
    st = Stats()
    _check(load().lt_render_multi(C.byref(cam), C.byref(metric), C.byref(opts), int(n_gpus), _np_ptr(dv), _np_ptr(bg), nch,
                                  _np_ptr(out.get("fa")), _np_ptr(out.get("winding")), _np_ptr(out.get("status")),
                                  _np_ptr(out.get("steps")), _np_ptr(out.get("rgb")), _np_ptr(out.get("rgba")),
                                  C.byref(st)))
    out["stats"] = stats_dict(st.counters, st.prologue_ms, st.integrate_ms, st.epilogue_ms)
    return out


def build_id():
    return load().lt_build_id().decode()


def pixel_angles(cam, axis_refine_frac=0.07, want_theta=True):
    """(alpha (H,W) f32, theta (H,W) f64 or None, axis_refine_cols (W,) bool) -- lt_pixel_angles."""
    H, W = cam.height, cam.width
    alpha = np.empty((H, W), dtype=np.float32)
    theta = np.empty((H, W), dtype=np.float64) if want_theta else None
    cols = np.zeros(W, dtype=np.uint8)
    _check(load().lt_pixel_angles(C.byref(cam), axis_refine_frac, _np_ptr(alpha), _np_ptr(theta), _np_ptr(cols)))
    return alpha, theta, cols.astype(bool)


def shade(cam, background, fa, winding=None, loop_around=False, want_rgba=False):
    """render_lensed_image twin (lt_shade): returns rgb like the background (and rgba8 if asked)."""
    bg = np.ascontiguousarray(background, dtype=np.float32)
    H, W = bg.shape[:2]
    if (H, W) != (cam.height, cam.width):
        raise ValueError("background and camera sizes differ")
    nch = 1 if bg.ndim == 2 else bg.shape[2]
    fa32 = np.ascontiguousarray(fa, dtype=np.float32)
    wd = None if winding is None else np.ascontiguousarray(winding, dtype=np.uint16)
    if fa32.shape != (H, W) or (wd is not None and wd.shape != (H, W)):
        raise ValueError("lookup shapes must be (H, W)")
    rgb = np.empty_like(bg)
    rgba = np.empty((H, W, 4), dtype=np.uint8) if want_rgba else None
    _check(load().lt_shade(C.byref(cam), int(bool(loop_around)), _np_ptr(bg), nch, _np_ptr(fa32), _np_ptr(wd),
                           _np_ptr(rgb), _np_ptr(rgba)))
    return (rgb, rgba) if want_rgba else rgb


def default_dense_opts(**kw):
    o = DenseOpts()
    load().lt_default_dense_opts(C.byref(o))
    for k, v in kw.items():
        if v is not None:
            setattr(o, k, v)
    return o


def integrate_dense(metric, state0, opts=None):
    """Batched integrate_geodesic (lt_integrate_dense): state0 (n, 8) ->
    (t (n, max_points), y (n, max_points, 8), count (n,), status (n,), nfev (n,)).
    Track i is t[i, :m], y[i, :m].T (= solution.t, solution.y of the reference) with m = min(count[i], max_points)."""
    o = opts or default_dense_opts()
    s0 = np.ascontiguousarray(state0, dtype=np.float64).reshape(-1, 8)
    n, mp = s0.shape[0], int(o.max_points)
    if mp < 2:
        raise LtraceError(ERR_INVALID_ARG, "max_points must be at least 2")
    t = np.empty((n, mp), dtype=np.float64)
    y = np.empty((n, mp, 8), dtype=np.float64)
    count = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.int8)
    nfev = np.zeros(n, dtype=np.int32)
    _check(load().lt_integrate_dense(C.byref(metric), C.byref(o), _np_ptr(s0), n, _np_ptr(t), _np_ptr(y),
                                     _np_ptr(count), _np_ptr(status), _np_ptr(nfev)))
    return t, y, count, status, nfev


def integrate_dense_dev(metric, opts, d_state0, n, d_t, d_y, d_count, d_status, d_nfev):
    """Device-pointer form (integers, e.g. tensor.data_ptr()); asynchronous on opts.stream."""
    _check(load().lt_integrate_dense_dev(C.byref(metric), C.byref(opts), C.c_void_p(d_state0), n, C.c_void_p(d_t),
                                         C.c_void_p(d_y), C.c_void_p(d_count), C.c_void_p(d_status),
                                         C.c_void_p(d_nfev)))


def dense_predict_lengths(metric, state0, opts=None):
    """Predicted step attempts per track (uint16, clamped to 2047): what the length-binned dense launch sorts by."""
    o = opts or default_dense_opts()
    s0 = np.ascontiguousarray(state0, dtype=np.float64).reshape(-1, 8)
    key = np.zeros(s0.shape[0], dtype=np.uint16)
    _check(load().lt_dense_predict_lengths(C.byref(metric), C.byref(o), _np_ptr(s0), s0.shape[0], _np_ptr(key)))
    return key


def rhs8_probe(metric, states):
    st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 8)
    out = np.empty_like(st)
    _check(load().lt_rhs8_probe(C.byref(metric), _np_ptr(st), st.shape[0], _np_ptr(out)))
    return out


def stats_dict(counters, prologue_ms=0.0, integrate_ms=0.0, epilogue_ms=0.0):
    c = [int(x) for x in counters]
    clk = c[STAT_CLK_CYCLES] / c[STAT_CLK_TICKS] * 100.0 if c[STAT_CLK_TICKS] else 0.0   # ticks are 100 MHz
    return dict(rays=c[STAT_RAYS], steps=c[STAT_STEPS], rhs_evals=c[STAT_RHS_EVALS], escaped=c[STAT_ESCAPED],
                captured=c[STAT_CAPTURED], invalid=c[STAT_INVALID], wave_iters=c[STAT_WAVE_ITERS],
                waves=c[STAT_WAVES], clock_mhz=clk, bg_tiles_lds=c[STAT_BG_TILES_LDS],
                bg_tiles_global=c[STAT_BG_TILES_GLOBAL],
                prologue_ms=prologue_ms, integrate_ms=integrate_ms, epilogue_ms=epilogue_ms)


def render_dev(cam, metric, opts, d_bg=0, bg_channels=3, d_fa=0, d_w=0, d_status=0, d_steps=0, d_rgb=0, d_rgba=0,
               d_stats=0):
    """Device-pointer frame render (lt_render_dev); pointers are integers (tensor.data_ptr()), 0 = NULL.
    Asynchronous on opts.stream."""
    p = lambda x: C.c_void_p(x) if x else None
    _check(load().lt_render_dev(C.byref(cam), C.byref(metric), C.byref(opts), p(d_bg), bg_channels, p(d_fa), p(d_w),
                                p(d_status), p(d_steps), p(d_rgb), p(d_rgba), p(d_stats)))


def scatter_rows_dev(d_part, d_full, height, width, elem_bytes, row_block, n_parts, part, stream=0):
    _check(load().lt_scatter_rows_dev(C.c_void_p(d_part), C.c_void_p(d_full), height, width, elem_bytes, row_block,
                                      n_parts, part, C.c_void_p(stream) if stream else None))


def scatter_rows_indexed_dev(d_rows, d_full, d_row_index, n_rows, height, row_bytes, stream=0):
    """Source row i (device, row_bytes each) -> row d_row_index[i] (device int64) of the full frame; one launch."""
    _check(load().lt_scatter_rows_indexed_dev(C.c_void_p(d_rows), C.c_void_p(d_full), C.c_void_p(d_row_index), n_rows, height,
                                              row_bytes, C.c_void_p(stream) if stream else None))


def timing_collect():
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    n = C.c_int32()
    _check(load().lt_timing_collect(C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
    return dict(prologue_ms=a.value, integrate_ms=b.value, epilogue_ms=c.value, calls=n.value)


def release_stream(stream_ptr):
    """Free the library's buffers of (current device, stream) -- before the stream is destroyed."""
    _check(load().lt_release_stream(C.c_void_p(stream_ptr) if stream_ptr else None))


def shutdown():
    if _lib is not None:
        _lib.lt_shutdown()
        with _pinned_lock:
            blocks = [addr for lst in _pinned_free.values() for addr in lst]
            _pinned_free.clear()
            globals()['_pinned_pooled'] = 0
        for addr in blocks:
            _lib.lt_host_free(C.c_void_p(addr))
