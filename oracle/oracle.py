"""ctypes front end of oracle/liblt_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package (light-path-tracer_amd/).  See the header of
oracle/lt_oracle.c for what the library restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB32 = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    if force or not (os.path.exists(os.path.join(_HERE, "liblt_oracle.so"))
                     and os.path.exists(os.path.join(_HERE, "liblt_oracle_f32.so"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liblt_oracle.so"))
        L.lto_kerr_rhs.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        L.lto_kerr_rhs.restype = None
        L.lto_kerr_ic.argtypes = [C.c_double] * 6 + [_dp, _dp, _dp]
        L.lto_kerr_ic.restype = C.c_int
        L.lto_trace_batch_schw.argtypes = [C.c_double, C.c_double, _dp, C.c_int64, C.c_double, C.c_double,
                                           _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int8),
                                           C.POINTER(C.c_uint32)]
        L.lto_trace_batch_schw.restype = C.c_int
        L.lto_trace_batch_kerr.argtypes = [C.c_double, C.c_double, C.c_double, _dp, _dp, C.c_double,
                                           C.c_double, C.POINTER(C.c_uint8), C.c_int, C.c_int64,
                                           _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int8),
                                           C.POINTER(C.c_uint32)]
        L.lto_trace_batch_kerr.restype = C.c_int
        L.lto_psi_frame.argtypes = [C.c_double, C.c_double, _dp, _dp, _dp, C.POINTER(C.c_int)]
        L.lto_psi_frame.restype = None
        L.lto_pixel_angles.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_double, _fp, _dp, C.POINTER(C.c_uint8)]
        L.lto_pixel_angles.restype = None
        L.lto_lookup.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_int, C.c_int,
                                 _fp, C.POINTER(C.c_uint16), C.POINTER(C.c_int8), C.POINTER(C.c_uint32)]
        L.lto_lookup.restype = C.c_int64
        L.lto_render.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.POINTER(C.c_uint16),
                                 C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _fp]
        L.lto_render.restype = None
        L.lto_rgba8.argtypes = [_fp, C.c_int64, C.c_int, C.POINTER(C.c_uint8)]
        L.lto_rgba8.restype = None
        L.lto_shadow_analytic.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        L.lto_shadow_analytic.restype = None
        L.lto_num_threads.restype = C.c_int
        L.lto_rhs8.argtypes = [C.c_int, C.c_double, C.c_double, _dp, _dp]
        L.lto_rhs8.restype = None
        L.lto_integrate_dense.argtypes = [C.c_int, C.c_double, C.c_double, _dp] + [C.c_double] * 6 + [
            C.c_int64, _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.lto_integrate_dense.restype = C.c_int64
        L.lto_dense_step_log.argtypes = [C.c_int, C.c_double, C.c_double, _dp] + [C.c_double] * 6 + [
            C.c_int64, _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.lto_dense_step_log.restype = C.c_int64
        _LIB = L
    return _LIB


_LIB_PERF = None
PERF_FLAGS = ["-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-std=c11"]


def lib_perf():
    """The same lt_oracle.c compiled for speed on THIS machine (-O3 -march=native, FMA contraction allowed), into a
    temporary directory: the timing build of bench.py's cpu_baseline leg.  Never used for parity --
    the parity build above is -O2 -ffp-contract=off -- and never shipped: -march=native code is only valid on
    the host that compiled it."""
    global _LIB_PERF
    if _LIB_PERF is None:
        import tempfile
        d = tempfile.mkdtemp(prefix="lt_oracle_perf_")
        so = os.path.join(d, "liblt_oracle_perf.so")
        subprocess.check_call(["gcc"] + PERF_FLAGS + ["-o", so, os.path.join(_HERE, "lt_oracle.c"), "-lm"])
        L = C.CDLL(so)
        L.lto_lookup.argtypes = lib().lto_lookup.argtypes
        L.lto_lookup.restype = C.c_int64
        L.lto_num_threads.restype = C.c_int
        L.lto_num_threads.argtypes = []
        _LIB_PERF = L
    return _LIB_PERF


def lib32():
    global _LIB32
    if _LIB32 is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liblt_oracle_f32.so"))
        L.lto_trace_batch_schw_f32.argtypes = [C.c_float, C.c_float, _fp, C.c_int64, C.c_float, C.c_float,
                                               _fp, C.POINTER(C.c_int64), C.POINTER(C.c_int8),
                                               C.POINTER(C.c_uint32)]
        L.lto_trace_batch_kerr_f32.argtypes = [C.c_float, C.c_float, C.c_float, _fp, _fp, C.c_float,
                                               C.c_float, C.POINTER(C.c_uint8), C.c_int, C.c_int64,
                                               _fp, C.POINTER(C.c_int64), C.POINTER(C.c_int8),
                                               C.POINTER(C.c_uint32)]
        _LIB32 = L
    return _LIB32


def num_threads():
    return int(lib().lto_num_threads())


def set_num_threads(n, perf_build=False):
    """OpenMP threads of the batch loops (libgomp may have been initialised long before, e.g. by torch)."""
    L = lib_perf() if perf_build else lib()
    L.lto_set_num_threads.argtypes = [C.c_int]
    L.lto_set_num_threads.restype = None
    L.lto_set_num_threads(int(n))


def kerr_rhs(state5, p_t, p_phi, M, a, r_plus):
    s = np.ascontiguousarray(state5, dtype=np.float64)
    out = np.empty(5)
    lib().lto_kerr_rhs(_ptr(s, _dp), p_t, p_phi, M, a, r_plus, _ptr(out, _dp))
    return out


def kerr_ic(M, a, r_obs, alpha, theta, theta_obs):
    st = np.zeros(5)
    pt = C.c_double()
    pp = C.c_double()
    ok = lib().lto_kerr_ic(M, a, r_obs, alpha, theta, theta_obs, _ptr(st, _dp), C.byref(pt), C.byref(pp))
    return bool(ok), st, pt.value, pp.value


def trace_batch_schw(M, r_obs, alphas, phi_max=50.0, h_max=0.05, f32=False):
    """-> (final_alpha[n] (NaN unless escaped), n_half[n] i64, status[n] i8, rhs_evals[n] u32)"""
    dt = np.float32 if f32 else np.float64
    al = np.ascontiguousarray(alphas, dtype=dt)
    n = al.size
    fa = np.full(n, np.nan, dtype=dt)
    w = np.zeros(n, dtype=np.int64)
    st = np.zeros(n, dtype=np.int8)
    ev = np.zeros(n, dtype=np.uint32)
    pt = _fp if f32 else _dp
    fn = lib32().lto_trace_batch_schw_f32 if f32 else lib().lto_trace_batch_schw
    fn(M, r_obs, _ptr(al, pt), n, phi_max, h_max, _ptr(fa, pt), _ptr(w, C.POINTER(C.c_int64)),
       _ptr(st, C.POINTER(C.c_int8)), _ptr(ev, C.POINTER(C.c_uint32)))
    return fa, w, st, ev


def trace_batch_kerr(M, a, r_obs, alphas, thetas, theta_obs=np.pi / 2, lambda_max=None,
                     axis_refines=None, integrator="dp45", f32=False):
    dt = np.float32 if f32 else np.float64
    al = np.ascontiguousarray(alphas, dtype=dt)
    th = np.ascontiguousarray(thetas, dtype=dt)
    n = al.size
    if lambda_max is None:
        lambda_max = max(5000.0, 6.0 * r_obs)
    ar = (np.zeros(n, dtype=np.uint8) if axis_refines is None
          else np.ascontiguousarray(axis_refines).astype(np.uint8))
    fa = np.full(n, np.nan, dtype=dt)
    w = np.zeros(n, dtype=np.int64)
    st = np.zeros(n, dtype=np.int8)
    ev = np.zeros(n, dtype=np.uint32)
    pt = _fp if f32 else _dp
    fn = lib32().lto_trace_batch_kerr_f32 if f32 else lib().lto_trace_batch_kerr
    rc = fn(M, a, r_obs, _ptr(al, pt), _ptr(th, pt), theta_obs, lambda_max,
            _ptr(ar, C.POINTER(C.c_uint8)), 0 if integrator == "dp45" else 1, n,
            _ptr(fa, pt), _ptr(w, C.POINTER(C.c_int64)), _ptr(st, C.POINTER(C.c_int8)),
            _ptr(ev, C.POINTER(C.c_uint32)))
    if rc != 0:
        raise ValueError("oracle: |a| exceeds M")
    return fa, w, st, ev


def psi_frame(psi):
    d, ex, ey = np.zeros(3), np.zeros(3), np.zeros(3)
    fr = C.c_int()
    lib().lto_psi_frame(psi[0], psi[1], _ptr(d, _dp), _ptr(ex, _dp), _ptr(ey, _dp), C.byref(fr))
    return d, ex, ey, bool(fr.value)


def pixel_angles(H, W, hfov, vfov, psi=(0.0, 0.0), axis_refine_frac=0.07):
    """-> alpha (H,W) f32 [image_lens.py:133-152], theta (H,W) f64 [:193-208], axis cols (W,) bool"""
    al = np.empty((H, W), dtype=np.float32)
    th = np.empty((H, W), dtype=np.float64)
    cols = np.zeros(W, dtype=np.uint8)
    lib().lto_pixel_angles(H, W, hfov, vfov, psi[0], psi[1], axis_refine_frac,
                           _ptr(al, _fp), _ptr(th, _dp), _ptr(cols, C.POINTER(C.c_uint8)))
    return al, th, cols.astype(bool)


def lookup(kind, M, a, r_obs, H, W, hfov, vfov, psi=(0.0, 0.0), theta_obs=np.pi / 2,
           integrator="dp45", tb_symmetry=False, axis_refine_frac=0.07, perf_build=False):
    """Per-pixel trace of a whole frame -> dict(fa f32, winding u16, status i8, evals u32, traced).
    perf_build: run the -O3 -march=native build (timing only, see lib_perf)."""
    fa = np.empty((H, W), dtype=np.float32)
    w = np.empty((H, W), dtype=np.uint16)
    st = np.empty((H, W), dtype=np.int8)
    ev = np.empty((H, W), dtype=np.uint32)
    traced = (lib_perf() if perf_build else lib()).lto_lookup(0 if kind == "schwarzschild" else 1, M, a, r_obs, theta_obs, H, W,
                              hfov, vfov, psi[0], psi[1], axis_refine_frac,
                              0 if integrator == "dp45" else 1, int(bool(tb_symmetry)),
                              _ptr(fa, _fp), _ptr(w, C.POINTER(C.c_uint16)),
                              _ptr(st, C.POINTER(C.c_int8)), _ptr(ev, C.POINTER(C.c_uint32)))
    return dict(fa=fa, winding=w, status=st, evals=ev, traced=int(traced))


def render(source, fa, winding, hfov, vfov, psi=(0.0, 0.0), loop_around=False):
    src = np.ascontiguousarray(source, dtype=np.float32)
    H, W = src.shape[:2]
    Cn = 1 if src.ndim == 2 else src.shape[2]
    out = np.zeros_like(src)
    fa = np.ascontiguousarray(fa, dtype=np.float32)
    wd = None if winding is None else np.ascontiguousarray(winding, dtype=np.uint16)
    lib().lto_render(_ptr(src, _fp), H, W, Cn, _ptr(fa, _fp), _ptr(wd, C.POINTER(C.c_uint16)),
                     hfov, vfov, psi[0], psi[1], int(bool(loop_around)), _ptr(out, _fp))
    return out


def rgba8(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    H, W = rgb.shape[:2]
    Cn = 1 if rgb.ndim == 2 else rgb.shape[2]
    out = np.empty((H, W, 4), dtype=np.uint8)
    lib().lto_rgba8(_ptr(rgb, _fp), H * W, Cn, _ptr(out, C.POINTER(C.c_uint8)))
    return out


def shadow_analytic(width, height, fov, alpha_crit):
    img = np.empty((width, height), dtype=np.float64)
    lib().lto_shadow_analytic(width, height, fov, alpha_crit, _ptr(img, _dp))
    return img


def rhs8(kind, M, a, state):
    """8-D right-hand side (metrics.py:763-790 kind 0, :946-1029 kind 1) of one state."""
    st = np.ascontiguousarray(state, dtype=np.float64)
    out = np.empty(8, dtype=np.float64)
    lib().lto_rhs8(int(kind), float(M), float(a), _ptr(st, _dp), _ptr(out, _dp))
    return out


def integrate_dense(kind, M, a, state0, lambda_max=1000.0, r_stop_inner=None, r_stop_outer=None,
                    rtol=1e-8, atol=1e-10, max_step=1.0, max_points=4096):
    """geodesic_tracer.integrate_geodesic (geodesic_tracer.py:22-71) for one 8-D initial state.
    -> (t (n,), y (8, n), status, nfev); status 1 capture event, 2 escape event, 0 lambda_max, -1 failed."""
    s0 = np.ascontiguousarray(state0, dtype=np.float64)
    t = np.empty(max_points, dtype=np.float64)
    y = np.empty((8, max_points), dtype=np.float64)
    st, nfev = C.c_int(0), C.c_int64(0)
    n = lib().lto_integrate_dense(int(kind), float(M), float(a), _ptr(s0, _dp), float(lambda_max), float(r_stop_inner),
                                  float(r_stop_outer), float(rtol), float(atol), float(max_step), int(max_points),
                                  _ptr(t, _dp), _ptr(y, _dp), C.byref(st), C.byref(nfev))
    m = min(int(n), max_points)
    return t[:m].copy(), y[:, :m].copy(), int(st.value), int(nfev.value)


def dense_predict_length(kind, M, a, state0, lambda_max=1000.0, r_stop_inner=None, r_stop_outer=None,
                         rtol=1e-8, atol=1e-10, max_step=1.0, rtol_loose=1e-4):
    """CPU twin of the dense kernel's length predictor (csrc/lt_dense.hpp, k_dense_predict): the same integrator at a loose
    tolerance with a free step size; every accepted step of length h and error norm err stands for
    max(h / max_step, err^(1/5) / (0.9 (rtol / rtol_loose)^(1/5))) steps of the real pass (the step holding the terminal
    event counted up to the event).  -> (predicted attempts, attempts of the loose pass)."""
    s0 = np.ascontiguousarray(state0, dtype=np.float64)
    cap = 4096
    h, e = np.empty(cap), np.empty(cap)
    st, nfev = C.c_int(0), C.c_int64(0)
    n = lib().lto_dense_step_log(int(kind), float(M), float(a), _ptr(s0, _dp), float(lambda_max), float(r_stop_inner),
                                 float(r_stop_outer), float(rtol_loose), float(rtol_loose * atol / rtol), 1e300, cap,
                                 _ptr(h, _dp), _ptr(e, _dp), C.byref(st), C.byref(nfev))
    n = min(int(n), cap)
    kappa = 0.9 * (rtol / rtol_loose) ** 0.2
    pred = np.maximum(h[:n] / max_step, np.maximum(e[:n], 0.0) ** 0.2 / kappa).sum()
    return float(pred), (int(nfev.value) - 2) // 6
