"""ctypes binding of the diagnostic build, light-path-tracer_amd/lib/libltrace_probes.so (include/ltrace_probes.h):
the product library plus the VALU issue-cost / bare-RK4-step / right-hand-side-piece microbenchmarks.  Build it with
`python __graft_entry__.py --probes`.  Not part of the product; nothing in the package imports this."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES_LIB = os.path.join(ROOT, "light-path-tracer_amd", "lib", "libltrace_probes.so")
_dp = C.POINTER(C.c_double)
_lib = None

SIGNATURES = {
    "lt_valu_peak_probe": (C.c_int, [C.c_int, C.c_int, _dp]),
    "lt_valu_issue_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, _dp, _dp]),
    "lt_valu_issue_probe_count": (C.c_int, []),
    "lt_rk4_step_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "lt_piece_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "lt_last_error": (C.c_char_p, []),
}


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(PROBES_LIB):
            sys.path.insert(0, ROOT)
            import __graft_entry__
            __graft_entry__.build_probes()
        lib = C.CDLL(PROBES_LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc):
    if rc:
        raise RuntimeError(f"probe failed ({rc}): {load().lt_last_error().decode()}")


def valu_peak_probe(mode=0, iters=4096):
    t = C.c_double()
    _check(load().lt_valu_peak_probe(mode, iters, C.byref(t)))
    return t.value


def valu_issue_probe(index, waves_per_simd=8, iters=2000, constant_data=False):
    """-> (mnemonic, ns per wave-instruction per SIMD, shader clock in MHz during the loop)"""
    name = C.create_string_buffer(64)
    t, clk = C.c_double(), C.c_double()
    _check(load().lt_valu_issue_probe(index, waves_per_simd, iters, int(constant_data), name, 64, C.byref(t), C.byref(clk)))
    return name.value.decode(), t.value, clk.value


def valu_issue_probe_count():
    return int(load().lt_valu_issue_probe_count())


def rk4_step_probe(precision=32, waves_per_simd=8, iters=20000):
    """-> (shader cycles one SIMD spends per wave-step of the bare Kerr RK4 step, clock MHz)"""
    c, clk = C.c_double(), C.c_double()
    _check(load().lt_rk4_step_probe(precision, waves_per_simd, iters, C.byref(c), C.byref(clk)))
    return c.value, clk.value


def piece_probe(piece, waves_per_simd=8, iters=20000):
    c, clk = C.c_double(), C.c_double()
    _check(load().lt_piece_probe(piece, waves_per_simd, iters, C.byref(c), C.byref(clk)))
    return c.value, clk.value
