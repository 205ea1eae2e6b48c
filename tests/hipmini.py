"""The few HIP runtime calls the GPU tests need besides the library under test (streams, raw device buffers),
through ctypes on the SAME HIP runtime instance libltrace_hip.so is linked against: the symbols are looked up
through the library's own handle, so the dynamic linker resolves them in its dependency (libamdhip64.so.7 of
/opt/rocm).  (torch bundles its own HIP runtime; once it is loaded, a dlopen of "libamdhip64.so" by name finds
that copy, and a second runtime instance in one process sees no device -- so in-process GPU tests go neither
through torch.cuda nor through a HIP library opened by name.)"""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        import ltrace
        _hip = C.CDLL(ltrace.LIB_PATH)       # same handle the binding uses; hip* resolve in its dependencies
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip.hipStreamDestroy.argtypes = [C.c_void_p]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


class DeviceArray:
    """A raw device allocation shaped like a numpy array (no arithmetic: only .ptr and .get())."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _ok(hip().hipMalloc(C.byref(p), max(self.nbytes, 1)), "hipMalloc")
        self.ptr = p.value

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _ok(hip().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2), "hipMemcpy D2H")   # 2 = hipMemcpyDeviceToHost
        return out

    def __del__(self):
        if getattr(self, "ptr", None) and _hip is not None:
            _hip.hipFree(self.ptr)


class Stream:
    def __init__(self, non_blocking=True):
        p = C.c_void_p()
        _ok(hip().hipStreamCreateWithFlags(C.byref(p), 1 if non_blocking else 0), "hipStreamCreateWithFlags")
        self.ptr = p.value

    def synchronize(self):
        _ok(hip().hipStreamSynchronize(self.ptr), "hipStreamSynchronize")

    def __del__(self):
        if getattr(self, "ptr", None) and _hip is not None:
            _hip.hipStreamDestroy(self.ptr)


def device_synchronize():
    _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
