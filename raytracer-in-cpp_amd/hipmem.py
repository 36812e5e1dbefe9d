"""Minimal device-memory helpers over the SAME HIP runtime librt_mi355x.so links against (ctypes on libamdhip64).

torch bundles its own libamdhip64; mixing the two runtimes in one process only works when torch initialises first
(bench.py does).  Code that does not need torch.distributed uses these helpers instead."""
import ctypes as C
import os

import numpy as np

_hip = None


def _lib():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64 not found")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


class DeviceBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = C.c_void_p()
        if _lib().hipMalloc(C.byref(self.ptr), self.nbytes) != 0:
            raise RuntimeError(f"hipMalloc({nbytes}) failed")
        _lib().hipMemset(self.ptr, 0, self.nbytes)

    @property
    def address(self):
        return self.ptr.value

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        if _lib().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2) != 0:   # hipMemcpyDeviceToHost
            raise RuntimeError("hipMemcpy D2H failed")
        return out

    def free(self):
        if self.ptr:
            _lib().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synchronize():
    _lib().hipDeviceSynchronize()
