"""Device context: one MI355X + one HIP stream (include/jpgpu.h: jpgpu_create)."""
import ctypes as C

import numpy as np

from . import _capi
from .errors import raise_for_status

_lib = _capi.lib


def device_count() -> int:
    return _lib.jpgpu_device_count()


class Context:
    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = _lib.jpgpu_create(device, C.byref(self._h))
        raise_for_status(rc, _lib.jpgpu_last_error(None))
        self.device = device
        self._pinned = {}

    def set_host_threads(self, threads: int):
        """Host threads upload() may use (0 = default: min(CPUs granted to the process by affinity mask and cgroup quota,
        16), or JPGPU_HOST_THREADS)."""
        raise_for_status(_lib.jpgpu_set_host_threads(self._h, threads), b"jpgpu_set_host_threads failed")

    def host_alloc(self, nbytes: int) -> "np.ndarray":
        """uint8 array over page-locked host memory (jpgpu_host_alloc): files read into it can be uploaded with
        pinned=True, i.e. DMA'd to HBM from where they lie.  Freed by host_free(array) or with the context."""
        p = C.c_void_p()
        raise_for_status(_lib.jpgpu_host_alloc(self._h, nbytes, C.byref(p)), _lib.jpgpu_last_error(self._h))
        buf = (C.c_uint8 * max(1, nbytes)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data, None)
        if p is not None:
            _lib.jpgpu_host_free(self._h, C.c_void_p(p))

    def host_register(self, arr):
        """Page-locks the caller's own array (jpgpu_host_register); undo with host_unregister."""
        raise_for_status(_lib.jpgpu_host_register(self._h, C.c_void_p(arr.ctypes.data), arr.nbytes), _lib.jpgpu_last_error(self._h))

    def host_unregister(self, arr):
        raise_for_status(_lib.jpgpu_host_unregister(self._h, C.c_void_p(arr.ctypes.data)), _lib.jpgpu_last_error(self._h))

    def last_error(self) -> str:
        return _lib.jpgpu_last_error(self._h).decode("utf-8", "replace")

    def close(self):
        if self._h:
            for p in list(getattr(self, "_pinned", {}).values()):
                _lib.jpgpu_host_free(self._h, C.c_void_p(p))
            self._pinned = {}
            _lib.jpgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_context(device: int = 0) -> Context:
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]
