"""Device context: one MI355X + one HIP stream (include/jpgpu.h: jpgpu_create)."""
import ctypes as C

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

    def set_host_threads(self, threads: int):
        """Host threads upload() may use (0 = default: min(hardware threads, 32) or JPGPU_HOST_THREADS)."""
        raise_for_status(_lib.jpgpu_set_host_threads(self._h, threads), b"jpgpu_set_host_threads failed")

    def last_error(self) -> str:
        return _lib.jpgpu_last_error(self._h).decode("utf-8", "replace")

    def close(self):
        if self._h:
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
