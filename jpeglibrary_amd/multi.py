"""Several devices behind one call: jpgpu_multi_* -- one context, one batch and one host thread per
device slot, image i on slot i mod G, no exchange between the devices.  The caller-side equivalent in the reference is "one
JpegDecoder per thread" over a shared file list."""
import ctypes as C

import numpy as np

from . import _capi
from .batch import FMT_INTERLEAVED_U8, Batch
from .errors import raise_for_status

_lib = _capi.lib


class MultiDecoder:
    def __init__(self, devices):
        devices = list(devices)
        arr = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        raise_for_status(_lib.jpgpu_multi_create(arr, len(devices), C.byref(self._h)), _lib.jpgpu_multi_last_error(None))
        self.devices = devices
        self.format = FMT_INTERLEAVED_U8
        self.upload_ms = self.decode_ms = 0.0
        self._n = 0
        self._keep = None

    def decode(self, files, fmt=FMT_INTERLEAVED_U8):
        """Uploads and decodes every shard concurrently; returns when all devices are done."""
        n = len(files)
        ptrs = (C.c_void_p * n)()
        lens = (C.c_size_t * n)()
        keep = []
        for i, f in enumerate(files):
            a = np.frombuffer(f, dtype=np.uint8) if not isinstance(f, np.ndarray) else np.ascontiguousarray(f)
            keep.append(a)
            ptrs[i] = a.ctypes.data
            lens[i] = a.size
        up, dec = C.c_double(), C.c_double()
        raise_for_status(_lib.jpgpu_multi_decode(self._h, ptrs, lens, n, fmt, C.byref(up), C.byref(dec)), _lib.jpgpu_multi_last_error(self._h))
        self._keep = keep
        self._n = n
        self.format = fmt
        self.upload_ms, self.decode_ms = up.value, dec.value
        return self

    def __len__(self):
        return self._n

    def locate(self, i):
        slot, local = C.c_int(), C.c_int()
        raise_for_status(_lib.jpgpu_multi_locate(self._h, i, C.byref(slot), C.byref(local)), b"image index out of range")
        return slot.value, local.value

    def shard(self, slot) -> Batch:
        """The slot's batch (borrowed): result(i) / output(i) / image_info(i) at the local index locate() gives."""
        return Batch._borrowed(_lib.jpgpu_multi_batch(self._h, slot), _lib.jpgpu_multi_context(self._h, slot), self.format)

    def result(self, i):
        slot, local = self.locate(i)
        return self.shard(slot).result(local)

    def output(self, i):
        slot, local = self.locate(i)
        return self.shard(slot).output(local)

    def close(self):
        if self._h:
            _lib.jpgpu_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
