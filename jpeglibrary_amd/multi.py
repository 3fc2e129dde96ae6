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
        self._ticket = None
        self._tickets = {}

    @staticmethod
    def _pointers(files, pinned=False):
        n = len(files)
        ptrs = (C.c_void_p * max(1, n))()
        lens = (C.c_size_t * max(1, n))()
        keep = []
        for i, f in enumerate(files):
            if pinned and not (isinstance(f, np.ndarray) and f.dtype == np.uint8 and f.flags.c_contiguous):
                # the device reads page-locked memory where it lies: a bytes object is pageable and a copy made here would be
                # too (same rule as Batch.upload_segments)
                raise ValueError("pinned=True needs contiguous uint8 arrays inside page-locked memory (Context.host_alloc / host_register)")
            a = np.frombuffer(f, dtype=np.uint8) if not isinstance(f, np.ndarray) else np.ascontiguousarray(f)
            keep.append(a)
            ptrs[i] = a.ctypes.data
            lens[i] = a.size
        return ptrs, lens, keep

    def decode(self, files, fmt=FMT_INTERLEAVED_U8):
        """Uploads and decodes every shard concurrently; returns when all devices are done."""
        ptrs, lens, keep = self._pointers(files)
        n = len(files)
        up, dec = C.c_double(), C.c_double()
        raise_for_status(_lib.jpgpu_multi_decode(self._h, ptrs, lens, n, fmt, C.byref(up), C.byref(dec)), _lib.jpgpu_multi_last_error(self._h))
        self._keep = keep
        self._n = n
        self.format = fmt
        self._ticket = None
        self.upload_ms, self.decode_ms = up.value, dec.value
        return self

    def submit(self, files, fmt=FMT_INTERLEAVED_U8, pinned=False) -> int:
        """Uploads every shard into its slot's idle batch and launches the decodes; returns a ticket while the devices work.
        Submitting call k + 1 before wait(k) puts its host parse + H2D beside call k's decode (two calls in flight at most)."""
        ptrs, lens, keep = self._pointers(files, pinned=pinned)
        t = C.c_int(-1)
        raise_for_status(_lib.jpgpu_multi_submit(self._h, ptrs, lens, len(files), fmt, _capi.UPLOAD_PINNED if pinned else 0, C.byref(t)),
                         _lib.jpgpu_multi_last_error(self._h))
        self._tickets[t.value & 1] = (t.value, len(files), fmt, keep)
        self._n, self.format, self._ticket = len(files), fmt, t.value
        return t.value

    def wait(self, ticket: int):
        up, dec = C.c_double(), C.c_double()
        raise_for_status(_lib.jpgpu_multi_wait(self._h, ticket, C.byref(up), C.byref(dec)), _lib.jpgpu_multi_last_error(self._h))
        self.upload_ms, self.decode_ms = up.value, dec.value
        return self

    def shard_of(self, ticket: int, slot: int) -> Batch:
        """The batch call `ticket` used on `slot` (valid until the second submit after it)."""
        fmt = self._tickets[ticket & 1][2]
        return Batch._borrowed(_lib.jpgpu_multi_batch_of(self._h, ticket, slot), _lib.jpgpu_multi_context(self._h, slot), fmt)

    def result_of(self, ticket: int, i: int):
        world = len(self.devices)
        return self.shard_of(ticket, i % world).result(i // world)

    def output_of(self, ticket: int, i: int):
        world = len(self.devices)
        return self.shard_of(ticket, i % world).output(i // world)

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
