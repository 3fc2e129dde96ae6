"""Batch JpegOptimizer on the GPU (ref: src/JpegLibrary/JpegOptimizer.cs): for every file

    optimizer = JpegOptimizer(); optimizer.SetInput(bytes); optimizer.Scan(); optimizer.SetOutput(buffer); optimizer.Optimize(strip)

i.e. the Huffman tables are rebuilt from the file's own symbol statistics and the scan is re-written symbol by symbol
(no coefficients are reconstructed).  Single-scan baseline files; see include/jpgpu.h section (5).
"""
import ctypes as C

import numpy as np

from . import _capi
from .context import Context, default_context
from .errors import raise_for_status

_lib = _capi.lib


class OptimizeBatch:
    def __init__(self, ctx: Context = None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        raise_for_status(_lib.jpgpu_optimizer_create(self.ctx._h, C.byref(self._h)), b"jpgpu_optimizer_create failed")
        self._n = 0
        self._keep = None

    def _check(self, rc):
        raise_for_status(rc, _lib.jpgpu_last_error(self.ctx._h))

    def set_most_optimal_coding(self, on=True):
        """JpegOptimizer.MostOptimalCoding"""
        self._check(_lib.jpgpu_optimizer_set_most_optimal_coding(self._h, 1 if on else 0))
        return self

    def upload(self, files, strip=True):
        n = len(files)
        ptrs = (C.c_void_p * n)()
        lens = (C.c_size_t * n)()
        keep = []
        for i, f in enumerate(files):
            buf = np.frombuffer(bytes(f), dtype=np.uint8)
            keep.append(buf)
            ptrs[i] = buf.ctypes.data if buf.size else None
            lens[i] = buf.size
        self._keep = keep
        self._check(_lib.jpgpu_optimizer_upload(self._h, ptrs, lens, n, 1 if strip else 0))
        self._n = n
        return self

    def run(self):
        self._check(_lib.jpgpu_optimizer_run(self._h))
        return self

    def __len__(self):
        return self._n

    def result(self, i):
        res = _capi.ImageResult()
        size = C.c_size_t()
        self._check(_lib.jpgpu_optimizer_result(self._h, i, C.byref(res), C.byref(size)))
        return res, size.value

    def output(self, i) -> bytes:
        """The bytes Optimize() wrote for file i; raises the reference's exception class when the file failed."""
        res, size = self.result(i)
        raise_for_status(res.status, _lib.jpgpu_last_error(self.ctx._h))
        out = np.empty(size, np.uint8)
        self._check(_lib.jpgpu_optimizer_download(self._h, i, out.ctypes.data, out.size))
        return out.tobytes()

    def statistics(self, i):
        """[(table_class, identifier, counts[256])] as Scan() collected them, in builder-creation order."""
        tables = []
        for t in range(8):
            cls = C.c_uint8()
            ident = C.c_uint8()
            counts = np.zeros(256, np.uint32)
            if _lib.jpgpu_optimizer_statistics(self._h, i, t, C.byref(cls), C.byref(ident), counts.ctypes.data) != 0:
                break
            tables.append((cls.value, ident.value, counts))
        return tables

    def last_ms(self) -> float:
        ms = C.c_float()
        _lib.jpgpu_optimizer_last_ms(self._h, C.byref(ms))
        return ms.value

    def close(self):
        if self._h:
            _lib.jpgpu_optimizer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class JpegOptimizer:
    """The reference's class surface (src/JpegLibrary/JpegOptimizer.cs:16-70, 523-548) over a one-file batch:

        optimizer = JpegOptimizer(); optimizer.SetInput(bytes); optimizer.Scan(); optimizer.SetOutput(buffer); optimizer.Optimize(strip)

    `buffer` is anything with `write(bytes)` (an IBufferWriter<byte> stand-in, e.g. io.BytesIO) or a bytearray."""

    def __init__(self, ctx: Context = None):
        self._ctx = ctx
        self._input = None
        self._output = None
        self._scanned = False
        self.MostOptimalCoding = False

    def SetInput(self, data):
        self._input = bytes(data)
        self._scanned = False

    def Scan(self):
        if not self._input:
            from .errors import InvalidOperationException
            raise InvalidOperationException("Input buffer is not specified.")
        self._scanned = True  # the symbol pass runs with Optimize(): both need the device, and Scan() alone has no observable result

    def SetOutput(self, output):
        if output is None:
            raise TypeError("output")
        self._output = output

    def Optimize(self, strip=True):
        from .errors import InvalidOperationException
        if not self._scanned or self._output is None:
            raise InvalidOperationException("Operation is not valid due to the current state of the object.")
        data = optimize_batch([self._input], strip, self._ctx, self.MostOptimalCoding)[0]
        if isinstance(self._output, bytearray):
            self._output += data
        else:
            self._output.write(data)


def optimize_batch(files, strip=True, ctx=None, most_optimal=False):
    """One-call helper: the optimized bytes of every file (raises on the first failing file)."""
    b = OptimizeBatch(ctx).set_most_optimal_coding(most_optimal).upload(files, strip).run()
    try:
        return [b.output(i) for i in range(len(b))]
    finally:
        b.close()


def build_optimal_huffman_table(counts, most_optimal=False):
    """JpegHuffmanEncodingTableBuilder.Build(most_optimal) (host code of the optimizer path): (bits[16], values[n], code[256], length[256])."""
    f = np.ascontiguousarray(counts, dtype=np.uint32).reshape(256)
    bits = np.zeros(16, np.uint8)
    values = np.zeros(256, np.uint8)
    code = np.zeros(256, np.uint16)
    length = np.zeros(256, np.uint8)
    n = C.c_int()
    rc = _lib.jpgpu_build_optimal_huffman_table(f.ctypes.data, 1 if most_optimal else 0, bits.ctypes.data, values.ctypes.data, C.byref(n),
                                                code.ctypes.data, length.ctypes.data)
    raise_for_status(rc, b"No symbol is recorded.")
    return bits, values[:n.value].copy(), code, length
