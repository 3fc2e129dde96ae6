"""ctypes binding of tools/libjpegsynth.so: deterministic synthetic baseline JPEGs (SURVEY.md 8d recipe).

Bench/test input generator only -- not part of the decode path and not an oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjpegsynth.so")
SUBSAMPLING = {"444": 0, "4:4:4": 0, "422": 1, "4:2:2": 1, "420": 2, "4:2:0": 2, "gray": 3, "400": 3}


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("subsampling", C.c_int), ("quality", C.c_int),
                ("restart_interval", C.c_int), ("seed", C.c_uint64), ("noninterleaved", C.c_int),
                ("samp", (C.c_int * 2) * 3), ("progressive", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "jpegsynth.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def _get():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.jsynth_encode.restype = C.c_long
        L.jsynth_encode.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t]
        L.jsynth_encode_batch.restype = C.c_int
        L.jsynth_encode_batch.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_long),
                                          C.c_int]
        _lib = L
    return _lib


def _cap(width, height):
    return int(width * height * 3 + (1 << 16))


def encode(width, height, subsampling="420", quality=75, restart_interval=0, seed=0, noninterleaved=False, sampling=None,
           progressive=False) -> bytes:
    """noninterleaved=True: three single-component scans instead of one interleaved scan, blocks in the order the reference
    reads them (h x v per frame MCU: T.81's order only for 4:4:4); noninterleaved=2: T.81's own order.
    sampling=((Hy, Vy), (Hcb, Vcb), (Hcr, Vcr)), factors 1..4 with whole ratios to the maximum, overrides `subsampling`.
    progressive=True: SOF2, fixed script (DC with Al = 1 + refinement, two AC bands per component, spectral selection only)."""
    L = _get()
    p = Params(width, height, SUBSAMPLING[str(subsampling)] if sampling is None else 0, quality, restart_interval, seed, int(noninterleaved))
    if sampling is not None:
        for c in range(3):
            p.samp[c][0], p.samp[c][1] = sampling[c]
    p.progressive = int(progressive)
    buf = np.empty(_cap(width, height), dtype=np.uint8)
    n = L.jsynth_encode(C.byref(p), buf.ctypes.data, buf.size)
    if n < 0:
        raise RuntimeError(f"jsynth_encode failed ({n})")
    return buf[:n].tobytes()


def encode_batch(n, width, height, subsampling="420", quality=75, restart_interval=0, seed0=0, nthreads=None,
                 stride=None):
    """Returns (buffer uint8[n*stride], sizes int64[n], stride). Image i uses seed0+i."""
    L = _get()
    nthreads = nthreads or os.cpu_count() or 1
    # ~1-2.5 bpp for the recipe at Q75..Q90 (<= 0.32 B/px); 0.4 B/px + slack is ample
    stride = stride or int(width * height * 0.4 + 65536)
    stride = (stride + 255) & ~255
    arr = (Params * n)()
    for i in range(n):
        arr[i] = Params(width, height, SUBSAMPLING[str(subsampling)], quality, restart_interval, seed0 + i, 0)
    buf = np.empty(n * stride, dtype=np.uint8)
    sizes = (C.c_long * n)()
    rc = L.jsynth_encode_batch(arr, n, buf.ctypes.data, stride, sizes, nthreads)
    sz = np.array(sizes[:], dtype=np.int64)
    if rc != 0:
        raise RuntimeError(f"jsynth_encode_batch failed: sizes min {sz.min()}")
    return buf, sz, stride
