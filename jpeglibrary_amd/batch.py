"""Device-resident batch decode (include/jpgpu.h level 1: jpgpu_batch_*).

Replaces, for a set of files at once, the canonical reference sequence
    new JpegDecoder(); SetInput; Identify; SetOutputWriter(JpegBufferOutputWriter8Bit); Decode()
(ref: apps/JpegDecode/DecodeAction.cs:26-56, tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:51-73).
"""
import ctypes as C

import numpy as np

from . import _capi
from .context import Context, default_context
from .errors import raise_for_status

_lib = _capi.lib
FMT_INTERLEAVED_U8, FMT_PLANAR_U8, FMT_PLANAR_I16 = _capi.FMT_INTERLEAVED_U8, _capi.FMT_PLANAR_U8, _capi.FMT_PLANAR_I16
FMT_RGB_U8, FMT_RGBA_U8, FMT_EXTENDED_U16 = _capi.FMT_RGB_U8, _capi.FMT_RGBA_U8, _capi.FMT_EXTENDED_U16


class _BorrowedContext:
    def __init__(self, handle):
        self._h = C.c_void_p(handle)


class Batch:
    def __init__(self, ctx: Context = None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        raise_for_status(_lib.jpgpu_batch_create(self.ctx._h, C.byref(self._h)), b"jpgpu_batch_create failed")
        self.format = FMT_INTERLEAVED_U8
        self._keep = None

    def _check(self, rc):
        raise_for_status(rc, _lib.jpgpu_last_error(self.ctx._h))

    def upload(self, files, fmt=FMT_INTERLEAVED_U8):
        """files: list of bytes-like objects or of (numpy uint8 array) views. Host parse + H2D."""
        n = len(files)
        ptrs = (C.c_void_p * n)()
        lens = (C.c_size_t * n)()
        keep = []
        for i, f in enumerate(files):
            a = np.frombuffer(f, dtype=np.uint8) if not isinstance(f, np.ndarray) else f
            if not a.flags["C_CONTIGUOUS"]:
                a = np.ascontiguousarray(a)
            keep.append(a)
            ptrs[i] = a.ctypes.data
            lens[i] = a.size
        self._check(_lib.jpgpu_batch_upload(self._h, ptrs, lens, n, fmt))
        self.format = fmt
        return self

    def upload_segments(self, files, fmt=FMT_INTERLEAVED_U8, pinned=False, arena=False):
        """files: list of files, each a LIST of bytes-like / uint8-array segments (the ReadOnlySequence<byte> a reference
        caller hands to SetInput, read in place).  pinned=True: every segment lies in page-locked memory
        (Context.host_alloc / host_register) and is DMA'd to HBM from where it lies.  arena=True (JPGPU_UPLOAD_PINNED_ARENA): all
        of them are views into ONE page-locked array, every file contiguous -- the span travels as a few large DMAs."""
        keep, segs, per = [], [], []
        for f in files:
            parts = f if isinstance(f, (list, tuple)) else [f]
            per.append(len(parts))
            for part in parts:
                a = np.frombuffer(part, dtype=np.uint8) if not isinstance(part, np.ndarray) else part
                if not a.flags["C_CONTIGUOUS"]:
                    if pinned or arena:
                        raise ValueError("a pinned segment must be contiguous (a copy would leave the page-locked memory)")
                    a = np.ascontiguousarray(a)
                keep.append(a)
                segs.append((a.ctypes.data if a.size else None, a.size))
        arr = (_capi.Segment * max(1, len(segs)))(*[_capi.Segment(d, n) for d, n in segs])
        cnt = (C.c_int * max(1, len(per)))(*per)
        self._check(_lib.jpgpu_batch_upload_segments(self._h, arr, cnt, len(per), fmt, (_capi.UPLOAD_PINNED_ARENA if arena else 0) | (_capi.UPLOAD_PINNED if pinned else 0)))
        self.format = fmt
        return self

    def upload_frames(self, frames, quant_tables, fmt=FMT_INTERLEAVED_U8):
        """Coefficient hand-off (progressive images, BASELINE config 5): frames = list of dicts
        {width, height, precision, components: [(id, h, v, tq), ...]}, quant_tables = uint16[n][4][64] (zig-zag order).
        Follow with set_coefficients(i, blocks in MCU scan order) and run_idct()."""
        n = len(frames)
        arr = (_capi.Frame * n)()
        for i, f in enumerate(frames):
            arr[i].width, arr[i].height, arr[i].precision = f["width"], f["height"], f["precision"]
            arr[i].num_components = len(f["components"])
            arr[i].sof = f.get("sof", 0xC2)
            for c, (cid, h, v, tq) in enumerate(f["components"]):
                arr[i].comp[c] = _capi.FrameComponent(cid, h, v, tq)
        qt = np.ascontiguousarray(quant_tables, dtype=np.uint16).reshape(n, 4, 64)
        self._check(_lib.jpgpu_batch_upload_frames(self._h, arr, qt.ctypes.data, n, fmt))
        self.format = fmt
        return self

    def decode(self):
        self._check(_lib.jpgpu_batch_decode(self._h))
        return self

    def run_entropy(self):
        self._check(_lib.jpgpu_batch_run_entropy(self._h))
        return self

    def run_idct(self):
        self._check(_lib.jpgpu_batch_run_idct(self._h))
        return self

    def sync(self):
        self._check(_lib.jpgpu_batch_sync(self._h))
        return self

    def __len__(self):
        return _lib.jpgpu_batch_size(self._h)

    def image_info(self, i) -> _capi.ImageInfo:
        info = _capi.ImageInfo()
        self._check(_lib.jpgpu_batch_image_info(self._h, i, C.byref(info)))
        return info

    def result(self, i) -> _capi.ImageResult:
        res = _capi.ImageResult()
        self._check(_lib.jpgpu_batch_result(self._h, i, C.byref(res)))
        return res

    def stage_ms(self):
        ms = (C.c_float * 4)()
        self._check(_lib.jpgpu_batch_stage_ms(self._h, ms))
        return {"marker_index": ms[0], "huffman": ms[1], "idct": ms[2], "total": ms[3]}

    def ingest_stats(self):
        """What the last upload() did: files planned from their headers alone vs full host walks, and where the time went."""
        st = _capi.IngestStats()
        self._check(_lib.jpgpu_batch_ingest_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _capi.IngestStats._fields_}

    def progressive_fallbacks(self):
        """Times the single-launch progressive path timed out and the step was re-issued level by level."""
        return _lib.jpgpu_batch_progressive_fallbacks(self._h)

    def subseq_rounds(self):
        """Synchronisation rounds the DRI = 0 subsequence decoder needed in the last decode (0 = not used)."""
        return _lib.jpgpu_batch_subseq_rounds(self._h)

    def set_partial_flush(self, on=True):
        """Failing progressive files: reproduce the reference's partial flush (default) or leave those frames' outputs unspecified."""
        self._check(_lib.jpgpu_batch_set_partial_flush(self._h, 1 if on else 0))
        return self

    def progressive_replays(self):
        """Times a partial-flush replay was issued for this batch."""
        return _lib.jpgpu_batch_progressive_replays(self._h)

    def marker_fallbacks(self):
        """Waits behind which a group of the one-pass marker index had run out of patience and counted its predecessors itself (expected: 0)."""
        return _lib.jpgpu_batch_marker_fallbacks(self._h)

    def subseq_fallbacks(self):
        """Times the enqueued K2S rounds did not converge and the step was issued again with host-checked rounds."""
        return _lib.jpgpu_batch_subseq_fallbacks(self._h)

    def totals(self):
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.jpgpu_batch_totals(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return {"compressed_bytes": a.value, "blocks": b.value, "pixels": c.value, "output_bytes": d.value}

    def output_device_ptr(self):
        total = C.c_uint64()
        p = _lib.jpgpu_batch_output_device(self._h, C.byref(total))
        return p, total.value

    def output(self, i):
        """Downloads image i. INTERLEAVED_U8 -> uint8[H,W,C]; RGB_U8 / RGBA_U8 -> uint8[H,W,3|4]; EXTENDED_U16 -> uint16[H,W,4];
        PLANAR_* -> list of per-component 2-D arrays (padded)."""
        info = self.image_info(i)
        raise_for_status(info.status, _lib.jpgpu_last_error(self.ctx._h))
        raw = np.empty(info.out_bytes, dtype=np.uint8)
        self._check(_lib.jpgpu_batch_download_output(self._h, i, raw.ctypes.data, raw.size))
        if self.format == FMT_INTERLEAVED_U8:
            return raw.reshape(info.height, info.width, info.num_components)
        if self.format in (FMT_RGB_U8, FMT_RGBA_U8):
            return raw.reshape(info.height, info.width, 4 if self.format == FMT_RGBA_U8 else 3)
        if self.format == FMT_EXTENDED_U16:  # the reference tests' JpegExtendingOutputWriter buffer: ushort x 4 per pixel
            return raw.view(np.uint16).reshape(info.height, info.width, 4)
        dt = np.int16 if self.format == FMT_PLANAR_I16 else np.uint8
        planes = []
        for c in range(info.num_components):
            p = info.plane[c]
            nbytes = p.pitch * p.height * np.dtype(dt).itemsize
            planes.append(raw[p.offset:p.offset + nbytes].view(dt).reshape(p.height, p.pitch)[:, :p.width])
        return planes

    def coefficients(self, i):
        """int16[blocks, 64] zig-zag order, MCU scan order (the buffer between the Huffman and IDCT stages)."""
        info = self.image_info(i)
        raise_for_status(info.status, _lib.jpgpu_last_error(self.ctx._h))
        out = np.empty((info.total_blocks, 64), dtype=np.int16)
        self._check(_lib.jpgpu_batch_download_coefficients(self._h, i, out.ctypes.data, info.total_blocks))
        return out

    def set_coefficients(self, i, coefs):
        coefs = np.ascontiguousarray(coefs, dtype=np.int16).reshape(-1, 64)
        self._check(_lib.jpgpu_batch_upload_coefficients(self._h, i, coefs.ctypes.data, coefs.shape[0]))

    @classmethod
    def _borrowed(cls, handle, ctx_handle, fmt):
        """A view on a batch some other object owns (MultiDecoder's shards): same accessors, close() leaves it alone."""
        self = cls.__new__(cls)
        self.ctx = _BorrowedContext(ctx_handle)
        self._h = C.c_void_p(handle)
        self.format = fmt
        self._keep = None
        self._owned = False
        return self

    def close(self):
        if self._h and getattr(self, "_owned", True):
            _lib.jpgpu_batch_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_batch(files, fmt=FMT_INTERLEAVED_U8, ctx=None):
    """One-call helper: returns (outputs, results)."""
    b = Batch(ctx).upload(files, fmt).decode().sync()
    outs, results = [], []
    for i in range(len(b)):
        r = b.result(i)
        results.append(r)
        outs.append(b.output(i) if b.image_info(i).status == 0 else None)
    b.close()
    return outs, results
