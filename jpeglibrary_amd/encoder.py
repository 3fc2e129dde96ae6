"""Batch encode on the GPU: the reference's JpegEncoder.Encode() with the EncodeAction call sequence
(ref: src/JpegLibrary/JpegEncoder.cs:255-291, apps/JpegEncode/EncodeAction.cs:38-63) for a set of images at once.

    encoder.SetQuantizationTable(ScaleByQuality(luminance / chrominance, quality)); SetHuffmanTable(standard tables);
    AddComponent(1, 0, 0, 0, h, v); AddComponent(2, 1, 1, 1, 1, 1); AddComponent(3, 1, 1, 1, 1, 1);
    SetInputReader(JpegBufferInputReader(width, height, components, pixels)); SetOutput(writer); Encode()
"""
import ctypes as C

import numpy as np

from . import _capi
from .context import Context, default_context
from .errors import raise_for_status

_lib = _capi.lib


class EncodeBatch:
    def __init__(self, ctx: Context = None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        raise_for_status(_lib.jpgpu_encoder_create(self.ctx._h, C.byref(self._h)), b"jpgpu_encoder_create failed")
        self._n = 0
        self._blocks = []

    def _check(self, rc):
        raise_for_status(rc, _lib.jpgpu_last_error(self.ctx._h))

    def upload(self, images, luma=(2, 2), quality=75, rgb=False, optimize_coding=False, restart_interval=0):
        """images: list of uint8 arrays (H, W, 3) or (H, W) / (H, W, 1); with rgb=True also (H, W, 4) = Rgba32 pixels, the alpha byte
        stepped over like ConvertRgba32ToYCbCr8 does (the reference's EncoderBenchmark).  luma = sampling factors of the first component.
        optimize_coding = EncodeAction's switch: Huffman tables built from each image's own statistics.
        restart_interval = MCUs between restart markers (0 = none, as the reference's encoder; an extension, see jpgpu.h)."""
        n = len(images)
        ptrs = (C.c_void_p * n)()
        params = (_capi.EncodeParams * n)()
        keep = []
        self._blocks = []
        for i, im in enumerate(images):
            a = np.ascontiguousarray(im, dtype=np.uint8)
            if a.ndim == 2:
                a = a.reshape(a.shape[0], a.shape[1], 1)
            keep.append(a)
            ptrs[i] = a.ctypes.data
            h, w, c = a.shape
            mode = 1 if rgb else 0
            if c == 4:
                if not rgb:
                    raise ValueError("four bytes per pixel are Rgba32 pixels: rgb=True")
                c, mode = 3, 2
            params[i] = _capi.EncodeParams(w, h, c, luma[0], luma[1], quality, mode, int(optimize_coding), int(restart_interval))
            mcus = (-(-w // (8 * luma[0]))) * (-(-h // (8 * luma[1])))
            self._blocks.append(mcus * (luma[0] * luma[1] + (2 if c == 3 else 0)))
        self._check(_lib.jpgpu_encoder_upload(self._h, ptrs, params, n))
        self._n = n
        return self

    def set_quantization_table(self, i, identifier, zigzag64):
        """SetQuantizationTable for image i: the caller's own table (zig-zag order, 1..255) instead of the scaled standard one."""
        q = np.ascontiguousarray(zigzag64, dtype=np.uint16).reshape(64)
        self._check(_lib.jpgpu_encoder_set_quantization_table(self._h, i, identifier, q.ctypes.data))
        return self

    def encode(self):
        self._check(_lib.jpgpu_encoder_encode(self._h))
        return self

    def __len__(self):
        return self._n

    def emit_passes(self):
        """(calls of encode() whose entropy stage ran as ONE pass over the blocks, those of them that fell back to the two kernels)."""
        a, b = C.c_int(0), C.c_int(0)
        self._check(_lib.jpgpu_encoder_emit_passes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stage_ms(self):
        """Device time of the last encode() by stage (HIP events)."""
        ms = (C.c_float * 5)()
        self._check(_lib.jpgpu_encoder_stage_ms(self._h, ms))
        return {"fdct_quant": ms[0], "block_bits": ms[1], "emit": ms[2], "stuff": ms[3], "total": ms[4]}

    def output(self, i) -> bytes:
        size = C.c_size_t()
        self._check(_lib.jpgpu_encoder_encoded_size(self._h, i, C.byref(size)))
        out = np.empty(size.value, np.uint8)
        self._check(_lib.jpgpu_encoder_download(self._h, i, out.ctypes.data, out.size))
        return out.tobytes()

    def coefficients(self, i):
        out = np.empty((self._blocks[i], 64), np.int16)
        self._check(_lib.jpgpu_encoder_download_coefficients(self._h, i, out.ctypes.data, out.shape[0]))
        return out

    def close(self):
        if self._h:
            _lib.jpgpu_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_batch(images, luma=(2, 2), quality=75, rgb=False, ctx=None, optimize_coding=False, restart_interval=0):
    """One-call helper: list of JPEG byte strings."""
    b = EncodeBatch(ctx).upload(images, luma, quality, rgb, optimize_coding, restart_interval).encode()
    outs = [b.output(i) for i in range(len(b))]
    b.close()
    return outs
