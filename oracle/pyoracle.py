"""ctypes binding of oracle/libjpegref.so -- the CPU restatement of the reference decoder.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (jpeglibrary_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjpegref.so")

JREF_OK, JREF_INVALID_DATA, JREF_INVALID_OPERATION, JREF_NOT_SUPPORTED, JREF_ARGUMENT = range(5)
STATUS_NAMES = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException",
                4: "ArgumentException"}


class Component(C.Structure):
    _fields_ = [("identifier", C.c_uint8), ("h", C.c_uint8), ("v", C.c_uint8), ("tq", C.c_uint8)]


class Info(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("precision", C.c_int), ("ncomp", C.c_int),
                ("sof", C.c_int), ("restart_interval", C.c_int), ("consumed", C.c_int), ("comp", Component * 4)]


WRITE_BLOCK_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int)
COEF_TAP_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int16), C.c_int, C.c_long)
PROG_TAP_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16))


def build(force=False):
    """Compile the oracle with gcc (seconds). Idempotent."""
    src = os.path.join(_HERE, "jpegref.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.jref_create.restype = C.c_void_p
        L.jref_destroy.argtypes = [C.c_void_p]
        L.jref_last_error.restype = C.c_char_p
        L.jref_last_error.argtypes = [C.c_void_p]
        L.jref_set_input.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.jref_identify.argtypes = [C.c_void_p, C.c_int, C.POINTER(Info)]
        L.jref_try_estimate_quality.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.jref_set_output_writer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.jref_set_coef_tap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.jref_set_progressive_tap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.jref_decode.argtypes = [C.c_void_p]
        L.jref_decode_to_8bit.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(Info),
                                          C.c_char_p, C.c_size_t]
        L.jref_decode_to_16bit.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(Info),
                                           C.c_char_p, C.c_size_t]
        L.jref_block_dequant_idct_shift.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.jref_build_huffman.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code, message):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {message}")
        self.code = code
        self.kind = STATUS_NAMES.get(code, str(code))
        self.message = message


def identify(data: bytes, load_quantization_tables=False):
    """JpegDecoder.SetInput + Identify. Returns (Info, estimated_quality|None)."""
    L = lib()
    d = L.jref_create()
    try:
        L.jref_set_input(d, data, len(data))
        info = Info()
        rc = L.jref_identify(d, int(load_quantization_tables), C.byref(info))
        if rc != 0:
            raise OracleError(rc, L.jref_last_error(d).decode())
        q = None
        if load_quantization_tables:
            qf = C.c_float()
            if L.jref_try_estimate_quality(d, C.byref(qf)):
                q = qf.value
        return info, q
    finally:
        L.jref_destroy(d)


def decode_8bit(data: bytes, component_count=None):
    """Identify + Decode into the app writer's interleaved u8 buffer (O2). Returns (array[H,W,C], Info)."""
    L = lib()
    info, _ = identify(data)
    cc = component_count or info.ncomp
    out = np.zeros((info.height, info.width, cc), dtype=np.uint8)
    err = C.create_string_buffer(256)
    rc = L.jref_decode_to_8bit(data, len(data), cc, out.ctypes.data, out.size, C.byref(info), err, 256)
    if rc != 0:
        raise OracleError(rc, err.value.decode())
    return out, info


def decode_8bit_partial(data: bytes, component_count=None):
    """decode_8bit that also hands back what a FAILING decode left in the writer's buffer: Decode()'s `finally` disposes the scan
    decoder, and a progressive one then flushes whatever its store holds (JpegDecoder.cs:545-549).  Returns (array, Info, error):
    error is None for a clean decode, else the OracleError that decode_8bit would have raised."""
    L = lib()
    info, _ = identify(data)
    cc = component_count or info.ncomp
    out = np.zeros((info.height, info.width, cc), dtype=np.uint8)
    err = C.create_string_buffer(256)
    rc = L.jref_decode_to_8bit(data, len(data), cc, out.ctypes.data, out.size, C.byref(info), err, 256)
    return out, info, (OracleError(rc, err.value.decode()) if rc != 0 else None)


def decode_16bit(data: bytes, component_count=4):
    """Identify + Decode into the xunit test writer's u16 x4 buffer (O3). Returns (array[H,W,C] u16, Info)."""
    L = lib()
    info, _ = identify(data)
    out = np.zeros((info.height, info.width, component_count), dtype=np.uint16)
    err = C.create_string_buffer(256)
    rc = L.jref_decode_to_16bit(data, len(data), component_count, out.ctypes.data, out.size, C.byref(info), err, 256)
    if rc != 0:
        raise OracleError(rc, err.value.decode())
    return out, info


def decode_with_callbacks(data: bytes, write_block=None, coef_tap=None, call_identify=True):
    """Run Decode() with Python callbacks.

    write_block(block: np.ndarray[64] int16, component_index, x, y)
    coef_tap(zigzag: np.ndarray[64] int16, component_index, block_index)
    Returns Info (from Identify when call_identify, else None).
    """
    L = lib()
    d = L.jref_create()
    try:
        L.jref_set_input(d, data, len(data))
        info = None
        if call_identify:
            info = Info()
            rc = L.jref_identify(d, 0, C.byref(info))
            if rc != 0:
                raise OracleError(rc, L.jref_last_error(d).decode())

        def _wb(_user, blk, ci, x, y):
            if write_block is not None:
                write_block(np.ctypeslib.as_array(blk, shape=(64,)).copy(), ci, x, y)

        def _tap(_user, blk, ci, bi):
            coef_tap(np.ctypeslib.as_array(blk, shape=(64,)).copy(), ci, bi)

        wb = WRITE_BLOCK_FN(_wb)
        L.jref_set_output_writer(d, C.cast(wb, C.c_void_p), None)
        tap = None
        if coef_tap is not None:
            tap = COEF_TAP_FN(_tap)
            L.jref_set_coef_tap(d, C.cast(tap, C.c_void_p), None)
        rc = L.jref_decode(d)
        if rc != 0:
            raise OracleError(rc, L.jref_last_error(d).decode())
        return info
    finally:
        L.jref_destroy(d)


def decode_coefficients(data: bytes):
    """Baseline only: all entropy-decoded blocks in scan order -> (coefs[nblocks,64] int16 zig-zag, comp[nblocks])."""
    blocks, comps = [], []

    def tap(z, ci, bi):
        blocks.append(z)
        comps.append(ci)

    decode_with_callbacks(data, None, tap)
    return np.stack(blocks) if blocks else np.zeros((0, 64), np.int16), np.array(comps, dtype=np.int32)


def decode_progressive_store(data: bytes):
    """Progressive (SOF2) files: the accumulated coefficient store right before the reference's Dispose() runs the IDCT
    pass.  Returns (Info, blocks: {component: {(bx, by): int16[64] zig-zag}}, quant: {component: uint16[64] zig-zag})."""
    L = lib()
    d = L.jref_create()
    blocks, quant = {}, {}
    try:
        L.jref_set_input(d, data, len(data))
        info = Info()
        rc = L.jref_identify(d, 0, C.byref(info))
        if rc != 0:
            raise OracleError(rc, L.jref_last_error(d).decode())

        def _tap(_u, blk, ci, bx, by, q):
            blocks.setdefault(ci, {})[(bx, by)] = np.ctypeslib.as_array(blk, shape=(64,)).copy()
            if ci not in quant:
                quant[ci] = np.ctypeslib.as_array(q, shape=(64,)).copy()

        tap = PROG_TAP_FN(_tap)
        wb = WRITE_BLOCK_FN(lambda *a: None)
        L.jref_set_output_writer(d, C.cast(wb, C.c_void_p), None)
        L.jref_set_progressive_tap(d, C.cast(tap, C.c_void_p), None)
        rc = L.jref_decode(d)
        if rc != 0:
            raise OracleError(rc, L.jref_last_error(d).decode())
        return info, blocks, quant
    finally:
        L.jref_destroy(d)


def ycbcr8_to_rgb(ycbcr: np.ndarray, rgba: bool = False, gray: bool = False) -> np.ndarray:
    """The reference callers' colour step on an interleaved YCbCr8 image (H, W, 3) -- or (H, W, 1) with gray=True, which
    is first widened with Cb = Cr = 128 like apps/JpegDecode/DecodeAction.cs:57-65."""
    L = lib()
    L.jref_ycbcr8_to_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.jref_ycbcr8_to_rgb.restype = None
    a = np.ascontiguousarray(ycbcr, dtype=np.uint8)
    if gray:
        a = np.concatenate([a.reshape(a.shape[0], a.shape[1], 1), np.full((a.shape[0], a.shape[1], 2), 128, np.uint8)], axis=2)
        a = np.ascontiguousarray(a)
    h, w = a.shape[0], a.shape[1]
    bpp = 4 if rgba else 3
    out = np.empty((h, w, bpp), np.uint8)
    L.jref_ycbcr8_to_rgb(a.ctypes.data, out.ctypes.data, h * w, bpp)
    return out


def encode_8bit(pixels: np.ndarray, luma_h: int = 2, luma_v: int = 2, quality: int = 75, want_coefficients: bool = False,
                optimize_coding: bool = False, restart_interval: int = 0, quant_tables=None):
    """The reference encoder's EncodeAction sequence (standard tables, no optimisation) on an interleaved 8-bit image
    (H, W, C) with C = 3 (Y, Cb, Cr) or 1.  Returns the JPEG bytes (and the quantised zig-zag blocks in encoding order).
    restart_interval != 0: the extension of jref_encode_8bit_dri (the reference encoder has no restart markers).
    quant_tables = (luminance, chrominance) uint16[64] in zig-zag order: SetQuantizationTable with the caller's own tables."""
    L = lib()
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim == 2:
        a = a.reshape(a.shape[0], a.shape[1], 1)
    h, w, c = a.shape
    L.jref_encode_8bit_tables.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    L.jref_encode_8bit_tables.restype = C.c_int
    ql = qc = None
    if quant_tables is not None:
        ql = np.ascontiguousarray(quant_tables[0], dtype=np.uint16).reshape(64)
        qc = np.ascontiguousarray(quant_tables[1], dtype=np.uint16).reshape(64)
    ncomp = 1 if c == 1 else 3
    mh, mv = luma_h, luma_v
    mcus = (-(-w // (8 * mh))) * (-(-h // (8 * mv)))
    nblocks = mcus * (mh * mv + (2 if ncomp == 3 else 0))
    coefs = np.zeros((nblocks, 64), np.int16) if want_coefficients else None
    cap = 2048 + nblocks * 512 + mcus * 3  # 64 symbols x (16 + 11) bits per block, every byte stuffed, with room to spare
    out = np.empty(cap, np.uint8)
    n = C.c_size_t(0)
    rc = L.jref_encode_8bit_tables(a.ctypes.data, w, h, c, luma_h, luma_v, quality, ql.ctypes.data if ql is not None else None,
                                   qc.ctypes.data if qc is not None else None, int(optimize_coding), int(restart_interval), out.ctypes.data, cap,
                                   C.byref(n), coefs.ctypes.data if coefs is not None else None)
    if rc == 2:
        raise OracleError(2, "No symbol is recorded.")
    if rc != 0:
        raise OracleError(4, "encoder output buffer too small")
    data = out[:n.value].tobytes()
    return (data, coefs) if want_coefficients else data


def fdct_quantize_block(samples: np.ndarray, quant_zigzag: np.ndarray) -> np.ndarray:
    L = lib()
    L.jref_fdct_quantize_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.jref_fdct_quantize_block.restype = None
    s = np.ascontiguousarray(samples, dtype=np.int16).reshape(-1, 64)
    q = np.ascontiguousarray(quant_zigzag, dtype=np.uint16).reshape(64)
    out = np.empty_like(s)
    for i in range(s.shape[0]):
        L.jref_fdct_quantize_block(s[i].ctypes.data, q.ctypes.data, out[i].ctypes.data)
    return out


def rgb_to_ycbcr8(rgb: np.ndarray) -> np.ndarray:
    L = lib()
    L.jref_rgb_to_ycbcr8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.jref_rgb_to_ycbcr8.restype = None
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.empty_like(a)
    L.jref_rgb_to_ycbcr8(a.ctypes.data, out.ctypes.data, a.shape[0] * a.shape[1])
    return out


def rgba_to_ycbcr8(rgba: np.ndarray) -> np.ndarray:
    """ConvertRgba32ToYCbCr8 (the reference's EncoderBenchmark): (H, W, 4) -> (H, W, 3)."""
    L = lib()
    L.jref_rgba_to_ycbcr8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.jref_rgba_to_ycbcr8.restype = None
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4
    out = np.empty((a.shape[0], a.shape[1], 3), dtype=np.uint8)
    L.jref_rgba_to_ycbcr8(a.ctypes.data, out.ctypes.data, a.shape[0] * a.shape[1])
    return out


def decode_blocks(data: bytes):
    """All WriteBlock calls in order: list of (component_index, x, y, block[64] int16)."""
    calls = []
    info = decode_with_callbacks(data, lambda b, ci, x, y: calls.append((ci, x, y, b)))
    return calls, info


def block_dequant_idct_shift(zigzag_coefs: np.ndarray, quant_zigzag: np.ndarray, level_shift: int) -> np.ndarray:
    """Vectorised over leading dims: int16[...,64] zig-zag -> int16[...,64] spatial."""
    L = lib()
    z = np.ascontiguousarray(zigzag_coefs, dtype=np.int16).reshape(-1, 64)
    q = np.ascontiguousarray(quant_zigzag, dtype=np.uint16).reshape(64)
    out = np.empty_like(z)
    for i in range(z.shape[0]):
        L.jref_block_dequant_idct_shift(z[i].ctypes.data, q.ctypes.data, level_shift, out[i].ctypes.data)
    return out.reshape(zigzag_coefs.shape)


def build_huffman(bits, values):
    L = lib()
    bits = bytes(bits)
    values = bytes(values)
    las = np.zeros(256, np.uint8)
    lasym = np.zeros(256, np.uint8)
    maxcode = np.zeros(18, np.uint16)
    valoff = np.zeros(19, np.uint8)
    vout = np.zeros(256, np.uint8)
    ok = L.jref_build_huffman(bits, values, len(values), las.ctypes.data, lasym.ctypes.data, maxcode.ctypes.data,
                              valoff.ctypes.data, vout.ctypes.data)
    if not ok:
        raise ValueError("huffman table rejected")
    return las, lasym, maxcode, valoff, vout


# ---------------------------------------------------------------------------------------------- JpegOptimizer restatement
def optimize(data: bytes, strip: bool = True, most_optimal: bool = False) -> bytes:
    """new JpegOptimizer(): SetInput(data); Scan(); SetOutput(buffer); Optimize(strip) -> the written bytes.
    Raises OracleError(code, message) where the reference throws."""
    L = lib()
    L.jref_optimize_ex.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.jref_optimize_ex.restype = C.c_int
    L.jref_free.argtypes = [C.c_void_p]
    L.jref_free.restype = None
    out = C.c_void_p()
    n = C.c_size_t(0)
    err = C.create_string_buffer(256)
    rc = L.jref_optimize_ex(data, len(data), 1 if strip else 0, 1 if most_optimal else 0, C.byref(out), C.byref(n), err, 256)
    if rc != 0:
        raise OracleError(rc, err.value.decode("utf-8", "replace"))
    try:
        return C.string_at(out.value, n.value)
    finally:
        L.jref_free(out)


def optimizer_statistics(data: bytes):
    """Scan() alone: [(table_class, identifier, counts[256])] in the order the reference creates its table builders."""
    L = lib()
    L.jref_optimizer_statistics.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_char_p,
                                            C.c_size_t]
    L.jref_optimizer_statistics.restype = C.c_int
    cls = np.zeros(8, np.uint8)
    ident = np.zeros(8, np.uint8)
    freq = np.zeros((8, 256), np.uint32)
    n = C.c_int(0)
    err = C.create_string_buffer(256)
    rc = L.jref_optimizer_statistics(data, len(data), cls.ctypes.data, ident.ctypes.data, freq.ctypes.data, C.byref(n), err, 256)
    if rc != 0:
        raise OracleError(rc, err.value.decode("utf-8", "replace"))
    return [(int(cls[i]), int(ident[i]), freq[i].copy()) for i in range(n.value)]


def build_optimal_table(freq: np.ndarray, most_optimal: bool = False):
    """JpegHuffmanEncodingTableBuilder.Build(most_optimal): (bits[16], values[n], code[256], length[256])."""
    L = lib()
    L.jref_build_optimal_table_ex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    L.jref_build_optimal_table_ex.restype = C.c_int
    f = np.ascontiguousarray(freq, dtype=np.uint32).reshape(256)
    bits = np.zeros(16, np.uint8)
    values = np.zeros(256, np.uint8)
    code = np.zeros(256, np.uint16)
    length = np.zeros(256, np.uint8)
    n = C.c_int(0)
    rc = L.jref_build_optimal_table_ex(f.ctypes.data, 1 if most_optimal else 0, bits.ctypes.data, values.ctypes.data, C.byref(n), code.ctypes.data,
                                       length.ctypes.data)
    if rc != 0:
        raise OracleError(4, "No symbol is recorded." if rc == -1 else "code size outside the reference's array")
    return bits, values[:n.value].copy(), code, length


def decode_batch_mt(files, component_count=3, threads=1, warm=True, rgba=False):
    """bench.py's CPU baseline (jref_decode_batch_mt): Identify + Decode into a YCbCr8 buffer, one decoder per native
    thread, thread t taking images t, t + threads, ...  files: list of numpy uint8 arrays (views are fine).
    rgba=True: ConvertYCbCr8ToRgba32 behind every Decode(), the reference benchmark's whole sequence (DecoderBenchmark.cs:51-73).
    Returns (seconds, pixels)."""
    L = lib()
    L.jref_decode_batch_mt_ex.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    L.jref_decode_batch_mt_ex.restype = C.c_int
    n = len(files)
    keep = [np.ascontiguousarray(np.frombuffer(f, np.uint8) if not isinstance(f, np.ndarray) else f) for f in files]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in keep])
    lens = (C.c_size_t * n)(*[a.size for a in keep])
    sec, px = C.c_double(), C.c_uint64()
    err = C.create_string_buffer(256)
    rc = L.jref_decode_batch_mt_ex(ptrs, lens, n, component_count, threads, 1 if warm else 0, 1 if rgba else 0, C.byref(sec), C.byref(px), err, 256)
    if rc != 0:
        raise OracleError(rc, err.value.decode("utf-8", "replace"))
    return sec.value, px.value
