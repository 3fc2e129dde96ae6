"""Host-side mirror of the reference's public decoder surface, over the C ABI (include/jpgpu.h level 3).

Names, argument meaning and error behaviour follow the reference so that tests read like its own:

    decoder = JpegDecoder()
    decoder.SetInput(jpeg_bytes)
    decoder.Identify()
    writer = JpegExtendingOutputWriter(decoder.Width, decoder.Height, 4, decoder.Precision, buffer)
    decoder.SetOutputWriter(writer)
    decoder.Decode()

ref: src/JpegLibrary/JpegDecoder.cs, src/JpegLibrary/JpegBlockOutputWriter.cs,
     apps/JpegDecode/JpegBufferOutputWriter8Bit.cs, tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs
All block arithmetic (Huffman decode, dequantise, IDCT, level shift) runs in the HIP kernels; this module only
marshals calls.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi
from .context import Context, default_context
from .errors import ArgumentException, InvalidOperationException, raise_for_status

_lib = _capi.lib


class JpegBlockOutputWriter:
    """ref: JpegBlockOutputWriter.cs:17 -- abstract sink: one 8x8 int16 block at (componentIndex, x, y)."""

    def WriteBlock(self, blockRef, componentIndex, x, y):  # noqa: N802,N803 (reference names)
        raise NotImplementedError


class JpegBufferOutputWriter8Bit(JpegBlockOutputWriter):
    """ref: apps/JpegDecode/JpegBufferOutputWriter8Bit.cs -- interleaved u8 buffer, signed clamp to [0,255].

    When its geometry equals the frame's, JpegDecoder produces this layout directly on the GPU
    (JPGPU_FMT_INTERLEAVED_U8) instead of replaying WriteBlock calls.
    """

    def __init__(self, width, height, componentCount, output):  # noqa: N803
        output = np.asarray(output)
        if output.dtype != np.uint8 or not output.flags["C_CONTIGUOUS"]:
            raise ArgumentException("output must be a contiguous uint8 array")
        if output.size < width * height * componentCount:
            raise ArgumentException("Destination buffer is too small.")
        self.width, self.height, self.componentCount, self.output = width, height, componentCount, output

    def WriteBlock(self, blockRef, componentIndex, x, y):  # noqa: N802,N803
        w, h, cc = self.width, self.height, self.componentCount
        if x > w or y > h:
            return
        ww, wh = min(w - x, 8), min(h - y, 8)
        if ww <= 0 or wh <= 0:
            return
        out = self.output.reshape(-1)[:w * h * cc].reshape(h, w, cc)
        blk = np.asarray(blockRef, dtype=np.int16).reshape(8, 8)[:wh, :ww]
        out[y:y + wh, x:x + ww, componentIndex] = np.clip(blk, 0, 255).astype(np.uint8)


class JpegExtendingOutputWriter(JpegBlockOutputWriter):
    """ref: tests/JpegLibrary.Tests/Utils/JpegExtendingOutputWriter.cs -- the xunit tests' u16 sink:
    (ushort) clamp to 2^P-1 (negative samples become max), then bit-replication to 16 bits."""

    def __init__(self, width, height, componentCount, precision, output):  # noqa: N803
        output = np.asarray(output)
        if output.dtype != np.uint16 or not output.flags["C_CONTIGUOUS"]:
            raise ArgumentException("output must be a contiguous uint16 array")
        if output.size < width * height * componentCount:
            raise ArgumentException("Destination buffer is too small.")
        self.width, self.height, self.componentCount, self.precision, self.output = width, height, componentCount, precision, output

    @staticmethod
    def _fast_expand(bits, bit_count):
        remaining = 16 - bit_count
        return (bits << remaining) | (bits & ((1 << remaining) - 1))

    def _expand(self, v):
        p = self.precision
        if p >= 8:
            return self._fast_expand(v, p)
        bits, cur = v, p
        while cur < 16:
            bits = (bits << p) | bits
            cur += p
        if cur > 16:
            bits = bits >> p
            cur -= p
            bits = self._fast_expand(bits, cur)
        return bits

    def WriteBlock(self, blockRef, componentIndex, x, y):  # noqa: N802,N803
        w, h, cc = self.width, self.height, self.componentCount
        if x > w or y > h:
            return
        ww, wh = min(w - x, 8), min(h - y, 8)
        if ww <= 0 or wh <= 0:
            return
        mx = (1 << self.precision) - 1
        out = self.output.reshape(-1)[:w * h * cc].reshape(h, w, cc)
        blk = np.asarray(blockRef, dtype=np.int16).reshape(8, 8)[:wh, :ww]
        v = np.minimum(blk.astype(np.uint16).astype(np.uint32), mx)
        out[y:y + wh, x:x + ww, componentIndex] = (self._expand(v) & 0xFFFF).astype(np.uint16)


class JpegFrameComponentSpecificationParameters:
    """ref: JpegFrameHeader.cs:243-249"""

    def __init__(self, identifier, horizontalSamplingFactor, verticalSamplingFactor, quantizationTableSelector):  # noqa: N803
        self.Identifier, self.HorizontalSamplingFactor = int(identifier), int(horizontalSamplingFactor)
        self.VerticalSamplingFactor, self.QuantizationTableSelector = int(verticalSamplingFactor), int(quantizationTableSelector)


class JpegFrameHeader:
    """ref: JpegFrameHeader.cs:22-29 (the SOF payload)."""

    def __init__(self, samplePrecision, numberOfLines, samplesPerLine, numberOfComponents, components):  # noqa: N803
        self.SamplePrecision, self.NumberOfLines, self.SamplesPerLine = int(samplePrecision), int(numberOfLines), int(samplesPerLine)
        self.NumberOfComponents, self.Components = int(numberOfComponents), list(components or [])

    def _c(self, sof=0):
        f = _capi.Frame()
        f.width, f.height, f.precision, f.num_components, f.sof = self.SamplesPerLine, self.NumberOfLines, self.SamplePrecision, self.NumberOfComponents, sof
        for i, c in enumerate(self.Components[:4]):
            f.comp[i] = _capi.FrameComponent(c.Identifier, c.HorizontalSamplingFactor, c.VerticalSamplingFactor, c.QuantizationTableSelector)
        return f


class JpegScanComponentSpecificationParameters:
    """ref: JpegScanHeader.cs:265-270"""

    def __init__(self, scanComponentSelector, dcEntropyCodingTableSelector, acEntropyCodingTableSelector):  # noqa: N803
        self.ScanComponentSelector = int(scanComponentSelector)
        self.DcEntropyCodingTableSelector, self.AcEntropyCodingTableSelector = int(dcEntropyCodingTableSelector), int(acEntropyCodingTableSelector)


class JpegScanHeader:
    """ref: JpegScanHeader.cs:23-31 (the SOS payload)."""

    def __init__(self, numberOfComponents, components, startOfSpectralSelection, endOfSpectralSelection,  # noqa: N803
                 successiveApproximationBitPositionHigh, successiveApproximationBitPositionLow):  # noqa: N803
        self.NumberOfComponents, self.Components = int(numberOfComponents), list(components or [])
        self.StartOfSpectralSelection, self.EndOfSpectralSelection = int(startOfSpectralSelection), int(endOfSpectralSelection)
        self.SuccessiveApproximationBitPositionHigh = int(successiveApproximationBitPositionHigh)
        self.SuccessiveApproximationBitPositionLow = int(successiveApproximationBitPositionLow)

    def _c(self):
        s = _capi.Scan()
        s.num_components, s.ss, s.se = self.NumberOfComponents, self.StartOfSpectralSelection, self.EndOfSpectralSelection
        s.ah, s.al = self.SuccessiveApproximationBitPositionHigh, self.SuccessiveApproximationBitPositionLow
        for i, c in enumerate(self.Components[:4]):
            s.comp[i] = _capi.ScanComponent(c.ScanComponentSelector, c.DcEntropyCodingTableSelector, c.AcEntropyCodingTableSelector, 0)
        return s


class JpegHuffmanDecodingTable:
    """ref: JpegHuffmanDecodingTable.cs -- the DHT payload behind the Tc/Th byte: BITS[16] + HUFFVAL (TryParse :249-291).
    The canonical codes and lookup tables are derived on the host side of the library, as Configure (:339-376) does."""

    def __init__(self, tableClass, identifier, bits, values):  # noqa: N803
        self.TableClass, self.Identifier = int(tableClass), int(identifier)
        self.bits, self.values = bytes(bits), bytes(values)
        if len(self.bits) != 16:
            raise ArgumentException("BITS must hold 16 counts.")


def _tables_c(quantizationTables, huffmanTables, scan_c=None, tq_slots=None):  # noqa: N803
    """(qt[4][64], qt_present[4], dht[2][4]) for the per-scan entry points, from the decoder-registry-style lists.

    The reference finds a table by its EXACT class and identifier (GetHuffmanTable / GetQuantizationTable, JpegDecoder.cs:869-884,
    910-925), identifiers being whatever byte the DHT / DQT / SOS / SOF carried (0..15 for well-formed files): a file that defines
    and selects table 5 decodes.  The C arrays have four slots per class -- as many as a scan of four components can select --, so
    with `scan_c` (a _capi.Scan, rewritten in place) every identifier the scan SELECTS is given a slot of its own and the scan's
    selectors are pointed at the slots (round 6; ADVICE r5: identifiers above 3 used to be dropped, and before that folded onto
    [identifier & 3] over live tables).  tq_slots: {quantisation table identifier: slot} fixed by the frame header
    (_frame_tq_slots).  A registry entry under a class other than 0 / 1 is never the one a scan uses and is not handed over."""
    qt = np.zeros((4, 64), np.uint16)
    present = np.zeros(4, np.uint8)
    if tq_slots is None:
        tq_slots = {i: i for i in range(4)}
    for q in quantizationTables or []:
        if q is None or q.IsEmpty or q.Identifier not in tq_slots:
            continue
        qt[tq_slots[q.Identifier]] = np.asarray(q.Elements, np.uint16)
        present[tq_slots[q.Identifier]] = 1
    slots = ({i: i for i in range(4)}, {i: i for i in range(4)})
    if scan_c is not None:
        slots = ({}, {})
        for i in range(min(int(scan_c.num_components), 4)):
            c = scan_c.comp[i]
            c.td = slots[0].setdefault(int(c.td), len(slots[0]))
            c.ta = slots[1].setdefault(int(c.ta), len(slots[1]))
    dht = ((_capi.Dht * 4) * 2)()
    for t in huffmanTables or []:
        if t.TableClass not in (0, 1) or t.Identifier not in slots[t.TableClass]:
            continue
        d = dht[t.TableClass][slots[t.TableClass][t.Identifier]]
        d.present = 1
        for i in range(16):
            d.bits[i] = t.bits[i]
        d.num_values = len(t.values)
        for i, v in enumerate(t.values[:256]):
            d.values[i] = v
    return qt, present, dht


def _frame_tq_slots(frame_c):
    """{quantisation table identifier: slot 0..3} for the identifiers a frame's (at most four) components select; the frame's
    selectors are rewritten to the slots."""
    slots = {}
    for i in range(min(int(frame_c.num_components), 4)):
        c = frame_c.comp[i]
        c.tq = slots.setdefault(int(c.tq), len(slots))
    return slots


class JpegGpuProgressiveScanDecoder:
    """The per-scan boundary for SOF2 frames (include/jpgpu.h 2b: jpgpu_progressive_*): what JpegDecoder does with the scan
    decoder JpegScanDecoder.Create(SOF2, ...) returns -- constructor at SOF, ProcessScan at every SOS with the tables and the
    restart interval in force there, Dispose at the end (ref: ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:23-90, 421-470;
    JpegDecoder.cs:562-570, 592-599, 545-549).  The coefficient store lives in HBM between the calls."""

    def __init__(self, frameHeader, ctx: Context = None):  # noqa: N803
        self._ctx = ctx or default_context()
        self._h = C.c_void_p()
        f = frameHeader._c(0xC2)
        self._tq_slots = _frame_tq_slots(f)
        raise_for_status(_lib.jpgpu_progressive_begin(self._ctx._h, C.byref(f), C.byref(self._h)), _lib.jpgpu_last_error(self._ctx._h))

    def ProcessScan(self, entropy, scanHeader, quantizationTables, huffmanTables, restartInterval=0):  # noqa: N802,N803
        """Decodes one scan into the store; raises the reference's exception for this scan.  Returns the reader advance (0)."""
        a = np.frombuffer(entropy, dtype=np.uint8) if not isinstance(entropy, np.ndarray) else np.ascontiguousarray(entropy, dtype=np.uint8)
        sc = scanHeader._c()
        qt, present, dht = _tables_c(quantizationTables, huffmanTables, sc, self._tq_slots)
        res = _capi.ImageResult()
        consumed = C.c_size_t()
        rc = _lib.jpgpu_progressive_scan(self._h, C.byref(sc), qt.ctypes.data, present.ctypes.data, C.cast(dht, C.c_void_p), int(restartInterval),
                                         a.ctypes.data if a.size else None, a.size, C.byref(res), C.byref(consumed))
        raise_for_status(rc, _lib.jpgpu_last_error(self._ctx._h))
        return consumed.value

    def output_size(self, fmt):
        n = C.c_size_t()
        raise_for_status(_lib.jpgpu_progressive_output_size(self._h, fmt, C.byref(n)), _lib.jpgpu_last_error(self._ctx._h))
        return n.value

    def Dispose(self, outputWriter=None, fmt=None):  # noqa: N802,N803
        """The IDCT pass + Flush.  outputWriter: a JpegBlockOutputWriter (WriteBlock calls in Flush's order); or fmt: one of the
        FMT_* device layouts, returned as a flat uint8 array."""
        if outputWriter is not None:
            def _cb(_user, blk, ci, x, y):
                outputWriter.WriteBlock(np.ctypeslib.as_array(blk, shape=(64,)), ci, x, y)

            cb = _capi.WRITE_BLOCK_FN(_cb)
            raise_for_status(_lib.jpgpu_progressive_dispose_to_writer(self._h, C.cast(cb, C.c_void_p), None), _lib.jpgpu_last_error(self._ctx._h))
            return None
        out = np.zeros(self.output_size(fmt), np.uint8)
        raise_for_status(_lib.jpgpu_progressive_dispose(self._h, fmt, out.ctypes.data, out.size), _lib.jpgpu_last_error(self._ctx._h))
        return out

    def close(self):
        if self._h:
            _lib.jpgpu_progressive_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class JpegDecoder:
    """ref: src/JpegLibrary/JpegDecoder.cs (public surface).  Scans are decoded on the MI355X."""

    def __init__(self, ctx: Context = None, host_only=False):
        """host_only=True builds a decoder without a device context: SetInput / Identify / metadata / table calls
        work anywhere, Decode() raises NoDeviceError (scans are decoded on the GPU only)."""
        self._ctx = None if host_only else (ctx or default_context())
        self._h = C.c_void_p()
        raise_for_status(_lib.jpgpu_decoder_create(self._ctx._h if self._ctx else None, C.byref(self._h)), b"jpgpu_decoder_create failed")
        self._input = None
        self._writer = None
        self._cb = None

    # -- helpers
    def _check(self, rc):
        raise_for_status(rc, _lib.jpgpu_decoder_last_error(self._h))

    def _need_header(self, v):
        if v < 0:
            raise InvalidOperationException("Call Identify() before this operation.")
        return v

    # -- input / identify
    def SetInput(self, input):  # noqa: N802,A002
        a = np.frombuffer(input, dtype=np.uint8) if not isinstance(input, np.ndarray) else np.ascontiguousarray(input, dtype=np.uint8)
        self._input = a  # the decoder never copies the caller's bytes; keep them alive
        self._check(_lib.jpgpu_decoder_set_input(self._h, a.ctypes.data if a.size else None, a.size))

    def Identify(self, loadQuantizationTables=False):  # noqa: N802,N803
        n = C.c_int()
        self._check(_lib.jpgpu_decoder_identify(self._h, int(bool(loadQuantizationTables)), C.byref(n)))
        return n.value

    def TryEstimateQuanlity(self):  # noqa: N802  (sic: the reference's spelling)
        q = C.c_float()
        ok = _lib.jpgpu_decoder_try_estimate_quality(self._h, C.byref(q))
        return bool(ok), q.value

    Width = property(lambda self: self._need_header(_lib.jpgpu_decoder_width(self._h)))
    Height = property(lambda self: self._need_header(_lib.jpgpu_decoder_height(self._h)))
    Precision = property(lambda self: self._need_header(_lib.jpgpu_decoder_precision(self._h)))
    NumberOfComponents = property(lambda self: self._need_header(_lib.jpgpu_decoder_number_of_components(self._h)))
    StartOfFrame = property(lambda self: _lib.jpgpu_decoder_start_of_frame(self._h),
                            lambda self, marker: self._check(_lib.jpgpu_decoder_set_start_of_frame(self._h, int(marker))))  # { get; set; } :43

    # -- the TIFF-style surface: the caller supplies frame header and tables, then hands over one scan's data
    def SetFrameHeader(self, frameHeader):  # noqa: N802,N803  (:404-407)
        f = frameHeader._c()
        self._check(_lib.jpgpu_decoder_set_frame_header(self._h, C.byref(f)))

    def SetHuffmanTable(self, table):  # noqa: N802  (:793-815)
        if table is None:
            raise ArgumentException("Value cannot be null. (Parameter 'table')")
        bits = (C.c_uint8 * 16)(*table.bits)
        vals = (C.c_uint8 * max(1, len(table.values)))(*table.values)
        self._check(_lib.jpgpu_decoder_set_huffman_table(self._h, table.TableClass, table.Identifier, bits, vals, len(table.values)))

    def SetQuantizationTable(self, table):  # noqa: N802  (:840-861)
        if table is None or table.IsEmpty:
            raise ArgumentException("No actual quantization table is provided. (Parameter 'table')")
        el = np.asarray(table.Elements, np.uint16)
        self._check(_lib.jpgpu_decoder_set_quantization_table(self._h, table.ElementPrecision, table.Identifier, el.ctypes.data))

    def ClearHuffmanTable(self):  # noqa: N802  (:768-771)
        self._check(_lib.jpgpu_decoder_clear_huffman_table(self._h))

    def ClearQuantizationTable(self):  # noqa: N802  (:784-787)
        self._check(_lib.jpgpu_decoder_clear_quantization_table(self._h))

    def ProcessScan(self, reader, scanHeader):  # noqa: N802,N803  (:624-632)
        """reader: the bytes behind the SOS header (JpegReader.RemainingBytes).  Returns how far the reference advances the reader."""
        a = np.frombuffer(reader, dtype=np.uint8) if not isinstance(reader, np.ndarray) else np.ascontiguousarray(reader, dtype=np.uint8)
        sc = scanHeader._c()
        n = C.c_size_t()
        self._check(_lib.jpgpu_decoder_process_scan(self._h, C.byref(sc), a.ctypes.data if a.size else None, a.size, C.byref(n)))
        return n.value

    def GetMaximumHorizontalSampling(self):  # noqa: N802
        v = _lib.jpgpu_decoder_get_maximum_horizontal_sampling(self._h)
        if v < 0:
            self._check(_capi.ERR_INVALID_OPERATION)
        return v

    def GetMaximumVerticalSampling(self):  # noqa: N802
        v = _lib.jpgpu_decoder_get_maximum_vertical_sampling(self._h)
        if v < 0:
            self._check(_capi.ERR_INVALID_OPERATION)
        return v

    def GetHorizontalSampling(self, componentIndex):  # noqa: N802,N803
        v = _lib.jpgpu_decoder_get_horizontal_sampling(self._h, componentIndex)
        if v < 0:
            self._check(_capi.ERR_ARGUMENT)
        return v

    def GetVerticalSampling(self, componentIndex):  # noqa: N802,N803
        v = _lib.jpgpu_decoder_get_vertical_sampling(self._h, componentIndex)
        if v < 0:
            self._check(_capi.ERR_ARGUMENT)
        return v

    def GetRestartInterval(self):  # noqa: N802
        return _lib.jpgpu_decoder_get_restart_interval(self._h)

    def SetRestartInterval(self, restartInterval):  # noqa: N802,N803
        self._check(_lib.jpgpu_decoder_set_restart_interval(self._h, restartInterval))

    def LoadTables(self, content):  # noqa: N802
        a = np.frombuffer(content, dtype=np.uint8)
        self._check(_lib.jpgpu_decoder_load_tables(self._h, a.ctypes.data, a.size))

    # -- output / decode
    def SetOutputWriter(self, outputWriter):  # noqa: N802,N803
        if outputWriter is None:
            raise ArgumentException("Value cannot be null. (Parameter 'outputWriter')")
        self._writer = outputWriter
        if type(outputWriter) is JpegBufferOutputWriter8Bit:
            w = outputWriter
            self._check(_lib.jpgpu_decoder_set_output_buffer8(self._h, w.width, w.height, w.componentCount, w.output.ctypes.data, w.output.size))
            self._cb = None
            return

        def _cb(_user, blk, ci, x, y):
            outputWriter.WriteBlock(np.ctypeslib.as_array(blk, shape=(64,)), ci, x, y)

        self._cb = _capi.WRITE_BLOCK_FN(_cb)
        self._check(_lib.jpgpu_decoder_set_output_writer(self._h, C.cast(self._cb, C.c_void_p), None))

    def Decode(self):  # noqa: N802
        self._check(_lib.jpgpu_decoder_decode(self._h))

    # -- resets
    def Reset(self):  # noqa: N802
        _lib.jpgpu_decoder_reset(self._h)
        self._input = self._writer = self._cb = None

    def ResetInput(self):  # noqa: N802
        _lib.jpgpu_decoder_reset_input(self._h)
        self._input = None

    def ResetHeader(self):  # noqa: N802
        _lib.jpgpu_decoder_reset_header(self._h)

    def ResetTables(self):  # noqa: N802
        _lib.jpgpu_decoder_reset_tables(self._h)

    def ResetOutputWriter(self):  # noqa: N802
        _lib.jpgpu_decoder_reset_output_writer(self._h)
        self._writer = self._cb = None

    def close(self):
        if self._h:
            _lib.jpgpu_decoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
