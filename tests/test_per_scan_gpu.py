"""The per-scan boundaries: a caller that keeps JpegDecoder's own marker loop and hands the library one scan at a time.

* jpgpu_progressive_begin / _scan / _dispose (include/jpgpu.h 2b) = JpegHuffmanProgressiveScanDecoder behind
  JpegScanDecoder.Create(SOF2, ...) -- constructor at SOF, ProcessScan at every SOS with the tables and the restart interval in
  force THERE, Dispose at the end (ref: ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:23-90, 421-470; JpegDecoder.cs:562-599).
* the TIFF-style decoder surface: SetFrameHeader / StartOfFrame / SetHuffmanTable / SetQuantizationTable / Clear*Table /
  ProcessScan (ref: JpegDecoder.cs:43, 404-407, 624-632, 768-861).

The marker loop of these tests is written here, in the test, the way JpegDecoder.Decode walks a file (TryReadMarker's forward
search included): the library sees headers, tables and entropy bytes only through the per-scan entry points.
"""
import numpy as np
import pytest

import jpeglibrary_amd as jl
from golden_util import load_reference_buffer, read_jpeg
from oracle import pyoracle as po
from tools import jpegsynth

pytestmark = pytest.mark.gpu


class Walk:
    """JpegDecoder.Decode's marker loop (JpegDecoder.cs:509-617) with the table registry (DQT :732-763, DHT :672-700, DRI
    :635-650) kept as the lists the reference keeps; on_frame(marker, frameHeader) / on_scan(entropy, scanHeader) are the two
    places it calls into a scan decoder.  on_scan returns how far the scan decoder advanced the reader."""

    def __init__(self, data):
        self.d = bytes(data)
        self.quant, self.huff, self.dri = {}, {}, 0

    def quantization_tables(self):
        return list(self.quant.values())

    def huffman_tables(self):
        return list(self.huff.values())

    def run(self, on_frame, on_scan):
        d = self.d
        assert d[:2] == b"\xff\xd8"
        p = 2
        while True:
            # JpegReader.TryReadMarker (JpegReader.cs:120-158): forward to the next FF xx with xx not in {00, FF}
            while True:
                q = d.find(b"\xff", p)
                if q < 0 or q + 1 >= len(d):
                    return
                if d[q + 1] == 0xFF:
                    p = q + 1
                    continue
                if d[q + 1] == 0x00:
                    p = q + 2
                    continue
                marker, p = d[q + 1], q + 2
                break
            if marker == 0xD9:
                return
            if 0xD0 <= marker <= 0xD7:
                continue
            length = (d[p] << 8) | d[p + 1]
            body = d[p + 2:p + length]
            p += length
            if marker in (0xC0, 0xC1, 0xC2):
                comps = [jl.JpegFrameComponentSpecificationParameters(body[6 + 3 * i], body[7 + 3 * i] >> 4, body[7 + 3 * i] & 15, body[8 + 3 * i])
                         for i in range(body[5])]
                on_frame(marker, jl.JpegFrameHeader(body[0], (body[1] << 8) | body[2], (body[3] << 8) | body[4], body[5], comps))
            elif marker == 0xC4:
                b = body
                while b:
                    n = sum(b[1:17])
                    t = jl.JpegHuffmanDecodingTable(b[0] >> 4, b[0] & 15, b[1:17], b[17:17 + n])
                    self.huff[(t.TableClass, t.Identifier)] = t
                    b = b[17 + n:]
            elif marker == 0xDB:
                b = body
                while b:
                    prec, ident = b[0] >> 4, b[0] & 15
                    if prec == 0:
                        el, b = list(b[1:65]), b[65:]
                    else:
                        el, b = [(b[1 + 2 * i] << 8) | b[2 + 2 * i] for i in range(64)], b[129:]
                    self.quant[ident] = jl.JpegQuantizationTable(prec, ident, el)
            elif marker == 0xDD:
                self.dri = (body[0] << 8) | body[1]
            elif marker == 0xDA:
                ns = body[0]
                comps = [jl.JpegScanComponentSpecificationParameters(body[1 + 2 * i], body[2 + 2 * i] >> 4, body[2 + 2 * i] & 15) for i in range(ns)]
                sh = jl.JpegScanHeader(ns, comps, body[1 + 2 * ns], body[2 + 2 * ns], body[3 + 2 * ns] >> 4, body[3 + 2 * ns] & 15)
                p += on_scan(d[p:], sh)


def _decode_progressive_scan_by_scan(data, deliver):
    w = Walk(data)
    state = {}

    def on_frame(marker, fh):
        assert marker == 0xC2
        state["fh"] = fh
        state["dec"] = jl.JpegGpuProgressiveScanDecoder(fh)
        state["dris"] = []

    def on_scan(entropy, sh):
        state["dris"].append(w.dri)
        return state["dec"].ProcessScan(entropy, sh, w.quantization_tables(), w.huffman_tables(), w.dri)

    w.run(on_frame, on_scan)
    out = deliver(state["dec"], state["fh"])
    state["dec"].close()
    return out, state


@pytest.mark.parametrize("name", ["progress.jpg", "yellowcat_progressive_restart.jpg"])
def test_progressive_goldens_scan_by_scan_through_the_c_abi(name):
    """Both progressive assets of the reference's tests, every SOS handed over on its own with the registry of that moment
    (yellowcat changes its restart interval between scans: yellowcat_progressive_restart.jpg.txt:118-121, 160-163), then
    Dispose into the reference tests' own sink layout: equal to the reference's golden PNG dumps, ushort for ushort."""
    data = read_jpeg(name)

    def deliver(dec, fh):
        return dec.Dispose(fmt=jl.FMT_EXTENDED_U16).view(np.uint16).reshape(fh.NumberOfLines, fh.SamplesPerLine, 4)

    out, state = _decode_progressive_scan_by_scan(data, deliver)
    fh = state["fh"]
    assert np.array_equal(out, load_reference_buffer(name, fh.SamplesPerLine, fh.NumberOfLines, fh.NumberOfComponents))
    if name.startswith("yellowcat"):
        assert len(set(state["dris"])) > 1, state["dris"]  # the per-scan restart interval really did change

    # ... and Dispose into an arbitrary JpegBlockOutputWriter: the WriteBlock calls of JpegBlockAllocator.Flush
    def deliver_writer(dec, fh):
        buf = np.zeros(fh.SamplesPerLine * fh.NumberOfLines * 4, np.uint16)
        dec.Dispose(outputWriter=jl.JpegExtendingOutputWriter(fh.SamplesPerLine, fh.NumberOfLines, 4, fh.SamplePrecision, buf))
        return buf.reshape(fh.NumberOfLines, fh.SamplesPerLine, 4)

    out2, _ = _decode_progressive_scan_by_scan(data, deliver_writer)
    assert np.array_equal(out2, out)


def test_progressive_scan_errors_belong_to_the_scan_that_fails():
    """ProcessScan decodes on the spot: a corrupted scan reports ITS failure from ITS call (class as the checker's for the
    whole file), the scans in front of it report success, and a missing table is refused before anything runs."""
    import io

    from PIL import Image

    rng = np.random.default_rng(11)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=80, progressive=True)
    good = buf.getvalue()
    # plain: scan by scan equals the whole-file decode and the checker
    out, state = _decode_progressive_scan_by_scan(good, lambda dec, fh: dec.Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, 3))
    assert np.array_equal(out, po.decode_8bit(good)[0])
    # corrupt the 4th scan's data: find the SOS markers
    sos = [i for i in range(len(good) - 1) if good[i] == 0xFF and good[i + 1] == 0xDA]
    assert len(sos) >= 5
    k = 3
    start = sos[k] + 2 + ((good[sos[k] + 2] << 8) | good[sos[k] + 3])
    bad = bytearray(good)
    bad[start + 8:start + 40] = b"\xff\xff" * 16  # ones: codes no table assigns
    bad = bytes(bad)
    try:
        po.decode_8bit(bad)
        expected = None
    except po.OracleError as e:
        expected = e.kind
    assert expected is not None
    w = Walk(bad)
    state = {"n": 0, "failed_at": None}

    def on_frame(marker, fh):
        state["dec"] = jl.JpegGpuProgressiveScanDecoder(fh)

    def on_scan(entropy, sh):
        if state["failed_at"] is not None:
            return 0
        try:
            state["dec"].ProcessScan(entropy, sh, w.quantization_tables(), w.huffman_tables(), w.dri)
        except jl.JpegError as e:
            state["failed_at"] = (state["n"], type(e).__name__)
        state["n"] += 1
        return 0

    w.run(on_frame, on_scan)
    assert state["failed_at"] == (k, expected), state["failed_at"]
    state["dec"].close()
    # a scan whose Huffman table was never defined: refused by ProcessScan's checks, nothing decoded
    w = Walk(good)
    seen = {}

    def on_frame2(marker, fh):
        seen["dec"] = jl.JpegGpuProgressiveScanDecoder(fh)

    def on_scan2(entropy, sh):
        if "done" not in seen:
            seen["done"] = True
            with pytest.raises(jl.InvalidDataException):
                seen["dec"].ProcessScan(entropy, sh, w.quantization_tables(), [], w.dri)
            with pytest.raises(jl.InvalidDataException):
                seen["dec"].ProcessScan(entropy, sh, [], w.huffman_tables(), w.dri)
        return 0

    w.run(on_frame2, on_scan2)
    seen["dec"].close()
    with pytest.raises(jl.NotSupportedException):  # Create(SOF0) is not this scan decoder
        fh = jl.JpegFrameHeader(8, 16, 16, 1, [jl.JpegFrameComponentSpecificationParameters(1, 1, 1, 0)])
        f = fh._c(0xC0)
        import ctypes as C
        h = C.c_void_p()
        jl.errors.raise_for_status(jl._capi.lib.jpgpu_progressive_begin(jl.default_context()._h, C.byref(f), C.byref(h)), b"")


@pytest.mark.parametrize("name", ["lake.jpg", "cramps.jpg", "testorig12.jpg"])
def test_tiff_style_surface_decodes_the_reference_goldens(name):
    """SetFrameHeader + StartOfFrame + SetQuantizationTable + SetHuffmanTable + SetOutputWriter + ProcessScan(reader, header):
    the JPEG-in-TIFF call pattern (tables, frame header and strip data arrive separately).  The reference's sequential goldens,
    through the reference tests' own writer, equal its PNG dumps; the reader advance is the scan's length."""
    data = read_jpeg(name)
    w = Walk(data)
    dec = jl.JpegDecoder()
    st = {}

    def on_frame(marker, fh):
        st["fh"], st["sof"] = fh, marker

    def on_scan(entropy, sh):
        fh = st["fh"]
        dec.ClearHuffmanTable()
        dec.ClearQuantizationTable()
        for q in w.quantization_tables():
            dec.SetQuantizationTable(q)
        for t in w.huffman_tables():
            dec.SetHuffmanTable(t)
        dec.SetFrameHeader(fh)
        dec.StartOfFrame = st["sof"]
        dec.SetRestartInterval(w.dri)
        st["buf"] = np.zeros(fh.SamplesPerLine * fh.NumberOfLines * 4, np.uint16)
        dec.SetOutputWriter(jl.JpegExtendingOutputWriter(fh.SamplesPerLine, fh.NumberOfLines, 4, fh.SamplePrecision, st["buf"]))
        st["adv"] = dec.ProcessScan(entropy, sh)
        st["left"] = len(entropy)
        return st["adv"]

    w.run(on_frame, on_scan)
    fh = st["fh"]
    assert dec.Width == fh.SamplesPerLine and dec.NumberOfComponents == fh.NumberOfComponents and dec.StartOfFrame == st["sof"]
    assert np.array_equal(st["buf"].reshape(fh.NumberOfLines, fh.SamplesPerLine, 4),
                          load_reference_buffer(name, fh.SamplesPerLine, fh.NumberOfLines, fh.NumberOfComponents))
    assert st["left"] - st["adv"] == 2  # the reader stands in front of the EOI (JpegHuffmanBaselineScanDecoder.cs:167-176)
    dec.close()


def test_tiff_style_surface_argument_checks_and_table_replacement():
    dec = jl.JpegDecoder()
    sh = jl.JpegScanHeader(1, [jl.JpegScanComponentSpecificationParameters(1, 0, 0)], 0, 63, 0, 0)
    with pytest.raises(jl.InvalidOperationException):  # GetFrameHeader(): "Call Identify() before this operation."
        dec.ProcessScan(b"\x00", sh)
    with pytest.raises(jl.ArgumentException):
        dec.SetHuffmanTable(None)
    with pytest.raises(jl.ArgumentException):
        dec.SetQuantizationTable(jl.JpegQuantizationTable())  # IsEmpty
    data = bytes(jpegsynth.encode(64, 48, "gray", 75, 0, seed=4))
    w = Walk(data)
    got = {}

    def on_frame(marker, fh):
        got["fh"], got["sof"] = fh, marker

    def on_scan(entropy, sh2):
        got["entropy"], got["sh"] = entropy, sh2
        return len(entropy)

    w.run(on_frame, on_scan)
    fh = got["fh"]
    out = np.zeros(fh.SamplesPerLine * fh.NumberOfLines, np.uint8)
    dec.SetFrameHeader(fh)
    dec.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(fh.SamplesPerLine, fh.NumberOfLines, 1, out))
    dec.StartOfFrame = 0xC3  # lossless: not a scan decoder of this path
    with pytest.raises(jl.NotSupportedException):
        dec.ProcessScan(got["entropy"], got["sh"])
    dec.StartOfFrame = got["sof"]
    with pytest.raises(jl.InvalidDataException):  # no tables yet
        dec.ProcessScan(got["entropy"], got["sh"])
    for q in w.quantization_tables():
        dec.SetQuantizationTable(jl.JpegQuantizationTable(0, q.Identifier, [1] * 64))  # a wrong table first ...
        dec.SetQuantizationTable(q)                                                    # ... replaced by identifier (:850-857)
    for t in w.huffman_tables():
        dec.SetHuffmanTable(t)
    dec.ProcessScan(got["entropy"], got["sh"])
    assert np.array_equal(out.reshape(fh.NumberOfLines, fh.SamplesPerLine, 1), po.decode_8bit(data)[0])
    dec.ClearHuffmanTable()
    with pytest.raises(jl.InvalidDataException):
        dec.ProcessScan(got["entropy"], got["sh"])
    dec.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_a_failing_scan_leaves_the_session_store_as_the_reference_leaves_it(seed):
    """JpegDecoder.Decode's `finally` disposes the scan decoder (JpegDecoder.cs:545-549) and the progressive one flushes its store
    exactly as the throwing ProcessScan left it (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:421-470).  The caller of the
    per-scan session does the same: stop at the failing scan, Dispose.  Round 4 reproduced that flush for whole files only; the
    session now re-issues a failing scan from a device copy of the store, coefficient by coefficient up to the throw, so its Dispose
    equals the restatement's writer buffer (po.decode_8bit_partial) on the corrupted corpus of tests/test_partial_flush_gpu.py."""
    from test_partial_flush_gpu import _corrupted_progressive

    files = _corrupted_progressive(70, seed)
    compared = failing = 0
    for k, data in enumerate(files):
        try:
            px, info, err = po.decode_8bit_partial(data)
        except po.OracleError:
            continue  # Identify failed: no scan decoder was ever created
        w = Walk(data)
        st = {"dec": None, "err": None, "fh": None}

        def on_frame(marker, fh):
            if marker != 0xC2:
                raise jl.NotSupportedException("not a progressive frame")
            st["fh"] = fh
            st["dec"] = jl.JpegGpuProgressiveScanDecoder(fh)

        def on_scan(entropy, sh):
            if st["err"] is not None or st["dec"] is None:
                return 0
            try:
                return st["dec"].ProcessScan(entropy, sh, w.quantization_tables(), w.huffman_tables(), w.dri)
            except jl.JpegError as e:
                st["err"] = e
                return 0

        try:
            w.run(on_frame, on_scan)
        except Exception:  # a corrupted header this test's own walk does not read the way the reference does: not the session's business
            if st["dec"] is not None:
                st["dec"].close()
            continue
        if st["dec"] is None:
            continue
        if (st["err"] is None) != (err is None):
            st["dec"].close()
            continue  # (the failure is the marker walk's -- a late error of Decode's loop --, not a scan's)
        if st["err"] is not None and (type(st["err"]).__name__ != err.kind or "at offset" in str(err)):
            st["dec"].close()
            continue  # ("... at offset N ...": the reference's marker walk gave up, not a scan)
        fh = st["fh"]
        if fh.NumberOfComponents != info.ncomp or fh.SamplesPerLine != info.width:
            st["dec"].close()
            continue
        try:
            out = st["dec"].Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, fh.NumberOfComponents)
        except jl.NotSupportedException:
            st["dec"].close()
            continue
        st["dec"].close()
        assert np.array_equal(out, px), (seed, k, None if err is None else err.kind, int((out != px).sum()))
        compared += 1
        failing += err is not None
    assert compared >= 30 and failing >= 15, (compared, failing)


def test_a_registry_entry_under_an_unknown_table_class_is_never_the_table_a_scan_uses():
    """tools/stress_parity.py (session mode, header corruptions), seed 52: a DHT whose class / identifier byte has a flipped bit (0x90:
    class 9) is registered but never FOUND by GetHuffmanTable(isDc, identifier) (JpegDecoder.cs:869-884, exact match) -- the scan
    behind it decodes with the OLD AC table and fails in its first restart interval.  The Python mirror of the C# twin's MarshalTables
    folded the entry into dht[class & 1][identifier & 3] and replaced the live table: the scan "succeeded"."""
    data = read_jpeg("stress/progressive_session_header_16.jpg")
    px, info, err = po.decode_8bit_partial(data)
    assert err is not None and err.kind == "InvalidOperationException"
    w = Walk(data)
    st = {"dec": None, "err": None, "n": 0}

    def on_frame(marker, fh):
        st["fh"], st["dec"] = fh, jl.JpegGpuProgressiveScanDecoder(fh)

    def on_scan(entropy, sh):
        if st["err"] is None:
            try:
                st["dec"].ProcessScan(entropy, sh, w.quantization_tables(), w.huffman_tables(), w.dri)
                st["n"] += 1
            except jl.JpegError as e:
                st["err"] = e
        return 0

    w.run(on_frame, on_scan)
    assert st["n"] == 3 and type(st["err"]).__name__ == "InvalidOperationException"  # the fourth scan, like the whole-file decode
    fh = st["fh"]
    out = st["dec"].Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, fh.NumberOfComponents)
    st["dec"].close()
    assert np.array_equal(out, px)


def _retag_tables(data, q_id, h_id):
    """The chroma tables of a tools/jpegsynth file moved from identifier 1 to `q_id` (DQT + SOF Tq) and `h_id` (DHT + SOS
    Td / Ta): identifiers are 4-bit fields, the reference matches them exactly and has no upper bound of 3
    (JpegDecoder.cs:869-884, 910-925; JpegHuffmanDecodingTable.cs:249-291, JpegQuantizationTable.cs:192-232)."""
    b = bytearray(data)
    p = 2
    while p + 4 <= len(b):
        assert b[p] == 0xFF
        m, ln = b[p + 1], (b[p + 2] << 8) | b[p + 3]
        body = p + 4
        if m == 0xDB and b[body] == 0x01:
            b[body] = q_id
        elif m in (0xC0, 0xC2):
            for i in range(b[body + 5]):
                if b[body + 8 + 3 * i] == 1:
                    b[body + 8 + 3 * i] = q_id
        elif m == 0xC4:
            q = body
            while q < p + 2 + ln:
                n = sum(b[q + 1:q + 17])
                if b[q] & 15 == 1:
                    b[q] = (b[q] & 0xF0) | h_id
                q += 17 + n
        elif m == 0xDA:
            for i in range(b[body]):
                if b[body + 2 + 2 * i] == 0x11:
                    b[body + 2 + 2 * i] = (h_id << 4) | h_id
            p += 2 + ln
            while p + 1 < len(b) and not (b[p] == 0xFF and b[p + 1] not in (0x00, 0xFF) and not 0xD0 <= b[p + 1] <= 0xD7):
                p += 1
            continue
        elif m == 0xD9:
            break
        p += 2 + ln
    return bytes(b)


def test_table_identifiers_above_3_through_every_entry_level():
    """the per-scan mirror handed over identifiers 0..3 only.  A file that defines and selects Huffman tables 5 / 12 and
    quantisation table 7 / 15: whole-file batch (level 1), JpegDecoder.Decode() (level 3), the progressive per-scan session
    (level 2b: the selected identifiers are given slots of their own) -- all equal to the checker."""
    for q_id, h_id in ((7, 5), (15, 12)):
        base = _retag_tables(jpegsynth.encode(120, 88, "420", 80, 5, seed=21), q_id, h_id)
        prog = _retag_tables(jpegsynth.encode(120, 88, "420", 80, 7, seed=22, progressive=True), q_id, h_id)
        for data in (base, prog):
            ref = po.decode_8bit(data)[0]
            assert ref[..., 1].std() > 1  # the chroma really was decoded with its tables
            outs, results = jl.decode_batch([data], jl.FMT_INTERLEAVED_U8)
            assert results[0].status == 0 and np.array_equal(outs[0], ref)
            d = jl.JpegDecoder()
            d.SetInput(data)
            d.Identify()
            out = np.zeros(d.Width * d.Height * 3, np.uint8)
            d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, 3, out))
            d.Decode()
            assert np.array_equal(out.reshape(ref.shape), ref)

        def deliver(dec, fh):
            return dec.Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, 3)

        out, _ = _decode_progressive_scan_by_scan(prog, deliver)
        assert np.array_equal(out, po.decode_8bit(prog)[0])
