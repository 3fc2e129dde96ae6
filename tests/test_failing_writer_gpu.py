"""What a FAILING sequential decode leaves in the writer (round 5).  The reference decodes block by block and calls WriteBlock
right behind each block's transform (ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:99-134): when ReadBlockBaseline or the restart
check throws (:153, JpegHuffmanScanDecoder.cs:103-110), every block in front of the failing one has reached the writer and none
behind it; a scan behind a failed scan is never started; and a later scan of a component an earlier scan has written wins
wherever it got to.  The batch decodes the restart intervals and the scans of an image side by side: the Huffman kernels report
the LOWEST failing block of a scan (DevScanStatus::pad[1]), K3 leaves out what lies behind it, and overlapping scans of one image
are transformed in file order.  Checked against the restatement's writer buffer (oracle.pyoracle.decode_8bit_partial)."""
import io
import os

import numpy as np
import pytest

import jpeglibrary_amd as jl
from golden_util import read_jpeg
from oracle import pyoracle as po
from tools import jpegsynth

pytestmark = pytest.mark.gpu
NAMES = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}


def _corrupt(data, rng):
    d = bytearray(data)
    sos = [k for k in range(len(d) - 1) if d[k] == 0xFF and d[k + 1] == 0xDA]
    first = sos[0] + 4 + d[sos[0] + 3]
    mode = int(rng.integers(0, 6))
    pos = int(rng.integers(first, len(d) - 3))
    if mode == 0:
        d[pos] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1:  # truncated, EOI kept
        d = d[:pos] + b"\xff\xd9"
    elif mode == 2:
        d[pos:pos + 6] = bytes(6)
    elif mode == 3:
        del d[pos:pos + int(rng.integers(1, 5))]
    elif mode == 4:  # a marker in the middle of the data: a wrong restart marker, an early EOI, a table
        d[pos:pos + 2] = bytes([0xFF, int(rng.choice([0xD0, 0xD5, 0xD9, 0xC4, 0xE0]))])
    else:
        d[pos:pos + 4] = b"\xff" * 4
    return bytes(d)


def _compare(files, fmt=jl.FMT_INTERLEAVED_U8):
    b = jl.Batch().upload(files, fmt).decode().sync()
    failing = 0
    for i, f in enumerate(files):
        try:
            px, _, err = po.decode_8bit_partial(f)
        except po.OracleError as e:  # Identify failed: nothing decoded, nothing to compare but the class
            assert NAMES.get(b.result(i).status) == e.kind, i
            continue
        res = b.result(i)
        kind = "OK" if err is None else err.kind
        if NAMES.get(res.status) == "NotSupportedException" and res.detail == 6 and kind != "NotSupportedException":
            continue  # one of the fences of DESIGN.md 5 (a middle scan that leaves a byte unread in front of its marker, ...): refused by design
        assert NAMES.get(res.status) == kind, (i, kind, res.status, res.detail)
        if b.image_info(i).status == 0:
            got = b.output(i)
            assert np.array_equal(got, px), (i, kind, int((got != px).sum()), np.argwhere((got != px).any(axis=2))[:3].tolist())
            failing += err is not None
    b.close()
    return failing


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("sub,dri,non", [("420", 4, False), ("420", 0, False), ("444", 1, False), ("422", 3, False), ("444", 0, True), ("444", 5, True), ("420", 7, False)])
def test_failing_baseline_files_leave_the_writer_as_the_reference_does(sub, dri, non, seed):
    import zlib

    rng = np.random.default_rng(zlib.crc32(repr((sub, dri, non, seed)).encode()))  # (not hash(): that differs from process to process)
    base = [jpegsynth.encode(int(rng.integers(40, 300)), int(rng.integers(40, 220)), sub, int(rng.integers(30, 95)), dri, seed=int(rng.integers(1, 1 << 20)),
                             noninterleaved=non) for _ in range(6)]
    files = [_corrupt(base[k % len(base)], rng) for k in range(60)] + base[:2]
    assert _compare(files) >= 15  # (most corruptions make the decode fail somewhere in the middle)


def test_failing_dri0_scan_of_many_subsequences():
    """a DRI = 0 scan long enough for the subsequence decoder (K2S), failing far from its start"""
    rng = np.random.default_rng(11)
    base = jpegsynth.encode(1024, 768, "420", 75, 0, seed=5)
    files = [_corrupt(base, rng) for _ in range(24)] + [base]
    assert _compare(files) >= 2  # (a flipped bit in a scan without restart markers mostly re-synchronises: wrong samples, no exception)


def test_gray_and_pillow_files():
    from PIL import Image

    rng = np.random.default_rng(5)
    files = []
    for k in range(24):
        w, h = int(rng.integers(17, 200)), int(rng.integers(17, 160))
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        kw = dict(format="JPEG", quality=int(rng.integers(20, 95)))
        if k % 3 == 0:
            img = img.convert("L")
        else:
            kw["subsampling"] = int(rng.integers(0, 3))
        if k % 2:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        buf = io.BytesIO()
        img.save(buf, **kw)
        files.append(_corrupt(buf.getvalue(), rng))
    assert _compare(files) >= 6


@pytest.mark.parametrize("name", ["a_later_scan_of_one_component", "b_later_scan_cut_short", "c_component_twice_cut_short"])
def test_overlapping_scans_are_written_in_file_order(name):
    """tests/golden/make_overlapping_scans.py: a later scan of a component an earlier scan wrote -- whole, cut short, and a component
    selected twice with the second scan cut short (round 4 left the earlier scan's transform out in that case, and did not order a
    partial overlap at all)."""
    data = read_jpeg(os.path.join("stress", f"baseline_overlapping_scans_{name}.jpg"))
    neighbour = jpegsynth.encode(64, 48, "420", 75, 2, seed=3)
    _compare([data, neighbour, data, data])
    px, _, err = po.decode_8bit_partial(data)
    for fmt in (jl.FMT_PLANAR_U8,):  # the planes of the same writer state
        b = jl.Batch().upload([data], fmt).decode().sync()
        planes = b.output(0)
        for c in range(3):
            assert np.array_equal(np.asarray(planes[c])[:px.shape[0], :px.shape[1]], px[..., c]), (name, c)
        b.close()


def test_decoder_mirror_leaves_the_callers_buffer_like_the_reference():
    """JpegDecoder.Decode() into a JpegBufferOutputWriter8Bit over the CALLER's buffer (each scan one device call over that canvas):
    when the scan fails inside an MCU, the blocks of that MCU in front of the failing one are in the buffer, and what the caller
    had put there before stays everywhere else."""
    rng = np.random.default_rng(3)
    done = 0
    for sub, dri in (("420", 0), ("420", 4), ("444", 2), ("422", 0)):
        base = jpegsynth.encode(200, 136, sub, 80, dri, seed=21)
        for _ in range(8):
            f = _corrupt(base, rng)
            try:
                px, _, err = po.decode_8bit_partial(f)
            except po.OracleError:
                continue
            d = jl.JpegDecoder()
            d.SetInput(f)
            d.Identify()
            out = np.zeros(d.Width * d.Height * 3, np.uint8)
            d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, 3, out))
            try:
                d.Decode()
                assert err is None
            except jl.JpegError as e:
                assert err is not None and type(e).__name__ == err.kind, (type(e).__name__, err)
            got = out.reshape(d.Height, d.Width, 3)
            assert np.array_equal(got, px), (sub, dri, int((got != px).sum()))
            done += err is not None
    assert done >= 8


def test_a_failing_dri0_scan_over_the_callers_canvas_that_needs_many_rounds():
    """tools/stress_parity.py, session mode, seed 7: a 4:2:2 DRI = 0 scan that fails near its end and whose subsequence states take
    more rounds than the device-driven budget.  The optimistic output stage then ran on unconverged states and wrote chroma blocks
    of MCUs the scan never reached into the caller's buffer; the correct second pass leaves a canvas alone where the scan did not
    get to, so they stayed.  Over a caller's canvas the rounds are host-checked now."""
    data = read_jpeg(os.path.join("stress", "baseline_failing_422_canvas_54.jpg"))
    px, _, err = po.decode_8bit_partial(data)
    assert err is not None
    d = jl.JpegDecoder()
    d.SetInput(data)
    d.Identify()
    out = np.zeros(d.Width * d.Height * 3, np.uint8)
    d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, 3, out))
    with pytest.raises(jl.InvalidDataException):
        d.Decode()
    assert np.array_equal(out.reshape(px.shape), px)
    b = jl.Batch().upload([data], jl.FMT_INTERLEAVED_U8).decode().sync()  # (the batch's own buffer: the second pass rewrites everything)
    assert np.array_equal(b.output(0), px) and b.subseq_fallbacks() == 1
    b.close()


def test_an_arbitrary_writer_receives_the_blocks_in_front_of_the_throw_and_no_others():
    """JpegDecoder.Decode() into a caller's own JpegBlockOutputWriter (WriteBlock calls replayed on the host from the device's int16
    planes): for a failing scan the calls stop at the block the reference threw in -- inside the failing restart interval and the
    failing MCU too (jpgpu_image_result.error_block; round 4 replayed whole restart intervals only)."""

    class Sink(jl.JpegBlockOutputWriter):
        def __init__(self, w, h, c):
            self.px = np.zeros((h, w, c), np.uint8)
            self.calls = 0

        def WriteBlock(self, blockRef, componentIndex, x, y):  # noqa: N802,N803
            self.calls += 1
            h, w, _ = self.px.shape
            if x >= w or y >= h:
                return
            ww, wh = min(w - x, 8), min(h - y, 8)
            self.px[y:y + wh, x:x + ww, componentIndex] = np.clip(np.asarray(blockRef, np.int16).reshape(8, 8)[:wh, :ww], 0, 255)

    rng = np.random.default_rng(17)
    done = 0
    for sub, dri in (("420", 5), ("444", 0), ("422", 3)):
        base = jpegsynth.encode(120, 88, sub, 80, dri, seed=31)
        for _ in range(10):
            f = _corrupt(base, rng)
            try:
                px, _, err = po.decode_8bit_partial(f)
            except po.OracleError:
                continue
            d = jl.JpegDecoder()
            d.SetInput(f)
            d.Identify()
            sink = Sink(d.Width, d.Height, 3)
            d.SetOutputWriter(sink)
            try:
                d.Decode()
                assert err is None
            except jl.JpegError as e:
                if type(e).__name__ == "NotSupportedException" and (err is None or err.kind != "NotSupportedException"):
                    d.close()
                    continue
                assert err is not None and type(e).__name__ == err.kind
            d.close()
            assert np.array_equal(sink.px, px), (sub, dri, int((sink.px != px).sum()))
            done += err is not None
    assert done >= 8
