"""The partial flush: a progressive file that FAILS in the reference still reaches the writer.  Decode()'s `finally` disposes the
scan decoder (JpegDecoder.cs:545-549), and the progressive one then transforms and flushes whatever its store holds at that moment
(ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:421-470): the scans before the failing one complete, the failing one up to the
coefficient where it threw, later ones never, the component slots as the failing scan left them.  DeviceBatch decodes the scans of a
frame side by side; when a frame has failed it issues the step once more, in file order and with the failing scan on the kernel
that stores coefficient by coefficient (DeviceBatch::replay_failed_progressive).  The exception is the same as before; the output
buffer now is the reference's, sample for sample."""
import io
import os

import numpy as np
import pytest

import jpeglibrary_amd as jl
from golden_util import read_jpeg
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
NAMES = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}


def _corrupted_progressive(n, seed):
    from PIL import Image

    rng = np.random.default_rng(seed)
    files = []
    for _ in range(n):
        w, h = int(rng.integers(16, 260)), int(rng.integers(16, 200))
        yy, xx = np.mgrid[0:h, 0:w]
        px = np.stack([128 + 100 * np.sin(xx / rng.uniform(3, 40) + yy / rng.uniform(3, 40)) for _ in range(3)], -1) + rng.normal(0, rng.uniform(0, 25), (h, w, 3))
        img = Image.fromarray(np.clip(px, 0, 255).astype(np.uint8))
        kw = dict(format="JPEG", quality=int(rng.integers(20, 98)), progressive=True, subsampling=int(rng.integers(0, 3)))
        if rng.random() < 0.2:
            img = img.convert("L")
            kw.pop("subsampling")
        if rng.random() < 0.35:  # restart intervals: the reference stops at the first failing interval, the later ones stay untouched
            kw["restart_marker_blocks"] = int(rng.integers(1, 12))
        buf = io.BytesIO()
        img.save(buf, **kw)
        d = bytearray(buf.getvalue())
        sos = [k for k in range(len(d) - 1) if d[k] == 0xFF and d[k + 1] == 0xDA]
        mode = int(rng.integers(0, 4))
        if mode == 0:  # a flipped bit inside one scan's entropy-coded data
            k = int(rng.integers(0, len(sos)))
            lo, hi = sos[k] + 14, (sos[k + 1] if k + 1 < len(sos) else len(d) - 2)
            if hi > lo:
                d[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:  # truncated inside a scan, EOI kept
            d = d[:int(rng.integers(sos[0] + 14, len(d) - 2))] + b"\xff\xd9"
        elif mode == 2:  # a run of zeros
            p = int(rng.integers(sos[0] + 14, len(d) - 8))
            d[p:p + 6] = bytes(6)
        else:  # bytes deleted
            p = int(rng.integers(sos[0] + 14, len(d) - 8))
            del d[p:p + int(rng.integers(1, 5))]
        files.append(bytes(d))
    return files


def _compare(files, fmt=jl.FMT_INTERLEAVED_U8):
    b = jl.Batch().upload(files, fmt).decode().sync()
    failed = 0
    for i, f in enumerate(files):
        try:
            ref, _, err = po.decode_8bit_partial(f)
        except po.OracleError:
            assert b.image_info(i).status != 0 or b.result(i).status != 0, i  # Identify failed: no scan decoder, nothing to flush
            continue
        r = b.result(i)
        if NAMES.get(r.status) == "NotSupportedException" and r.detail == 6 and (err is None or err.kind != "NotSupportedException"):
            continue  # one of the fences of DESIGN.md 5 (a spectral selection beyond 63, a DC category above 16, ...): refused by design
        assert NAMES.get(r.status) == ("OK" if err is None else err.kind), (i, r.status, r.detail, err)
        if b.image_info(i).status != 0:
            continue
        want = ref if fmt == jl.FMT_INTERLEAVED_U8 else po.ycbcr8_to_rgb(ref, rgba=True, gray=(ref.shape[2] == 1))
        assert np.array_equal(b.output(i), want), (i, "clean" if err is None else str(err), r.detail)
        failed += err is not None
    b.close()
    return failed


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_failing_progressive_files_leave_the_references_partial_output(seed):
    files = _corrupted_progressive(150, seed)
    assert _compare(files) >= 40  # (about half of the edits make the file fail)


def test_partial_flush_as_rgba_and_next_to_clean_files():
    files = _corrupted_progressive(40, 9) + [read_jpeg("progress.jpg"), read_jpeg("cramps.jpg")]
    assert _compare(files, jl.FMT_RGBA_U8) >= 10


@pytest.mark.parametrize("name", ["progressive_partial_flush_161.jpg", "progressive_partial_flush_243.jpg"])
def test_the_data_ending_inside_a_correction_field(name):
    """the reference reads correction bits one at a time: the ones in front of the end of the data are applied before it throws
    (the lane-per-interval kernel used to read them sixteen at a time and apply none)"""
    data = read_jpeg(os.path.join("stress", name))
    ref, _, err = po.decode_8bit_partial(data)
    assert err is not None and "Unexpected end" in str(err)
    b = jl.Batch().upload([data], jl.FMT_INTERLEAVED_U8).decode().sync()
    assert NAMES.get(b.result(0).status) == err.kind
    assert np.array_equal(b.output(0), ref)
    b.close()


def test_a_file_that_fails_in_its_first_scan_header_still_flushes():
    """SOF2, then a first SOS whose Huffman table was never defined: InitDecodeComponents has filled the decoder's slots by the
    time ProcessScan throws, so Dispose() transforms the zeroed store -- every sample is the level shift -- and flushes it."""
    data = bytearray(read_jpeg("progress.jpg"))
    first_dht = data.index(b"\xff\xc4")
    n = (data[first_dht + 2] << 8) | data[first_dht + 3]
    data[first_dht + 1] = 0xEC  # the first DHT segment becomes an APP12 nobody reads
    ref, _, err = po.decode_8bit_partial(bytes(data))
    assert err is not None
    b = jl.Batch().upload([bytes(data)], jl.FMT_INTERLEAVED_U8).decode().sync()
    assert NAMES.get(b.result(0).status) == err.kind
    assert np.array_equal(b.output(0), ref)
    assert n > 0 and ref.any()  # (not the untouched buffer)
    b.close()


def test_the_replay_touches_only_the_failed_frames_whatever_is_asked_for_first(monkeypatch):
    """the replay used to repeat the WHOLE batch scan by scan, was only triggered by jpgpu_batch_result (a caller
    that downloaded first got the un-replayed store), left the batch in the slow launch mode, and could not be switched off."""
    bad = [f for f in _corrupted_progressive(30, 5) if _fails_in_a_scan(f)][:3]
    assert len(bad) == 3
    from tools import jpegsynth

    clean = [read_jpeg("progress.jpg"), jpegsynth.encode(320, 200, "420", 75, 4, seed=8), jpegsynth.encode(300, 180, "444", 80, 0, seed=9),
             read_jpeg("yellowcat_progressive_restart.jpg")]
    files = [clean[0], bad[0], clean[1], bad[1], clean[2], clean[3], bad[2]]
    refs = [po.decode_8bit_partial(f)[0] for f in files]
    # download FIRST (no jpgpu_batch_result before it): the partial flush is there
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode()
    for i in (1, 3, 6, 0, 2, 4, 5):
        assert np.array_equal(b.output(i), refs[i]), i
    assert b.progressive_replays() == 1
    for i, f in enumerate(files):
        assert (b.result(i).status != 0) == (f in bad), i
    # a second decode of the same upload: a whole pass from the batch's own work lists again, then one more replay -- same answers
    b.decode().sync()
    for i in range(len(files)):
        assert (b.result(i).status != 0) == (files[i] in bad), i
        assert np.array_equal(b.output(i), refs[i]), i
    assert b.progressive_replays() == 2
    # the same batch object, another upload: nothing of the replay is left behind
    b.upload(clean, jl.FMT_INTERLEAVED_U8).decode().sync()
    for i, f in enumerate(clean):
        assert b.result(i).status == 0 and np.array_equal(b.output(i), po.decode_8bit(f)[0]), i
    assert b.progressive_fallbacks() == 0
    b.close()
    # switched off: statuses as before, the clean images' outputs as before, no replay
    b2 = jl.Batch().set_partial_flush(False).upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
    for i, f in enumerate(files):
        assert (b2.result(i).status != 0) == (f in bad), i
        if f not in bad:
            assert np.array_equal(b2.output(i), refs[i]), i
    assert b2.progressive_replays() == 0
    b2.close()


def _fails_in_a_scan(f):
    try:
        _, _, err = po.decode_8bit_partial(f)
    except po.OracleError:
        return False
    return err is not None


def test_the_decoder_mirror_flushes_the_partial_store_before_the_exception_leaves():
    """JpegDecoder.Decode() of a failing progressive file (level 3, one image): the mirror collects the scans during the walk and
    decodes them at Dispose; when one fails -- or the walk itself fails behind some scans -- the writer now receives the flush the
    reference's `finally` makes (JpegDecoder.cs:545-549) and then the exception leaves (round 4: "we flush nothing")."""
    done = 0
    for f in _corrupted_progressive(60, 21):
        try:
            px, info, err = po.decode_8bit_partial(f)
        except po.OracleError:
            continue
        if err is None or "at offset" in str(err):
            continue  # (clean, or the walk's own failures: their order against a later scan's is the batch entry points' business)
        d = jl.JpegDecoder()
        d.SetInput(f)
        d.Identify()
        buf = np.zeros(d.Width * d.Height * d.NumberOfComponents, np.uint8)
        d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, d.NumberOfComponents, buf))
        try:
            d.Decode()
            raised = None
        except jl.JpegError as e:
            raised = type(e).__name__
        d.close()
        if raised == "NotSupportedException" and err.kind != raised:
            continue
        assert raised == err.kind, (raised, err)
        assert np.array_equal(buf.reshape(px.shape), px), (str(err), int((buf.reshape(px.shape) != px).sum()))
        done += 1
    assert done >= 12, done
