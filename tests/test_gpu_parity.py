"""Parity tests proper: the HIP path (through the C ABI) against the oracle and the reference's golden vectors.

Bit-exact everywhere: coefficients (Huffman stage), samples (IDCT stage), and every output layout.
Mirrors the reference's own decode tests (tests/JpegLibrary.Tests/Decoder/HuffmanSequentialDecodeTests.cs:23-43).
"""
import os

import numpy as np
import pytest

import jpeglibrary_amd as jl
from golden_util import load_reference_buffer, read_jpeg
from oracle import pyoracle as po
from tools import jpegsynth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ reference goldens

@pytest.mark.parametrize("name", ["cramps.jpg", "lake.jpg", "testorig12.jpg", "progress.jpg", "yellowcat_progressive_restart.jpg"])
def test_decode_matches_reference_golden(name):
    jpeg_bytes = read_jpeg(name)
    decoder = jl.JpegDecoder()
    decoder.SetInput(jpeg_bytes)
    decoder.Identify()
    buffer = np.zeros(decoder.Width * decoder.Height * 4, dtype=np.uint16)
    output_writer = jl.JpegExtendingOutputWriter(decoder.Width, decoder.Height, 4, decoder.Precision, buffer)
    decoder.SetOutputWriter(output_writer)
    decoder.Decode()
    reference = load_reference_buffer(name, decoder.Width, decoder.Height, decoder.NumberOfComponents)
    assert np.array_equal(reference.reshape(-1), buffer)


def test_extended_u16_device_format_matches_the_reference_goldens():
    """JPGPU_FMT_EXTENDED_U16 = the buffer of the reference tests' JpegExtendingOutputWriter (componentCount 4), produced on
    the device: all five golden assets in ONE batch, one download each, compared with the reference's own PNG dumps --
    and with the WriteBlock-callback path of the decoder mirror on a 4:2:2 and a clipped 12-bit-like case."""
    names = ["cramps.jpg", "lake.jpg", "testorig12.jpg", "progress.jpg", "yellowcat_progressive_restart.jpg"]
    files = [read_jpeg(n) for n in names]
    b = jl.Batch().upload(files, jl.FMT_EXTENDED_U16).decode().sync()
    for i, n in enumerate(names):
        assert b.result(i).status == 0, n
        info = b.image_info(i)
        reference = load_reference_buffer(n, info.width, info.height, info.num_components)
        assert np.array_equal(b.output(i), reference), n
    b.close()
    # synthetic cases against the oracle's own 16-bit sink (samples outside [0, 2^P - 1] included: Q30 overshoots)
    extra = [jpegsynth.encode(331, 177, "422", 30, 3, seed=7), jpegsynth.encode(100, 75, "gray", 60, 1, seed=8),
             jpegsynth.encode(96, 64, "444", 75, 0, seed=5, noninterleaved=True), jpegsynth.encode(17, 9, "420", 20, 1, seed=9)]
    b = jl.Batch().upload(extra, jl.FMT_EXTENDED_U16).decode().sync()
    for i, f in enumerate(extra):
        assert b.result(i).status == 0
        assert np.array_equal(b.output(i), po.decode_16bit(bytes(f), component_count=4)[0]), i
    b.close()


@pytest.mark.parametrize("name", ["cramps.jpg", "lake.jpg", "HETissueSlide.jpg", "progress.jpg", "yellowcat_progressive_restart.jpg"])
def test_buffer8_writer_matches_oracle(name):
    """The app writer (JpegBufferOutputWriter8Bit) fast path: interleaved u8 produced on the GPU."""
    data = read_jpeg(name)
    decoder = jl.JpegDecoder()
    decoder.SetInput(data)
    decoder.Identify()
    out = np.zeros(decoder.Width * decoder.Height * decoder.NumberOfComponents, dtype=np.uint8)
    decoder.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(decoder.Width, decoder.Height, decoder.NumberOfComponents, out))
    decoder.Decode()
    ref, _ = po.decode_8bit(data)
    assert np.array_equal(out.reshape(ref.shape), ref)


def test_buffer8_writer_with_foreign_geometry_replays_blocks():
    """componentCount = 4 for a 3-component image (as the reference's tests do): falls back to WriteBlock replay."""
    data = read_jpeg("lake.jpg")
    decoder = jl.JpegDecoder()
    decoder.SetInput(data)
    decoder.Identify()
    out = np.zeros(decoder.Width * decoder.Height * 4, dtype=np.uint8)
    decoder.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(decoder.Width, decoder.Height, 4, out))
    decoder.Decode()
    ref, _ = po.decode_8bit(data, component_count=4)
    assert np.array_equal(out.reshape(ref.shape), ref)


# ------------------------------------------------------------------------------------------------ synthetic matrix

CASES = [
    # (w, h, subsampling, quality, dri)
    (512, 512, "444", 75, 0),      # BASELINE config 1
    (640, 368, "420", 75, 4),      # config-2 shape, small
    (640, 368, "420", 75, 0),      # config-3 shape, small
    (480, 272, "420", 90, 4),      # config-4 shape (Q90), height not a multiple of 16
    (331, 177, "422", 80, 3),      # odd size, 4:2:2
    (100, 75, "gray", 60, 1),      # grayscale, DRI=1
    (17, 9, "420", 75, 1),         # tiny, partial MCUs in both directions
    (341, 486, "420", 50, 22),     # DRI = one MCU row
    (64, 64, "444", 100, 5),       # Q100: all-ones tables, long codes
    (1920, 1080, "420", 90, 4),    # one full config-4 image
]


def _decode_gpu(files, fmt):
    b = jl.Batch().upload(files, fmt).decode().sync()
    return b


@pytest.mark.parametrize("w,h,ss,q,dri", CASES)
def test_interleaved_u8_matches_oracle(w, h, ss, q, dri):
    data = jpegsynth.encode(w, h, ss, q, dri, seed=w * 31 + h)
    b = _decode_gpu([data], jl.FMT_INTERLEAVED_U8)
    res = b.result(0)
    assert (res.status, res.detail) == (0, 0)
    ref, info = po.decode_8bit(data)
    out = b.output(0)
    assert out.shape == ref.shape
    assert np.array_equal(out, ref), f"{(out != ref).sum()} mismatching samples"
    assert res.terminator == 0xD9 and res.decoded_mcus == b.image_info(0).mcus_per_line * b.image_info(0).mcus_per_column


@pytest.mark.parametrize("w,h,ss,q,dri", CASES[:8])
def test_coefficients_match_oracle(w, h, ss, q, dri):
    """Huffman stage alone: the coefficient buffer equals the reference's ReadBlockBaseline output, block for block."""
    data = jpegsynth.encode(w, h, ss, q, dri, seed=7 + w)
    b = jl.Batch().upload([data], jl.FMT_PLANAR_I16).run_entropy().sync()
    assert b.result(0).status == 0
    coefs = b.coefficients(0)
    ref, _ = po.decode_coefficients(data)
    assert coefs.shape == ref.shape
    assert np.array_equal(coefs, ref)


@pytest.mark.parametrize("w,h,ss,q,dri", CASES[:8])
def test_planar_i16_matches_writeblock_arguments(w, h, ss, q, dri):
    """O1: unclamped int16 planes == the blocks WriteBlock receives (before chroma expansion)."""
    data = jpegsynth.encode(w, h, ss, q, dri, seed=11 + h)
    b = _decode_gpu([data], jl.FMT_PLANAR_I16)
    assert b.result(0).status == 0
    planes = b.output(0)
    info = b.image_info(0)
    # oracle: collect the un-expanded blocks: dequant/IDCT of its own coefficient tap
    coefs, comps = po.decode_coefficients(data)
    oinfo, _ = po.identify(data)
    # rebuild planes from the oracle's WriteBlock calls: for components with hs=vs=1 the calls are the blocks themselves;
    # for sub-sampled ones the (0,0) sample of each 2x2 (or 2x1) replica group is the native sample.
    calls, _ = po.decode_blocks(data)
    max_h = max(oinfo.comp[i].h for i in range(oinfo.ncomp))
    max_v = max(oinfo.comp[i].v for i in range(oinfo.ncomp))
    ref_planes = [np.zeros((p.shape[0], p.shape[1]), np.int16) for p in planes]
    full = [np.zeros((info.mcus_per_column * max_v * 8, info.mcus_per_line * max_h * 8), np.int16) for _ in planes]
    for ci, x, y, blk in calls:
        full[ci][y:y + 8, x:x + 8] = blk.reshape(8, 8)
    for ci in range(oinfo.ncomp):
        hs, vs = max_h // oinfo.comp[ci].h, max_v // oinfo.comp[ci].v
        ref_planes[ci] = full[ci][::vs, ::hs]
    for ci in range(oinfo.ncomp):
        assert np.array_equal(planes[ci], ref_planes[ci]), ci


def test_planar_u8_is_clamped_planar_i16():
    data = jpegsynth.encode(352, 240, "420", 75, 4, seed=99)
    p8 = _decode_gpu([data], jl.FMT_PLANAR_U8).output(0)
    p16 = _decode_gpu([data], jl.FMT_PLANAR_I16).output(0)
    for a, b in zip(p8, p16):
        assert np.array_equal(a, np.clip(b, 0, 255).astype(np.uint8))


def test_idct_stage_on_random_coefficients():
    """IDCT kernel alone on adversarial coefficient blocks (full int16 range products, dense blocks)."""
    data = jpegsynth.encode(256, 256, "444", 75, 0, seed=5)
    b = jl.Batch().upload([data], jl.FMT_PLANAR_I16)
    info = b.image_info(0)
    rng = np.random.default_rng(123)
    n = info.total_blocks
    coefs = np.zeros((n, 64), np.int16)
    coefs[: n // 2] = rng.integers(-1024, 1024, size=(n // 2, 64))
    coefs[n // 2:, :10] = rng.integers(-2048, 2048, size=(n - n // 2, 10))
    coefs[0] = 32767
    coefs[1] = -32768
    b.set_coefficients(0, coefs)
    b.run_idct().sync()
    planes = b.output(0)
    # oracle per block; quant tables from a reference-side parse of the same file
    from jpeglibrary_amd import _capi  # noqa: F401
    qts = _quant_tables(data)
    bpm = info.blocks_per_mcu
    for bi in rng.choice(n, size=600, replace=False).tolist() + [0, 1]:
        mcu, k = divmod(bi, bpm)
        ci = k  # 4:4:4: block k of the MCU is component k
        q = qts[0] if ci == 0 else qts[1]
        ref = po.block_dequant_idct_shift(coefs[bi], q, 128).reshape(8, 8)
        my, mx = divmod(mcu, info.mcus_per_line)
        got = planes[ci][my * 8:my * 8 + 8, mx * 8:mx * 8 + 8]
        assert np.array_equal(got, ref), bi


def _quant_tables(data):
    """DQT payloads (zig-zag order) by table id, read straight from the file."""
    i, out = 2, {}
    while i < len(data):
        m, ln = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        if m == 0xDB:
            p = i + 4
            while p < i + 2 + ln:
                pq, tq = data[p] >> 4, data[p] & 15
                if pq == 0:
                    out[tq] = np.frombuffer(data[p + 1:p + 65], np.uint8).astype(np.uint16)
                    p += 65
                else:
                    out[tq] = np.frombuffer(data[p + 1:p + 129], ">u2").astype(np.uint16)
                    p += 129
        if m == 0xDA:
            break
        i += 2 + ln
    return out


def test_batch_of_mixed_images():
    files = [jpegsynth.encode(w, h, ss, q, dri, seed=i) for i, (w, h, ss, q, dri) in enumerate(CASES[:8])]
    files += [read_jpeg("cramps.jpg"), read_jpeg("lake.jpg")]
    outs, results = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
    for f, out, res in zip(files, outs, results):
        assert res.status == 0
        ref, _ = po.decode_8bit(f)
        assert np.array_equal(out, ref)


def test_pillow_encoded_files_with_restart_markers():
    """Files from an independent encoder (libjpeg-turbo via Pillow): optimised + standard tables, DRI in blocks/rows."""
    import io

    from PIL import Image

    rng = np.random.default_rng(1)
    img = Image.fromarray(rng.integers(0, 255, size=(200, 300, 3), dtype=np.uint8).astype(np.uint8)).resize((301, 203))
    for kwargs in (dict(quality=75, subsampling=2, restart_marker_blocks=4), dict(quality=85, subsampling=1, restart_marker_rows=1),
                   dict(quality=60, subsampling=0, optimize=True), dict(quality=90, subsampling=2, optimize=True, restart_marker_blocks=1)):
        buf = io.BytesIO()
        img.save(buf, "JPEG", **kwargs)
        data = buf.getvalue()
        outs, results = jl.decode_batch([data], jl.FMT_INTERLEAVED_U8)
        assert results[0].status == 0, kwargs
        ref, _ = po.decode_8bit(data)
        assert np.array_equal(outs[0], ref), kwargs


# ------------------------------------------------------------------------------------------------ malformed streams

def _status_of_oracle(data):
    try:
        po.decode_8bit(data)
        return "OK"
    except po.OracleError as e:
        return e.kind


def _status_of_gpu(data):
    outs, results = jl.decode_batch([data], jl.FMT_INTERLEAVED_U8)
    return {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException"}[results[0].status], results[0]


def test_truncated_and_corrupted_streams_fail_like_the_reference():
    good = jpegsynth.encode(160, 96, "420", 75, 2, seed=42)
    sos = good.index(b"\xff\xda")
    entropy0 = sos + 14
    rsts = [i for i in range(entropy0, len(good) - 1) if good[i] == 0xFF and 0xD0 <= good[i + 1] <= 0xD7]
    assert len(rsts) == 29
    cases = {
        "truncated_mid_interval": good[:rsts[10] + 9],
        "truncated_at_restart": good[:rsts[10]] + b"\xff\xd9",   # EOI at a restart boundary: early return, no error
        "rst_removed": good[:rsts[5]] + good[rsts[5] + 2:],
        "garbage_before_rst": good[:rsts[7]] + b"\x12\x34" + good[rsts[7]:],
        "marker_in_data": good[:rsts[3] + 5] + b"\xff\xc4" + good[rsts[3] + 7:],
        "no_eoi": good[:-2],
    }
    cases.update({
        # Identify() passes (EOI present) and the failure happens inside the scan, on the device
        "cut_mid_interval_eoi": good[:rsts[10] + 9] + b"\xff\xd9",
        "cut_mid_interval_eoi2": good[:rsts[10] + 30] + b"\xff\xd9",
        "cut_last_interval_eoi": good[:len(good) - 12] + b"\xff\xd9",
        "interval_zeroed": good[:rsts[4] + 2] + bytes(rsts[5] - rsts[4] - 2) + good[rsts[5]:],
        "interval_ones": good[:rsts[4] + 2] + b"\xff\x00" * ((rsts[5] - rsts[4] - 2) // 2) + good[rsts[5]:],
        "extra_rst_mid": good[:rsts[6] + 10] + b"\xff\xd3" + good[rsts[6] + 10:],
        "non_rst_marker_as_restart": good[:rsts[8]] + b"\xff\xc8" + good[rsts[8] + 2:],
        "trailing_rst_before_eoi": good[:-2] + b"\xff\xd5\xff\xd9",
        "fill_bytes_before_rst": good[:rsts[9]] + b"\xff\xff\xff" + good[rsts[9]:],
    })
    from jpeglibrary_amd import _capi
    for name, data in cases.items():
        try:
            po.decode_8bit(data)
            ref, ref_msg = "OK", ""
        except po.OracleError as e:
            ref, ref_msg = e.kind, e.message
        mine, res = _status_of_gpu(data)
        assert mine == ref, (name, ref, ref_msg, mine, res.detail)
        if res.detail in (1, 2, 3, 4):  # device-reported failures carry the reference's exception text
            assert _capi.lib.jpgpu_detail_string(res.detail).decode() == ref_msg, (name, ref_msg, res.detail)
        if ref == "OK":
            out = jl.decode_batch([data], jl.FMT_INTERLEAVED_U8)[0][0]
            assert np.array_equal(out, po.decode_8bit(data)[0]), name  # MCUs behind an early EOI: zero, as in a fresh buffer
    # early EOI decodes exactly the intervals before it and leaves the rest of the caller's buffer untouched
    data = cases["truncated_at_restart"]
    d = jl.JpegDecoder()
    d.SetInput(data)
    d.Identify()
    out = np.full(160 * 96 * 3, 77, np.uint8)
    d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(160, 96, 3, out))
    d.Decode()
    ref = np.full((96, 160, 3), 77, np.uint8)
    L = po.lib()
    import ctypes as C
    err = C.create_string_buffer(256)
    info = po.Info()
    assert L.jref_decode_to_8bit(data, len(data), 3, ref.ctypes.data, ref.size, C.byref(info), err, 256) == 0
    assert np.array_equal(out.reshape(96, 160, 3), ref)


def test_dri_latched_at_sof_like_the_reference():
    """without Identify(), a DRI that follows SOF is not seen by the baseline scan decoder."""
    data = jpegsynth.encode(64, 48, "420", 75, 2, seed=8)   # jpegsynth writes DRI after SOF, like libjpeg
    d = jl.JpegDecoder()
    d.SetInput(data)
    out = np.zeros(64 * 48 * 3, np.uint8)
    d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(64, 48, 3, out))
    with pytest.raises(jl.JpegError):
        d.Decode()   # restart interval still 0 at SOF time -> the first RST marker ends the bit stream
    d2 = jl.JpegDecoder()
    d2.SetInput(data)
    d2.Identify()
    d2.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(64, 48, 3, out))
    d2.Decode()
    ref, _ = po.decode_8bit(data)
    assert np.array_equal(out.reshape(ref.shape), ref)


def test_decode_scan_level2_entry():
    """jpgpu_decode_scan with pre-parsed tables (what a C# JpegScanDecoder replacement would P/Invoke)."""
    import ctypes as C

    from jpeglibrary_amd import _capi

    data = jpegsynth.encode(96, 80, "420", 75, 3, seed=21)
    frame = _capi.Frame(96, 80, 8, 3, 0xC0, 0)
    for i, (cid, h, v, tq) in enumerate([(1, 2, 2, 0), (2, 1, 1, 1), (3, 1, 1, 1)]):
        frame.comp[i] = _capi.FrameComponent(cid, h, v, tq)
    scan = _capi.Scan(3, 0, 63, 0, 0)
    for i, (sel, td, ta) in enumerate([(1, 0, 0), (2, 1, 1), (3, 1, 1)]):
        scan.comp[i] = _capi.ScanComponent(sel, td, ta, 0)
    qts = _quant_tables(data)
    qt = np.zeros((4, 64), np.uint16)
    qt[0], qt[1] = qts[0], qts[1]
    present = np.array([1, 1, 0, 0], np.uint8)
    dht = ((_capi.Dht * 4) * 2)()
    i = 2
    while True:
        m, ln = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        if m == 0xC4:
            p = i + 4
            while p < i + 2 + ln:
                tc, th = data[p] >> 4, data[p] & 15
                bits = data[p + 1:p + 17]
                n = sum(bits)
                e = dht[tc][th]
                e.present = 1
                e.num_values = n
                for k in range(16):
                    e.bits[k] = bits[k]
                for k in range(n):
                    e.values[k] = data[p + 17 + k]
                p += 17 + n
        if m == 0xDA:
            entropy = data[i + 2 + ln:]
            break
        i += 2 + ln
    ctx = jl.default_context()
    out = np.zeros(96 * 80 * 3, np.uint8)
    res = _capi.ImageResult()
    consumed = C.c_size_t()
    ebuf = np.frombuffer(entropy, np.uint8)
    rc = _capi.lib.jpgpu_decode_scan(ctx._h, C.byref(frame), C.byref(scan), qt.ctypes.data, present.ctypes.data, C.addressof(dht), 3,
                                     ebuf.ctypes.data, ebuf.size, jl.FMT_INTERLEAVED_U8, out.ctypes.data, out.size, C.byref(res), C.byref(consumed))
    assert rc == 0, ctx.last_error()
    ref, _ = po.decode_8bit(data)
    assert np.array_equal(out.reshape(ref.shape), ref)
    assert consumed.value == len(entropy) - 2  # reader left just before EOI


def test_dri0_uses_self_synchronising_decoder():
    """Scans without restart intervals are decoded by parallel 1024-bit subsequences that synchronise in a few rounds;
    the interval decoder (JPGPU_NO_SUBSEQ=1 path) and the oracle give the same coefficients."""
    data = jpegsynth.encode(1024, 768, "420", 75, 0, seed=77)
    b = jl.Batch().upload([data], jl.FMT_INTERLEAVED_U8).decode().sync()
    assert b.result(0).status == 0
    rounds = b.subseq_rounds()
    assert 2 <= rounds <= 16, rounds
    coefs = jl.Batch().upload([data], jl.FMT_PLANAR_I16).run_entropy().sync().coefficients(0)
    ref, _ = po.decode_coefficients(data)
    assert np.array_equal(coefs, ref)
    assert np.array_equal(b.output(0), po.decode_8bit(data)[0])
    # truncated DRI = 0 stream (EOI kept so that Identify passes): same exception class as the reference
    cut = data[:len(data) // 2] + b"\xff\xd9"
    assert _status_of_gpu(cut)[0] == _status_of_oracle(cut) != "OK"


def test_dri0_rounds_are_enqueued_ahead_and_checked_when_the_caller_waits(monkeypatch):
    """Round 5: jpgpu_batch_decode does not wait for anything on a DRI = 0 batch.  The K2S rounds are enqueued ahead (16 the
    first time, then as many as the upload's last decode used), a round behind the converged one leaves at once, and sync()
    reads whether they sufficed; if not, it issues the step again with the host reading the counts between rounds.
    Same coefficients and samples either way (ref: ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:99-134, 179-222)."""
    files = [jpegsynth.encode(1024, 768, "420", 75, 0, seed=77 + i) for i in range(3)] + [jpegsynth.encode(640, 368, "420", 75, 4, seed=5),
                                                                                      jpegsynth.encode(416, 240, "444", 90, 0, seed=6)]
    refs = [po.decode_8bit(f)[0] for f in files]
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().decode().sync()  # two decodes in flight, one check
    rounds = b.subseq_rounds()
    assert 2 <= rounds <= 16 and b.subseq_fallbacks() == 0, (rounds, b.subseq_fallbacks())
    for i, r in enumerate(refs):
        assert b.result(i).status == 0 and np.array_equal(b.output(i), r), i
    b.decode().sync()  # the learned budget: exactly `rounds` rounds, the last of them the one that changes nothing
    assert b.subseq_rounds() == rounds and b.subseq_fallbacks() == 0
    for i, r in enumerate(refs):
        assert np.array_equal(b.output(i), r), i
    # a budget that cannot suffice: noticed at the sync, the step repeated with host-checked rounds, the upload stays that way
    monkeypatch.setenv("JPGPU_SUBSEQ_BUDGET", "2")
    b2 = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
    assert b2.subseq_fallbacks() == 1 and b2.subseq_rounds() >= rounds
    for i, r in enumerate(refs):
        assert b2.result(i).status == 0 and np.array_equal(b2.output(i), r), i
    b2.decode().sync()
    assert b2.subseq_fallbacks() == 1
    assert np.array_equal(b2.output(0), refs[0])
    # ... the entropy stage alone, and the output stage issued behind it without a sync in between
    b3 = jl.Batch().upload(files, jl.FMT_PLANAR_I16).run_entropy().sync()
    assert b3.subseq_fallbacks() == 1
    assert np.array_equal(b3.coefficients(0), po.decode_coefficients(files[0])[0])
    b4 = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).run_entropy().run_idct().sync()
    assert b4.subseq_fallbacks() == 1
    for i, r in enumerate(refs):
        assert np.array_equal(b4.output(i), r), i


def test_marker_index_in_one_pass_and_its_bounded_wait(monkeypatch):
    """Round 5: K1 reads and classifies the entropy segments once -- a workgroup per four 4 KiB chunks publishes its summary and finds the
    running sums of the groups in front of it by looking back (marker_onepass_kernel) -- where round 4 counted, prefixed and wrote in
    three kernels that read them twice.  Same ends / unstuffed data / statuses (ref: JpegBitReader.cs:95-138, the restart hand-off of
    JpegHuffmanBaselineScanDecoder.cs:139-163): every case below against the restatement, in one batch and decode after decode.
    A spin budget of zero makes any wait a give-up: the group then counts its predecessors itself -- same results."""
    files = [jpegsynth.encode(1024, 768, "420", 75, 4, seed=31), jpegsynth.encode(640, 368, "444", 90, 1, seed=32), jpegsynth.encode(800, 600, "422", 60, 0, seed=33),
             jpegsynth.encode(333, 211, "444", 60, 8, seed=10, noninterleaved=True), read_jpeg("progress.jpg"), read_jpeg("yellowcat_progressive_restart.jpg"),
             jpegsynth.encode(1920, 1080, "420", 90, 4, seed=34)]
    files.append(files[0][:len(files[0]) // 2] + b"\xff\xd9")  # truncated: the terminator in the middle of the scan
    refs = []
    for f in files:
        px, _, err = po.decode_8bit_partial(f)
        refs.append((px, err))
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
    for _ in range(3):  # (the descriptors are not cleared between decodes: tags)
        b.decode().sync()
        for i, (px, err) in enumerate(refs):
            assert (b.result(i).status != 0) == (err is not None), i
            assert np.array_equal(b.output(i), px), i
    assert b.marker_fallbacks() == 0
    b.close()
    # no patience at all: a group that finds a predecessor's record missing counts the data in front of it itself and goes on
    # (whether any group HAD to wait is the machine's business: the counter says how many waits saw it happen)
    monkeypatch.setenv("JPGPU_K1_SPIN_BUDGET", "0")
    b = jl.Batch().upload(files * 4, jl.FMT_INTERLEAVED_U8)
    for rep in range(3):
        b.decode().sync()
        assert 0 <= b.marker_fallbacks() <= rep + 1
        for i in range(len(files) * 4):
            px, err = refs[i % len(files)]
            assert (b.result(i).status != 0) == (err is not None), i
            assert np.array_equal(b.output(i), px), i
    b.close()
    monkeypatch.delenv("JPGPU_K1_SPIN_BUDGET")
    monkeypatch.setenv("JPGPU_K1_THREE_PASS", "1")
    outs, res = jl.decode_batch(files)
    for i, (px, err) in enumerate(refs):
        assert np.array_equal(outs[i], px), i


def test_marker_index_tile_writer_on_dense_markers_at_every_alignment():
    """Round 6: a group of the one-pass marker index in the middle of its scan assembles its udata bytes in one 16 KiB tile -- dropped bytes
    taken out of the lane's four words, five aligned LDS ORs per lane, the second FF of a marker pair ORed over the code byte's place
    (k1_markers.hip).  Its corner cases are places: a marker's FF in the last byte of a lane / a wave's 1 KiB / a chunk / a group, a
    stuffed zero in a lane's first byte, two dropped bytes in one lane.  Files with a restart marker every MCU (tens of thousands of them
    over ~1 MB of noisy entropy data, i.e. at every phase of 16 / 1024 / 4096 / 16384 many times over), shifted through all 16 byte
    alignments of the segment by a comment in front -- every sample against the restatement (ref: JpegBitReader.cs:95-138, the restart
    hand-off of JpegHuffmanBaselineScanDecoder.cs:139-163)."""
    files = []
    for k, (w, h, ss, q, dri) in enumerate(((2048, 1536, "444", 95, 1), (1920, 1080, "420", 98, 1), (2048, 1024, "422", 92, 2), (1600, 1200, "444", 97, 0))):
        f = jpegsynth.encode(w, h, ss, q, dri, seed=600 + k)
        for pad in (0, 5, 10, 15) if k == 0 else ((k * 3) % 16, (k * 7 + 1) % 16):
            com = b"\xff\xfe" + (pad + 2).to_bytes(2, "big") + bytes(pad)
            files.append(f[:2] + com + f[2:])
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
    for _ in range(2):
        b.decode().sync()
        for i, f in enumerate(files):
            ref, _ = po.decode_8bit(f)
            assert b.result(i).status == 0, i
            assert np.array_equal(b.output(i), ref), i
    assert b.marker_fallbacks() == 0
    b.close()


def test_full_size_properties_4k():
    """BASELINE config-2 geometry at full size: every image of a small 4K batch is bit-exact vs the oracle, and the
    DRI=4 and DRI=0 encodings of the same pixels decode to identical output (restart markers carry no information)."""
    files4 = [jpegsynth.encode(3840, 2160, "420", 75, 4, seed=1000 + i) for i in range(2)]
    files0 = [jpegsynth.encode(3840, 2160, "420", 75, 0, seed=1000 + i) for i in range(2)]
    outs4, res4 = jl.decode_batch(files4, jl.FMT_INTERLEAVED_U8)
    outs0, res0 = jl.decode_batch(files0, jl.FMT_INTERLEAVED_U8)
    for i in range(2):
        assert res4[i].status == 0 and res0[i].status == 0
        assert np.array_equal(outs4[i], outs0[i])
        # each encoding against the checker's decode of ITS OWN bytes (not only transitively through the other one)
        assert np.array_equal(outs4[i], po.decode_8bit(files4[i])[0]), i
        assert np.array_equal(outs0[i], po.decode_8bit(files0[i])[0]), i


def test_multi_scan_and_four_component_files():
    """Non-interleaved baseline files (three single-component scans = three scan jobs writing one image) and a
    4-component (CMYK) file exercise the generic bytewise output path and the multi-scan plumbing."""
    import io

    from PIL import Image

    files = [
        jpegsynth.encode(120, 88, "444", 80, 0, seed=9, noninterleaved=True),
        jpegsynth.encode(120, 88, "444", 80, 4, seed=9, noninterleaved=True),
        jpegsynth.encode(333, 211, "444", 60, 8, seed=10, noninterleaved=True),
    ]
    rng = np.random.default_rng(3)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 255, size=(70, 90, 4), dtype=np.uint8), mode="CMYK").save(buf, "JPEG", quality=80)
    files.append(buf.getvalue())
    outs, results = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
    for f, out, res in zip(files, outs, results):
        assert res.status == 0
        assert np.array_equal(out, po.decode_8bit(f)[0])
    # the interleaved and non-interleaved encodings of the same pixels decode identically
    assert np.array_equal(outs[0], jl.decode_batch([jpegsynth.encode(120, 88, "444", 80, 0, seed=9)])[0][0])
    # reference quirk: when the MCU count is a multiple of DRI the restart check after the last MCU demands EOI/RSTn,
    # so a scan followed by another SOS fails with "Expect restart marker." (BaselineScanDecoder.cs:139-154)
    quirk = jpegsynth.encode(120, 88, "444", 80, 5, seed=9, noninterleaved=True)  # 165 MCUs = 33 x 5
    assert _status_of_oracle(quirk) == "InvalidOperationException"
    assert _status_of_gpu(quirk)[0] == "InvalidOperationException"
    # same files through the JpegDecoder mirror (each SOS is one GPU scan decode into the caller's buffer)
    d = jl.JpegDecoder()
    d.SetInput(files[1])
    d.Identify()
    out = np.zeros(d.Width * d.Height * 3, np.uint8)
    d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, 3, out))
    d.Decode()
    assert np.array_equal(out.reshape(88, 120, 3), po.decode_8bit(files[1])[0])


@pytest.mark.parametrize("name", ["progress.jpg", "yellowcat_progressive_restart.jpg"])
def test_progressive_idct_pass_on_gpu(name):
    """BASELINE config 5 shape: multi-scan (SOF2) coefficient accumulate, then ONE dequantise + IDCT + flush pass on the
    GPU (what JpegHuffmanProgressiveScanDecoder.Dispose + JpegBlockAllocator.Flush do).  The accumulated store comes
    from the oracle's progressive entropy decoder; the GPU output must equal the reference's golden PNG dump."""
    data = read_jpeg(name)
    info, blocks, quant = po.decode_progressive_store(data)
    comps = [(info.comp[i].identifier, info.comp[i].h, info.comp[i].v, i) for i in range(info.ncomp)]  # tq := component index
    max_h, max_v = max(c[1] for c in comps), max(c[2] for c in comps)
    mcus_x = -(-info.width // (8 * max_h))
    mcus_y = -(-info.height // (8 * max_v))
    qt = np.zeros((1, 4, 64), np.uint16)
    for ci in range(info.ncomp):
        qt[0, ci] = quant[ci]
    # MCU scan order: MCU raster, component order, block raster inside the MCU; blocks outside a component's store are zero
    coefs = []
    for my in range(mcus_y):
        for mx in range(mcus_x):
            for ci, (_, h, v, _) in enumerate(comps):
                for y in range(v):
                    for x in range(h):
                        coefs.append(blocks[ci].get((mx * h + x, my * v + y), np.zeros(64, np.int16)))
    coefs = np.stack(coefs)
    frame = {"width": info.width, "height": info.height, "precision": info.precision, "components": comps, "sof": 0xC2}
    b = jl.Batch().upload_frames([frame], qt, jl.FMT_INTERLEAVED_U8)
    b.set_coefficients(0, coefs)
    b.run_idct().sync()
    out = b.output(0)
    assert np.array_equal(out, po.decode_8bit(data)[0])
    # and against the reference's own golden dump (test writer semantics: (ushort) clamp -> u16 -> high byte)
    b16 = jl.Batch().upload_frames([frame], qt, jl.FMT_PLANAR_I16)
    b16.set_coefficients(0, coefs)
    b16.run_idct().sync()
    planes = b16.output(0)
    golden = load_reference_buffer(name, info.width, info.height, info.ncomp)
    for ci, (_, h, v, _) in enumerate(comps):
        hs, vs = max_h // h, max_v // v
        full = np.repeat(np.repeat(planes[ci], vs, axis=0), hs, axis=1)[:info.height, :info.width]
        clamped = np.minimum(full.astype(np.uint16), 255).astype(np.uint16)  # (ushort) cast: negatives become max
        assert np.array_equal(clamped * 257, golden[..., ci]), ci


# ------------------------------------------------------------------------------------------------ progressive frames (SOF2)

def _pillow_progressive(w, h, subsampling, quality, seed, restart_blocks=0, gray=False):
    import io
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 70 * np.sin(xx / 37.0 + seed) * np.cos(yy / 53.0), 128 + 60 * np.cos(xx / 91.0 + yy / 29.0),
                    128 + 90 * np.sin((xx + yy) / 67.0)], axis=-1) + rng.normal(0, 8, (h, w, 3))
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    im = Image.fromarray(img[..., 0]) if gray else Image.fromarray(img)
    buf = io.BytesIO()
    kw = dict(format="JPEG", quality=quality, progressive=True)
    if not gray:
        kw["subsampling"] = subsampling
    if restart_blocks:
        kw["restart_marker_blocks"] = restart_blocks
    im.save(buf, **kw)
    return buf.getvalue()


PROGRESSIVE_CASES = [
    (64, 64, "4:4:4", 75, 0, False), (333, 211, "4:2:0", 75, 0, False), (333, 211, "4:2:2", 90, 0, False),
    (257, 129, "4:2:0", 50, 3, False), (640, 480, "4:2:0", 95, 0, False), (199, 301, "4:4:4", 30, 5, False),
    (123, 77, None, 80, 0, True), (1024, 768, "4:2:0", 85, 16, False), (250, 130, "4:2:0", 60, 7, False),
    (96, 80, "4:4:4", 70, 11, False),
]


@pytest.mark.parametrize("w,h,ss,q,rst,gray", PROGRESSIVE_CASES)
def test_progressive_files_match_oracle(w, h, ss, q, rst, gray):
    """Every scan kind of libjpeg's progressive script (DC first/refine, AC first with EOB runs, AC refinement), with
    and without restart markers, decoded on the GPU through the batch API: pixels AND the accumulated coefficient store."""
    data = _pillow_progressive(w, h, ss, q, seed=w + h, restart_blocks=rst, gray=gray)
    assert b"\xff\xc2" in data
    try:
        ref, info = po.decode_8bit(data)
    except po.OracleError as e:
        # the reference's restart check also runs after the LAST unit of a scan when the unit count is a multiple of
        # DRI, and then trips over the next scan's DHT/SOS ("Expect restart marker."): same failure on the GPU
        mine, res = _status_of_gpu(data)
        assert mine == e.kind, (e.message, mine, res.detail)
        from jpeglibrary_amd import _capi
        assert _capi.lib.jpgpu_detail_string(res.detail).decode() == e.message
        return
    outs, results = jl.decode_batch([data])
    assert results[0].status == 0, results[0].detail
    assert np.array_equal(outs[0], ref)
    # the coefficient store right before the IDCT pass
    _, blocks, _ = po.decode_progressive_store(data)
    b = jl.Batch().upload([data]).decode().sync()
    coefs = b.coefficients(0)
    comps = [(info.comp[i].h, info.comp[i].v) for i in range(info.ncomp)]
    max_h, max_v = max(c[0] for c in comps), max(c[1] for c in comps)
    mcus_x = -(-info.width // (8 * max_h))
    bpm = sum(c[0] * c[1] for c in comps)
    base = 0
    for ci, (ch, cv) in enumerate(comps):
        for (bx, by), blk in blocks[ci].items():
            idx = ((by // cv) * mcus_x + bx // ch) * bpm + base + (by % cv) * ch + bx % ch
            assert np.array_equal(coefs[idx], blk), (ci, bx, by)
        base += ch * cv


def test_progressive_4k_images_match_the_oracle():
    """BASELINE config 5 at full size: 3840x2160 4:2:0 progressive (libjpeg's 10-scan script, DRI = 0), three images in one
    batch, all scans of all frames in the single pipelined launch; samples against the oracle."""
    files = [_pillow_progressive(3840, 2160, "4:2:0", 75, seed) for seed in (11, 12, 13)]
    b = jl.Batch().upload(files).decode().sync()
    for i, f in enumerate(files):
        assert b.result(i).status == 0, (i, b.result(i).detail)
        ref, info = po.decode_8bit(f)
        assert info.sof == 0xC2 and ref.shape == (2160, 3840, 3)
        assert np.array_equal(b.output(i), ref), i
    assert b.progressive_fallbacks() == 0
    b.close()


def _many_small_progressive(n):
    distinct = [_pillow_progressive(64 + 8 * (k % 5), 48 + 8 * (k % 3), ["4:2:0", "4:4:4", "4:2:2"][k % 3], 60 + k, 500 + k) for k in range(24)]
    refs = [po.decode_8bit(d)[0] for d in distinct]
    return [distinct[i % 24] for i in range(n)], [refs[i % 24] for i in range(n)]


@pytest.mark.parametrize("frames,force", [(800, False), (2400, False), (2400, True)])
def test_more_progressive_streams_than_the_machine_keeps_resident(frames, force, monkeypatch):
    """Ten scans per frame, one workgroup each.  800 frames = 8 000 workgroups, 2 400 frames = 24 000: more than 256 CUs hold at
    once, so the scans run as chain launches on streams of their own (the DC scans; each component's AC scans: no waiting inside
    a kernel) -- or, forced, as one pipelined launch without the count-in gate (JPGPU_PROG_FORCE_PIPELINE: followers are
    dispatched behind their producers).  Every frame exact, no fallback needed."""
    if force:
        monkeypatch.setenv("JPGPU_PROG_FORCE_PIPELINE", "1")
    files, refs = _many_small_progressive(frames)
    b = jl.Batch().upload(files)
    for _ in range(3):
        b.decode()
    b.sync()
    bad = [i for i in range(len(files)) if b.result(i).status != 0 or not np.array_equal(b.output(i), refs[i])]
    assert not bad, bad[:10]
    assert b.progressive_fallbacks() == 0
    b.close()


def test_dc_refinement_beside_ac_scans_of_the_same_frame():
    """1024 x 4K progressive frames in one pipelined launch at 14 workgroups per CU: the regime in which the DC refinement
    scan and the AC scans of one frame run neck and neck on the same blocks.  The DC refinement used to be a 32-bit atomic
    OR on the block's first word and now and then put an old coefficient 1 back over the AC scan's store: 7-16 frames per
    pass failed with "invalid Huffman code" in the next refinement of that band.  The 16 source frames are compared with the
    checker's samples, the 1008 copies with their sources.  (Own process: the LDS shape is read once.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JPGPU_PS_RING="4096", JPGPU_PS_CHUNK="32", JPGPU_PROG_FORCE_PIPELINE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "trace", "progressive_oversubscribed.py"), "1024"], env=env,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("n=1024")]
    assert r.returncode == 0 and len(lines) == 3, r.stdout[-2000:] + r.stderr[-2000:]
    assert all("failed 0 [] differing []" in ln for ln in lines), lines
    # the 16 sources against the CHECKER (the tool itself only compares copies with their source): same generator, same seeds
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    from bench import progressive_batch
    src = progressive_batch(16, 3840, 2160, 75, 1, 16)
    with ThreadPoolExecutor(8) as ex:
        want = list(ex.map(lambda f: hashlib.sha256(po.decode_8bit(f)[0].tobytes()).hexdigest(), src))
    got = {int(ln.split()[1]): ln.split()[2] for ln in r.stdout.splitlines() if ln.startswith("sha256 ")}
    assert [got.get(i) for i in range(16)] == want


def _scan_lengths(data):
    """bytes from each SOS marker to the next marker segment (FF DA cannot occur inside entropy-coded data)"""
    at = [i for i in range(len(data) - 1) if data[i] == 0xFF and data[i + 1] == 0xDA]
    ends = at[1:] + [len(data)]
    return [e - a for a, e in zip(at, ends)]


@pytest.mark.parametrize("slow_scan", [1, 0, 4])
def test_a_scan_inside_an_end_of_band_run_still_follows_its_producers(slow_scan, monkeypatch):
    """libjpeg's 10-scan script on a smooth frame: scan 4 (Y AC 6-63, first pass, Al = 2) is ONE end-of-band run over the whole
    frame -- a few bytes.  The host keeps only direct dependencies: scan 5 (the Y AC 1-63 refinement) follows scan 4 alone,
    which follows scan 1 (Y AC 1-5), so scan 4 has to pass scan 1's progress on.  Its skip of a block inside an end-of-band
    run once came BEFORE it looked at its producer: scan 4 announced the whole frame at once, and wherever the refinement
    caught up with scan 1 it ran over coefficients 1-5 that were not there yet (wrong samples, or "invalid Huffman code" one
    scan later; in production only the forced oversubscribed launch hit it, a few frames per thousand).
    JPGPU_DEBUG_DELAY_SCAN=k:ms makes scan k of every frame slow (it idles ms at its start and after every progress word),
    which turns the rare interleaving into a certain one: tools/trace/eob_skip_regression.sh runs this test against a build
    with the old order, where the slow-scan-1 case fails."""
    import time

    from bench import progressive_batch
    files = progressive_batch(6, 1024, 768, 75, 900, 6)
    for f in files:
        lens = _scan_lengths(f)
        assert len(lens) == 10 and lens[4] < 64, lens  # the precondition: scan 4 is one end-of-band run
    refs = [po.decode_8bit(f)[0] for f in files]
    monkeypatch.delenv("JPGPU_PROG_NO_PIPELINE", raising=False)
    b0 = jl.Batch().upload(files).decode().sync()  # (first decode of the process: module load, LDS shape ...)
    undelayed = 1e9
    for _ in range(3):  # (the fastest of three: a hiccup here must not pass for the hook being dead)
        t0 = time.perf_counter()
        b0.decode().sync()
        undelayed = min(undelayed, time.perf_counter() - t0)
    b0.close()
    monkeypatch.setenv("JPGPU_DEBUG_DELAY_SCAN", "%d:4" % slow_scan)
    b = jl.Batch().upload(files)
    t0 = time.perf_counter()
    b.decode().sync()
    delayed = time.perf_counter() - t0
    assert delayed > undelayed + 0.002, (undelayed, delayed)  # the hook is live: ~4 ms at the start and per progress word of that scan
    bad = [i for i in range(len(files)) if b.result(i).status != 0 or not np.array_equal(b.output(i), refs[i])]
    where = [(i, b.result(i).detail, [int(x) for x in np.argwhere((b.output(i) != refs[i]).any(axis=2))[0]] if b.result(i).status == 0 else None) for i in bad]
    assert not bad, (where, b.progressive_fallbacks())
    assert b.progressive_fallbacks() == 0
    b.close()


def test_progressive_spin_budget_exhausted_falls_back_level_by_level(monkeypatch):
    """With no polls to spend (JPGPU_PROG_SPIN_BUDGET=0) every follower that is not already satisfied gives up; the host
    sees the internal time-out status and re-issues the step scan level by scan level in fresh launches: same samples."""
    monkeypatch.setenv("JPGPU_PROG_SPIN_BUDGET", "0")
    monkeypatch.delenv("JPGPU_PROG_NO_PIPELINE", raising=False)  # (the suite is also run under the A/B switches)
    files, refs = _many_small_progressive(300)
    files += [_pillow_progressive(1024, 768, "4:2:0", 85, 77)]
    refs += [po.decode_8bit(files[-1])[0]]
    b = jl.Batch().upload(files).decode().sync()
    bad = [i for i in range(len(files)) if b.result(i).status != 0 or not np.array_equal(b.output(i), refs[i])]
    assert not bad, bad[:10]
    assert b.progressive_fallbacks() == 1
    b.decode().sync()  # stays level by level
    assert all(b.result(i).status == 0 for i in range(len(files))) and b.progressive_fallbacks() == 1
    b.close()


def test_progressive_batch_mixed_with_baseline():
    files = [read_jpeg("progress.jpg"), read_jpeg("lake.jpg"), _pillow_progressive(160, 120, "4:2:0", 70, 3),
             bytes(jpegsynth.encode(96, 64, "420", 75, 2, seed=5)), read_jpeg("yellowcat_progressive_restart.jpg")]
    outs, results = jl.decode_batch(files)
    for f, o, r in zip(files, outs, results):
        assert r.status == 0
        assert np.array_equal(o, po.decode_8bit(bytes(f))[0])


def test_truncated_progressive_stream_fails_like_the_reference():
    from jpeglibrary_amd import _capi
    data = _pillow_progressive(320, 240, "4:2:0", 85, 9)
    for cut in (len(data) // 3, len(data) // 2, len(data) - 40):
        bad = data[:cut] + b"\xff\xd9"
        try:
            po.decode_8bit(bad)
            ref, ref_msg = "OK", ""
        except po.OracleError as e:
            ref, ref_msg = e.kind, e.message
        mine, res = _status_of_gpu(bad)
        assert mine == ref, (cut, ref, ref_msg, mine, res.detail)
        if res.detail in (1, 2, 3, 4, 9):
            assert _capi.lib.jpgpu_detail_string(res.detail).decode() == ref_msg, (cut, ref_msg, res.detail)


# ------------------------------------------------------------------------------------------------ YCbCr -> RGB(A) in the writer

RGB_CASES = [
    (256, 128, "420", 75, 4), (256, 128, "422", 80, 0), (256, 128, "444", 90, 2),   # fused layouts
    (333, 211, "420", 75, 8), (100, 60, "444", 60, 0),                             # widths without a fused path
    (64, 64, "gray", 75, 0), (70, 50, "gray", 75, 0), (640, 480, "420", 95, 4),
]


@pytest.mark.parametrize("w,h,ss,q,dri", RGB_CASES)
@pytest.mark.parametrize("fmt", ["rgb", "rgba"])
def test_rgb_output_matches_reference_converter(w, h, ss, q, dri, fmt):
    """FMT_RGB_U8 / FMT_RGBA_U8 = JpegYCbCrToRgbConverter applied to the decoded YCbCr8 buffer (what DecodeAction and the
    reference benchmark do after Decode()), fused into the writer kernel where the layout has a fast path."""
    data = bytes(jpegsynth.encode(w, h, ss, q, dri, seed=w * 3 + h))
    ycc, _ = po.decode_8bit(data)
    gray = ss == "gray"
    ref = po.ycbcr8_to_rgb(ycc, rgba=(fmt == "rgba"), gray=gray)
    outs, results = jl.decode_batch([data], jl.FMT_RGBA_U8 if fmt == "rgba" else jl.FMT_RGB_U8)
    assert results[0].status == 0
    assert np.array_equal(outs[0], ref)


def test_rgb_output_of_reference_assets_and_progressive():
    for name in ["lake.jpg", "cramps.jpg", "HETissueSlide.jpg", "progress.jpg"]:
        data = read_jpeg(name)
        ycc, info = po.decode_8bit(data)
        ref = po.ycbcr8_to_rgb(ycc, rgba=True, gray=(info.ncomp == 1))
        outs, results = jl.decode_batch([data], jl.FMT_RGBA_U8)
        assert results[0].status == 0, name
        assert np.array_equal(outs[0], ref), name


def test_rgb_output_rejects_other_colour_spaces():
    # 12-bit samples and 4-component frames: the reference's callers refuse them too (DecodeAction.cs:29-33)
    _, results = jl.decode_batch([read_jpeg("testorig12.jpg")], jl.FMT_RGB_U8)
    assert results[0].status == 3


# ------------------------------------------------------------------------------------------------ randomised corruption

def _mutate(data: bytes, rng) -> bytes:
    """One random edit of the entropy-coded part (headers stay parseable most of the time): flip bits, drop / duplicate /
    insert bytes, plant markers, truncate."""
    sos = data.index(b"\xff\xda")
    lo = sos + 4 + data[sos + 3]
    b = bytearray(data)
    kind = rng.integers(0, 7)
    pos = int(rng.integers(lo, len(b) - 2))
    if kind == 0:
        b[pos] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        del b[pos:pos + int(rng.integers(1, 6))]
    elif kind == 2:
        b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 6))).astype(np.uint8))
    elif kind == 3:
        b[pos:pos + 2] = bytes([0xFF, int(rng.choice([0xD0, 0xD3, 0xD7, 0xD9, 0xC4, 0xDA, 0x00, 0xFF, 0xE1]))])
    elif kind == 4:
        b = b[:pos] + b"\xff\xd9"
    elif kind == 5:
        n = int(rng.integers(1, 40))
        b[pos:pos + n] = bytes(n)
    else:
        n = int(rng.integers(1, 40))
        b[pos:pos + n] = b"\xff" * n
    return bytes(b)


@pytest.mark.parametrize("variant", ["dri4_420", "dri0_444", "dri1_gray", "progressive"])
def test_random_corruptions_fail_like_the_reference(variant):
    """200 random edits per stream kind: the GPU path must report the reference's exception class (and message, for
    device-reported failures) or, when the reference still decodes, produce the same pixels."""
    from jpeglibrary_amd import _capi
    rng = np.random.default_rng({"dri4_420": 1, "dri0_444": 2, "dri1_gray": 3, "progressive": 4}[variant])
    if variant == "dri4_420":
        good = bytes(jpegsynth.encode(112, 80, "420", 75, 4, seed=11))
    elif variant == "dri0_444":
        good = bytes(jpegsynth.encode(72, 56, "444", 85, 0, seed=12))
    elif variant == "dri1_gray":
        good = bytes(jpegsynth.encode(96, 64, "gray", 60, 1, seed=13))
    else:
        good = _pillow_progressive(96, 72, "4:2:0", 80, 14, restart_blocks=0)
    cases = [_mutate(good, rng) for _ in range(200)]
    refs = []
    for data in cases:
        try:
            refs.append(("OK", "", po.decode_8bit(data)[0]))
        except po.OracleError as e:
            refs.append((e.kind, e.message, None))
    outs, results = jl.decode_batch(cases, jl.FMT_INTERLEAVED_U8)
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    mismatches = []
    for i, ((ref, ref_msg, ref_px), out, res) in enumerate(zip(refs, outs, results)):
        mine = names.get(res.status, str(res.status))
        if ref == "OK" and variant == "progressive" and mine == "NotSupportedException" and res.detail == 6:
            # documented fence: a damaged scan sequence can leave the reference's Dispose() pass transforming
            # one component twice and another never; it "succeeds" with an artefact we refuse to reproduce
            continue
        if mine != ref:
            mismatches.append((i, ref, ref_msg, mine, res.detail))
        elif ref == "OK" and not np.array_equal(out, ref_px):
            mismatches.append((i, "pixels differ", "", mine, res.detail))
        elif ref != "OK" and res.detail in (1, 2, 3, 4, 9) and _capi.lib.jpgpu_detail_string(res.detail).decode() != ref_msg:
            mismatches.append((i, ref, ref_msg, mine, res.detail))
    assert not mismatches, mismatches[:10]


# ------------------------------------------------------------------------------------------------ edge sizes, one batch

def test_edge_sizes_and_samplings_in_one_batch():
    """Tiny and ragged geometries (1x1 up to one-and-a-bit MCUs) for every sampling, with and without restart intervals,
    sequential and progressive, decoded as ONE batch in every output format that has an oracle counterpart."""
    files = []
    for (w, h) in [(1, 1), (7, 9), (8, 8), (9, 7), (16, 16), (17, 1), (1, 33), (31, 17), (48, 40)]:
        for ss in ("444", "422", "420", "gray"):
            for dri in (0, 1, 3):
                files.append(bytes(jpegsynth.encode(w, h, ss, 70, dri, seed=w * 131 + h * 7 + dri)))
        files.append(_pillow_progressive(w, h, "4:2:0", 75, w + h))
        files.append(_pillow_progressive(w, h, None, 75, w * h, gray=True))
    refs = [po.decode_8bit(f) for f in files]
    outs, results = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
    for i, ((ref, info), out, res) in enumerate(zip(refs, outs, results)):
        assert res.status == 0, (i, res.status, res.detail)
        assert np.array_equal(out, ref), i
    outs, results = jl.decode_batch(files, jl.FMT_RGBA_U8)
    for i, ((ref, info), out, res) in enumerate(zip(refs, outs, results)):
        assert res.status == 0, (i, res.status, res.detail)
        assert np.array_equal(out, po.ycbcr8_to_rgb(ref, rgba=True, gray=(info.ncomp == 1))), i
    # unclamped samples: every WriteBlock call of the reference (sequential files: one per block, before chroma
    # expansion only for full-resolution components) must be found in the PLANAR_I16 planes
    b = jl.Batch().upload(files, jl.FMT_PLANAR_I16).decode().sync()
    for i, f in enumerate(files):
        assert b.result(i).status == 0
        planes = b.output(i)
        info = refs[i][1]
        calls, _ = po.decode_blocks(f)
        max_h = max(info.comp[c].h for c in range(info.ncomp))
        max_v = max(info.comp[c].v for c in range(info.ncomp))
        for (ci, x, y, blk) in calls:
            if info.comp[ci].h != max_h or info.comp[ci].v != max_v:
                continue  # expanded chroma blocks: covered by the interleaved comparison above
            tile = planes[ci][y:y + 8, x:x + 8]
            if tile.shape == (8, 8):
                assert np.array_equal(tile.reshape(-1), np.asarray(blk)), (i, ci, x, y)


# ------------------------------------------------------------------------------------------------ encoder

def _enc_image(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 70 * np.sin(xx / 37 + seed) * np.cos(yy / 53), 128 + 60 * np.cos(xx / 91 + yy / 29),
                    128 + 90 * np.sin((xx + yy) / 67)], -1) + rng.normal(0, 8, (h, w, 3))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _rgba(rgb, seed):
    """Rgba32 pixels as the reference's EncoderBenchmark hands them over (ConvertRgba32ToYCbCr8 reads R, G, B): the alpha bytes are
    noise here, so that a reader that looked at them would show."""
    a = np.random.default_rng(seed).integers(0, 256, rgb.shape[:2] + (1,), dtype=np.uint8)
    out = np.ascontiguousarray(np.concatenate([rgb, a], axis=-1))
    assert np.array_equal(po.rgba_to_ycbcr8(out), po.rgb_to_ycbcr8(rgb))
    return out


ENC_CASES = [(64, 48, (2, 2), 75), (333, 211, (2, 2), 75), (333, 211, (2, 1), 90), (100, 75, (1, 1), 50), (17, 9, (2, 2), 30),
             (1, 1, (2, 2), 75), (640, 368, (2, 2), 100), (1024, 768, (2, 2), 85), (31, 65, (4, 1), 60)]


@pytest.mark.parametrize("w,h,luma,q", ENC_CASES)
def test_encoder_matches_the_reference_restatement_byte_for_byte(w, h, luma, q):
    """GPU encode (FDCT + quantise, Huffman lengths, bit emission, byte stuffing) == oracle/jpegenc.c: the quantised
    blocks and the whole byte stream, for YCbCr input and for RGB input (colour conversion fused into the reader)."""
    rgb = _enc_image(w, h, w + h)
    ycc = po.rgb_to_ycbcr8(rgb)
    ref, ref_coefs = po.encode_8bit(ycc, luma[0], luma[1], q, want_coefficients=True)
    b = jl.EncodeBatch().upload([ycc], luma, q).encode()
    assert np.array_equal(b.coefficients(0), ref_coefs)
    assert b.output(0) == ref
    assert jl.encode_batch([rgb], luma, q, rgb=True)[0] == ref
    assert jl.encode_batch([_rgba(rgb, w)], luma, q, rgb=True)[0] == ref  # Rgba32 pixels: the alpha byte is stepped over
    # and it is a JPEG the decoder path reads back: pixels equal to the oracle decode of the oracle stream
    outs, results = jl.decode_batch([b.output(0)])
    assert results[0].status == 0 and np.array_equal(outs[0], po.decode_8bit(ref)[0])


FUSED_SHAPES = [(100, 80), (20, 16), (12, 40), (48, 1), (16, 16), (272, 33), (1600, 48)]


@pytest.mark.parametrize("w,h", FUSED_SHAPES)
@pytest.mark.parametrize("opt", [False, True])
@pytest.mark.parametrize("luma", [(2, 2), (2, 1), (1, 1)])
def test_encoder_fused_e1_row_alignments_and_edges(w, h, opt, luma):
    """fdct_fused_kernel<H, V> (three components; 4:2:0, 4:2:2, 4:4:4) on the shapes its paths split by: rows that start on 16 bytes / on 4 / on
    nothing, images narrower than one MCU, partial MCUs right and below, more than 16 MCUs -- coefficients and stream equal
    the checker's for YCbCr and for RGB pixels, with the standard tables and with optimizeCoding (no carry-over from the
    previous block, JpegEncoder.cs:414-485)."""
    rgb = _enc_image(w, h, 3 * w + h)
    ycc = po.rgb_to_ycbcr8(rgb)
    ref, ref_coefs = po.encode_8bit(ycc, luma[0], luma[1], 77, want_coefficients=True, optimize_coding=opt)
    b = jl.EncodeBatch().upload([ycc], luma, 77, optimize_coding=opt).encode()
    assert np.array_equal(b.coefficients(0), ref_coefs)
    assert b.output(0) == ref
    assert jl.encode_batch([rgb], luma, 77, rgb=True, optimize_coding=opt)[0] == ref
    # fdct_fused_kernel<H, V, 4>: the same rows at four bytes per pixel (16-byte row starts when the width is a multiple of four)
    assert jl.encode_batch([_rgba(rgb, w)], luma, 77, rgb=True, optimize_coding=opt)[0] == ref


def test_encoder_rgba_pixels_in_a_mixed_batch():
    """One batch holding Rgba32, RGB and YCbCr images of fused and two-kernel shapes: every instance of E1 leaves the others'
    images alone, and the four-byte reader of the two-kernel path (luma 4 x 1, 1 x 2) steps over the alpha byte too."""
    rgbs = [_enc_image(w, h, w * 3 + h) for (w, h) in [(160, 96), (33, 47), (640, 64), (75, 50), (50, 70)]]
    lumas = [(2, 2), (2, 2), (2, 2), (4, 1), (1, 2)]
    for luma in set(lumas):
        idx = [i for i, l in enumerate(lumas) if l == luma]
        imgs = []
        for n, i in enumerate(idx):
            imgs.append(_rgba(rgbs[i], i) if n % 2 == 0 else rgbs[i])
        outs = jl.encode_batch(imgs, luma, 70, rgb=True)
        for i, o in zip(idx, outs):
            assert o == po.encode_8bit(po.rgb_to_ycbcr8(rgbs[i]), luma[0], luma[1], 70)
    with pytest.raises(ValueError):
        jl.EncodeBatch().upload([_rgba(rgbs[0], 0)], (2, 2), 70, rgb=False)


def test_encoder_fused_and_two_kernel_images_in_one_batch():
    """A batch whose images go two ways through E1: colour images (fdct_fused_kernel) between gray ones (E1a + E1b), each
    kernel leaving the other's images alone."""
    col = [po.rgb_to_ycbcr8(_enc_image(w, h, w)) for (w, h) in [(160, 96), (33, 47), (640, 64)]]
    gray = [po.rgb_to_ycbcr8(_enc_image(w, h, h))[..., 0] for (w, h) in [(64, 64), (50, 30)]]
    imgs = [col[0], gray[0], col[1], gray[1], col[2]]
    outs = jl.encode_batch(imgs, (2, 2), 70)
    for im, o in zip(imgs, outs):
        assert o == po.encode_8bit(im, 2, 2, 70)


def test_encoder_batch_gray_and_mixed_sizes():
    imgs = [_enc_image(96, 64, 1), _enc_image(50, 70, 2), _enc_image(256, 256, 3)]
    outs = jl.encode_batch([po.rgb_to_ycbcr8(i) for i in imgs], (2, 2), 80)
    for im, o in zip(imgs, outs):
        assert o == po.encode_8bit(po.rgb_to_ycbcr8(im), 2, 2, 80)
    gray = [po.rgb_to_ycbcr8(i)[..., 0] for i in imgs]
    outs = jl.encode_batch(gray, (1, 1), 60)
    for g, o in zip(gray, outs):
        assert o == po.encode_8bit(g, 1, 1, 60)


def test_encoder_adversarial_content():
    """Content that stresses the entropy stage: white noise at quality 100 (long codes, many FF bytes to stuff), flat
    images (EOB-only blocks), full-swing checkerboards (largest coefficients), saturated edges; all in one batch."""
    rng = np.random.default_rng(5)
    h, w = 120, 136
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = [
        rng.integers(0, 256, (h, w, 3)).astype(np.uint8),
        np.full((h, w, 3), 255, np.uint8),
        np.zeros((h, w, 3), np.uint8),
        (((xx + yy) & 1) * 255).astype(np.uint8)[..., None].repeat(3, axis=2),
        (((xx // 8 + yy // 8) & 1) * 255).astype(np.uint8)[..., None].repeat(3, axis=2),
        np.where(xx[..., None] < w // 2, np.uint8(0), np.uint8(255)).repeat(3, axis=2).astype(np.uint8),
    ]
    for q in (1, 50, 100):
        for luma in ((2, 2), (1, 1)):
            outs = jl.encode_batch(imgs, luma, q)
            for k, (im, o) in enumerate(zip(imgs, outs)):
                ref = po.encode_8bit(im, luma[0], luma[1], q)
                assert o == ref, (q, luma, k, len(o), len(ref))
    assert b"\xff\x00" in jl.encode_batch([imgs[0]], (1, 1), 100)[0]


def test_encoder_counts_and_emits_the_bits_in_one_pass_and_falls_back_when_a_stretch_does_not_fit():
    """bits_emit_kernel (E2 + E3 as one pass, round 5): uploads of two or more images without restart intervals take it -- the
    first encode() with a default on-chip buffer, the next ones with a buffer sized from the one before -- and the streams are the
    checker's byte for byte whichever way they were made: images of one workgroup and of hundreds in one batch, gray, optimizeCoding,
    a one-pixel image (a chain of one record whose only word is the padded partial one).  A stretch that does not fit the buffer
    (white noise at quality 100: ~700 bits per block) makes the host issue the two kernels behind it, once; single images and
    restart intervals never take the one pass."""
    rng = np.random.default_rng(77)
    sizes = [(640, 480), (1, 1), (17, 9), (256, 256), (1000, 31), (8, 8)]
    imgs = [_enc_image(w, h, w + 3 * h) for (w, h) in sizes]
    ycc = [po.rgb_to_ycbcr8(im) for im in imgs]
    for luma, opt in (((2, 2), False), ((1, 1), False), ((2, 1), True)):
        b = jl.EncodeBatch().upload(imgs, luma, 75, rgb=True, optimize_coding=opt)
        for rep in range(3):  # (the second and third encode() size the buffer from the one before)
            b.encode()
            assert b.emit_passes() == (rep + 1, 0)
            for k, y in enumerate(ycc):
                assert b.output(k) == po.encode_8bit(y, luma[0], luma[1], 75, optimize_coding=opt), (luma, opt, rep, sizes[k])
        b.close()
    gray = [y[..., 0].copy() for y in ycc[:3]]
    b = jl.EncodeBatch().upload(gray, (1, 1), 60).encode()
    assert b.emit_passes() == (1, 0)
    for k, g in enumerate(gray):
        assert b.output(k) == po.encode_8bit(g, 1, 1, 60)
    b.close()
    # a stretch beyond the buffer: the two kernels behind the one pass, and no second attempt for this upload
    noise = [rng.integers(0, 256, (264, 520, 3)).astype(np.uint8) for _ in range(3)]
    b = jl.EncodeBatch().upload(noise + [imgs[0]], (1, 1), 100)
    for rep in range(2):
        b.encode()
        assert b.emit_passes() == (1, 1)
        for k, im in enumerate(noise + [imgs[0]]):
            assert b.output(k) == po.encode_8bit(im, 1, 1, 100)
    b.close()
    one = jl.EncodeBatch().upload(imgs[:1], (2, 2), 75, rgb=True).encode()
    assert one.emit_passes() == (0, 0) and one.output(0) == po.encode_8bit(ycc[0], 2, 2, 75)
    one.close()
    ri = jl.EncodeBatch().upload(imgs[:2], (2, 2), 75, rgb=True, restart_interval=5).encode()
    assert ri.emit_passes() == (0, 0) and ri.output(1) == po.encode_8bit(ycc[1], 2, 2, 75, restart_interval=5)
    ri.close()


@pytest.mark.parametrize("env", [
    {"JPGPU_PROG_NO_PIPELINE": "1"},                    # wave-per-stream kernel, chain launches (a stream per chain of scans)
    {"JPGPU_PROG_NO_PIPELINE": "1", "JPGPU_PROG_NO_CHAINS": "1"},  # ... one launch per dependency level
    {"JPGPU_PROG_STREAM_MAX_INTERVALS": "0"},           # lane-per-interval kernel for every scan
    {"JPGPU_PROG_STREAM_MAX_INTERVALS": "1000000"},     # wave-per-stream kernel even for scans with many restart intervals
])
@pytest.mark.parametrize("name", ["progress.jpg", "yellowcat_progressive_restart.jpg"])
def test_progressive_kernel_variants_agree_with_the_oracle(name, env, monkeypatch):
    """The progressive path picks its kernel per scan (lanes per restart interval / one wave per stream) and runs the
    stream kernel either as one pipelined launch or level by level: every combination must give the reference's samples."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    data = read_jpeg(name)
    ref, _ = po.decode_8bit(data)
    outs, results = jl.decode_batch([data, data])
    for out in outs:
        assert np.array_equal(np.asarray(out).reshape(ref.shape), ref)


# ------------------------------------------------------------------------------------------------ optimizer

def _synth_jpeg(w, h, quality=75, subsampling=2, restart=0, seed=0, gray=False):
    import io
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 70 * np.sin(xx / 37 + 1) * np.cos(yy / 53), 128 + 60 * np.cos(xx / 91 + yy / 29), 128 + 90 * np.sin((xx + yy) / 67)], -1)
    img = np.clip(np.rint(img + rng.normal(0, 8, img.shape)), 0, 255).astype(np.uint8)
    out = io.BytesIO()
    kw = dict(format="JPEG", quality=quality)
    if not gray:
        kw["subsampling"] = subsampling
    if restart:
        kw["restart_marker_blocks"] = restart
    Image.fromarray(img[..., 0] if gray else img).save(out, **kw)
    return out.getvalue()


def _optimizer_files():
    return [read_jpeg("lake.jpg"), read_jpeg("cramps.jpg"), read_jpeg("HETissueSlide.jpg"), _synth_jpeg(200, 136, restart=4),
            _synth_jpeg(333, 77, subsampling=0, restart=11, seed=2), _synth_jpeg(64, 64, gray=True, seed=3),
            _synth_jpeg(640, 360, quality=95, subsampling=1, restart=7, seed=4), _synth_jpeg(1920, 1088, restart=7, seed=5),
            _synth_jpeg(1280, 720, quality=90, subsampling=0, seed=6), _synth_jpeg(333, 222, quality=50, subsampling=1, seed=7)]


@pytest.mark.parametrize("strip", [True, False])
def test_optimizer_matches_the_reference_restatement_byte_for_byte(strip):
    """JpegOptimizer.Scan() + Optimize(strip) for a mixed batch (DRI and no DRI, gray, 4:4:4 / 4:2:2 / 4:2:0): the statistics,
    the rebuilt DHT, the re-written scan and every copied segment must equal the restatement's output."""
    files = _optimizer_files()
    b = jl.OptimizeBatch().upload(files, strip).run()
    for i, f in enumerate(files):
        ref = po.optimize(f, strip)
        got = b.output(i)
        assert got == ref, (i, len(got), len(ref))
        stats = b.statistics(i)
        ref_stats = po.optimizer_statistics(f)
        assert [(c, t) for c, t, _ in stats] == [(c, t) for c, t, _ in ref_stats]
        for (_, _, a), (_, _, r) in zip(stats, ref_stats):
            assert np.array_equal(a, r)
    b.close()


@pytest.mark.parametrize("strip", [True, False])
def test_reference_optimizer_test_on_the_gpu(strip):
    """tests/JpegLibrary.Tests/Optimizer/OptimizerTests.cs:26-47 (TestOptimize on Assets/baseline/lake.jpg, strip = true /
    false), line by line through the mirror of the reference's class."""
    import io
    jpeg_bytes = read_jpeg("lake.jpg")
    ref_image, _ = po.decode_8bit(jpeg_bytes)

    optimizer = jl.JpegOptimizer()
    optimizer.SetInput(jpeg_bytes)
    optimizer.Scan()

    buffer = io.BytesIO()
    optimizer.SetOutput(buffer)
    optimizer.Optimize(strip)

    assert buffer.tell() < len(jpeg_bytes)
    test_image, _ = po.decode_8bit(buffer.getvalue())
    assert np.array_equal(ref_image, test_image)
    outs, results = jl.decode_batch([buffer.getvalue()])  # ... and through the GPU decoder as well
    assert results[0].status == 0 and np.array_equal(np.asarray(outs[0]), ref_image)


def test_optimizer_failures_follow_the_reference():
    good = read_jpeg("lake.jpg")
    files = [good, _synth_jpeg(200, 120, restart=4), read_jpeg("progress.jpg"), good[:100000], b"\xff\xd8\xff\xd9", good]
    b = jl.OptimizeBatch().upload(files, True).run()
    assert b.output(0) == po.optimize(good, True) and b.output(5) == b.output(0)
    # MCU count a multiple of DRI: Scan() gives up at the EOI it meets in the restart check, Optimize() throws
    with pytest.raises(jl.InvalidOperationException):
        b.output(1)
    with pytest.raises(po.OracleError):
        po.optimize(files[1], True)
    for i in (2, 3, 4):  # progressive: Scan() skips SOF2 like an APPn segment, the SOS then finds no frame header
        with pytest.raises(po.OracleError) as ref:
            po.optimize(files[i], True)
        with pytest.raises(getattr(jl, ref.value.kind)):
            b.output(i)
    b.close()


@pytest.mark.parametrize("luma,w,h", [((2, 2), 64, 48), ((2, 2), 100, 70), ((2, 2), 37, 29), ((1, 1), 61, 35), ((2, 1), 90, 41), ((4, 1), 130, 33),
                                      ((2, 2), 640, 360)])
def test_encoder_optimize_coding_matches_the_restatement(luma, w, h):
    """EncodeAction's optimizeCoding: TransformBlocks into the block allocator (own block per position, dummy block for
    positions outside a component's grid), BuildHuffmanTables, WritePreparedScanData -- coefficients, DHT and scan bytes."""
    rng = np.random.default_rng(w * 1000 + h)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([128 + 90 * np.sin(xx / 17) * np.cos(yy / 23), 128 + 70 * np.cos(xx / 31 + yy / 11), 128 + 80 * np.sin((xx + yy) / 19)], -1)
    imgs = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8), np.clip(np.rint(smooth + rng.normal(0, 6, smooth.shape)), 0, 255).astype(np.uint8)]
    for q in (30, 75, 95):
        b = jl.EncodeBatch().upload(imgs, luma, q, optimize_coding=True).encode()
        for k, im in enumerate(imgs):
            ref, ref_coefs = po.encode_8bit(im, luma[0], luma[1], q, want_coefficients=True, optimize_coding=True)
            assert np.array_equal(b.coefficients(k), ref_coefs), (q, k)
            got = b.output(k)
            assert got == ref, (q, k, len(got), len(ref))
            assert len(got) <= len(po.encode_8bit(im, luma[0], luma[1], q)) + 64
        b.close()


def test_encoder_optimize_coding_mixed_batch_and_gray_failure():
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, (50, 70, 3)).astype(np.uint8)
    gray = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    out = jl.encode_batch([rgb], (2, 2), 80, rgb=True, optimize_coding=True)[0]
    assert out == po.encode_8bit(po.rgb_to_ycbcr8(rgb), 2, 2, 80, optimize_coding=True)
    b = jl.EncodeBatch().upload([gray, rgb], (1, 1), 80, optimize_coding=True).encode()
    with pytest.raises(jl.InvalidOperationException):  # "No symbol is recorded.": the chrominance builders of a gray image stay empty
        b.output(0)
    with pytest.raises(po.OracleError):
        po.encode_8bit(gray, 1, 1, 80, optimize_coding=True)
    assert b.output(1) == po.encode_8bit(rgb, 1, 1, 80, optimize_coding=True)
    b.close()


@pytest.mark.parametrize("variant", ["dri7_420", "dri0_444", "dri3_gray"])
def test_optimizer_random_corruptions_follow_the_reference(variant):
    """150 random edits per stream kind through the optimizer: where the reference's JpegOptimizer still produces a
    stream the GPU path must produce the same bytes; where it throws, the same exception class.  (Files the device path
    refuses by design -- several scans, progressive, a DRI that changes after the scan -- are skipped.)"""
    rng = np.random.default_rng({"dri7_420": 21, "dri0_444": 22, "dri3_gray": 23}[variant])
    if variant == "dri7_420":
        good = bytes(jpegsynth.encode(112, 80, "420", 75, 7, seed=31))   # 35 MCUs
    elif variant == "dri0_444":
        good = bytes(jpegsynth.encode(72, 56, "444", 85, 0, seed=32))
    else:
        good = bytes(jpegsynth.encode(88, 64, "gray", 60, 3, seed=33))   # 88 MCUs
    cases = [good] + [_mutate(good, rng) for _ in range(150)]
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    b = jl.OptimizeBatch().upload(cases, True).run()
    mismatches, compared = [], 0
    for i, data in enumerate(cases):
        try:
            ref, ref_kind = po.optimize(data, True), "OK"
        except po.OracleError as e:
            ref, ref_kind = None, e.kind
        res, size = b.result(i)
        mine = names.get(res.status, str(res.status))
        if mine == "NotSupportedException":
            continue
        compared += 1
        if mine != ref_kind:
            mismatches.append((i, ref_kind, mine, res.detail))
        elif ref is not None and b.output(i) != ref:
            mismatches.append((i, "bytes differ", len(ref), size))
    b.close()
    assert compared > 100
    assert not mismatches, mismatches[:10]


def _structural_edits(good):
    """Hand-made edits of one baseline file aimed at the marker walk around the scan (found by tools/stress_parity.py)."""
    def seg(marker):
        i = good.index(bytes([0xFF, marker]))
        return i, i + 2 + int.from_bytes(good[i + 2:i + 4], "big")
    assert good[-2:] == b"\xff\xd9"
    body = good[:-2]
    edits = {
        "no_eoi": body,
        "zeros_for_eoi": body + bytes(2),
        "trailing_zeros_no_marker": body + bytes(24),
        "trailing_zeros_then_eoi": body + bytes(24) + b"\xff\xd9",
        "trailing_fill_then_eoi": body + b"\xff" * 9 + b"\xd9",
        "garbage_after_eoi": good + bytes(range(1, 40)),
        "second_eoi": good + b"\xff\xd9",
    }
    # whole bytes left unread in front of the terminating marker: with exactly one the reference resumes its walk one byte
    # INTO the marker (the bit reader holds the marker but only shows it once its buffer is empty)
    for k in (1, 2, 3, 4, 7):
        edits[f"{k}_unread_then_eoi"] = body + bytes([0x5A] * k) + b"\xff\xd9"
        edits[f"{k}_unread_eoi_eoi"] = body + bytes([0x5A] * k) + b"\xff\xd9\xff\xd9"
        edits[f"{k}_unread_eoi_com"] = body + bytes([0x5A] * k) + b"\xff\xd9\xff\xfe\x00\x04ab"
        edits[f"{k}_unread_eoi_badseg"] = body + bytes([0x5A] * k) + b"\xff\xd9\xff\xfe\x00\x09ab"
    edits["1_unread_stuffed_then_eoi"] = body + b"\xff\x00\xff\xd9"
    edits["1_unread_fill_then_eoi"] = body + b"\x5a\xff\xff\xff\xd9"
    a, b = seg(0xC4)
    edits["no_dht"] = good[:a] + good[b:]
    edits["dht_as_app5"] = good[:a + 1] + b"\xe5" + good[a + 2:]
    a, b = seg(0xDB)
    edits["no_dqt"] = good[:a] + good[b:]
    a, b = seg(0xDA)
    edits["scan_names_no_component"] = good[:a + 4] + b"\x00" + good[a + 5:]  # ProcessScan walks the MCUs without reading a bit
    edits["second_scan_names_no_component"] = body + b"\xff\xda\x00\x08\x00\x01\x00\x00\x3f\x00" + bytes(range(1, 60)) + b"\xff\xd9"
    edits["scan_selector_not_in_frame"] = good[:a + 5] + b"\x77" + good[a + 6:]
    if good[a + 4] == 3:  # two scan components resolve to frame component 3, none to component 1: written twice / never
        edits["first_selector_is_the_third"] = good[:a + 5] + good[a + 9:a + 10] + good[a + 6:]
    f0, _ = seg(0xC0)
    gray_frame = good[:f0 + 9] + b"\x01" + good[f0 + 10:]  # a frame header that only admits to its first component
    edits["more_scan_components_than_frame"] = gray_frame[:f0 + 2] + (good[f0 + 2:f0 + 4]) + gray_frame[f0 + 4:]
    # a frame header that asks for fewer MCU rows than the scan carries: 28 MCUs, with DRI = 4 the restart check behind the
    # last MCU finds one more RSTn (the optimizer copies it), the rest of the scan is skipped as fill
    edits["frame_height_64"] = good[:f0 + 5] + b"\x00\x40" + good[f0 + 7:]
    edits["frame_height_33"] = good[:f0 + 5] + b"\x00\x21" + good[f0 + 7:]
    edits["sof2_in_scan_tail"] = body + b"\xff\xc2" + bytes(12)
    edits["second_sos_behind_scan"] = body + good[a:b] + bytes(30) + b"\xff\xd9"
    edits["truncated_in_scan"] = good[:b + (len(good) - b) // 2]
    edits["truncated_in_scan_then_eoi"] = good[:b + (len(good) - b) // 2] + b"\xff\xd9"
    return edits


@pytest.mark.parametrize("restart", [0, 4])
def test_marker_walk_around_the_scan_follows_the_reference(restart):
    """The reference reads the next marker only after the scan decoded, and its optimizer walks the file twice with
    looser rules (JpegOptimizer.cs Scan() / Optimize()): trailing bytes, missing tables and markers planted behind the
    scan must end in the same exception class -- or the same bytes -- through both the decoder and the optimizer."""
    good = bytes(jpegsynth.encode(104, 72, "420", 80, restart, seed=91))   # 35 MCUs: not a multiple of 4
    edits = _structural_edits(good)
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    keys = list(edits)
    ob = jl.OptimizeBatch().upload([edits[k] for k in keys], False).run()
    db = jl.Batch().upload([edits[k] for k in keys], jl.FMT_INTERLEAVED_U8).decode().sync()
    problems = []
    for i, k in enumerate(keys):
        try:
            ref, ref_kind = po.optimize(edits[k], False), "OK"
        except po.OracleError as e:
            ref, ref_kind = None, e.kind
        res, _ = ob.result(i)
        mine = names.get(res.status, str(res.status))
        if mine == "NotSupportedException" and k in ("second_sos_behind_scan", "scan_names_no_component", "second_scan_names_no_component", "1_unread_eoi_eoi", "1_unread_eoi_com"):
            pass  # several scans / a swallowed terminator the walk survives: refused by design (optimizer fences)
        elif mine != ref_kind:
            problems.append(("optimize", k, ref_kind, mine, res.detail))
        elif ref is not None and ob.output(i) != ref:
            problems.append(("optimize bytes", k))
        try:
            ref, ref_kind = po.decode_8bit(edits[k])[0], "OK"
        except po.OracleError as e:
            ref, ref_kind = None, e.kind
        res = db.result(i)
        mine = names.get(res.status, str(res.status))
        if mine != ref_kind:
            problems.append(("decode", k, ref_kind, mine, res.detail))
        elif ref is not None and not np.array_equal(db.output(i), ref):
            problems.append(("decode samples", k))
    ob.close()
    db.close()
    assert not problems, problems


def test_a_middle_scan_that_leaves_one_byte_unread_is_planned_again():
    """Round 6 (a fence until then: NotSupportedException).  One whole byte between the last MCU of a MIDDLE scan and the marker behind
    it: the reference's reader resumes one byte INTO that marker (JpegHuffmanBaselineScanDecoder.cs:167-176), so the next scan's SOS --
    or the DHT in front of it -- is stepped over as fill and the walk goes on with whatever marker follows: which scans exist depends
    on the decode.  The batch plans such a file again with that knowledge (DeviceBatch::redo_swallowed): status class and pixels of
    the checker, for each of the three scans of a multi-scan baseline frame, with and without restart intervals, beside files that do
    not need it; and whatever is asked for first (result or pixels)."""
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    files, tags = [], []
    for dri in (0, 4):
        good = bytes(jpegsynth.encode(96, 72, "444", 80, dri, seed=31 + dri, noninterleaved=True))
        sos = [i for i in range(len(good) - 1) if good[i] == 0xFF and good[i + 1] == 0xDA]
        eoi = len(good) - 2
        assert len(sos) == 3
        for k in (1, 2, 3):
            for where, at in (("second_sos", sos[1]), ("third_sos", sos[2]), ("eoi", eoi)):
                files.append(good[:at] + bytes([0x5A] * k) + good[at:])
                tags.append((dri, k, where))
        # a DHT of its own in front of the third scan: the swallowed marker is then the DHT's, the third SOS is found, and the scan
        # decodes with the tables in force before (the same ones here)
        dht = good[good.index(b"\xff\xc4"):]
        dht = dht[:2 + int.from_bytes(dht[2:4], "big")]
        files.append(good[:sos[2]] + b"\x5a" + dht + good[sos[2]:])
        tags.append((dri, 1, "dht_before_third_sos"))
        # both middle scans at once: the re-plan is re-planned
        files.append(good[:sos[1]] + b"\x5a" + good[sos[1]:sos[2]] + b"\x5a" + good[sos[2]:])
        tags.append((dri, 1, "second_and_third_sos"))
    refs = []
    for f in files:
        px, _, err = po.decode_8bit_partial(f)
        refs.append(("OK" if err is None else err.kind, px))
    for first in ("result", "pixels"):
        b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
        for i, (kind, px) in enumerate(refs):
            if first == "pixels":
                out = b.output(i)
                res = b.result(i)
            else:
                res = b.result(i)
                out = b.output(i)
            assert names[res.status] == kind, (first, tags[i], kind, res.status, res.detail)
            assert np.array_equal(np.asarray(out), px), (first, tags[i])
        # a second decode of the same upload gives the same again (the re-plans are made per decode)
        b.decode().sync()
        for i in (0, 1, 9):
            assert names[b.result(i).status] == refs[i][0] and np.array_equal(np.asarray(b.output(i)), refs[i][1]), tags[i]
        b.close()
    assert any(t[1] == 1 and t[2] != "eoi" and r[0] == "OK" for t, r in zip(tags, refs))  # (the case exists: the reference decodes on)
    # ... and through JpegDecoder.Decode() (level 3: one device call per scan, the mirror's own marker walk): the scan's reader advance
    # ends one byte into the marker, the walk goes on from there
    for f, t, (kind, px) in zip(files, tags, refs):
        if t[1] != 1:
            continue
        d = jl.JpegDecoder()
        d.SetInput(f)
        d.Identify()
        out = np.zeros(d.Width * d.Height * 3, np.uint8)
        d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, 3, out))
        try:
            d.Decode()
            mine = "OK"
        except jl.JpegError as e:
            mine = type(e).__name__
        assert mine == kind, (t, mine, kind)
        assert np.array_equal(out.reshape(px.shape), px), t


def test_progressive_scan_failures_come_before_later_walk_failures():
    """Every ProcessScan of a progressive frame decodes its scan on the spot, so a failure inside an early scan wins over
    a broken segment further down the file (found by tools/stress_parity.py on corrupted progressive files)."""
    good = read_jpeg("yellowcat_progressive_restart.jpg")
    sos = [i for i in range(len(good) - 1) if good[i] == 0xFF and good[i + 1] == 0xDA]
    assert len(sos) >= 3
    rst = good.index(b"\xff\xd0", sos[0])
    broken_scan = good[:rst] + b"\x12\x34" + good[rst + 2:]
    cases = {
        "broken_first_scan": broken_scan,
        "broken_first_scan_cut_in_third_header": broken_scan[:sos[2] + 5],
        "cut_in_third_header": good[:sos[2] + 5],
        "broken_first_scan_no_tables_for_second": broken_scan[:sos[0]] + broken_scan[sos[0]:sos[1]].replace(b"\xff\xc4", b"\xff\xe9") + broken_scan[sos[1]:],
        "no_tables_for_second": good[:sos[0]] + good[sos[0]:sos[1]].replace(b"\xff\xc4", b"\xff\xe9") + good[sos[1]:],
        "cut_in_second_scan": good[:(sos[1] + sos[2]) // 2],
        # a frame header whose length field swallows every scan: Decode() succeeds without writing a sample
        "frame_header_covers_the_scans": good[:good.index(b"\xff\xc2") + 2] + (len(good) - good.index(b"\xff\xc2") - 6).to_bytes(2, "big") + good[good.index(b"\xff\xc2") + 4:],
    }
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    keys = list(cases)
    outs, results = jl.decode_batch([cases[k] for k in keys], jl.FMT_INTERLEAVED_U8)
    kinds = {}
    for k, out, res in zip(keys, outs, results):
        try:
            ref, kind = po.decode_8bit(cases[k])[0], "OK"
        except po.OracleError as e:
            ref, kind = None, e.kind
        kinds[k] = kind
        assert names.get(res.status) == kind, (k, kind, res.status, res.detail)
        if ref is not None:
            assert np.array_equal(np.asarray(out), ref), k
    # the order is what is tested (a cut header is no such case: Identify() already walks into it before Decode() starts)
    assert kinds["broken_first_scan_no_tables_for_second"] != kinds["no_tables_for_second"], kinds


def test_decoder_mirror_reports_a_bad_progressive_scan_before_a_walk_failure_behind_it():
    """The same order through JpegDecoder.Decode() (level 3, jpgpu_decoder_decode): the mirror collects a progressive frame's scans
    during the walk and decodes them in one device pass at Dispose; a failure of the walk BEHIND a bad scan used to be reported in
    the scan's place (DESIGN.md 5 until round 6).  The reference ran the scan at its SOS (JpegDecoder.cs:592-599): the scan's
    exception -- class and message -- is the one that leaves Decode(), and the writer holds the partial flush of the store."""
    good = read_jpeg("yellowcat_progressive_restart.jpg")
    sos = [i for i in range(len(good) - 1) if good[i] == 0xFF and good[i + 1] == 0xDA]
    rst = good.index(b"\xff\xd0", sos[0])
    broken_scan = good[:rst] + b"\x12\x34" + good[rst + 2:]
    cases = {
        "broken_first_scan_no_tables_for_second": broken_scan[:sos[0]] + broken_scan[sos[0]:sos[1]].replace(b"\xff\xc4", b"\xff\xe9") + broken_scan[sos[1]:],
        "no_tables_for_second": good[:sos[0]] + good[sos[0]:sos[1]].replace(b"\xff\xc4", b"\xff\xe9") + good[sos[1]:],
        # a Huffman table segment too short for its counts in front of the third scan: Identify() steps over it, Decode() parses it
        "broken_first_scan_bad_dht_before_third": broken_scan[:sos[2]] + b"\xff\xc4\x00\x05\x00\xff\xff" + broken_scan[sos[2]:],
        "bad_dht_before_third": good[:sos[2]] + b"\xff\xc4\x00\x05\x00\xff\xff" + good[sos[2]:],
        "broken_first_scan": broken_scan,
    }
    seen = {}
    for k, data in cases.items():
        ref, info, err = po.decode_8bit_partial(data)
        d = jl.JpegDecoder()
        d.SetInput(data)
        d.Identify()
        out = np.zeros(d.Width * d.Height * d.NumberOfComponents, np.uint8)
        d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, d.NumberOfComponents, out))
        try:
            d.Decode()
            mine = None
        except jl.JpegError as e:
            mine = e
        assert (mine is None) == (err is None), (k, mine, err)
        if err is not None:
            assert type(mine).__name__ == err.kind, (k, mine, err)
            assert err.message in str(mine) or str(mine) in err.message, (k, str(mine), err.message)
            seen[k] = (err.kind, err.message)
        assert np.array_equal(out.reshape(ref.shape), ref), k
    assert seen["broken_first_scan_bad_dht_before_third"] == seen["broken_first_scan"] != seen["bad_dht_before_third"], seen


def test_progressive_band_overrun_is_ordered_like_the_file():
    """A corrupted first-pass AC scan (band 1-5) stores a coefficient behind its band, where the 6-63 scan of the same
    component also stores: file order decides.  The two scans used to share a dependency level, and which store won
    depended on where the image sat in the batch (found by tools/stress_parity.py, 24 x 1500 files)."""
    data = read_jpeg(os.path.join("stress", "progressive_band_overrun.jpg"))
    ref = po.decode_8bit(data)[0]
    outs, results = jl.decode_batch([data] * 12, jl.FMT_INTERLEAVED_U8)
    for k, (o, r) in enumerate(zip(outs, results)):
        assert r.status == 0
        assert np.array_equal(np.asarray(o), ref), k


def test_encoder_one_pixel_wide_noise_at_quality_100():
    """Columns of noise one pixel wide: the padded blocks carry eight times the pixels' entropy (the case that overran the
    checker's own output buffer in tools/stress_parity.py)."""
    rng = np.random.default_rng(13)
    imgs = [rng.integers(0, 256, (h, 1)).astype(np.uint8) for h in (188, 75, 1, 8, 9)]
    for mode in (0, 1):
        e = jl.EncodeBatch().upload(imgs, (1, 1), 100, optimize_coding=mode).encode()
        for k, im in enumerate(imgs):
            try:
                ref = po.encode_8bit(im, 1, 1, 100, optimize_coding=mode)
            except po.OracleError:
                ref = None
            try:
                got = e.output(k)
            except jl.JpegError:
                got = None
            assert got == ref, (mode, k)
        e.close()


def test_most_optimal_coding_optimizer_and_encoder():
    """MostOptimalCoding (the package-merge table builder) through both callers: JpegOptimizer and JpegEncoder."""
    files = _optimizer_files()[:5]
    b = jl.OptimizeBatch().set_most_optimal_coding(True).upload(files, False).run()
    for i, f in enumerate(files):
        assert b.output(i) == po.optimize(f, False, most_optimal=True), i
    b.close()
    rng = np.random.default_rng(12)
    imgs = [rng.integers(0, 256, (70, 100, 3)).astype(np.uint8), (rng.integers(0, 64, (45, 61, 3)) * 3).astype(np.uint8)]
    for luma in ((2, 2), (1, 1)):
        e = jl.EncodeBatch().upload(imgs, luma, 80, optimize_coding=2).encode()
        for k, im in enumerate(imgs):
            assert e.output(k) == po.encode_8bit(im, luma[0], luma[1], 80, optimize_coding=2), (luma, k)
        e.close()


def test_optimizer_edge_sizes_and_samplings_in_one_batch():
    """Tiny and ragged geometries for every sampling, with and without restart intervals, through the optimizer as ONE batch:
    the bytes of the restatement, or its failure class (an MCU count that is a multiple of DRI makes the reference give up)."""
    files = []
    for (w, h) in [(1, 1), (7, 9), (8, 8), (16, 16), (17, 1), (1, 33), (31, 17), (48, 40), (129, 65)]:
        for ss in ("444", "422", "420", "gray"):
            for dri in (0, 1, 3):
                files.append(bytes(jpegsynth.encode(w, h, ss, 70, dri, seed=w * 131 + h * 7 + dri)))
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    for strip in (True, False):
        b = jl.OptimizeBatch().upload(files, strip).run()
        n_ok = 0
        for i, f in enumerate(files):
            try:
                ref, kind = po.optimize(f, strip), "OK"
            except po.OracleError as e:
                ref, kind = None, e.kind
            res, size = b.result(i)
            assert names.get(res.status) == kind, (i, strip, kind, res.status, res.detail)
            if ref is not None:
                assert b.output(i) == ref, (i, strip)
                n_ok += 1
        assert n_ok > len(files) // 2
        b.close()


def test_optimizer_low_quality_tables_with_ff_bytes():
    """Optimize() does not skip the DQT / DHT payloads (JpegOptimizer.cs:596-609): its marker search runs through them, and a
    low-quality quantisation table holds FF bytes -- followed by C2 the reference throws "Progressive JPEG is not supported
    currently." on a baseline file, followed by other values it skips or copies phantom segments.  Whatever it does, the GPU
    path must do the same."""
    files = []
    for q in range(3, 31):
        files.append(_synth_jpeg(200, 152, quality=q, seed=q, gray=(q % 2 == 0)))
        files.append(_synth_jpeg(918, 866, quality=q, seed=100 + q, gray=True))
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    kinds = set()
    for strip in (True, False):
        b = jl.OptimizeBatch().upload(files, strip).run()
        for i, f in enumerate(files):
            try:
                ref, kind = po.optimize(f, strip), "OK"
            except po.OracleError as e:
                ref, kind = None, e.kind
            kinds.add(kind)
            res, size = b.result(i)
            assert names.get(res.status) == kind, (i, strip, kind, res.status, res.detail)
            if ref is not None:
                assert b.output(i) == ref, (i, strip)
        b.close()
    assert "OK" in kinds


# ------------------------------------------------------------------------------------------------ stress reproducers

def test_a_frame_beyond_the_devices_memory_fails_by_itself():
    """A corrupted header (here: an APP0 length that makes the parser find `FF C1` inside the quantisation tables) can declare a
    frame of 62 868 x 62 968 x 183 components.  The reference's caller would fail allocating the writer's buffer; here the
    image gets status 7 and the rest of the batch decodes (it used to take the batch's allocation, and every image, with it)."""
    big = read_jpeg("stress_oversize_frame_460.jpg")
    good = jpegsynth.encode(64, 48, "420", 75, 2, seed=3)
    outs, results = jl.decode_batch([good, big, good], jl.FMT_INTERLEAVED_U8)
    assert results[0].status == 0 and results[2].status == 0
    assert results[1].status == 7, results[1].status
    ref = po.decode_8bit(good)[0]
    assert np.array_equal(outs[0], ref) and np.array_equal(outs[2], ref)



def _stress_files():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stress")
    return sorted(f for f in os.listdir(d) if f.endswith(".jpg"))


@pytest.mark.parametrize("name", _stress_files())
def test_stress_reproducers_match_the_oracle(name):
    """Inputs on which tools/stress_parity.py once found the GPU path and the restatement apart (batch-position dependent
    progressive levels, frame heights smaller than the scan, optimizer pieces, refused envelopes): each one alone and
    inside a batch of copies and neighbours, through the decoder (YCbCr8 and RGBA) and the optimizer."""
    data = read_jpeg(os.path.join("stress", name))
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    try:
        ref, ref_kind = po.decode_8bit(data)[0], "OK"
    except po.OracleError as e:
        ref, ref_kind = None, e.kind
    partial = None  # what a FAILING decode left in the writer (progressive: the disposed store; sequential: the blocks in front of the throw)
    if ref is None:
        try:
            partial = po.decode_8bit_partial(data)[0]
        except po.OracleError:
            partial = None
    neighbours = [jpegsynth.encode(64, 48, "420", 75, 2, seed=3), read_jpeg("progress.jpg")]
    files = [data, neighbours[0], data, neighbours[1], data]
    for fmt in (jl.FMT_INTERLEAVED_U8, jl.FMT_RGBA_U8):
        b = jl.Batch().upload(files, fmt).decode().sync()
        for i in (0, 2, 4):
            res = b.result(i)
            if ref_kind == "OK" and fmt == jl.FMT_RGBA_U8 and ref.shape[2] not in (1, 3):
                continue
            assert names.get(res.status) == ref_kind, (name, i, fmt, ref_kind, res.status, res.detail)
            if ref is not None:
                want = ref if fmt == jl.FMT_INTERLEAVED_U8 else po.ycbcr8_to_rgb(ref, rgba=True, gray=(ref.shape[2] == 1))
                assert np.array_equal(b.output(i), want), (name, i, fmt)
            elif partial is not None and fmt == jl.FMT_INTERLEAVED_U8 and b.image_info(i).status == 0:
                assert np.array_equal(b.output(i), partial), (name, i, "writer state of the failing decode")
        b.close()
    for strip in (False, True):
        try:
            oref, okind = po.optimize(data, strip), "OK"
        except po.OracleError as e:
            oref, okind = None, e.kind
        ob = jl.OptimizeBatch().upload([data, neighbours[0], data], strip).run()
        for i in (0, 2):
            res, _ = ob.result(i)
            mine = names.get(res.status)
            if mine == "NotSupportedException" and okind != mine:
                continue  # optimizer fences (several scans, progressive): refused by design, DESIGN.md
            assert mine == okind, (name, strip, okind, mine, res.detail)
            if oref is not None:
                assert ob.output(i) == oref, (name, strip)
        ob.close()


def test_offsets_inside_one_image_pass_4_gib():
    """38 000 x 38 000 4:2:0 (1.44 Gpixel, 8.5 M MCUs): the image's samples (4.33 GB) and its coefficient blocks (4.33 GB) both
    reach past 2^32 bytes, its entropy segment is ~180 MB -- every in-image offset of K1 / K2 / K3 has to be 64-bit.  With
    restart intervals (K2) here; the DRI = 0 form and the format's maximum, 65 535 x 65 535, go through
    tools/trace/huge_image.py (both exact)."""
    w = h = 38000
    data = bytes(jpegsynth.encode(w, h, "420", 75, 8, seed=9))
    outs, results = jl.decode_batch([data])
    assert results[0].status == 0
    ref = po.decode_8bit(data)[0]
    assert outs[0].shape == ref.shape == (h, w, 3)
    for y0 in range(0, h, 4096):  # in slabs: no 4 GB comparison mask
        assert np.array_equal(outs[0][y0:y0 + 4096], ref[y0:y0 + 4096]), y0


def test_empty_and_header_only_inputs_fail_like_the_reference():
    """No files at all, files of zero / one / two bytes, SOI alone, SOI + EOI, headers without a scan, a frame of zero lines:
    the same exception class and message as the checker, per file, beside a good neighbour in the same batch."""
    outs, results = jl.decode_batch([])
    assert outs == [] and results == []
    good = bytes(jpegsynth.encode(64, 48, "420", 75, 2, seed=3))
    sos = good.index(b"\xff\xda")
    sof = good.index(b"\xff\xc0")
    cases = {
        "zero_bytes": b"",
        "one_byte": b"\xff",
        "soi_only": b"\xff\xd8",
        "soi_eoi": b"\xff\xd8\xff\xd9",
        "not_a_jpeg": b"GIF89a" + bytes(32),
        "headers_then_eoi": good[:sos] + b"\xff\xd9",
        "headers_cut": good[:sos],
        "zero_lines": good[:sof + 5] + b"\x00\x00" + good[sof + 7:],
        "zero_width": good[:sof + 7] + b"\x00\x00" + good[sof + 9:],
    }
    names = list(cases)
    files = [good] + [cases[n] for n in names] + [good]
    outs, results = jl.decode_batch(files)
    ref_good = po.decode_8bit(good)[0]
    assert results[0].status == 0 and results[-1].status == 0
    assert np.array_equal(outs[0], ref_good) and np.array_equal(outs[-1], ref_good)
    kinds = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException"}
    for n, res, out in zip(names, results[1:-1], outs[1:-1]):
        try:
            ref = po.decode_8bit(cases[n])[0]
            kind, msg = "OK", ""
        except po.OracleError as e:
            ref, kind, msg = None, e.kind, e.message
        assert kinds[res.status] == kind, (n, kind, msg, res.status, res.detail)
        if kind == "OK":
            assert np.array_equal(out, ref), n
            continue
        # the message: through the JpegDecoder mirror (SetInput / Identify / Decode raise the reference's exceptions)
        with pytest.raises(jl.JpegError) as ei:
            d = jl.JpegDecoder()
            d.SetInput(cases[n])
            d.Identify()
            buf = np.zeros(max(1, d.Width * d.Height * 3), np.uint8)
            d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, 3, buf))
            d.Decode()
        assert type(ei.value).__name__ == kind and str(ei.value) == msg, (n, kind, msg, type(ei.value).__name__, str(ei.value))


@pytest.mark.parametrize("ri", [1, 2, 5, 8, 64, 5000])
@pytest.mark.parametrize("w,h,luma,q,opt", [(90, 70, (2, 2), 75, 0), (257, 131, (2, 1), 92, 0), (64, 64, (1, 1), 40, 1), (301, 203, (2, 2), 100, 2),
                                            (33, 17, (1, 2), 60, 0)])
def test_encoder_restart_intervals(w, h, luma, q, opt, ri):
    """The encoder's restart-interval mode (an extension: JpegEncoder has none; SURVEY 8f N3 "+ DRI emission"): DRI in front of
    SOF0, one-bit padding + RSTm + predictor reset every ri MCUs.  Byte-exact against the checker's definition (T.81 / libjpeg),
    with standard and optimised tables; the stream decodes -- checker and GPU -- to the pixels of the checker's own stream, its
    coefficients are the ones that went in, and libjpeg-turbo reads it."""
    import io

    from PIL import Image
    rgb = _enc_image(w, h, w * 3 + h + ri)
    ycc = po.rgb_to_ycbcr8(rgb)
    ref, ref_coefs = po.encode_8bit(ycc, luma[0], luma[1], q, want_coefficients=True, optimize_coding=opt, restart_interval=ri)
    b = jl.EncodeBatch().upload([ycc], luma, q, optimize_coding=opt, restart_interval=ri).encode()
    assert np.array_equal(b.coefficients(0), ref_coefs)
    out = b.output(0)
    assert out == ref
    mcus = -(-w // (8 * luma[0])) * -(-h // (8 * luma[1]))
    assert out.count(b"\xff\xdd\x00\x04" + bytes([ri >> 8, ri & 255])) == 1
    outs, results = jl.decode_batch([out])
    assert results[0].status == 0 and results[0].terminator == 0xD9 and np.array_equal(outs[0], po.decode_8bit(ref)[0])
    if opt == 0:
        dec_coefs = po.decode_coefficients(out)[0]
        assert np.array_equal(np.asarray(dec_coefs).reshape(-1, 64), ref_coefs)
    im = Image.open(io.BytesIO(out))
    im.load()
    assert im.size == (w, h)
    assert (mcus - 1) // ri == sum(1 for _ in _rst_markers(out))


def _rst_markers(data):
    """Offsets of the RSTm markers of a baseline stream's entropy segment (FF Dn behind SOS; FF 00 is stuffing)."""
    i = data.index(b"\xff\xda")
    i += 2 + ((data[i + 2] << 8) | data[i + 3])
    while i + 1 < len(data):
        if data[i] == 0xFF and 0xD0 <= data[i + 1] <= 0xD7:
            yield i
            i += 2
        else:
            i += 1


def test_encoder_restart_intervals_in_a_mixed_batch():
    """Images with and without restart intervals, gray and colour, in one batch (lane per block and lane per interval side by
    side in the same launches)."""
    imgs = [po.rgb_to_ycbcr8(_enc_image(96, 64, 1)), po.rgb_to_ycbcr8(_enc_image(50, 70, 2)), po.rgb_to_ycbcr8(_enc_image(256, 256, 3))[..., 0],
            po.rgb_to_ycbcr8(_enc_image(400, 300, 4))]
    ris = [0, 3, 4, 7]
    b = jl.EncodeBatch()
    n = len(imgs)
    import ctypes as C
    from jpeglibrary_amd import _capi
    ptrs = (C.c_void_p * n)()
    params = (_capi.EncodeParams * n)()
    keep = []
    for i, (im, ri) in enumerate(zip(imgs, ris)):
        a = np.ascontiguousarray(im if im.ndim == 3 else im[..., None])
        keep.append(a)
        ptrs[i] = a.ctypes.data
        params[i] = _capi.EncodeParams(a.shape[1], a.shape[0], a.shape[2], 2 if a.shape[2] == 3 else 1, 2 if a.shape[2] == 3 else 1, 80, 0, 0, ri)
    b._check(_capi.lib.jpgpu_encoder_upload(b._h, ptrs, params, n))
    b._n = n
    b.encode()
    for i, (im, ri) in enumerate(zip(imgs, ris)):
        lum = 2 if im.ndim == 3 else 1
        assert b.output(i) == po.encode_8bit(im, lum, lum, 80, restart_interval=ri), i
    b.close()




def _encode_action(ycbcr, quality, optimize_coding, luma=(2, 2), most_optimal=False, tables=None):
    """apps/JpegEncode/EncodeAction.cs:38-63, line by line, on the JpegEncoder mirror."""
    h, w, _ = ycbcr.shape
    encoder = jl.JpegEncoder()
    if tables is None:
        encoder.SetQuantizationTable(jl.JpegStandardQuantizationTable.ScaleByQuality(jl.JpegStandardQuantizationTable.GetLuminanceTable(0, 0), quality))
        encoder.SetQuantizationTable(jl.JpegStandardQuantizationTable.ScaleByQuality(jl.JpegStandardQuantizationTable.GetChrominanceTable(0, 1), quality))
    else:
        encoder.SetQuantizationTable(jl.JpegQuantizationTable(0, 0, tables[0]))
        encoder.SetQuantizationTable(jl.JpegQuantizationTable(0, 1, tables[1]))
    if optimize_coding:
        encoder.SetHuffmanTable(True, 0)
        encoder.SetHuffmanTable(False, 0)
        encoder.SetHuffmanTable(True, 1)
        encoder.SetHuffmanTable(False, 1)
    else:
        encoder.SetHuffmanTable(True, 0, jl.JpegStandardHuffmanEncodingTable.GetLuminanceDCTable())
        encoder.SetHuffmanTable(False, 0, jl.JpegStandardHuffmanEncodingTable.GetLuminanceACTable())
        encoder.SetHuffmanTable(True, 1, jl.JpegStandardHuffmanEncodingTable.GetChrominanceDCTable())
        encoder.SetHuffmanTable(False, 1, jl.JpegStandardHuffmanEncodingTable.GetChrominanceACTable())
    encoder.MostOptimalCoding = most_optimal
    encoder.AddComponent(1, 0, 0, 0, luma[0], luma[1])  # Y component
    encoder.AddComponent(2, 1, 1, 1, 1, 1)  # Cb component
    encoder.AddComponent(3, 1, 1, 1, 1, 1)  # Cr component
    encoder.SetInputReader(jl.JpegBufferInputReader(w, h, 3, ycbcr))
    writer = bytearray()
    encoder.SetOutput(writer)
    encoder.Encode()
    return bytes(writer)


@pytest.mark.parametrize("quality,optimize_coding,most_optimal", [(75, False, False), (30, True, False), (92, True, True)])
def test_jpeg_encoder_mirror_runs_the_encode_action_sequence(quality, optimize_coding, most_optimal):
    """The reference's own caller (EncodeAction) transcribed onto the JpegEncoder mirror: the bytes of the checker."""
    ycc = po.rgb_to_ycbcr8(_enc_image(150, 98, quality))
    out = _encode_action(ycc, quality, optimize_coding, most_optimal=most_optimal)
    assert out == po.encode_8bit(ycc, 2, 2, quality, optimize_coding=(2 if most_optimal else 1) if optimize_coding else 0)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_quantisation_quotients_on_random_tables_and_full_range_samples(seed):
    """E1b does not divide: it runs the five instructions hipcc's IEEE division ends in, with the divisor's refined reciprocal
    taken from LDS (quant_pair / quant_divide, encode_kernels.hip).  White noise puts the DCT coefficients all over their
    range, random tables put every divisor 1..255 under them: 390 000 quotients per case, each rounded half-to-even right
    behind the division -- one quotient an ulp off near a tie changes a coefficient and the bytes of the stream."""
    rng = np.random.default_rng(1000 + seed)
    img = rng.integers(0, 256, (512, 512, 3)).astype(np.uint8)
    img[:64] = (img[:64] // 128) * 255  # and some blocks at the extremes
    lum, chr_ = rng.integers(1, 256, 64), rng.integers(1, 256, 64)
    if seed == 4:
        lum, chr_ = np.full(64, 255), np.arange(1, 65)
    for luma in ((2, 2), (1, 1)):
        ref = po.encode_8bit(img, luma[0], luma[1], 50, optimize_coding=False, quant_tables=(lum, chr_))
        assert _encode_action(img, 50, False, luma=luma, tables=(lum.tolist(), chr_.tolist())) == ref


def test_jpeg_encoder_mirror_takes_the_callers_quantization_tables():
    """SetQuantizationTable with tables that are no scaled standard table: flat, steep, and libjpeg's quality-100 all-ones."""
    ycc = po.rgb_to_ycbcr8(_enc_image(97, 61, 3))
    rng = np.random.default_rng(8)
    for lum, chr_ in [(np.full(64, 7), np.full(64, 19)), (np.arange(1, 65), np.arange(64, 0, -1) * 3), (np.ones(64), np.ones(64)),
                      (rng.integers(1, 256, 64), rng.integers(1, 256, 64))]:
        for opt in (False, True):
            ref = po.encode_8bit(ycc, 2, 1, 50, optimize_coding=opt, quant_tables=(lum, chr_))
            assert _encode_action(ycc, 50, opt, luma=(2, 1), tables=(lum.tolist(), chr_.tolist())) == ref
            outs, results = jl.decode_batch([ref])
            assert results[0].status == 0 and np.array_equal(outs[0], po.decode_8bit(ref)[0])
    b = jl.EncodeBatch().upload([ycc], (2, 2), 75)
    with pytest.raises(jl.ArgumentException):
        b.set_quantization_table(0, 0, np.zeros(64))
    with pytest.raises(jl.ArgumentException):
        b.set_quantization_table(0, 1, np.full(64, 256))
    with pytest.raises(jl.NotSupportedException):
        b.set_quantization_table(0, 2, np.ones(64))
    b.close()

