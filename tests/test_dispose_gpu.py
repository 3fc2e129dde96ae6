"""Dispose() as the reference runs it (ScanDecoder/JpegHuffmanProgressiveScanDecoder.cs:421-470): EVERY component slot of the scan
decoder, as the last scans left it, dequantises + transforms + level-shifts its component's blocks in place, then the allocator
flushes all blocks to the writer.  In libjpeg's scan order the slots end up as {Y, Cb, Cr}.  In other -- perfectly legal -- orders
they do not: a file whose last luma scan comes before the chroma refinements ends with slots {Cb, Cb, Cr}: Cb is transformed
twice (the second time reading its own samples as zig-zag coefficients), Y never (its quantised coefficients reach the writer as
samples).  Rounds 1-3 refused such files; now dispose_pass_kernel does literally what the reference does."""
import io

import numpy as np
import pytest

import jpeglibrary_amd as jl
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _units(data):
    """the file as [header + first scan, (tables +) second scan, ...] and the trailer: every unit ends behind a scan's entropy data"""
    d = bytes(data)
    assert d[:2] == b"\xff\xd8"
    p, start, units = 2, 0, []
    while p + 1 < len(d):
        assert d[p] == 0xFF, p
        m = d[p + 1]
        if m == 0xD9:
            break
        n = (d[p + 2] << 8) | d[p + 3]
        p += 2 + n
        if m == 0xDA:
            while not (d[p] == 0xFF and d[p + 1] != 0x00 and not 0xD0 <= d[p + 1] <= 0xD7 and d[p + 1] != 0xFF):
                p += 1
            units.append(d[start:p])
            start = p
    return units, d[start:]


def _progressive(w=200, h=136, subsampling="4:2:0", seed=3):
    from PIL import Image

    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    px = np.stack([128 + 90 * np.sin(x / 17.0) * np.cos(y / 11.0), 128 + 70 * np.cos((x + y) / 23.0), 60 + (x * 3 + y * 5) % 140], -1)
    px = np.clip(px + rng.normal(0, 12, px.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(px).save(buf, format="JPEG", quality=80, progressive=True, subsampling=subsampling)
    return buf.getvalue()


def _check(data, fmts=("interleaved", "rgba", "planar_i16")):
    ref, _ = po.decode_8bit(data)
    outs, res = jl.decode_batch([data], jl.FMT_INTERLEAVED_U8)
    assert res[0].status == 0, (res[0].status, res[0].detail)
    assert np.array_equal(outs[0], ref)
    if "rgba" in fmts:
        outs, res = jl.decode_batch([data], jl.FMT_RGBA_U8)
        assert res[0].status == 0 and np.array_equal(outs[0], po.ycbcr8_to_rgb(ref, rgba=True))
    return ref


@pytest.mark.parametrize("subsampling", ["4:2:0", "4:4:4", "4:2:2"])
def test_last_luma_scan_before_the_chroma_refinements(subsampling):
    data = _progressive(subsampling=subsampling)
    units, tail = _units(data)
    assert len(units) == 10  # libjpeg's script
    straight = _check(data)
    # ... DC refinement, Y final refinement, Cr refinement, Cb refinement: the decoder's slots end as {Cb, Cb, Cr}
    odd = b"".join(units[:7] + [units[9], units[7], units[8]]) + tail
    mangled = _check(odd)
    assert not np.array_equal(mangled, straight)  # (what the reference makes of it is NOT the picture)
    # Cr keeps its single transform
    assert np.array_equal(mangled[..., 2], straight[..., 2])


def test_script_cut_short_behind_a_chroma_scan_and_two_frames_in_one_batch():
    data = _progressive(232, 120, "4:2:0", seed=9)
    units, tail = _units(data)
    cut = b"".join(units[:3]) + tail  # DC of all, Y AC 1-5, Cr AC 1-63: slots {Cr, Cb, Cr}
    ref_cut, ref_full = po.decode_8bit(cut)[0], po.decode_8bit(data)[0]
    outs, res = jl.decode_batch([data, cut, data], jl.FMT_INTERLEAVED_U8)
    assert [r.status for r in res] == [0, 0, 0]
    assert np.array_equal(outs[0], ref_full) and np.array_equal(outs[1], ref_cut) and np.array_equal(outs[2], ref_full)
    for fmt, conv in ((jl.FMT_RGB_U8, False), (jl.FMT_RGBA_U8, True)):
        outs, res = jl.decode_batch([cut], fmt)
        assert res[0].status == 0 and np.array_equal(outs[0], po.ycbcr8_to_rgb(ref_cut, rgba=conv))


def test_dispose_without_any_scan_flushes_the_zeroed_store():
    """jpgpu_progressive_begin + _dispose with no ProcessScan in between: every block reaches the writer as the allocator left it
    (zeros: the slots' sampling factors are still 0, nothing is transformed) -- ADVICE r3: it used to deliver nothing."""
    from test_per_scan_gpu import Walk

    data = _progressive(96, 64, "4:2:0", seed=5)
    holder = {}

    class Stop(Exception):
        pass

    def on_frame(marker, fh):
        holder["fh"] = fh
        raise Stop

    try:
        Walk(data).run(on_frame, lambda e, sh: 0)
    except Stop:
        pass
    fh = holder["fh"]
    dec = jl.JpegGpuProgressiveScanDecoder(fh)
    out = dec.Dispose(fmt=jl.FMT_INTERLEAVED_U8)
    assert out.size == fh.SamplesPerLine * fh.NumberOfLines * 3 and not np.any(out)
    dec.close()
    # ... and through a JpegBlockOutputWriter: one WriteBlock per block of every component, all zeros
    calls = []

    class Sink:
        def WriteBlock(self, blk, ci, x, y):  # noqa: N802
            calls.append((ci, x, y, int(np.abs(blk).max())))

    dec = jl.JpegGpuProgressiveScanDecoder(fh)
    dec.Dispose(outputWriter=Sink())
    dec.close()
    # (a block of a subsampled component reaches the writer expanded: one WriteBlock per 8 x 8 samples of the frame, per component)
    n_expected = 3 * (fh.SamplesPerLine // 8) * (fh.NumberOfLines // 8)
    assert len(calls) == n_expected and all(c[3] == 0 for c in calls)


def test_a_second_dispose_or_output_stage_flushes_the_same_samples():
    """dispose_pass_kernel transforms the store in place, and it ran in front of EVERY output stage -- a second
    jpgpu_batch_run_idct, or a second Dispose of a session (first into a device layout, then into a writer), transformed the samples
    again.  The pass now runs once per entropy stage."""
    from test_per_scan_gpu import Walk

    data = _progressive(200, 136, "4:2:0", seed=3)
    units, tail = _units(data)
    odd = b"".join(units[:7] + [units[9], units[7], units[8]]) + tail  # slots end as {Cb, Cb, Cr}: the literal Dispose()
    ref = po.decode_8bit(odd)[0]
    b = jl.Batch().upload([odd], jl.FMT_INTERLEAVED_U8).decode().sync()
    assert np.array_equal(b.output(0), ref)
    b.run_idct().sync()
    assert np.array_equal(b.output(0), ref)
    b.decode().sync()  # (a whole pass again: coefficients, one transform)
    assert np.array_equal(b.output(0), ref)
    b.close()
    w = Walk(odd)
    st = {}

    def on_frame(marker, fh):
        st["fh"] = fh
        st["dec"] = jl.JpegGpuProgressiveScanDecoder(fh)

    w.run(on_frame, lambda entropy, sh: st["dec"].ProcessScan(entropy, sh, w.quantization_tables(), w.huffman_tables(), w.dri))
    fh = st["fh"]
    first = st["dec"].Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, 3)
    again = st["dec"].Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, 3)
    buf = np.zeros(fh.SamplesPerLine * fh.NumberOfLines * 3, np.uint8)
    st["dec"].Dispose(outputWriter=jl.JpegBufferOutputWriter8Bit(fh.SamplesPerLine, fh.NumberOfLines, 3, buf))
    st["dec"].close()
    assert np.array_equal(first, ref) and np.array_equal(again, ref)
    assert np.array_equal(buf.reshape(fh.NumberOfLines, fh.SamplesPerLine, 3), ref)
