"""The reference's own benchmark input (tests/JpegLibrary.Benchmarks/DecoderBenchmark.cs:19-43, rebuilt by bench.het_canvas):
an 8192 x 8192 baseline 4:2:0 Q75 frame without restart markers -- ONE 67-Mpixel scan, i.e. one K2S chain, of real content
(HETissueSlide.jpg) beside three quarters of black -- decoded the way the benchmark does it (Identify + Decode + RGBA)."""
import numpy as np
import pytest

import bench
import jpeglibrary_amd as jl
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def canvas():
    return bench.het_canvas(75)


def test_the_benchmark_canvas_is_what_the_reference_builds(canvas):
    info, _ = po.identify(canvas)
    assert (info.width, info.height, info.ncomp) == (8192, 8192, 3)
    assert info.sof == 0xC0 and info.restart_interval == 0  # baseline, one entropy-coded segment for the whole frame
    assert [(info.comp[c].h, info.comp[c].v) for c in range(3)] == [(2, 2), (1, 1), (1, 1)]  # 4:2:0


def test_the_benchmark_canvas_decodes_bit_exact_as_rgba_and_as_ycbcr(canvas):
    ref, _ = po.decode_8bit(canvas)
    outs, res = jl.decode_batch([canvas], jl.FMT_RGBA_U8)
    assert res[0].status == 0, (res[0].status, res[0].detail)
    assert np.array_equal(outs[0], po.ycbcr8_to_rgb(ref, rgba=True))
    del outs
    outs, res = jl.decode_batch([canvas], jl.FMT_INTERLEAVED_U8)
    assert res[0].status == 0
    assert np.array_equal(outs[0], ref)


def test_a_batch_of_benchmark_canvases_beside_small_frames(canvas):
    """the 67-Mpixel scan (tens of thousands of subsequences) in one batch with frames of a few subsequences"""
    from tools import jpegsynth

    small = [jpegsynth.encode(640, 368, "420", 75, 0, seed=5), jpegsynth.encode(96, 64, "444", 90, 0, seed=6)]
    files = [small[0], canvas, small[1], canvas]
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
    ref = po.decode_8bit(canvas)[0]
    for i, f in enumerate(files):
        assert b.result(i).status == 0, i
        assert np.array_equal(b.output(i), ref if f is canvas else po.decode_8bit(f)[0]), i


@pytest.mark.parametrize("subsampling,optimize,mode", [("4:2:0", False, "RGB"), ("4:4:4", False, "RGB"), ("4:2:2", True, "RGB"), ("4:4:4", True, "L")])
def test_flat_regions_do_not_cost_a_round_per_subsequence(subsampling, optimize, mode):
    """A constant region is the same few bits over and over: nothing for a self-synchronising decoder to lock on to, so the
    right state used to advance one subsequence per round (4 107 rounds on the benchmark canvas).  Twin subsequences (identical
    bits, the same entry state: the same result without decoding) resolve such a region in one walk, whatever the period of the
    flat MCU is against the subsequence length (32 bits for 4:2:0 under the standard tables, 14 for 4:4:4, other values under
    optimised tables): bit-exact, and in a bounded number of rounds."""
    import io

    from PIL import Image

    rng = np.random.default_rng(7)
    px = np.full((1536, 2048, 3), (200, 30, 90), np.uint8)
    px[:192, :320] = rng.integers(0, 256, (192, 320, 3), dtype=np.uint8)      # content, then a long flat stretch in every MCU row
    px[700:760, 900:1500] = rng.integers(0, 256, (60, 600, 3), dtype=np.uint8)  # an island inside the flat region
    img = Image.fromarray(px).convert(mode)
    buf = io.BytesIO()
    kw = {} if mode == "L" else {"subsampling": subsampling}
    img.save(buf, format="JPEG", quality=75, optimize=optimize, **kw)
    data = buf.getvalue()
    fmt = jl.FMT_INTERLEAVED_U8
    b = jl.Batch().upload([data], fmt).decode().sync()
    assert b.result(0).status == 0
    ref = po.decode_8bit(data)[0]
    assert np.array_equal(b.output(0), ref)
    assert b.subseq_rounds() <= 48, b.subseq_rounds()
    # the optimizer's symbol transcode rides on the same synchronisation
    assert jl.optimize_batch([data], strip=False)[0] == po.optimize(data, False)


def test_progressive_crops_of_the_benchmark_image_match_the_oracle():
    """bench.py --workload het_progressive: config 5 on real content (crops of the reference benchmark's image re-encoded as
    progressive 4:2:0 by Pillow: coefficients in every band, long refinement scans) -- three frames against the restatement."""
    files = bench.het_progressive_batch(3, 3840, 2160, 75, 7, 3)
    info, _ = po.identify(files[0])
    assert (info.width, info.height, info.sof) == (3840, 2160, 0xC2)
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
    for i, f in enumerate(files):
        assert b.result(i).status == 0, i
        assert np.array_equal(b.output(i), po.decode_8bit(f)[0]), i
    assert b.progressive_fallbacks() == 0
    assert files[0] != files[1] != files[2]
    b.close()


@pytest.mark.parametrize("luma,optimize", [((2, 2), False), ((1, 1), False), ((2, 2), True)])
def test_the_encoder_benchmarks_canvas_encodes_byte_exact(canvas, luma, optimize):
    """tests/JpegLibrary.Benchmarks/EncoderBenchmark.cs:21-58, 77-135: the decoded 8192 x 8192 canvas through the encoder, 4:2:0 and
    4:4:4, Q75, standard tables (and EncodeAction's optimizeCoding): the finished stream equals the restatement's byte for byte --
    real content at the reference's own size (268 MB of pixels, a million MCUs, three quarters of them black)."""
    # Setup(): decode + ConvertYCbCr8ToRgba32 -> Rgba32 pixels; the benchmark: ConvertRgba32ToYCbCr8 + Encode()
    rgba = np.ascontiguousarray(jl.decode_batch([canvas], jl.FMT_RGBA_U8)[0][0])
    assert rgba.shape == (8192, 8192, 4)
    got = jl.encode_batch([rgba], luma, 75, rgb=True, optimize_coding=optimize)[0]
    ref = po.encode_8bit(po.rgba_to_ycbcr8(rgba), luma[0], luma[1], 75, optimize_coding=optimize)
    assert len(got) == len(ref) and got == ref
    if luma == (2, 2) and not optimize:  # three-byte pixels of the same image: the same stream
        assert jl.encode_batch([np.ascontiguousarray(rgba[..., :3])], luma, 75, rgb=True)[0] == ref
