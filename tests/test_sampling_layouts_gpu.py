"""Well-formed sampling layouts outside 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 (round 6; VERDICT round 5, "What's missing" 3).

The reference accepts factors 1..4 per component (JpegFrameHeader.cs:234-...) and places every block at `(offsetX + x) * 8`
(ScanDecoder/JpegHuffmanBaselineScanDecoder.cs:104, 134): right when a component's factor is the maximum or 1, "wrong" --
deterministic, overlapping, the later WriteBlock winning -- when it is neither.  Both are the contract; the oracle restates them
(oracle/jpegref.c read_scan_baseline), the HIP path must equal it in every output layout.
"""
import numpy as np
import pytest

import jpeglibrary_amd as jl
from oracle import pyoracle as po
from tools import jpegsynth

pytestmark = pytest.mark.gpu

LAYOUTS = [
    ((4, 1), (1, 1), (1, 1)),  # 4:1:1
    ((1, 4), (1, 1), (1, 1)),  # 1 x 4
    ((4, 2), (1, 1), (1, 1)),  # 4 x 2: ten blocks per MCU
    ((2, 4), (1, 1), (1, 1)),
    ((1, 2), (1, 1), (1, 1)),  # 4:4:0
    ((2, 2), (2, 1), (1, 1)),  # Cb at half the luma's height only
    ((2, 2), (1, 2), (2, 1)),
    ((4, 1), (2, 1), (1, 1)),  # Cb's factor is neither the maximum nor 1: the (offsetX + x) * 8 placement overlaps
    ((1, 4), (1, 2), (1, 1)),  # the same vertically (blockOffsetY = (offsetY + y) * 8)
    ((4, 2), (2, 2), (1, 1)),  # thirteen blocks, middle factor in x, maximum in y
    ((4, 2), (2, 1), (4, 1)),  # Cr as wide as the luma
    ((1, 1), (2, 2), (2, 2)),  # the luma BELOW the chroma
    ((2, 1), (1, 1), (2, 1)),
    ((4, 4), (1, 1), (1, 1)),  # eighteen blocks per MCU: fenced (NotSupported), the reference has no limit
]
DRIS = [0, 1, 5]
SIZES = [(203, 117), (64, 64)]


def _files(layout):
    out = []
    for (w, h) in SIZES:
        for dri in DRIS:
            out.append(bytes(jpegsynth.encode(w, h, quality=80, restart_interval=dri, seed=w + dri * 7 + layout[0][0] * 31 + layout[1][0], sampling=layout)))
    # three single-component scans, the blocks in the order the reference reads them
    out.append(bytes(jpegsynth.encode(131, 70, quality=70, restart_interval=0, seed=5, sampling=layout, noninterleaved=True)))
    out.append(bytes(jpegsynth.encode(131, 70, quality=70, restart_interval=4, seed=6, sampling=layout, noninterleaved=True)))
    return out


def _blocks_per_mcu(layout):
    return sum(h * v for h, v in layout)



@pytest.mark.parametrize("layout", LAYOUTS, ids=lambda l: "_".join(f"{h}x{v}" for h, v in l))
def test_interleaved_u8_and_rgba_of_every_layout(layout):
    files = _files(layout)
    if _blocks_per_mcu(layout) > 16:
        # fenced: more than 16 blocks per MCU (T.81 allows 10; the reference has no limit).  Per SCAN: the single-component scans
        # of such a frame decode
        outs, results = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
        for f, o, r in zip(files, outs, results):
            if _is_noninterleaved(f):
                assert r.status == 0 and np.array_equal(o, po.decode_8bit(f)[0])
            else:
                assert r.status == 3
        return
    refs = [po.decode_8bit(f)[0] for f in files]
    outs, results = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
    for i, (o, r, res) in enumerate(zip(outs, refs, results)):
        assert (res.status, res.detail) == (0, 0), i
        assert o.shape == r.shape
        assert np.array_equal(o, r), (i, int((o != r).sum()))
    outs, results = jl.decode_batch(files, jl.FMT_RGBA_U8)
    for i, (o, r, res) in enumerate(zip(outs, refs, results)):
        assert res.status == 0, i
        assert np.array_equal(o, po.ycbcr8_to_rgb(r, rgba=True)), i
    outs, results = jl.decode_batch(files, jl.FMT_RGB_U8)
    for i, (o, r, res) in enumerate(zip(outs, refs, results)):
        assert res.status == 0, i
        assert np.array_equal(o, po.ycbcr8_to_rgb(r, rgba=False)), i


@pytest.mark.parametrize("layout", [l for l in LAYOUTS if _blocks_per_mcu(l) <= 16], ids=lambda l: "_".join(f"{h}x{v}" for h, v in l))
def test_coefficients_planar_i16_and_extended_u16_of_every_layout(layout):
    files = _files(layout)
    # Huffman stage: the coefficient buffer = ReadBlockBaseline's output in scan order
    b = jl.Batch().upload(files, jl.FMT_PLANAR_I16).decode().sync()
    for i, f in enumerate(files):
        assert b.result(i).status == 0, i
        ref, _ = po.decode_coefficients(f)
        assert np.array_equal(b.coefficients(i), ref), i
        # O1: the blocks WriteBlock receives, BEFORE WriteBlockSlow's replication, on the component's own grid
        info, _ = po.identify(f)
        max_h = max(info.comp[c].h for c in range(info.ncomp))
        max_v = max(info.comp[c].v for c in range(info.ncomp))
        calls, _ = po.decode_blocks(f)
        planes = b.output(i)
        bi = b.image_info(i)
        ref_planes = [np.zeros_like(p) for p in planes]
        k = 0
        # the calls come MCU by MCU, scan component by scan component, hs * vs calls per decoded block
        # (ref: ...BaselineScanDecoder.cs:99-134, 238-268): undo the replication call by call
        scans = [[0], [1], [2]] if _is_noninterleaved(f) else [[0, 1, 2]]
        for comps in scans:
            for my in range(bi.mcus_per_column):
                for mx in range(bi.mcus_per_line):
                    for c in comps:
                        h, v = info.comp[c].h, info.comp[c].v
                        hs, vs = max_h // h, max_v // v
                        for y in range(v):
                            for x in range(h):
                                native = np.zeros((8, 8), np.int16)
                                for vv in range(vs):
                                    for hh in range(hs):
                                        ci, cx, cy, blk = calls[k]
                                        k += 1
                                        assert ci == c and cx == (mx * max_h + x) * 8 + 8 * hh and cy == (my * max_v + y) * 8 + 8 * vv
                                        blk = blk.reshape(8, 8)
                                        # block (vv, hh) [i][j] = src[(8 vv + i) >> vshift][(8 hh + j) >> hshift]
                                        native[(8 * vv) // vs:(8 * vv + 8) // vs, (8 * hh) // hs:(8 * hh + 8) // hs] = blk[::vs, ::hs]
                                ref_planes[c][(my * v + y) * 8:(my * v + y) * 8 + 8, (mx * h + x) * 8:(mx * h + x) * 8 + 8] = native
        assert k == len(calls)
        for c in range(3):
            assert np.array_equal(planes[c], ref_planes[c]), (i, c)
    b.close()
    # the xunit tests' sink (JpegExtendingOutputWriter, componentCount 4)
    b = jl.Batch().upload(files, jl.FMT_EXTENDED_U16).decode().sync()
    for i, f in enumerate(files):
        assert b.result(i).status == 0, i
        assert np.array_equal(b.output(i), po.decode_16bit(f, component_count=4)[0]), i
    b.close()


def _is_noninterleaved(f):
    return f.count(b"\xff\xda\x00\x08\x01") == 3


@pytest.mark.parametrize("layout", [LAYOUTS[0], LAYOUTS[7], LAYOUTS[9], LAYOUTS[11]], ids=lambda l: "_".join(f"{h}x{v}" for h, v in l))
def test_decoder_mirror_replays_writeblock_calls_of_unusual_layouts(layout):
    """JpegDecoder.Decode() with an arbitrary writer: the WriteBlock calls (arguments and ORDER) of the reference."""
    f = bytes(jpegsynth.encode(75, 41, quality=75, restart_interval=2, seed=3, sampling=layout))
    ref_calls, _ = po.decode_blocks(f)

    class Recorder(jl.JpegBlockOutputWriter):
        def __init__(self):
            self.calls = []

        def WriteBlock(self, block, component_index, x, y):
            self.calls.append((component_index, x, y, np.array(block, np.int16).copy()))

    d = jl.JpegDecoder()
    d.SetInput(f)
    d.Identify()
    w = Recorder()
    d.SetOutputWriter(w)
    d.Decode()
    assert len(w.calls) == len(ref_calls)
    for a, b_ in zip(w.calls, ref_calls):
        assert a[:3] == tuple(b_[:3])
        assert np.array_equal(a[3].reshape(-1), np.asarray(b_[3]).reshape(-1))
    # and the stock sinks through the mirror (fast paths or replay, whichever the mirror picks)
    out = np.zeros(d.Width * d.Height * 3, np.uint8)
    d2 = jl.JpegDecoder()
    d2.SetInput(f)
    d2.Identify()
    d2.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d2.Width, d2.Height, 3, out))
    d2.Decode()
    assert np.array_equal(out.reshape(d2.Height, d2.Width, 3), po.decode_8bit(f)[0])


PROG_LAYOUTS = [l for l in LAYOUTS if _blocks_per_mcu(l) <= 16] + [((3, 1), (3, 1), (3, 1)), ((1, 3), (1, 3), (1, 3))]


@pytest.mark.parametrize("layout", PROG_LAYOUTS, ids=lambda l: "_".join(f"{h}x{v}" for h, v in l))
def test_progressive_frames_of_every_layout(layout):
    """The same layouts as SOF2 frames: JpegHuffmanProgressiveScanDecoder addresses the blocks of an interleaved scan as
    (colMcu * h + x, rowMcu * v + y) (:112-126) and JpegBlockAllocator.Flush places block (col, row) at (col * hs * 8, row * vs * 8)
    (JpegBlockAllocator.cs:120-149): no overlap here, whatever the factors.  A restart interval that divides a scan's unit count
    makes the reference expect a restart marker where the next SOS stands: the same exception from both."""
    files = [bytes(jpegsynth.encode(w, h, quality=80, restart_interval=dri, seed=w + dri, sampling=layout, progressive=True))
             for (w, h) in SIZES for dri in (0, 7, 11)]
    names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
    n_ok = 0
    for fmt in (jl.FMT_INTERLEAVED_U8, jl.FMT_RGBA_U8, jl.FMT_EXTENDED_U16):
        b = jl.Batch().upload(files, fmt).decode().sync()
        for i, f in enumerate(files):
            try:
                ref, kind = (po.decode_16bit(f, component_count=4)[0] if fmt == jl.FMT_EXTENDED_U16 else po.decode_8bit(f)[0]), "OK"
            except po.OracleError as e:
                ref, kind = None, e.kind
            assert names[b.result(i).status] == kind, (i, b.result(i).detail)
            if ref is None:
                continue
            n_ok += 1
            if fmt == jl.FMT_RGBA_U8:
                ref = po.ycbcr8_to_rgb(ref, rgba=True)
            assert np.array_equal(b.output(i), ref), (fmt, i)
        b.close()
    assert n_ok >= 9


def test_gpu_encoder_luma_shapes_decode_back():
    """The GPU encoder's (4,1) / (1,2) / (1,4) / (4,2) luma shapes (tests/test_gpu_parity.py: byte-exact against the encoder
    restatement) read back by the GPU decoder = by the decoder restatement."""
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (90, 134, 3)).astype(np.uint8)
    img[20:60, 30:100] = np.linspace(0, 255, 70).astype(np.uint8)[None, :, None]
    for luma in [(4, 1), (1, 2), (1, 4), (4, 2), (2, 4), (1, 1), (2, 1), (2, 2)]:
        enc = jl.encode_batch([img], luma, 85)[0]
        assert enc == po.encode_8bit(img, luma[0], luma[1], 85), luma
        for fmt, conv in [(jl.FMT_INTERLEAVED_U8, None), (jl.FMT_RGBA_U8, True)]:
            outs, results = jl.decode_batch([enc], fmt)
            assert results[0].status == 0, luma
            ref = po.decode_8bit(enc)[0]
            assert np.array_equal(outs[0], ref if conv is None else po.ycbcr8_to_rgb(ref, rgba=True)), (luma, fmt)
