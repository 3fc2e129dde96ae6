"""Host logic of the product (marker / table parsing, Identify, metadata) through the C ABI -- runs without a GPU.

Mirrors tests/JpegLibrary.Tests/Decoder/MetadataIdentifyTests.cs:19-154 of the reference and cross-checks the
product's host parser against the oracle's restatement on synthetic files.
"""
import numpy as np
import pytest

import jpeglibrary_amd as jl
from golden_util import read_jpeg
from oracle import pyoracle as po
from tools import jpegsynth

METADATA = [
    ("cramps.jpg", 800, 607, 8, 1, 90, 137766),
    ("HETissueSlide.jpg", 2048, 2048, 8, 3, 75, 783426),
    ("testorig12.jpg", 227, 149, 12, 3, 75, 12394),
    ("progress.jpg", 341, 486, 8, 3, 85, 44884),
    ("yellowcat_progressive_restart.jpg", 720, 540, 8, 3, 75, 45703),
]


@pytest.mark.parametrize("name,w,h,p,c,q,length", METADATA)
def test_decoder_identify(name, w, h, p, c, q, length):
    decoder = jl.JpegDecoder(host_only=True)
    decoder.SetInput(read_jpeg(name))
    size = decoder.Identify(loadQuantizationTables=True)
    assert decoder.Width == w and decoder.Height == h
    assert decoder.NumberOfComponents == c and decoder.Precision == p
    ok, quality = decoder.TryEstimateQuanlity()
    assert ok and round(quality) == q
    assert size == length


def test_identify_latches_last_dri_and_matches_oracle():
    for ss, dri in (("420", 4), ("444", 0), ("422", 7), ("gray", 1)):
        data = jpegsynth.encode(97, 61, ss, 80, dri, seed=3)
        d = jl.JpegDecoder(host_only=True)
        d.SetInput(data)
        n = d.Identify()
        info, _ = po.identify(data)
        assert (d.Width, d.Height, d.Precision, d.NumberOfComponents) == (info.width, info.height, info.precision, info.ncomp)
        assert d.GetRestartInterval() == info.restart_interval == dri
        assert n == info.consumed == len(data)
        assert d.StartOfFrame == info.sof == 0xC0
        assert d.GetMaximumHorizontalSampling() == max(info.comp[i].h for i in range(info.ncomp))
        for i in range(info.ncomp):
            assert d.GetHorizontalSampling(i) == info.comp[i].h and d.GetVerticalSampling(i) == info.comp[i].v


def test_api_misuse_errors_match_reference():
    d = jl.JpegDecoder(host_only=True)
    with pytest.raises(jl.InvalidOperationException, match="Input buffer is not specified"):
        d.Identify()
    with pytest.raises(jl.InvalidOperationException, match="Call Identify"):
        _ = d.Width
    d.SetInput(b"\xff\xd8\xff\xd9")
    with pytest.raises(jl.InvalidOperationException, match="Frame header was not found"):
        d.Identify()
    with pytest.raises(jl.InvalidOperationException, match="output buffer is not specified"):
        d.Decode()
    with pytest.raises(jl.ArgumentException):
        d.SetRestartInterval(70000)


def test_malformed_headers_match_oracle_messages():
    good = jpegsynth.encode(33, 17, "420", 75, 2, seed=5)
    cases = {
        "truncated_in_sof": good[:170],
        "no_soi_garbage": b"\x00\x01\x02\x03" * 8,
        "double_sof": good[:-2] + good[2:],
    }
    for name, data in cases.items():
        d = jl.JpegDecoder(host_only=True)
        d.SetInput(data)
        try:
            d.Identify()
            mine = None
        except jl.JpegError as e:
            mine = (type(e).__name__, str(e))
        try:
            po.identify(data)
            ref = None
        except po.OracleError as e:
            ref = (e.kind, e.message)
        assert mine == ref, name



def test_jpeg_encoder_mirror_argument_checks_need_no_device():
    """The setters of the JpegEncoder mirror raise what JpegEncoder.cs:102-239 raises, before anything touches a device."""
    import jpeglibrary_amd as jl

    e = jl.JpegEncoder()
    with pytest.raises(jl.ArgumentException, match="Quantization table is not initialized."):
        e.SetQuantizationTable(jl.JpegQuantizationTable())
    with pytest.raises(jl.InvalidOperationException, match="Only baseline JPEG is supported."):
        e.SetQuantizationTable(jl.JpegStandardQuantizationTable.GetLuminanceTable(1, 0))
    with pytest.raises(jl.ArgumentException, match="The length of elements must be 64."):
        jl.JpegQuantizationTable(0, 0, [1] * 63)
    lum = jl.JpegStandardQuantizationTable.ScaleByQuality(jl.JpegStandardQuantizationTable.GetLuminanceTable(0, 0), 75)
    assert lum.Elements[:4] == (8, 6, 6, 7) and jl.JpegStandardQuantizationTable.ScaleByQuality(lum, 100).Elements == (1,) * 64
    e.SetQuantizationTable(lum)
    with pytest.raises(jl.ArgumentException, match="Subsampling factor can only be 1, 2 or 4."):
        e.AddComponent(1, 0, 0, 0, 3, 1)
    with pytest.raises(jl.ArgumentException, match="Quantization table is not defined."):
        e.AddComponent(1, 1, 0, 0, 2, 2)
    with pytest.raises(jl.ArgumentException, match="Huffman table is not defined."):
        e.AddComponent(1, 0, 0, 0, 2, 2)
    e.SetHuffmanTable(True, 0, jl.JpegStandardHuffmanEncodingTable.GetLuminanceDCTable())
    e.SetHuffmanTable(False, 0, jl.JpegStandardHuffmanEncodingTable.GetLuminanceACTable())
    e.AddComponent(1, 0, 0, 0, 2, 2)
    with pytest.raises(jl.ArgumentException, match="The component index is already used by another component."):
        e.AddComponent(1, 0, 0, 0, 1, 1)
    with pytest.raises(jl.InvalidOperationException, match="Output is not specified."):
        e.Encode()
    e.SetOutput(bytearray())
    with pytest.raises(jl.InvalidOperationException, match="Input is not specified."):
        e.Encode()
    e2 = jl.JpegEncoder()
    e2.SetOutput(bytearray())
    e2.SetInputReader(jl.JpegBufferInputReader(8, 8, 1, bytes(64)))
    with pytest.raises(jl.InvalidOperationException, match="No component is specified."):
        e2.Encode()

