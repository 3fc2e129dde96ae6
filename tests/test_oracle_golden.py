"""Pins the oracle (oracle/jpegref.c) against the reference's own golden vectors.

Mirrors the reference's xunit tests:
  tests/JpegLibrary.Tests/Decoder/HuffmanSequentialDecodeTests.cs:13-43
  tests/JpegLibrary.Tests/Decoder/HuffmanProgressiveDecodeTests.cs:13-40
  tests/JpegLibrary.Tests/Decoder/MetadataIdentifyTests.cs:19-154
"""
import numpy as np
import pytest

from golden_util import load_reference_buffer, read_jpeg
from oracle import pyoracle as po

DECODE_ASSETS = ["cramps.jpg", "lake.jpg", "testorig12.jpg", "progress.jpg", "yellowcat_progressive_restart.jpg"]


@pytest.mark.parametrize("name", DECODE_ASSETS)
def test_decode_matches_reference_golden(name):
    data = read_jpeg(name)
    # decoder.SetInput; Identify; JpegExtendingOutputWriter(w, h, 4, precision, buffer); Decode
    out, info = po.decode_16bit(data, 4)
    reference = load_reference_buffer(name, info.width, info.height, info.ncomp)
    assert np.array_equal(reference, out)  # Assert.True(reference.AsSpan().SequenceEqual(buffer))


# (file, width, height, precision, components, estimated quality, stream length) -- MetadataIdentifyTests.cs
METADATA = [
    ("cramps.jpg", 800, 607, 8, 1, 90, 137766),
    ("HETissueSlide.jpg", 2048, 2048, 8, 3, 75, 783426),
    ("testorig12.jpg", 227, 149, 12, 3, 75, 12394),
    ("progress.jpg", 341, 486, 8, 3, 85, 44884),
    ("yellowcat_progressive_restart.jpg", 720, 540, 8, 3, 75, 45703),
]


@pytest.mark.parametrize("name,w,h,p,c,q,length", METADATA)
def test_identify_metadata(name, w, h, p, c, q, length):
    info, quality = po.identify(read_jpeg(name), load_quantization_tables=True)
    assert (info.width, info.height, info.precision, info.ncomp) == (w, h, p, c)
    assert info.consumed == length
    assert quality is not None and round(quality) == q


def test_unclamped_samples_exist():
    """decoder output is unclamped int16 (the golden probe saw -17..271 on these assets)."""
    calls, _ = po.decode_blocks(read_jpeg("lake.jpg"))
    lo = min(int(b.min()) for _, _, _, b in calls)
    hi = max(int(b.max()) for _, _, _, b in calls)
    assert lo < 0 and hi > 255
