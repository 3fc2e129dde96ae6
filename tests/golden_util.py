"""Reading the reference's golden PNG pairs.

Restates tests/JpegLibrary.Tests/Utils/ImageHelper.cs:12-91 of the reference:
u16[y][x][n] = (HIGH[y][x][n] << 8) | (HIGH[y][x][n] XOR LOWDIFF[y][x][n]) for n < numberOfComponents, 4 ushorts
per pixel (unused channels stay 0).  The PNG pairs under tests/golden/ are data files copied from the reference's
tests/Assets (they are bit-exact dumps of the reference decoder's output made by apps/JpegDebugDump).
"""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name)


def read_jpeg(name) -> bytes:
    with open(golden_path(name), "rb") as f:
        return f.read()


def load_reference_buffer(name, width, height, ncomp) -> np.ndarray:
    """Returns uint16 [H, W, 4] exactly as ImageHelper.LoadBuffer builds it."""
    from PIL import Image

    high = np.asarray(Image.open(golden_path(name + ".high.png")).convert("RGBA"), dtype=np.uint16)
    low = np.asarray(Image.open(golden_path(name + ".low-diff.png")).convert("RGBA"), dtype=np.uint16)
    assert high.shape == (height, width, 4) and low.shape == (height, width, 4), (high.shape, low.shape)
    buf = np.zeros((height, width, 4), dtype=np.uint16)
    buf[..., :ncomp] = (high[..., :ncomp] << 8) | (high[..., :ncomp] ^ low[..., :ncomp])
    return buf
