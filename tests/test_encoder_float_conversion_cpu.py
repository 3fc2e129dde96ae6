"""fdct_fused_kernel converts RGB to YCbCr in float: `floor(fma(k0', R, fma(k1', G, fma(k2', B, o'))))` with the reference's
Fix() factors scaled by 2^-16 (jpeglibrary_amd/csrc/encode_kernels.hip, EfConvert).  The claim behind it -- every product and
partial sum, taken in the kernel's order, is exactly representable in binary32, so the floor IS
`(i * Fix(x) + ... + offset) >> 16` of apps/JpegEncode/JpegRgbToYCbCrConverter.cs:64-96 -- is checked here for all 2^24 pixels
in numpy float32 (no fused multiply-add needed: an exact product plus an exact sum rounds nowhere).  Also the rounding trick
`(short)MathF.Round(x)` = low 16 bits of `x + 1.5 * 2^23` and the chroma sample `((t + 2) >> 2) - 128 = floor(t / 4 - 127.5)`."""
import numpy as np

from oracle import pyoracle as po

FIX = lambda x: int(x * 65536.0 + 0.5)
K = [FIX(0.299), FIX(0.587), FIX(0.114), FIX(0.168735892), FIX(0.331264108), FIX(0.5), FIX(0.418687589), FIX(0.081312411)]


def _chain(terms, offset):
    """the kernel's order: offset + first term, then the second, then the third -- each step checked for exactness"""
    acc = np.float32(offset)
    acc64 = np.float64(offset)
    for k, b in terms:
        prod = np.float32(k) * b.astype(np.float32)
        assert np.array_equal(prod.astype(np.float64), np.float64(k) * b.astype(np.float64))
        acc = (acc + prod).astype(np.float32)
        acc64 = acc64 + np.float64(k) * b.astype(np.float64)
        assert np.array_equal(acc.astype(np.float64), acc64)  # the partial sum did not round
    return np.floor(acc).astype(np.int64)


def test_float_colour_conversion_equals_the_integer_tables_for_every_pixel():
    s = np.float32(1.0 / 65536.0)
    ky = [np.float32(K[0]) * s, np.float32(K[1]) * s, np.float32(K[2]) * s]
    kb = [np.float32(-K[3]) * s, np.float32(-K[4]) * s, np.float32(K[5]) * s]
    kr = [np.float32(K[5]) * s, np.float32(-K[6]) * s, np.float32(-K[7]) * s]
    oy = np.float32(32768 - (128 << 16)) * s
    oc = np.float32((128 << 16) + 32767) * s
    g, b = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    for r in range(256):
        rr = np.full_like(g, r)
        y = _chain([(ky[2], b), (ky[1], g), (ky[0], rr)], oy)
        cb = _chain([(kb[0], rr), (kb[1], g), (kb[2], b)], oc)
        cr = _chain([(kr[1], g), (kr[2], b), (kr[0], rr)], oc)
        assert np.array_equal(y, ((K[0] * rr + K[1] * g + K[2] * b + 32768) >> 16) - 128)
        assert np.array_equal(cb, (K[5] * b - K[3] * rr - K[4] * g + (128 << 16) + 32767) >> 16)
        assert np.array_equal(cr, (K[5] * rr - K[6] * g - K[7] * b + (128 << 16) + 32767) >> 16)
        assert cb.min() >= 0 and cb.max() <= 255 and cr.min() >= 0 and cr.max() <= 255
    # and the integer tables are the checker's (oracle/jpegenc.c through pyoracle)
    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    ycc = po.rgb_to_ycbcr8(px).astype(np.int64)
    r_, g_, b_ = (px[..., i].astype(np.int64) for i in range(3))
    assert np.array_equal(ycc[..., 0], (K[0] * r_ + K[1] * g_ + K[2] * b_ + 32768) >> 16)
    assert np.array_equal(ycc[..., 1], (K[5] * b_ - K[3] * r_ - K[4] * g_ + (128 << 16) + 32767) >> 16)
    assert np.array_equal(ycc[..., 2], (K[5] * r_ - K[6] * g_ - K[7] * b_ + (128 << 16) + 32767) >> 16)


def test_rounding_and_short_cast_as_one_addition():
    """(short)MathF.Round(x) (ZigZagAndQuantizeBlock, JpegEncoder.cs:812-826) = low 16 bits of float32(x + 1.5 * 2^23) for
    |x| < 2^22: ties to even, negative values in two's complement, wrap-around beyond 16 bits."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-70000, 70000, 2_000_000), np.arange(-40000, 40000) + 0.5, np.arange(-40000, 40000) - 0.5,
                        np.array([0.0, -0.0, 0.49999997, -0.49999997, 32767.5, -32768.5, 4194303.0, -4194303.0])]).astype(np.float32)
    magic = (x + np.float32(12582912.0)).astype(np.float32)
    low16 = (magic.view(np.uint32) & 0xFFFF).astype(np.uint16).view(np.int16)
    want = np.rint(x).astype(np.int64).astype(np.int16)  # numpy's rint is half-to-even; int64 -> int16 wraps
    assert np.array_equal(low16, want)


def test_chroma_sample_from_sum_and_carry_in_float():
    """ReadBlockWithSubsample's sample for 2 x 2: ((short)(previous + sum) + 2) >> 2, then ShiftDataLevel's - 128
    (JpegEncoder.cs:788-805): the kernel takes floor(t * 0.25 - 127.5) on the 16-bit wrapped t."""
    t = np.arange(-32768, 32768, dtype=np.int64)
    f = np.floor(t.astype(np.float32) * np.float32(0.25) + np.float32(-127.5))
    assert np.array_equal(f.astype(np.int64), ((t + 2) >> 2) - 128)
