"""CPU tests of the encoder restatement (oracle/jpegenc.c).  The reference holds no encoder vectors (PARITY UNPINNED):
what can be checked is the wire format it documents, the arithmetic against closed forms, and the round trip through
the golden-pinned decoder oracle."""
import io

import numpy as np
import pytest

from oracle import pyoracle as po


def _image(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 70 * np.sin(xx / 37 + 1) * np.cos(yy / 53), 128 + 60 * np.cos(xx / 91 + yy / 29),
                    128 + 90 * np.sin((xx + yy) / 67)], -1) + rng.normal(0, 8, (h, w, 3))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_rgb_to_ycbcr_matches_the_fixed_point_closed_form():
    g = np.mgrid[0:256, 0:256]
    rgb = np.stack([g[0], g[1], (g[0] * 5 + g[1] * 3) % 256], -1).astype(np.uint8)
    out = po.rgb_to_ycbcr8(rgb).astype(np.int64)
    r, gg, b = (rgb[..., i].astype(np.int64) for i in range(3))
    fix = lambda x: int(np.float32(x) * np.float32(65536) + np.float32(0.5))
    y = (fix(0.299) * r + fix(0.587) * gg + fix(0.114) * b + 32768) >> 16
    cb = (-fix(0.168735892) * r - fix(0.331264108) * gg + fix(0.5) * b + (128 << 16) + 32767) >> 16
    cr = (fix(0.5) * r - fix(0.418687589) * gg - fix(0.081312411) * b + (128 << 16) + 32767) >> 16
    assert np.array_equal(out[..., 0], y) and np.array_equal(out[..., 1], cb) and np.array_equal(out[..., 2], cr)


def test_rgba_pixels_convert_like_rgb_pixels_with_the_alpha_stepped_over():
    """ConvertRgba32ToYCbCr8 (the benchmark project's copy of the converter, JpegRgbToYCbCrConverter.cs:95-124): the Rgb24 tables,
    the source advancing four bytes per pixel."""
    rng = np.random.default_rng(11)
    rgba = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    assert np.array_equal(po.rgba_to_ycbcr8(rgba), po.rgb_to_ycbcr8(np.ascontiguousarray(rgba[..., :3])))


def test_fdct_of_constant_and_basis_blocks():
    q = np.ones(64, np.uint16)
    flat = po.fdct_quantize_block(np.full(64, 200, np.int16), q)[0]
    assert flat[0] == (200 - 128) * 8 and not flat[1:].any()          # DC = 8 x mean, no AC
    # forward then inverse transform of the reference returns the block (all-ones tables, values well inside the range)
    rng = np.random.default_rng(0)
    blocks = rng.integers(0, 256, (200, 64)).astype(np.int16)
    coefs = po.fdct_quantize_block(blocks, q)
    back = po.block_dequant_idct_shift(coefs, q, 128)
    assert np.abs(back.astype(int) - blocks.astype(int)).max() <= 1


@pytest.mark.parametrize("w,h,mh,mv,q", [(333, 211, 2, 2, 75), (64, 48, 2, 1, 90), (17, 9, 1, 1, 50), (1, 1, 2, 2, 75), (640, 368, 2, 2, 30)])
def test_encoded_stream_layout_and_round_trip(w, h, mh, mv, q):
    ycc = po.rgb_to_ycbcr8(_image(w, h, w + h))
    data, coefs = po.encode_8bit(ycc, mh, mv, q, want_coefficients=True)
    # wire format of JpegEncoder.Encode: SOI, one DQT (2 tables), SOF0, one DHT (4 tables), SOS, data, EOI -- no APPn
    assert data[:4] == b"\xff\xd8\xff\xdb" and data[4:6] == (2 * 65 + 2).to_bytes(2, "big")
    sof = 4 + 2 + 130
    assert data[sof:sof + 2] == b"\xff\xc0" and data[sof + 4] == 8
    assert int.from_bytes(data[sof + 5:sof + 7], "big") == h and int.from_bytes(data[sof + 7:sof + 9], "big") == w
    assert data[sof + 10:sof + 13] == bytes([1, (mh << 4) | mv, 0]) and data[sof + 13:sof + 16] == bytes([2, 0x11, 1])
    dht = sof + 2 + 2 + 6 + 9
    assert data[dht:dht + 2] == b"\xff\xc4" and int.from_bytes(data[dht + 2:dht + 4], "big") == 2 + 4 * 17 + 12 + 162 + 12 + 162
    assert data[-2:] == b"\xff\xd9"
    # the pinned decoder reads back exactly the coefficients the encoder quantised ...
    dec = po.decode_coefficients(data)
    dec = np.asarray(dec[0] if isinstance(dec, tuple) else dec).reshape(-1, 64)
    assert np.array_equal(dec, coefs)
    # ... and an image close to the input (luma is not sub-sampled: tight bound at high quality)
    out, info = po.decode_8bit(data)
    assert (info.width, info.height, info.ncomp) == (w, h, 3)
    if q >= 75 and w >= 16:
        err = out[..., 0].astype(int) - ycc[..., 0].astype(int)
        assert 10 * np.log10(255 ** 2 / max(np.mean(err ** 2), 1e-9)) > 30
    # libjpeg-turbo accepts the file too
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.load()
    assert im.size == (w, h)


def test_subsampled_blocks_accumulate_on_the_previous_blocks_coefficients():
    """The behaviour restated on purpose (oracle/jpegenc.c header): WriteScanData reuses ONE block buffer and the
    sub-sampling reader adds into it, so Cb starts from the 4th luma block's quantised coefficients."""
    ycc = po.rgb_to_ycbcr8(_image(16, 16, 3))
    _, coefs = po.encode_8bit(ycc, 2, 2, 75, want_coefficients=True)
    lum_q = np.zeros(64, np.uint16)
    chr_q = np.zeros(64, np.uint16)
    L = po.lib()
    import ctypes as C
    std_lum = np.array([16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40, 26, 24, 22, 22, 24, 49, 35, 37, 29, 40, 58, 51, 61, 60,
                        57, 51, 56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56, 80, 109, 81, 87, 95, 98, 103, 104, 103, 62, 77, 113, 121, 112,
                        100, 120, 92, 101, 103, 99], np.uint16)
    std_chr = np.array([17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66] + [99] * 50, np.uint16)
    L.jref_scale_quant_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.jref_scale_quant_table(std_lum.ctypes.data, 75, lum_q.ctypes.data)
    L.jref_scale_quant_table(std_chr.ctypes.data, 75, chr_q.ctypes.data)
    cb = ycc[..., 1].astype(np.int32)
    box = cb.reshape(8, 2, 8, 2).sum(axis=(1, 3))                       # 2x2 sums of the 16x16 plane
    stale = coefs[3].astype(np.int32).reshape(8, 8)                      # what the buffer held: Y3's quantised block
    samples = ((box + stale + 2) >> 2).astype(np.int16)
    assert np.array_equal(po.fdct_quantize_block(samples.reshape(64), chr_q)[0], coefs[4])
    clean = ((box + 2) >> 2).astype(np.int16)
    assert not np.array_equal(po.fdct_quantize_block(clean.reshape(64), chr_q)[0], coefs[4]) or not stale.any()


def test_optimize_coding_round_trips_and_shrinks():
    """EncodeAction's optimizeCoding branch of the restatement: the stream decodes back to the tapped coefficients (with
    the restated decoder and with Pillow's), and is not larger than the standard-table stream."""
    import io
    from PIL import Image
    rng = np.random.default_rng(3)
    for (w, h), luma in (((64, 48), (2, 2)), ((37, 29), (2, 2)), ((90, 41), (2, 1)), ((61, 35), (1, 1))):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        std = po.encode_8bit(img, luma[0], luma[1], 75)
        opt, coefs = po.encode_8bit(img, luma[0], luma[1], 75, want_coefficients=True, optimize_coding=True)
        assert len(opt) < len(std)
        assert np.array_equal(po.decode_coefficients(opt)[0], coefs)
        assert Image.open(io.BytesIO(opt)).size == (w, h)
    with pytest.raises(po.OracleError):
        po.encode_8bit(img[..., 0], 1, 1, 75, optimize_coding=True)


@pytest.mark.parametrize("ri", [1, 3, 8, 1000])
@pytest.mark.parametrize("opt", [0, 1])
def test_restart_interval_extension_round_trips(ri, opt):
    """The checker's definition of the encoder's restart-interval mode (an extension: the reference's JpegEncoder has none): one DRI
    segment in front of SOF0, RSTm with m counting modulo 8 in front of every MCU that is a non-zero multiple of the interval,
    and a stream that the golden-pinned decoder restatement and libjpeg-turbo both read back -- the restatement to exactly the
    coefficients that went in, and, decoded, to the pixels of the same image encoded without restart markers."""
    from PIL import Image
    for (w, h), luma in (((90, 70), (2, 2)), ((37, 29), (2, 1)), ((64, 40), (1, 1))):
        img = _image(w, h, w + h)
        data, coefs = po.encode_8bit(img, luma[0], luma[1], 80, want_coefficients=True, optimize_coding=opt, restart_interval=ri)
        plain = po.encode_8bit(img, luma[0], luma[1], 80, optimize_coding=opt)
        sof = data.index(b"\xff\xc0")
        assert data.count(b"\xff\xdd\x00\x04" + bytes([ri >> 8, ri & 255])) == 1 and data.index(b"\xff\xdd") < sof
        mcus = -(-w // (8 * luma[0])) * -(-h // (8 * luma[1]))
        i = data.index(b"\xff\xda")
        i += 2 + ((data[i + 2] << 8) | data[i + 3])
        seen = []
        while i + 1 < len(data):
            if data[i] == 0xFF and 0xD0 <= data[i + 1] <= 0xD7:
                seen.append(data[i + 1] - 0xD0)
                i += 2
            else:
                i += 1
        assert seen == [k & 7 for k in range((mcus - 1) // ri)]
        if not opt:
            assert np.array_equal(po.decode_coefficients(data)[0], coefs)
        assert np.array_equal(po.decode_8bit(data)[0], po.decode_8bit(plain)[0])
        assert Image.open(io.BytesIO(data)).size == (w, h)

