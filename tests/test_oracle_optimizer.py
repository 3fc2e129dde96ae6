"""CPU tests of the JpegOptimizer restatement (oracle/jpegopt.inc) and of the product's host-side table builder.

The reference's own test (tests/JpegLibrary.Tests/Optimizer/OptimizerTests.cs:26-47) optimizes baseline/lake.jpg with
strip = true / false and asserts (a) the output is smaller and (b) it decodes to the same pixels.  The same assertions
run here on the restatement -- that is all the pinning the reference offers for this path (no golden bytes)."""
import io
import os

import numpy as np
import pytest
from PIL import Image

import jpeglibrary_amd as jl
from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def read(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def pillow_pixels(data):
    return np.asarray(Image.open(io.BytesIO(data)))


def synth_jpeg(w, h, quality=75, subsampling=2, restart=0, seed=0, gray=False):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 70 * np.sin(xx / 37 + 1) * np.cos(yy / 53), 128 + 60 * np.cos(xx / 91 + yy / 29), 128 + 90 * np.sin((xx + yy) / 67)], -1)
    img = np.clip(np.rint(img + rng.normal(0, 8, img.shape)), 0, 255).astype(np.uint8)
    out = io.BytesIO()
    im = Image.fromarray(img[..., 0] if gray else img)
    kw = dict(format="JPEG", quality=quality)
    if not gray:
        kw["subsampling"] = subsampling
    if restart:
        kw["restart_marker_blocks"] = restart
    im.save(out, **kw)
    return out.getvalue()


@pytest.mark.parametrize("strip", [True, False])
def test_reference_optimizer_test_on_the_restatement(strip):
    data = read("lake.jpg")
    out = po.optimize(data, strip)
    assert len(out) < len(data)
    a, _ = po.decode_8bit(data)
    b, _ = po.decode_8bit(out)
    assert np.array_equal(a, b)
    assert np.array_equal(pillow_pixels(data), pillow_pixels(out))


def test_gray_and_restart_files_keep_their_coefficients():
    for data in (read("cramps.jpg"), synth_jpeg(200, 136, restart=4), synth_jpeg(333, 77, subsampling=0, restart=11, seed=2),
                 synth_jpeg(64, 64, gray=True, seed=3)):
        out = po.optimize(data, strip=False)  # strip = True drops the DRI segment with the other "default" markers (JpegOptimizer.cs:627-637)
        assert np.array_equal(po.decode_coefficients(data)[0], po.decode_coefficients(out)[0])
        assert np.array_equal(pillow_pixels(data), pillow_pixels(out))
        assert len(out) <= len(data) + 16  # a file that already carries optimal tables cannot shrink (cramps.jpg grows by one byte)


def test_scan_gives_up_when_the_restart_check_meets_eoi():
    """MCU count a multiple of DRI: the restart check after the last MCU finds EOI, Scan() returns before building its
    tables (JpegOptimizer.cs:437-442) and Optimize() throws InvalidOperationException."""
    data = synth_jpeg(200, 120, restart=4)  # 13 x 8 = 104 MCUs
    with pytest.raises(po.OracleError) as e:
        po.optimize(data, strip=False)
    assert "InvalidOperation" in str(e.value)


def test_strip_drops_every_segment_the_switch_does_not_name():
    data = synth_jpeg(96, 80, restart=2)  # 6 x 5 = 30 MCUs... 4:2:0 -> 6 x 5; use an odd count
    data = synth_jpeg(112, 80, restart=2)  # 7 x 5 = 35 MCUs
    assert b"\xff\xdd" in data
    kept = po.optimize(data, strip=False)
    stripped = po.optimize(data, strip=True)
    assert b"\xff\xdd\x00\x04" in kept and b"\xff\xdd\x00\x04" not in stripped  # the reference strips DRI as well
    assert stripped.count(b"\xff\xd0") >= 1  # ... while the scan keeps its RSTn markers


def test_statistics_count_every_symbol_once():
    data = read("lake.jpg")
    tables = po.optimizer_statistics(data)
    assert [(c, i) for c, i, _ in tables] == [(0, 0), (1, 0), (0, 1), (1, 1)]
    blocks = po.decode_coefficients(data)[0].shape[0]
    assert int(tables[0][2].sum() + tables[2][2].sum()) == blocks  # one DC symbol per block


@pytest.mark.parametrize("seed", range(6))
def test_table_builder_of_the_product_matches_the_restatement(seed):
    rng = np.random.default_rng(seed)
    freq = np.zeros(256, np.uint32)
    n = [1, 2, 5, 12, 40, 162][seed]
    symbols = rng.choice(256, n, replace=False)
    scale = [3, 10, 1000, 10 ** 6, 10 ** 7, 10 ** 8][seed]
    freq[symbols] = np.maximum(1, (rng.pareto(0.8, n) * scale).astype(np.uint64).clip(1, 2 ** 31)).astype(np.uint32)
    bits, values, code, length = po.build_optimal_table(freq)
    b2, v2, c2, l2 = jl.build_optimal_huffman_table(freq)
    assert np.array_equal(bits, b2) and np.array_equal(values, v2) and np.array_equal(code, c2) and np.array_equal(length, l2)
    # MostOptimalCoding (package merge): product == restatement, never costlier than a 16-bit-limited code has to be
    pb, pv, pc, pl = po.build_optimal_table(freq, most_optimal=True)
    qb, qv, qc, ql = jl.build_optimal_huffman_table(freq, most_optimal=True)
    assert np.array_equal(pb, qb) and np.array_equal(pv, qv) and np.array_equal(pc, qc) and np.array_equal(pl, ql)
    assert pl[symbols].max() <= 16 and sum(int(pb[l - 1]) * 2.0 ** -l for l in range(1, 17)) < 1.0
    if n > 1:
        assert int((freq[symbols].astype(np.int64) * pl[symbols]).sum()) <= int((freq[symbols].astype(np.int64) * length[symbols]).sum())
    # a prefix code of at most 16 bits that never uses the all-ones code word
    assert bits.sum() == n and set(values.tolist()) == set(symbols.tolist())
    kraft = sum(int(bits[l - 1]) * 2.0 ** -l for l in range(1, 17))
    assert kraft < 1.0
    order = np.argsort(length[symbols], kind="stable")
    assert np.all(np.diff(freq[symbols][order].astype(np.int64))[np.diff(length[symbols][order]) > 0] <= 0) or n > 40


def test_length_limiting_kicks_in_for_fibonacci_counts():
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    freq = np.zeros(256, np.uint32)
    freq[:40] = fib
    bits, values, code, length = po.build_optimal_table(freq)
    assert length[:40].max() == 16 and bits.sum() == 40
    b2, v2, c2, l2 = jl.build_optimal_huffman_table(freq)
    assert np.array_equal(bits, b2) and np.array_equal(values, v2) and np.array_equal(length, l2)


def test_most_optimal_coding_on_the_reference_asset():
    data = read("lake.jpg")
    std = po.optimize(data, True)
    best = po.optimize(data, True, most_optimal=True)
    assert len(best) <= len(std) < len(data)
    assert np.array_equal(pillow_pixels(data), pillow_pixels(best))


def test_errors_of_the_restatement():
    with pytest.raises(po.OracleError):
        po.optimize(b"", True)
    with pytest.raises(po.OracleError) as e:
        po.optimize(b"\xff\xd8\xff\xd9", True)
    assert "No image data is read." in str(e.value)
    with pytest.raises(po.OracleError):
        po.optimize(read("progress.jpg"), True)
    truncated = read("lake.jpg")[:100000]
    with pytest.raises(po.OracleError):
        po.optimize(truncated, True)
