"""Huffman tables of unusual SHAPE through the K2 family's two-level lookups and the K2S round kernel's parked lanes.

The lookups (kernels_device.h, "Lookups of the K2 family") decide codes of up to 11 bits in one step, long codes through a second
level that covers the last 256 of the 65 536 sixteen-bit values -- where a canonical table's long codes sit when most of the
code space is spent on short codes (the standard tables: 192 values) -- and everything else through the reference's maxcode
walk.  The tables of libjpeg-turbo's files, standard or optimised, never leave the second level.  These do:
  deep    one code each of 2 .. 11 bits for the ten most frequent symbols, 16 bits for ALL the others: the long codes start at
          0x7FF0, far below the second level's window -- every rare symbol takes the exact walk;
  flat    every symbol 8 bits (DC: 4 bits): nothing is long, the first level decides everything;
  wide    short codes for the LARGE DC categories: DC symbols whose magnitude leaves the round kernel's 10-bit prefix.
A tiny baseline entropy coder in the test writes the files (the reference encoder takes no caller's Huffman tables with its
standard action, and the oracle's restatement follows it); coefficients and samples are compared with the oracle's decode."""
import numpy as np
import pytest

import jpeglibrary_amd as jl
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _canonical(lengths):
    """{symbol: length} -> ({symbol: (code, length)}, BITS[16], HUFFVAL) the way Annex C assigns codes"""
    order = sorted(lengths.items(), key=lambda kv: (kv[1], kv[0]))
    bits, codes, code, prev = [0] * 16, {}, 0, order[0][1]
    for sym, ln in order:
        code <<= ln - prev
        prev = ln
        codes[sym] = (code, ln)
        assert code < (1 << ln) - (1 if ln == 16 else 0), "code space exhausted"
        code += 1
        bits[ln - 1] += 1
    return codes, bits, [s for s, _ in order]


def _shape(freq, shape, is_dc):
    syms = sorted(freq, key=lambda s: (-freq[s], s))
    if shape == "deep":
        return {s: (i + 2 if i < 10 else 16) for i, s in enumerate(syms)}
    if shape == "flat":
        return {s: (4 if is_dc else 8) for s in syms}
    if shape == "wide":  # (DC tables only) the larger the category the shorter the code
        by_cat = sorted(syms, key=lambda s: -s)
        return {s: min(16, i + 1 + (1 if i == len(by_cat) - 1 else 0)) if i < 15 else 16 for i, s in enumerate(by_cat)}
    raise ValueError(shape)


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | (code & ((1 << ln) - 1))
        self.n += ln
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _symbols(block, pred):
    """(DC category, DC bits), [(AC symbol, magnitude bits, size)] of one zig-zag block"""
    def mag(v):
        s = int(abs(v)).bit_length()
        return s, (v if v >= 0 else v + (1 << s) - 1) & ((1 << s) - 1)

    d = int(block[0]) - pred
    out, run = [], 0
    for k in range(1, 64):
        v = int(block[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            out.append((0xF0, 0, 0))
            run -= 16
        s, m = mag(v)
        out.append(((run << 4) | s, m, s))
        run = 0
    if run:
        out.append((0x00, 0, 0))
    return mag(d), out


def _recode(pixels, luma, quality, dri, dc_shape, ac_shape):
    """the oracle encoder's coefficients (and its DQT / SOF0 / SOS bytes) written again with Huffman tables of the given shapes,
    restart markers every `dri` MCUs (DC predictors back to zero there)"""
    lh, lv = luma
    ref_file, coefs = po.encode_8bit(pixels, lh, lv, quality, want_coefficients=True)
    ncomp = 3 if pixels.ndim == 3 and pixels.shape[2] == 3 else 1
    per_mcu = [0] * (lh * lv) + ([1, 2] if ncomp == 3 else [])
    per_interval = len(per_mcu) * dri if dri else len(coefs) + 1
    freq, syms, pred = [{}, {}, {}, {}], [], [0, 0, 0]  # tables: DC luma, AC luma, DC chroma, AC chroma
    for i, blk in enumerate(coefs):
        c = per_mcu[i % len(per_mcu)]
        if i % per_interval == 0:
            pred = [0, 0, 0]
        (dcat, dbits), ac = _symbols(blk, pred[c])
        pred[c] = int(blk[0])
        t = 0 if c == 0 else 2
        freq[t][dcat] = freq[t].get(dcat, 0) + 1
        for sym, _, _ in ac:
            freq[t + 1][sym] = freq[t + 1].get(sym, 0) + 1
        syms.append((t, dcat, dbits, ac))
    tables = [_canonical(_shape(freq[t], dc_shape if t % 2 == 0 else ac_shape, t % 2 == 0)) if freq[t] else None for t in range(4)]
    d, p, out = bytes(ref_file), 2, bytearray(b"\xff\xd8")
    while d[p + 1] != 0xDA:
        n = (d[p + 2] << 8) | d[p + 3]
        if d[p + 1] not in (0xC4, 0xDD):
            out += d[p:p + 2 + n]
        p += 2 + n
    sos = d[p:p + 2 + ((d[p + 2] << 8) | d[p + 3])]
    for t, tab in enumerate(tables):
        if tab is not None:
            payload = bytes([((t % 2) << 4) | (t // 2)]) + bytes(tab[1]) + bytes(tab[2])
            out += b"\xff\xc4" + (len(payload) + 2).to_bytes(2, "big") + payload
    if dri:
        out += b"\xff\xdd\x00\x04" + dri.to_bytes(2, "big")
    out += sos
    w = _Bits()
    for i, (t, dcat, dbits, ac) in enumerate(syms):
        if i and i % per_interval == 0:
            w.flush()
            w.out += bytes([0xFF, 0xD0 + ((i // per_interval - 1) & 7)])
        w.put(*tables[t][0][dcat])
        if dcat:
            w.put(dbits, dcat)
        for sym, m, s in ac:
            w.put(*tables[t + 1][0][sym])
            if s:
                w.put(m, s)
    w.flush()
    return bytes(out) + bytes(w.out) + b"\xff\xd9", coefs


def _image(w, h, seed, gray=False):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 128 + 100 * np.sin(x / 9.0 + seed) * np.cos(y / 13.0)
    px = np.stack([base + rng.normal(0, 40, base.shape) for _ in range(1 if gray else 3)], -1)
    # DC differences of every size: blocks of very different brightness side by side
    px[(y // 8 + x // 8) % 5 == 0] = rng.integers(0, 256)
    px = np.clip(px, 0, 255).astype(np.uint8)
    return px[..., 0] if gray else px


@pytest.mark.parametrize("dri", [0, 4])  # (357 MCUs: an interval that divides the MCU count makes the reference optimizer give up)
@pytest.mark.parametrize("dc_shape,ac_shape", [("deep", "deep"), ("flat", "flat"), ("wide", "deep"), ("deep", "flat")])
def test_tables_of_unusual_shape(dc_shape, ac_shape, dri):
    px = _image(328, 264, 4)
    data, coefs_in = _recode(px, (2, 2), 92, dri, dc_shape, ac_shape)
    ref_coefs, _ = po.decode_coefficients(data)  # (the oracle reads the file: the recoder above wrote what it meant to)
    assert np.array_equal(ref_coefs.reshape(-1, 64), coefs_in.reshape(-1, 64))
    ref, _ = po.decode_8bit(data)
    outs, res = jl.decode_batch([data], jl.FMT_INTERLEAVED_U8)
    assert res[0].status == 0, (res[0].status, res[0].detail)
    assert np.array_equal(outs[0], ref)
    coefs = jl.Batch().upload([data], jl.FMT_PLANAR_I16).run_entropy().sync().coefficients(0)
    assert np.array_equal(coefs, ref_coefs)
    # and through the optimizer (its symbol transcode reads the same streams)
    assert jl.optimize_batch([data], strip=False)[0] == po.optimize(data, False)


def test_gray_444_and_a_batch_of_shapes():
    files = [_recode(_image(200, 120, 7, gray=True), (1, 1), 85, 0, "deep", "deep")[0],
             _recode(_image(136, 200, 8), (1, 1), 97, 2, "wide", "deep")[0],
             _recode(_image(256, 160, 9), (2, 1), 60, 0, "flat", "deep")[0]]
    outs, res = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
    for f, o, r in zip(files, outs, res):
        assert r.status == 0
        assert np.array_equal(o, po.decode_8bit(f)[0])


def test_dc_categories_above_16_are_a_fence_with_a_stated_behaviour():
    """A DHT that assigns a DC code a category of 17..255 (a corrupted table; T.81 allows 0..11 / 0..15).  The reference has no
    check: ReceiveAndExtend asks TryReadBits for that many bits (ScanDecoder/JpegHuffmanScanDecoder.cs:100-115,
    JpegBitReader.cs:190-204) -- 17..32 succeed with C#'s masked shifts ((1 << n) with n mod 32), 33..39 depend on how many
    whole bytes the 64-bit buffer happens to hold at that symbol, more can never be loaded ("The bit stream ended prematurely").
    The HIP path does NOT follow it there: every table kind fails the scan at the FIRST symbol of such a category
    with the reference's "Invalid Huffman code" verdict (InvalidDataException, detail 1).  Same exception class as the reference
    whenever the reference fails too; where the reference decodes on (category 17 here), the classes differ -- pinned so that a
    change of either side shows."""
    from tools import jpegsynth
    base = bytearray(jpegsynth.encode(64, 64, "444", 75, 0, seed=3))
    vals = base.index(b"\xff\xc4") + 4 + 1 + 16  # the luma DC table's values: categories 0..11
    verdicts = {}
    for cat in (17, 20, 32, 33, 64, 255):
        f = bytearray(base)
        f[vals + 2] = cat  # the code of category 2 now means `cat`
        try:
            po.decode_8bit(bytes(f))
            ref = "OK"
        except po.OracleError as e:
            ref = e.kind
        _, results = jl.decode_batch([bytes(f)], jl.FMT_INTERLEAVED_U8)
        assert (results[0].status, results[0].detail) == (1, 1), cat
        verdicts[cat] = ref
    assert verdicts == {17: "OK", 20: "InvalidDataException", 32: "InvalidDataException", 33: "InvalidDataException", 64: "InvalidDataException",
                        255: "InvalidDataException"}, verdicts
