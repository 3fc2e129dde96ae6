#!/usr/bin/env python3
"""tools/stress_parity.py [n] [seed] -- random corpus through the GPU paths vs the CPU restatements (checker only).

Pillow / libjpeg-turbo files of random size, quality, sampling, restart interval, progressive / optimize flags:
decode (YCbCr8 + RGBA) and, for the single-scan baseline ones, the optimizer (both strip settings).  Prints a summary;
exit code 1 on any mismatch.  Not part of the test suite (it takes minutes with large n).  STRESS_SCALE=k multiplies the image
dimensions (1..300 -> k..300k pixels a side).  STRESS_HEADER=1: every corrupted file has its flipped bit(s) in a header.
STRESS_SAMPLING=1 (round 6): every second file comes from tools/jpegsynth with per-component sampling factors drawn from 1..4
(whole power-of-two ratios, at most 16 blocks per MCU), interleaved, as three single-component scans or as a progressive frame."""
import io
import os
import sys

import numpy as np
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 27  # Pillow sizes its encoder buffer from this for optimize / progressive (large with STRESS_SCALE)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeglibrary_amd as jl
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
scale = int(os.environ.get("STRESS_SCALE", "1"))
HEADER_ONLY = os.environ.get("STRESS_HEADER") is not None
CMYK = os.environ.get("STRESS_CMYK") is not None
files, kinds = [], []
for i in range(n):
    w, h = int(rng.integers(1, 301)), int(rng.integers(1, 301))
    if rng.random() < 0.1:
        w, h = int(rng.integers(300, 1200)), int(rng.integers(300, 900))
    if scale > 1:  # STRESS_SCALE=n: large images (scans of many workgroups / subsequences / streams); use a small n
        w, h = w * scale, h * scale
    gray = rng.random() < 0.2
    cmyk = CMYK and not gray and rng.random() < 0.35  # STRESS_CMYK=1: four components (Adobe CMYK as Pillow writes it)
    base = rng.integers(0, 256, (h, w, 1 if gray else (4 if cmyk else 3)))
    smooth = rng.random() < 0.6
    if smooth:
        yy, xx = np.mgrid[0:h, 0:w]
        base = (128 + 100 * np.sin(xx / rng.uniform(3, 60) + yy / rng.uniform(3, 60)))[..., None] + rng.normal(0, rng.uniform(0, 20), base.shape)
    img = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    kw = dict(format="JPEG", quality=int(rng.integers(3, 101)))
    if not gray:
        kw["subsampling"] = int(rng.integers(0, 3))
    prog = rng.random() < 0.25
    if prog:
        kw["progressive"] = True
    if rng.random() < 0.5:
        kw["restart_marker_blocks"] = int(rng.integers(1, 12))
    if rng.random() < 0.3:
        kw["optimize"] = True
    out = io.BytesIO()
    (Image.fromarray(img, "CMYK") if cmyk else Image.fromarray(img[..., 0] if gray else img)).save(out, **kw)
    files.append(out.getvalue())
    kinds.append((w, h, gray, prog, kw))

if os.environ.get("STRESS_SYNTH") is not None:
    # STRESS_SYNTH=1: every third file comes from the tree's own generator instead (tools/jpegsynth): baseline with restart
    # intervals of any length, 4:4:4 as three single-component scans (a multi-scan baseline frame: Pillow writes none)
    from tools import jpegsynth
    for i in range(0, n, 3):
        sub = str(rng.choice(["420", "422", "444", "444"]))
        w, h = int(rng.integers(1, 260)) * scale, int(rng.integers(1, 260)) * scale
        non = sub == "444" and rng.random() < 0.6
        files[i] = jpegsynth.encode(w, h, sub, int(rng.integers(5, 101)), int(rng.integers(0, 40)) if rng.random() < 0.6 else 0,
                                    seed=int(rng.integers(0, 1 << 30)), noninterleaved=non)
        kinds[i] = (w, h, False, False, {"synth": sub, "noninterleaved": non})

if os.environ.get("STRESS_SAMPLING") is not None:
    from tools import jpegsynth

    def draw_layout():
        while True:
            lay = tuple((int(rng.choice([1, 1, 2, 2, 4, 3])), int(rng.choice([1, 1, 2, 2, 4, 3]))) for _ in range(3))
            mh, mv = max(l[0] for l in lay), max(l[1] for l in lay)
            if any(mh % l[0] or mv % l[1] or (mh // l[0]) not in (1, 2, 4) or (mv // l[1]) not in (1, 2, 4) for l in lay):
                continue
            if sum(l[0] * l[1] for l in lay) <= 16:
                return lay

    for i in range(0, n, 2):
        lay = draw_layout()
        w, h = int(rng.integers(1, 260)) * scale, int(rng.integers(1, 260)) * scale
        mode = int(rng.integers(0, 4))  # 0, 1: interleaved baseline; 2: three scans; 3: progressive
        files[i] = jpegsynth.encode(w, h, quality=int(rng.integers(5, 101)), restart_interval=int(rng.integers(1, 40)) if rng.random() < 0.5 else 0,
                                    seed=int(rng.integers(0, 1 << 30)), sampling=lay, noninterleaved=(int(rng.integers(1, 3)) if mode == 2 else 0),
                                    progressive=(mode == 3))
        kinds[i] = (w, h, False, mode == 3, {"sampling": lay, "mode": mode})

bad = 0
names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException", 4: "ArgumentException"}
# ---- decode
refs = []
for f in files:
    try:
        refs.append(("OK", po.decode_8bit(f)))
    except po.OracleError as e:
        refs.append((e.kind, None))
outs, results = jl.decode_batch(files, jl.FMT_INTERLEAVED_U8)
outs4, results4 = jl.decode_batch(files, jl.FMT_RGBA_U8)
for i, ((kind, ref), out, res, out4, res4) in enumerate(zip(refs, outs, results, outs4, results4)):
    mine = names.get(res.status, str(res.status))
    if mine != kind:
        bad += 1
        print("decode status", i, kinds[i], kind, mine, res.detail)
    elif kind == "OK":
        px, info = ref
        if not np.array_equal(np.asarray(out), px):
            bad += 1
            print("decode pixels", i, kinds[i])
        if info.ncomp in (1, 3) and (res4.status != 0 or not np.array_equal(np.asarray(out4), po.ycbcr8_to_rgb(px, rgba=True, gray=(info.ncomp == 1)))):
            bad += 1
            print("decode rgba", i, kinds[i])
# ---- optimizer
n_opt = 0
for strip in (True, False):
    b = jl.OptimizeBatch().upload(files, strip).run()
    for i, f in enumerate(files):
        res, size = b.result(i)
        mine = names.get(res.status, str(res.status))
        if mine == "NotSupportedException":
            continue
        try:
            ref, kind = po.optimize(f, strip), "OK"
        except po.OracleError as e:
            ref, kind = None, e.kind
        if mine != kind:
            bad += 1
            print("optimize status", i, strip, kinds[i], kind, mine, res.detail)
        elif ref is not None:
            n_opt += 1
            if b.output(i) != ref:
                bad += 1
                print("optimize bytes", i, strip, kinds[i])
    b.close()
# ---- corrupted streams: random edits of the entropy-coded part (and sometimes of a header byte) of baseline files
def header_bytes(data):
    """positions of every byte of every marker segment (all SOS headers of a progressive file too), entropy data left out"""
    pos, i, n = [], 2, len(data)
    while i + 4 <= n:
        if data[i] != 0xFF:
            i += 1
            continue
        m = data[i + 1]
        if m in (0x00, 0xFF) or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        if m == 0xD9:
            break
        ln = (data[i + 2] << 8) | data[i + 3]
        pos.extend(range(i + 1, min(i + 2 + ln, n)))
        i += 2 + ln
        if m == 0xDA:  # entropy-coded data: up to the next marker that is not a restart marker
            while i + 1 < n and not (data[i] == 0xFF and data[i + 1] not in (0x00, 0xFF) and not 0xD0 <= data[i + 1] <= 0xD7):
                i += 1
    return pos


def mutate(data, rng):
    if HEADER_ONLY:
        b = bytearray(data)
        hb = header_bytes(data)
        for _ in range(2 if rng.random() < 0.3 else 1):
            if hb:
                b[hb[int(rng.integers(0, len(hb)))]] ^= 1 << int(rng.integers(0, 8))
        return bytes(b)
    sos = data.index(b"\xff\xda")
    lo = sos + 4 + data[sos + 3]
    b = bytearray(data)
    kind = int(rng.integers(0, 8))
    if len(b) - 2 <= lo:
        return bytes(b)
    pos = int(rng.integers(lo, len(b) - 2))
    if kind == 0:
        b[pos] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        del b[pos:pos + int(rng.integers(1, 6))]
    elif kind == 2:
        b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 6))).astype(np.uint8))
    elif kind == 3:
        b[pos:pos + 2] = bytes([0xFF, int(rng.choice([0xD0, 0xD3, 0xD7, 0xD9, 0xC4, 0xDA, 0x00, 0xFF, 0xE1, 0xC2]))])
    elif kind == 4:
        b = b[:pos] + b"\xff\xd9"
    elif kind == 5:
        k = int(rng.integers(1, 40))
        b[pos:pos + k] = bytes(k)
    elif kind == 6:
        k = int(rng.integers(1, 40))
        b[pos:pos + k] = b"\xff" * k
    else:
        hp = int(rng.integers(2, lo))  # a header byte
        b[hp] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)


def dc_category_above_16(data):
    """A DHT that gives a DC table a symbol above 16: outside the verified envelope."""
    i = 2
    while i + 4 <= len(data):
        if data[i] != 0xFF:
            i += 1
            continue
        m = data[i + 1]
        if m in (0x00, 0xFF) or 0xD0 <= m <= 0xD9:
            i += 2
            continue
        ln = (data[i + 2] << 8) | data[i + 3]
        if m == 0xC4:
            seg, j = data[i + 4:i + 2 + ln], 0
            while j + 17 <= len(seg):
                cnt = sum(seg[j + 1:j + 17])
                if (seg[j] >> 4) == 0 and any(v > 16 for v in seg[j + 17:j + 17 + cnt]):
                    return True
                j += 17 + cnt
        if m == 0xDA:
            return False
        i += 2 + ln
    return False


def frame_beyond_any_buffer(data):
    """A corrupted frame header can ask for hundreds of gigabytes (two 16-bit sizes x up to 255 components): neither the
    checker's numpy buffer nor the device has them.  The library fails such an image by itself (status 7); nothing to compare."""
    try:
        info, _ = po.identify(data)
    except po.OracleError:
        return False
    return info.width * info.height * max(info.ncomp, 1) > (1 << 31)


def keep(tag, i, data):
    """Mismatching inputs go to gpurun_out/stress/ (merged back from the GPU box) for a CPU-side look."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "stress")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"seed{sys.argv[2] if len(sys.argv) > 2 else 1}_{tag}_{i}.jpg"), "wb") as fh:
        fh.write(data)


n_mut = 0
mut = [mutate(f, rng) for f, k in zip(files, kinds) if not k[3]][: max(100, n // 3)]
mut = [f for f in mut if not dc_category_above_16(f) and not frame_beyond_any_buffer(f)]
refs = []
for f in mut:
    # (round 5: a failing baseline decode has called WriteBlock for every block in front of the one it threw in, none behind it
    # -- JpegHuffmanBaselineScanDecoder.cs:99-134, 153 -- and the writer's buffer is compared for those files too)
    try:
        px, _, err = po.decode_8bit_partial(f)
        refs.append(("OK" if err is None else err.kind, px))
    except po.OracleError as e:  # Identify failed: nothing was decoded
        refs.append((e.kind, None))
outs, results = jl.decode_batch(mut, jl.FMT_INTERLEAVED_U8)
n_partial_baseline = 0
for i, ((kind, px), out, res) in enumerate(zip(refs, outs, results)):
    mine = names.get(res.status, str(res.status))
    n_mut += 1
    if mine == "NotSupportedException" and kind != mine and res.detail == 6:
        continue  # documented fence: frame types / scan sequences outside the path
    if mine != kind:
        bad += 1
        print("mutated decode status", i, kind, mine, res.detail)
        keep("dec", i, mut[i])
    elif px is not None and out is not None and not np.array_equal(np.asarray(out), px):
        bad += 1
        print("mutated decode pixels", i, kind)
        keep("decpx", i, mut[i])
    n_partial_baseline += kind != "OK" and px is not None and out is not None
b = jl.OptimizeBatch().upload(mut, True).run()
for i, f in enumerate(mut):
    res, size = b.result(i)
    mine = names.get(res.status, str(res.status))
    if mine == "NotSupportedException":
        continue
    try:
        ref, kind = po.optimize(f, True), "OK"
    except po.OracleError as e:
        ref, kind = None, e.kind
    if mine != kind:
        bad += 1
        print("mutated optimize status", i, kind, mine, res.detail)
        keep("opt", i, f)
    elif ref is not None and b.output(i) != ref:
        bad += 1
        print("mutated optimize bytes", i)
        keep("optb", i, f)
b.close()

# ---- corrupted progressive streams: the same edits anywhere behind the first SOS (entropy data, later DHT / SOS headers)
pmut = [mutate(f, rng) for f, k in zip(files, kinds) if k[3]][: max(60, n // 6)]
pmut = [f for f in pmut if not dc_category_above_16(f) and not frame_beyond_any_buffer(f)]
prefs = []
for f in pmut:
    # (a failing progressive decode still flushes its partial store to the writer: the buffer is compared too, round 4)
    try:
        px, _, err = po.decode_8bit_partial(f)
        prefs.append(("OK" if err is None else err.kind, px))
    except po.OracleError as e:  # Identify failed: no scan decoder, nothing flushed
        prefs.append((e.kind, None))
outs, results = jl.decode_batch(pmut, jl.FMT_INTERLEAVED_U8)
n_partial = 0
for i, ((kind, px), out, res) in enumerate(zip(prefs, outs, results)):
    mine = names.get(res.status, str(res.status))
    n_mut += 1
    if mine == "NotSupportedException" and kind != mine and res.detail == 6:
        continue
    if mine != kind:
        bad += 1
        print("mutated progressive status", i, kind, mine, res.detail)
        keep("pdec", i, pmut[i])
    elif px is not None and out is not None and not np.array_equal(np.asarray(out), px):
        bad += 1
        print("mutated progressive pixels", i, kind)
        keep("pdecpx", i, pmut[i])
    n_partial += kind != "OK" and px is not None and out is not None

# ---- the per-scan boundaries (round 5): the same corrupted files through the entry points a caller with its own marker loop
# uses -- JpegDecoder.Decode() into a JpegBufferOutputWriter8Bit over the caller's buffer (one device call per scan over that
# canvas), and for progressive files jpgpu_progressive_begin / _scan / _dispose driven by the marker walk of
# tests/test_per_scan_gpu.py -- against the restatement's writer buffer, failing files included
n_session = 0
if os.environ.get("STRESS_NO_SESSION") is None:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_per_scan_gpu import Walk

    for i, f in enumerate(mut[:120]):
        try:
            px, info, err = po.decode_8bit_partial(f)
        except po.OracleError:
            continue
        d = jl.JpegDecoder()
        try:
            d.SetInput(f)
            d.Identify()
            if d.NumberOfComponents != info.ncomp or d.Width != info.width or d.Height != info.height:
                continue
            buf = np.zeros(d.Width * d.Height * d.NumberOfComponents, np.uint8)
            d.SetOutputWriter(jl.JpegBufferOutputWriter8Bit(d.Width, d.Height, d.NumberOfComponents, buf))
            kind = "OK"
            try:
                d.Decode()
            except jl.JpegError as e:
                kind = type(e).__name__
                if kind == "NotSupportedException" and (err is None or err.kind != kind):
                    continue  # a documented fence
            want = "OK" if err is None else err.kind
            n_session += 1
            if kind != want:
                bad += 1
                print("session decode status", i, want, kind)
                keep("sdec", i, f)
            elif not np.array_equal(buf.reshape(px.shape), px):
                bad += 1
                print("session decode pixels", i, want)
                keep("sdecpx", i, f)
        except jl.JpegError:
            continue
        finally:
            d.close()
    for i, f in enumerate(pmut[:80]):
        try:
            px, info, err = po.decode_8bit_partial(f)
        except po.OracleError:
            continue
        w = Walk(f)
        st = {"dec": None, "err": None, "fh": None}

        def on_frame(marker, fh):
            if marker != 0xC2:
                raise jl.NotSupportedException("not a progressive frame")
            st["fh"], st["dec"] = fh, jl.JpegGpuProgressiveScanDecoder(fh)

        def on_scan(entropy, sh):
            if st["err"] is not None or st["dec"] is None:
                return 0
            try:
                return st["dec"].ProcessScan(entropy, sh, w.quantization_tables(), w.huffman_tables(), w.dri)
            except jl.JpegError as e:
                st["err"] = e
                return 0

        try:
            w.run(on_frame, on_scan)
        except Exception:
            pass  # a header this walk does not read the way the reference does: the batch entry points cover those files
        else:
            fh = st["fh"]
            # (only where the failure is a SCAN's: "... at offset N. ..." is the reference's marker walk giving up -- a table that
            # does not parse, a segment that runs past the file -- and this tool's own walk is not that walk)
            same_failure = (st["err"] is None) == (err is None) and (st["err"] is None or (type(st["err"]).__name__ == err.kind and "at offset" not in str(err)))
            if st["dec"] is not None and same_failure and fh.NumberOfComponents == info.ncomp and fh.SamplesPerLine == info.width:
                try:
                    out = st["dec"].Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, fh.NumberOfComponents)
                    n_session += 1
                    if not np.array_equal(out, px):
                        bad += 1
                        print("session progressive pixels", i, None if err is None else err.kind)
                        keep("spdecpx", i, f)
                except jl.NotSupportedException:
                    pass
        if st["dec"] is not None:
            st["dec"].close()

# ---- encoder: random images / samplings / qualities / table modes, grouped by the parameters one batch shares
n_enc = 0
n_one_pass = n_fell_back = 0
groups = {}
for i in range(max(60, n // 4)):
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 200))
    gray = rng.random() < 0.15
    luma = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (4, 2)][int(rng.integers(0, 6))] if not gray else (1, 1)
    # (mostly a few qualities: images that share sampling / quality / tables / pixel form go up as ONE batch, and batches of two or
    # more restart-free images are what takes the encoder's one-pass entropy stage)
    q = int(rng.integers(1, 101)) if rng.random() < 0.3 else int([1, 30, 75, 90, 100][int(rng.integers(0, 5))])
    mode = int(rng.integers(0, 3))
    rgb = int((not gray) and rng.random() < 0.5)
    ri = int(rng.integers(1, 40)) if rng.random() < 0.4 else 0  # restart interval (the encoder's extension)
    img = rng.integers(0, 256, (h, w) if gray else (h, w, 3)).astype(np.uint8)
    if rng.random() < 0.5:
        img = (img.astype(np.int32) // int(rng.integers(1, 40)) * int(rng.integers(1, 8))).clip(0, 255).astype(np.uint8)
    if rgb and i % 2 == 1:  # Rgba32 pixels (the reference's EncoderBenchmark input): an alpha byte of noise that nobody may read
        rgb = 2
        img = np.ascontiguousarray(np.concatenate([img, np.random.default_rng(i).integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=-1))
    groups.setdefault((luma, q, mode, rgb, ri), []).append(img)
for (luma, q, mode, rgb, ri), imgs in groups.items():
    e = jl.EncodeBatch().upload(imgs, luma, q, rgb=bool(rgb), optimize_coding=mode, restart_interval=ri).encode()
    for k, im in enumerate(imgs):
        src = po.rgba_to_ycbcr8(im) if rgb == 2 else (po.rgb_to_ycbcr8(im) if rgb else im)
        try:
            ref = po.encode_8bit(src, luma[0], luma[1], q, optimize_coding=mode, restart_interval=ri)
        except po.OracleError:
            ref = None
        try:
            got = e.output(k)
        except jl.JpegError:
            got = None
        n_enc += 1
        if got != ref:
            bad += 1
            print("encode", luma, q, mode, rgb, ri, im.shape, None if got is None else len(got), None if ref is None else len(ref))
    a, b_ = e.emit_passes()
    n_one_pass += a
    n_fell_back += b_
    e.close()
print(f"stress: {n} files, {sum(k[3] for k in kinds)} progressive, {n_opt} optimizer outputs compared, {n_enc} encoder outputs compared ({n_one_pass} batches through the one-pass entropy stage, {n_fell_back} of them fell back), {n_mut} corrupted files ({n_partial_baseline} failing baseline writers + {n_partial} partial progressive flushes compared), {n_session} per-scan sessions compared, mismatches: {bad}")
sys.exit(1 if bad else 0)
