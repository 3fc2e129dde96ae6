#!/usr/bin/env python3
"""tools/trace/partial_flush_probe.py [n] [seed] -- corrupted progressive files: the GPU's output for a FAILING file against what
the reference leaves in the writer's buffer (Decode()'s finally -> Dispose(): the partial store transformed and flushed)."""
import io
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
files = []
for i in range(n):
    w, h = int(rng.integers(16, 260)), int(rng.integers(16, 200))
    yy, xx = np.mgrid[0:h, 0:w]
    px = np.stack([128 + 100 * np.sin(xx / rng.uniform(3, 40) + yy / rng.uniform(3, 40)) for _ in range(3)], -1) + rng.normal(0, rng.uniform(0, 25), (h, w, 3))
    buf = io.BytesIO()
    kw = dict(format="JPEG", quality=int(rng.integers(20, 98)), progressive=True, subsampling=int(rng.integers(0, 3)))
    img = Image.fromarray(np.clip(px, 0, 255).astype(np.uint8))
    if rng.random() < 0.2:
        img = img.convert("L")
        kw.pop("subsampling")
    img.save(buf, **kw)
    d = bytearray(buf.getvalue())
    sos = [k for k in range(len(d) - 1) if d[k] == 0xFF and d[k + 1] == 0xDA]
    mode = rng.integers(0, 4)
    if mode == 0:  # a flipped byte inside some scan's entropy data
        k = int(rng.integers(0, len(sos)))
        lo = sos[k] + 14
        hi = sos[k + 1] if k + 1 < len(sos) else len(d) - 2
        if hi > lo:
            p = int(rng.integers(lo, hi))
            d[p] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1:  # truncated inside a scan, EOI kept
        p = int(rng.integers(sos[0] + 14, len(d) - 2))
        d = d[:p] + b"\xff\xd9"
    elif mode == 2:  # a run of zeros
        p = int(rng.integers(sos[0] + 14, len(d) - 8))
        d[p:p + 6] = bytes(6)
    else:  # bytes deleted
        p = int(rng.integers(sos[0] + 14, len(d) - 8))
        del d[p:p + int(rng.integers(1, 5))]
    files.append(bytes(d))
b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
stats = {"ok_both": 0, "fail_both_same_output": 0, "fail_both_other_output": 0, "class_mismatch": 0, "host_fail": 0}
bad = []
for i, f in enumerate(files):
    r = b.result(i)
    try:
        ref, info, err = po.decode_8bit_partial(f)
    except po.OracleError:
        stats["host_fail"] += 1  # Identify failed: no scan decoder, nothing flushed
        continue
    if b.image_info(i).status != 0:
        stats["host_fail"] += 1
        continue
    out = b.output(i)
    if err is None and r.status == 0:
        stats["ok_both"] += int(np.array_equal(out, ref))
        if not np.array_equal(out, ref):
            bad.append((i, "clean decode differs"))
    elif err is not None and r.status != 0:
        if np.array_equal(out, ref):
            stats["fail_both_same_output"] += 1
        else:
            stats["fail_both_other_output"] += 1
            bad.append((i, f"{err} | detail {r.detail} | {int((out != ref).sum())} samples differ of {out.size}"))
    else:
        stats["class_mismatch"] += 1
        bad.append((i, f"oracle {err} gpu {r.status}"))
print(stats)
for i, why in bad[:12]:
    print(" ", i, len(files[i]), why)
if bad and len(sys.argv) > 3:
    os.makedirs(sys.argv[3], exist_ok=True)
    for i, _ in bad[:8]:
        open(os.path.join(sys.argv[3], f"partial_{i}.jpg"), "wb").write(files[i])
