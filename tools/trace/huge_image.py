#!/usr/bin/env python3
"""One image large enough that byte offsets inside it pass 4 GiB (40000 x 40000 4:2:0: 4.8 GB of samples, 9.4 M MCUs): GPU
decode against the oracle.  Usage: huge_image.py [width height dri]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tools import jpegsynth  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
h = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
dri = int(sys.argv[3]) if len(sys.argv) > 3 else 8
t = time.time()
data = bytes(jpegsynth.encode(w, h, "420", 75, dri, seed=9))
print(f"{w}x{h} DRI={dri}: {len(data) / 1e6:.1f} MB in {time.time() - t:.1f} s", flush=True)
t = time.time()
outs, results = jl.decode_batch([data])
print(f"gpu decode status {results[0].status} in {time.time() - t:.1f} s, output {outs[0].shape}", flush=True)
t = time.time()
ref = po.decode_8bit(data)[0]
print(f"oracle in {time.time() - t:.1f} s", flush=True)
same = outs[0].shape == ref.shape
for y0 in range(0, h, 2048):  # in slabs: a 12.9 GB comparison mask is not needed
    if same and not np.array_equal(outs[0][y0:y0 + 2048], ref[y0:y0 + 2048]):
        same = False
        d = np.argwhere(outs[0][y0:y0 + 2048] != ref[y0:y0 + 2048])
        print("first differences (y, x, c):", [(int(y0 + a), int(b_), int(c)) for a, b_, c in d[:5]], "in this slab:", len(d))
print("identical:", same)
sys.exit(0 if same else 1)
