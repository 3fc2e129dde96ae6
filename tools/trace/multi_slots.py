#!/usr/bin/env python3
"""jpgpu_multi_* at benchmark scale on whatever devices there are: n 4K images over `slots` device slots (slot s on device
s mod device_count); prints upload / decode times of the slowest shard and checks a sample of images against a plain batch.
Usage: multi_slots.py [images] [slots]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl  # noqa: E402
from tools import jpegsynth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = max(1, jl.device_count())
buf, sizes, stride = jpegsynth.encode_batch(n, 3840, 2160, "420", 75, 4, seed0=1, nthreads=16)
files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
m = jl.MultiDecoder([s % g for s in range(slots)])
for rep in range(3):
    t = time.perf_counter()
    m.decode(files)
    dt = time.perf_counter() - t
    print(f"{n} x 4K over {slots} slots on {g} device(s): call {dt * 1e3:.1f} ms (upload {m.upload_ms:.1f} + decode {m.decode_ms:.1f} of the slowest shard) "
          f"= {n * 3840 * 2160 / dt / 1e6:.0f} Mpixels/s with the host in the loop", flush=True)
outs, results = jl.decode_batch(files[:8])
assert all(np.array_equal(m.output(i), outs[i]) for i in range(8))
print("sample of 8 images equals a plain batch")
