#!/usr/bin/env python3
"""tools/trace/k1_concurrency_probe.py [images] -- K1 takes a group's place in its list from the workgroup index (round 6): does any group
ever run out of patience when kernels of OTHER contexts compete for the CUs?  Two / four contexts decode their own batches at once,
DRI = 4 and DRI = 0; prints the time per round of decodes, jpgpu_batch_marker_fallbacks of every batch (expected: 0), and compares the
first and last image of every batch with the checker."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from oracle import pyoracle as po
from tools import jpegsynth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for dri in (4, 0):
    buf, sizes, stride = jpegsynth.encode_batch(n, 3840, 2160, "420", 75, dri, seed0=1, nthreads=os.cpu_count())
    files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
    for n_ctx in (2, 4):
        ctxs = [jl.Context(0) for _ in range(n_ctx)]
        step = n // n_ctx
        bs = [jl.Batch(ctxs[k]).upload(files[k * step:(k + 1) * step], jl.FMT_INTERLEAVED_U8) for k in range(n_ctx)]
        t0 = time.perf_counter()
        for _ in range(12):
            for b in bs:
                b.decode()
        for b in bs:
            b.sync()
        ms = (time.perf_counter() - t0) / 12 * 1e3
        ok = True
        for k, b in enumerate(bs):
            for i in (0, step - 1):
                ref, _ = po.decode_8bit(bytes(files[k * step + i]))
                ok = ok and b.result(i).status == 0 and np.array_equal(b.output(i), ref)
        print(f"DRI={dri} {n_ctx} contexts x {step} images: {ms:.2f} ms per round, marker_fallbacks {[b.marker_fallbacks() for b in bs]}, equal to the checker: {ok}", flush=True)
        for b in bs:
            b.close()
