"""Soak: decode one batch of progressive files many times and compare every run with the oracle-checked first run
(the scans of a frame run as one launch and follow each other through device-scope fences: any ordering bug shows up as
a run that differs)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import jpeglibrary_amd as jl
from oracle import pyoracle as po

n, w, h, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 96, 1280, 720, int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = bench.progressive_batch(n, w, h, 80, 100, os.cpu_count() or 8)
b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
b.decode().sync()
first = [np.asarray(b.output(i)).copy() for i in range(n)]
for i in (0, n // 2, n - 1):
    assert np.array_equal(first[i], po.decode_8bit(files[i])[0]), i
bad = 0
for r in range(reps):
    b.decode().sync()
    for i in range(n):
        if not np.array_equal(np.asarray(b.output(i)), first[i]):
            bad += 1
            print("run", r, "image", i, "differs")
print("soak:", reps, "runs x", n, "images; differing outputs:", bad)
