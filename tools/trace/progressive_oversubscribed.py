#!/usr/bin/env python3
"""Decode n copies of a few 4K progressive images under different stream-kernel LDS shapes / launch modes and list the
images that fail or differ from the first copy of their source; the sha256 of every first copy's samples is printed
("sha256 <source> <hex>", last pass) so that the test that runs this tool can hold them against the checker's samples.
Debugging aid for the pipelined progressive launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl  # noqa: E402
from bench import progressive_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
distinct = 16
src = progressive_batch(distinct, 3840, 2160, 75, 1, 16)
files = [src[i % distinct] for i in range(n)]
import hashlib  # noqa: E402
b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
for rep in range(3):
    b.decode().sync()
    bad = [(i, b.result(i).status, b.result(i).detail, b.result(i).error_interval) for i in range(n) if b.result(i).status != 0]
    diff = []
    if not bad:
        first = [b.output(i) for i in range(distinct)]
        if rep == 2:
            for i in range(distinct):
                print(f"sha256 {i} {hashlib.sha256(first[i].tobytes()).hexdigest()}", flush=True)
        for i in range(distinct, n, 37):
            if not np.array_equal(b.output(i), first[i % distinct]):
                diff.append(i)
    for (i, _, _, _) in bad[:2]:
        good, mine = b.coefficients(i % distinct), b.coefficients(i)
        d = np.argwhere(good != mine)
        blocks = np.unique(d[:, 0])
        print(f"  image {i}: {len(d)} coefficients differ in {len(blocks)} blocks; first blocks {blocks[:8].tolist()} (Y blocks per MCU row 960+480); "
              f"zig-zag positions {np.unique(d[:, 1])[:20].tolist()}; first diffs {[(int(a), int(c), int(good[a, c]), int(mine[a, c])) for a, c in d[:6]]}", flush=True)
    print(f"n={n} ring={os.environ.get('JPGPU_PS_RING')} chunk={os.environ.get('JPGPU_PS_CHUNK')} nopipe={os.environ.get('JPGPU_PROG_NO_PIPELINE')} "
          f"rep={rep}: failed {len(bad)} {bad[:6]} differing {diff[:6]} fallbacks {b.progressive_fallbacks()}", flush=True)
b.close()
