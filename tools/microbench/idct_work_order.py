#!/usr/bin/env python3
"""A/B inside one process: K3's work-list order (XCD interleave) and tile width (line-aligned), same files, batches
decoded alternately.  The switches are read when a batch is laid out (upload)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from tools import jpegsynth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
buf, sizes, stride = jpegsynth.encode_batch(n, 3840, 2160, "420", 75, 4, seed0=1, nthreads=os.cpu_count())
files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
ctx = jl.Context(0)
configs = {"plain": {"JPGPU_XCD_MAP": "0", "JPGPU_TILE_ALIGN": "0"}, "xcd8": {"JPGPU_XCD_MAP": "8", "JPGPU_TILE_ALIGN": "0"},
           "xcd8+aligned (default)": {"JPGPU_XCD_MAP": "8", "JPGPU_TILE_ALIGN": "1"}, "aligned": {"JPGPU_XCD_MAP": "0", "JPGPU_TILE_ALIGN": "1"}}
batches = {}
for name, env in configs.items():
    for k in ("JPGPU_XCD_MAP", "JPGPU_TILE_ALIGN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    batches[name] = jl.Batch(ctx).upload(files, jl.FMT_INTERLEAVED_U8)
for b in batches.values():
    b.decode()
    b.sync()
    b.stage_ms()
for rnd in range(4):
    line = []
    for name, b in batches.items():
        for _ in range(5):
            b.decode()
        b.sync()
        st = b.stage_ms()
        line.append(f"{name}: K1 {st['marker_index']:.2f} K2 {st['huffman']:.2f} K3 {st['idct']:.2f} total {st['total']:.2f}")
    print(" | ".join(line))
