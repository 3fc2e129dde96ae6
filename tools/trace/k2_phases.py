#!/usr/bin/env python3
"""tools/trace/k2_phases.py [n_images] -- where a K2 wave spends its cycles (needs a library built with -DJPGPU_K2_PROFILE:
tools/trace/ab_build.sh "-DJPGPU_K2_PROFILE" python tools/trace/k2_phases.py).  Per-wave cycle counters of huffman_decode_kernel
(lane 0 of every wave adds its own): table staging, stream init, and per block: symbol decode, ring top-up, flush."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from jpeglibrary_amd import _capi
from tools import jpegsynth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = sys.argv[2] if len(sys.argv) > 2 else ""
w, h, q = (1920, 1080, 90) if mode == "q90" else (3840, 2160, 75)
buf, sizes, stride = jpegsynth.encode_batch(n, w, h, "420", q, 0 if mode == "dri0" else 4, seed0=1000)
files = [bytes(buf[i * stride:i * stride + int(sizes[i])]) for i in range(n)]
b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
b.decode().sync()
lib = C.CDLL(_capi.LIB_PATH)
out = (C.c_ulonglong * 8)()
if mode == "dri0":  # the K2S final pass (sf_wave): the same shares, + how much of the wave's block steps its lanes fill
    b.decode().sync()
    assert lib.jpgpu_debug_sf_profile(out, 1) == 0
    for _ in range(3):
        b.decode().sync()
    assert lib.jpgpu_debug_sf_profile(out, 1) == 0
    waves, skip, dec, top, fl, total, steps, lane_blocks = [out[i] for i in range(8)]
    print(f"{n} x {w}x{h} Q{q} DRI=0, final pass: {waves // 3} waves per decode; cycles per wave {total / waves:.0f}; block steps per wave {steps / waves:.1f}, "
          f"lanes filled {100.0 * lane_blocks / (64.0 * steps):.1f} %")
    for name, v in (("open + skip", skip), ("symbol decode", dec), ("ring top-up", top), ("flush", fl)):
        print(f"  {name:14s} {100.0 * v / total:5.1f} %   {v / steps:9.0f} cycles per block step")
    print(f"  other          {100.0 * (total - skip - dec - top - fl) / total:5.1f} %")
    sys.exit(0)
assert lib.jpgpu_debug_k2_profile(out, 1) == 0
for _ in range(3):
    b.decode().sync()
assert lib.jpgpu_debug_k2_profile(out, 1) == 0
waves, stage, init, dec, top, fl, total = out[0], out[1], out[2], out[3], out[4], out[5], out[6]
print(f"{n} x {w}x{h} Q{q}: {waves // 3} waves per decode; cycles per wave {total / waves:.0f}")
for name, v in (("table staging", stage), ("stream init", init), ("symbol decode", dec), ("ring top-up", top), ("flush", fl)):
    print(f"  {name:14s} {100.0 * v / total:5.1f} %   {v / waves:9.0f} cycles per wave")
print(f"  other          {100.0 * (total - stage - init - dec - top - fl) / total:5.1f} %")
print("stage_ms", b.stage_times() if hasattr(b, "stage_times") else "")
