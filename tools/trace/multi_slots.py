#!/usr/bin/env python3
"""jpgpu_multi_* at benchmark scale on whatever devices there are: `per_slot` 4K images PER SLOT (weak scaling, as the
8-GPU run shards: 1024 per GPU) over 1 .. `slots` device slots (slot s on device s mod device_count).  Three ways of feeding:
  sync      jpgpu_multi_decode: upload -> decode -> wait, pageable input through the staging ring
  pipelined jpgpu_multi_submit / _wait with two calls in flight (call k + 1 uploaded beside call k's decode), pageable input
  pinned    the same from page-locked memory (JPGPU_UPLOAD_PINNED: DMA from where the files lie, no staging copy)
Prints one JSON line per (slots, mode): ms per call, the slowest shard's upload time, Mpixels/s with the host in the loop.
With one device in the box all slots share its PCIe link and its HBM, so the per-slot ingest time flat in `slots` is the
HOST side's scaling (crew threads = granted CPUs / slots; the pinned path uses none for the entropy bytes).
Usage: multi_slots.py [per_slot] [slots] [out.jsonl]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl  # noqa: E402
from tools import jpegsynth  # noqa: E402

per_slot = int(sys.argv[1]) if len(sys.argv) > 1 else 256
max_slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out_path = sys.argv[3] if len(sys.argv) > 3 else None
g = max(1, jl.device_count())
n_max = per_slot * max_slots
buf, sizes, stride = jpegsynth.encode_batch(n_max, 3840, 2160, "420", 75, 4, seed0=1, nthreads=16)
lines = []
for slots in range(1, max_slots + 1):
    n = per_slot * slots
    m = jl.MultiDecoder([s % g for s in range(slots)])
    ctx0 = jl.Batch._borrowed(0, jl._capi.lib.jpgpu_multi_context(m._h, 0), 0).ctx
    import ctypes as C

    p = C.c_void_p()
    assert jl._capi.lib.jpgpu_host_alloc(ctx0._h, int(stride) * n, C.byref(p)) == 0
    arena = np.frombuffer((C.c_uint8 * (int(stride) * n)).from_address(p.value), dtype=np.uint8)
    arena[:] = buf[:arena.size]
    pageable = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
    pinned = [arena[i * stride:i * stride + int(sizes[i])] for i in range(n)]
    for mode in ("sync", "pipelined", "pinned"):
        reps = 6
        ups = []
        if mode == "sync":
            m.decode(pageable)
            t = time.perf_counter()
            for _ in range(reps):
                m.decode(pageable)
                ups.append(m.upload_ms)
            dt = (time.perf_counter() - t) / reps
        else:
            files = pinned if mode == "pinned" else pageable
            m.wait(m.submit(files, pinned=(mode == "pinned")))
            t = time.perf_counter()
            prev = m.submit(files, pinned=(mode == "pinned"))
            for _ in range(reps - 1):
                cur = m.submit(files, pinned=(mode == "pinned"))
                m.wait(prev)
                ups.append(m.upload_ms)
                prev = cur
            m.wait(prev)
            ups.append(m.upload_ms)
            dt = (time.perf_counter() - t) / reps
        line = {"slots": slots, "devices": g, "mode": mode, "images_per_call": n, "ms_per_call": round(dt * 1e3, 2),
                "upload_ms_slowest_shard": round(float(np.median(ups)), 2), "Mpixels/s": round(n * 3840 * 2160 / dt / 1e6, 0)}
        lines.append(line)
        print(json.dumps(line), flush=True)
    outs, results = jl.decode_batch(pageable[:4])
    assert all(np.array_equal(m.output(i), outs[i]) for i in range(4))
    jl._capi.lib.jpgpu_host_free(ctx0._h, p)
    m.close()
if out_path:
    with open(out_path, "w") as f:
        for ln in lines:
            f.write(json.dumps(ln) + "\n")
print("samples equal a plain batch")
