#!/usr/bin/env python3
"""Does running the entropy stage of one half of a batch beside the IDCT/output stage of the other half help?
Two contexts (= two HIP streams), half the images each, decode() issued on both before either is synchronised,
against one batch over all images.  Measurement only ('what bounds what')."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from tools import jpegsynth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dri = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = 10
buf, sizes, stride = jpegsynth.encode_batch(n, 3840, 2160, "420", 75, dri, seed0=1, nthreads=os.cpu_count())
files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
ctx_a, ctx_b = jl.Context(0), jl.Context(0)


def run(batches, order):
    for b in batches:
        b.decode()
    for b in batches:
        b.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        order(batches)
    for b in batches:
        b.sync()
    return (time.perf_counter() - t0) / steps * 1e3


whole = jl.Batch(ctx_a).upload(files, jl.FMT_INTERLEAVED_U8)
t_whole = run([whole], lambda bs: bs[0].decode())
whole.close()
halves = [jl.Batch(ctx_a).upload(files[:n // 2], jl.FMT_INTERLEAVED_U8), jl.Batch(ctx_b).upload(files[n // 2:], jl.FMT_INTERLEAVED_U8)]
t_halves = run(halves, lambda bs: [b.decode() for b in bs])


def staged(bs):  # entropy of B is issued right behind entropy of A, so it runs beside IDCT of A
    bs[0].run_entropy()
    bs[1].run_entropy()
    bs[0].run_idct()
    bs[1].run_idct()


t_staged = run(halves, staged)
for b in halves:
    b.close()
same = [jl.Batch(ctx_a).upload(files[:n // 2], jl.FMT_INTERLEAVED_U8), jl.Batch(ctx_a).upload(files[n // 2:], jl.FMT_INTERLEAVED_U8)]
t_serial = run(same, lambda bs: [b.decode() for b in bs])
for b in same:
    b.close()
ctxs = [ctx_a, ctx_b, jl.Context(0), jl.Context(0)]
more = {}
for parts, streams in ((4, 2), (4, 4), (8, 2), (3, 3)):
    step = (n + parts - 1) // parts
    bs = [jl.Batch(ctxs[k % streams]).upload(files[k * step:(k + 1) * step], jl.FMT_INTERLEAVED_U8) for k in range(parts)]
    more[(parts, streams)] = run(bs, lambda b_: [x.decode() for x in b_])
    for x in bs:
        x.close()
print("parts x streams:", {k: round(v, 2) for k, v in more.items()})
print(f"{n} x 4K DRI={dri}: one batch {t_whole:.2f} ms; two halves on two streams {t_halves:.2f} ms (staged issue {t_staged:.2f} ms); "
      f"two halves on one stream {t_serial:.2f} ms")
