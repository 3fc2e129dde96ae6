#!/usr/bin/env python3
"""jpgpu_batch_upload of the headline batch (1024 x 4K 4:2:0 Q75 DRI=4) under different host crew sizes: where the
ingest time goes (IngestStats) and what the staging ring sustains.  Ring geometry comes from JPGPU_STAGING_SLOTS /
JPGPU_STAGING_SLOT_MB (read when the context is created)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl  # noqa: E402
from tools import jpegsynth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    buf, sizes, stride = jpegsynth.encode_batch(n, 3840, 2160, "420", 75, 4, seed0=1, nthreads=os.cpu_count())
    files = [buf[i * stride:i * stride + int(sizes[i])] for i in range(n)]
    total = float(sizes.sum())
    rows = []
    for slots, mb in ((4, 32), (8, 32), (16, 16), (8, 8)):
        os.environ["JPGPU_STAGING_SLOTS"], os.environ["JPGPU_STAGING_SLOT_MB"] = str(slots), str(mb)
        ctx = jl.Context(0)
        b = jl.Batch(ctx)
        for threads in (0, 4, 8, 12, 16, 24, 32):
            ctx.set_host_threads(threads)
            b.upload(files, jl.FMT_INTERLEAVED_U8)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                b.upload(files, jl.FMT_INTERLEAVED_U8)
                dt = time.perf_counter() - t0
                st = b.ingest_stats()
                if best is None or dt < best[0]:
                    best = (dt, st)
            rows.append({"slots": slots, "slot_mb": mb, "threads": best[1]["threads"], "ms": round(best[0] * 1e3, 2), "GBps": round(total / best[0] / 1e9, 1),
                         "parse_ms": round(best[1]["parse_ms"], 2), "copy_ms": round(best[1]["copy_ms"], 2), "walk_ms": round(best[1]["full_walk_ms"], 2),
                         "layout_ms": round(best[1]["layout_ms"], 2)})
            print(json.dumps(rows[-1]), flush=True)
        b.close()
        ctx.close()


if __name__ == "__main__":
    main()
