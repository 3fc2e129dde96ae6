import sys, time, os
sys.path.insert(0, os.getcwd())
import jpeglibrary_amd as jl
from bench import progressive_batch
src = progressive_batch(16, 3840, 2160, 75, 1, 16)
for n in (384, 416, 448, 512, 576):
    files = [src[i % 16] for i in range(n)]
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
    b.decode().sync()
    t = time.perf_counter(); b.decode().sync(); dt = time.perf_counter() - t
    print(n, "frames", round(dt * 1e3, 1), "ms", "fallbacks", b.progressive_fallbacks(), flush=True)
    b.close()
