"""Debug aid: compare the lane-per-interval and wave-per-stream progressive kernels level by level on one file
(coefficient store: int16[blocks, 64], MCU order)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl


def store(data):
    b = jl.Batch().upload([data], jl.FMT_INTERLEAVED_U8).decode().sync()
    c = b.coefficients(0).copy()
    b.close()
    return c


data = open(sys.argv[1], "rb").read()
for levels in (1, 2, 3):
    os.environ["JPGPU_DEBUG_MAX_PROGRESSIVE_SCANS"] = str(levels)
    os.environ["JPGPU_PROG_STREAM_MAX_INTERVALS"] = "0"
    a = store(data)
    os.environ["JPGPU_PROG_STREAM_MAX_INTERVALS"] = "16"
    b = store(data)
    bad = np.argwhere(a != b)
    print("levels", levels, a.shape, "mismatching coefficients", len(bad), "first (block, k)", bad[:6].tolist())
    if len(bad):
        blk = bad[0][0]
        print(" lanes :", a[blk].tolist())
        print(" stream:", b[blk].tolist())
        ks = np.bincount(bad[:, 1], minlength=64)
        print(" by k:", ks.tolist())
        break
