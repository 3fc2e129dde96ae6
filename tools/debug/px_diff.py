import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
d = open(os.path.join(os.path.dirname(__file__), "px121.jpg"), "rb").read()
ref = po.decode_8bit(d)[0]
for env in (None, "1"):
    outs, res = jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)
    out = np.asarray(outs[0])
    diff = np.argwhere((out != ref).any(axis=2))
    print("status", res[0].status, res[0].detail, "diff pixels", len(diff))
    if len(diff):
        ys, xs = diff[:, 0], diff[:, 1]
        print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
        mcus = sorted({(int(y) // 8) * 34 + int(x) // 8 for y, x in diff})
        print("mcus", mcus[:40], len(mcus))
        print("intervals", sorted({m // 7 for m in mcus}))
    break
b = jl.Batch().upload([d], jl.FMT_PLANAR_I16).run_entropy().sync()
co = b.coefficients(0)
rc = po.decode_coefficients(d)
print(type(co), type(rc))
