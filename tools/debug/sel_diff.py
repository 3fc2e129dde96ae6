import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
from tools import jpegsynth
good = bytes(jpegsynth.encode(104, 72, "420", 80, 0, seed=91))
a = good.index(b"\xff\xda")
d = good[:a + 5] + good[a + 9:a + 10] + good[a + 6:]
ref = po.decode_8bit(d)[0]
outs, res = jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)
out = np.asarray(outs[0])
print("status", res[0].status, res[0].detail)
for c in range(3):
    diff = np.argwhere(out[..., c] != ref[..., c])
    print("chan", c, "diff", len(diff), (diff[:, 0].min(), diff[:, 0].max(), diff[:, 1].min(), diff[:, 1].max()) if len(diff) else None)
    if len(diff):
        y, x = diff[0]
        print("  first", y, x, out[y, x, c], ref[y, x, c], "mcu", (y // 16) * 7 + x // 16)
        mcus = sorted({(int(y) // 16) * 7 + int(x) // 16 for y, x in diff})
        print("  mcus", mcus)
print("chan0 ref unique", np.unique(ref[..., 0])[:5], "out", np.unique(out[..., 0])[:5])
