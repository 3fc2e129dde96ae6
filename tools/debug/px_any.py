import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
d = open(os.path.join(os.path.dirname(__file__), sys.argv[1]), "rb").read()
ref = po.decode_8bit(d)[0]
outs, res = jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)
out = np.asarray(outs[0])
print("status", res[0].status, res[0].detail, out.shape, ref.shape)
for c in range(out.shape[2]):
    diff = np.argwhere(out[..., c] != ref[..., c])
    print("chan", c, "diff", len(diff))
    for y, x in diff[:6]:
        print("   ", y, x, "got", out[y, x, c], "ref", ref[y, x, c], "block", (x // 8, y // 8))
blocks = sorted({(int(x) // 8, int(y) // 8) for y, x in np.argwhere((out != ref).any(axis=2))})
print("blocks", blocks[:20], len(blocks))
if blocks:
    info, store, quant = po.decode_progressive_store(d)
    bx, by = blocks[0]
    print("coefs", store[0][(bx, by)])
    print("quant", quant[0])
    print("got block\n", out[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8, 0])
    print("ref block\n", ref[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8, 0])
