import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
d = open(os.path.join(os.path.dirname(__file__), sys.argv[1]), "rb").read()
ref = po.decode_8bit(d)[0]
outs, res = jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)
out = np.asarray(outs[0])
print("status", res[0].status, res[0].detail, out.shape)
for c in range(out.shape[2]):
    diff = np.argwhere(out[..., c] != ref[..., c])
    print("chan", c, "diff", len(diff), (diff[:, 0].min(), diff[:, 0].max(), diff[:, 1].min(), diff[:, 1].max()) if len(diff) else None)
store = po.decode_progressive_store(d)
b = jl.Batch().upload([d], jl.FMT_PLANAR_I16).run_entropy().sync()
co = b.coefficients(0)
print(type(store), getattr(store, "shape", None), type(co), getattr(co, "shape", None))
try:
    rs = np.asarray(store[0] if isinstance(store, tuple) else store).reshape(-1, 64)
    cs = np.asarray(co).reshape(-1, 64)
    n = min(len(rs), len(cs))
    bad = np.argwhere((rs[:n] != cs[:n]).any(axis=1)).ravel()
    print("blocks", len(rs), len(cs), "bad blocks", len(bad), bad[:20])
    for bk in bad[:3]:
        idx = np.argwhere(rs[bk] != cs[bk]).ravel()
        print(" block", bk, "coef idx", idx[:16], "ref", rs[bk][idx[:8]], "got", cs[bk][idx[:8]])
except Exception as e:
    print("coef compare failed", e)
