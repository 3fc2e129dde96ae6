import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import jpeglibrary_amd as jl
from oracle import pyoracle as po
from golden_util import read_jpeg
data = read_jpeg(sys.argv[1]) if len(sys.argv) > 1 else read_jpeg("progress.jpg")
info, blocks, _ = po.decode_progressive_store(data)
b = jl.Batch().upload([data]).decode().sync()
print("result", b.result(0).status, b.result(0).detail)
coefs = b.coefficients(0)
comps = [(info.comp[i].h, info.comp[i].v) for i in range(info.ncomp)]
max_h, max_v = max(c[0] for c in comps), max(c[1] for c in comps)
mcus_x = -(-info.width // (8 * max_h))
bpm = sum(c[0] * c[1] for c in comps)
base = 0
for ci, (ch, cv) in enumerate(comps):
    bad = np.zeros(64, int); n = 0; first = None
    for (bx, by), blk in blocks[ci].items():
        idx = ((by // cv) * mcus_x + bx // ch) * bpm + base + (by % cv) * ch + bx % ch
        d = coefs[idx] != blk
        bad += d; n += 1
        if d.any() and first is None: first = (bx, by, coefs[idx].tolist(), blk.tolist())
    print("comp", ci, "blocks", n, "mismatch per zigzag index:", bad.tolist())
    if first: print(" first bad block", first)
    base += ch * cv
