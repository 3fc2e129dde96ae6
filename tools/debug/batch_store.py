import sys, os, glob, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
here = os.path.dirname(os.path.abspath(__file__))
files = sorted(glob.glob(os.path.join(here, sys.argv[1], "*.jpg")), key=lambda f: int(re.findall(r"(\d+)\.jpg", f)[0]))
victim = int(sys.argv[2])
datas = [open(f, "rb").read() for f in files]
info, blocks, _ = po.decode_progressive_store(datas[victim])
ref = po.decode_8bit(datas[victim])[0]
for rnd in range(3):
    b = jl.Batch().upload(datas).decode().sync()
    co = b.coefficients(victim)
    out = np.asarray(b.output(victim))
    bad = [(bx, by) for (bx, by), blk in sorted(blocks[0].items()) if not np.array_equal(co[by * 20 + bx], blk)]
    print("round", rnd, "status", b.result(victim).status, "bad coef blocks", len(bad), bad[:8], "pixel diff", int((out != ref).sum()))
    if bad:
        bx, by = bad[0]
        k = np.argwhere(co[by * 20 + bx] != blocks[0][(bx, by)]).ravel()
        print("   idx", k[:16], "ref", blocks[0][(bx, by)][k[:8]], "got", co[by * 20 + bx][k[:8]])
    b.close()
