import sys, os, glob, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
here = os.path.dirname(os.path.abspath(__file__))
files = sorted(glob.glob(os.path.join(here, sys.argv[1], "*.jpg")), key=lambda f: int(re.findall(r"(\d+)\.jpg", f)[0]))
datas = [open(f, "rb").read() for f in files]
v = 19
ref = po.decode_8bit(datas[v])[0]
def run(lst, where=0):
    outs, res = jl.decode_batch(lst, jl.FMT_INTERLEAVED_U8)
    return int((np.asarray(outs[where]) != ref).sum())
big = datas[81]
for k in (1, 2, 4, 8, 16, 32, 64):
    print("victim first +", k, "x file81:", run([datas[v]] + [big] * k), " | victim last:", run([big] * k + [datas[v]], k))
small = datas[0]
for k in (64, 128, 256, 512):
    print("victim +", k, "x file0 (", len(small), "B ):", run([datas[v]] + [small] * k))
cum = np.cumsum([len(d) for d in datas])
print("cumulative bytes at 80, 81:", cum[80], cum[81])
