import sys, os, glob, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
here = os.path.dirname(os.path.abspath(__file__))
files = sorted(glob.glob(os.path.join(here, sys.argv[1], "*.jpg")), key=lambda f: int(re.findall(r"(\d+)\.jpg", f)[0]))
victim = int(sys.argv[2])
datas = [open(f, "rb").read() for f in files]
ref = po.decode_8bit(datas[victim])[0]
def bad(sel, n=3):
    r = []
    for _ in range(n):
        outs, _ = jl.decode_batch([datas[i] for i in sel], jl.FMT_INTERLEAVED_U8)
        r.append(int((np.asarray(outs[sel.index(victim)]) != ref).sum()))
    return r
print("alone", bad([victim]))
print("x2 (victim twice + filler copies)", bad([victim] + [victim] * 0))
for k in (2, 4, 8, 16, 32, 64, 128, 249):
    sel = sorted(set(list(range(0, k)) + [victim]))
    print("first", k, "->", bad(sel))
for k in (20, 24, 28, 40, 48, 56):
    sel = sorted(set(list(range(0, k)) + [victim]))
    print("first", k, "->", bad(sel))
