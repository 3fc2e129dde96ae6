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
def bad(sel):
    outs, res = jl.decode_batch([datas[i] for i in sel], jl.FMT_INTERLEAVED_U8)
    return not np.array_equal(np.asarray(outs[sel.index(v)]), ref)
lo, hi = 64, 128
while hi - lo > 1:
    mid = (lo + hi) // 2
    if bad(sorted(set(range(mid)) | {v})): hi = mid
    else: lo = mid
print("breaks when file", hi - 1, "joins; files 0..", hi - 1)
def tables(d):
    t = set(); i = 0
    while True:
        i = d.find(b"\xff\xc4", i)
        if i < 0: break
        ln = (d[i + 2] << 8) | d[i + 3]
        t.add(bytes(d[i + 4:i + 2 + ln])); i += 2
    return t
allt = set()
for i in range(hi):
    allt |= tables(datas[i])
    if i in (hi - 2, hi - 1): print("after file", i, "distinct DHT payloads", len(allt))
d = datas[hi - 1]
sof = d.find(b"\xff\xc2"); print("file", hi - 1, len(d), "SOF", d[sof:sof + 19].hex(" "), "sos", d.count(b"\xff\xda"))
# is it that file, or the count?  replace it by a copy of file 0
print("with file", hi - 1, "dropped but", hi, "kept:", bad(sorted((set(range(hi + 1)) - {hi - 1}) | {v})))
print("only victim +", hi - 1, ":", bad(sorted({v, hi - 1})))
