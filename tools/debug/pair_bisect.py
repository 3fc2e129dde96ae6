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
lo, hi = 0, len(datas)
def bad(sel):
    for _ in range(3):
        outs, _ = jl.decode_batch([datas[i] for i in sel], jl.FMT_INTERLEAVED_U8)
        if not np.array_equal(np.asarray(outs[sel.index(victim)]), ref):
            return True
    return False
others = [i for i in range(len(datas)) if i != victim]
print("all:", bad(sorted(others + [victim])))
cand = others
while len(cand) > 1:
    half = cand[:len(cand) // 2]
    if bad(sorted(half + [victim])):
        cand = half
    else:
        rest = cand[len(cand) // 2:]
        if bad(sorted(rest + [victim])):
            cand = rest
        else:
            print("needs files from both halves; stop at", len(cand), cand[:20])
            break
print("culprit candidates", cand[:10])
if len(cand) == 1:
    c = cand[0]
    d = datas[c]
    print("pair bad:", bad(sorted([c, victim])), os.path.basename(files[c]), len(d))
    sof = d.find(b"\xff\xc2"); print(" SOF", d[sof:sof + 19].hex(" "))
    i = 0
    while True:
        i = d.find(b"\xff\xda", i)
        if i < 0: break
        ln = (d[i + 2] << 8) | d[i + 3]; print(" SOS", i, d[i + 4:i + 2 + ln].hex(" ")); i += 2
    try:
        po.decode_8bit(d); print(" oracle OK")
    except po.OracleError as e:
        print(" oracle", e)
    r = jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)[1][0]
    print(" gpu status", r.status, r.detail)
