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
def bad(sel):
    for _ in range(2):
        outs, _ = jl.decode_batch([datas[i] for i in sel], jl.FMT_INTERLEAVED_U8)
        if not np.array_equal(np.asarray(outs[sel.index(victim)]), ref):
            return True
    return False
sel = sorted(set(range(128)) | {victim})
assert bad(sel)
for i in list(sel):
    if i == victim:
        continue
    t = [x for x in sel if x != i]
    if bad(t):
        sel = t
print("minimal set", sel)
for i in sel:
    d = datas[i]
    sof = d.find(b"\xff\xc2")
    dri = d.find(b"\xff\xdd")
    nsos = d.count(b"\xff\xda")
    ndht = d.count(b"\xff\xc4")
    try:
        po.decode_8bit(d); o = "OK"
    except po.OracleError as e:
        o = str(e)[:60]
    r = jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)[1][0]
    print(i, len(d), "SOF", d[sof:sof + 19].hex(" "), "DRI", d[dri + 4:dri + 6].hex() if dri >= 0 else None, "sos", nsos, "dht", ndht, "| oracle", o, "| gpu", r.status, r.detail)
