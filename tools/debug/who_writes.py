import sys, os, glob, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
here = os.path.dirname(os.path.abspath(__file__))
files = sorted(glob.glob(os.path.join(here, sys.argv[1], "*.jpg")), key=lambda f: int(re.findall(r"(\d+)\.jpg", f)[0]))
datas = [open(f, "rb").read() for f in files]
v = datas[19]
big = datas[81]
gold = open(os.path.join(os.path.dirname(os.path.dirname(here)), "tests", "golden", "yellowcat_progressive_restart.jpg"), "rb").read()
gold2 = open(os.path.join(os.path.dirname(os.path.dirname(here)), "tests", "golden", "progress.jpg"), "rb").read()
a = jl.Batch().upload([v], jl.FMT_PLANAR_I16).run_entropy().sync()
ca = a.coefficients(0).copy()
def cmp(lst, where, tag):
    b = jl.Batch().upload(lst, jl.FMT_PLANAR_I16).run_entropy().sync()
    cb = b.coefficients(where).copy()
    bad = np.argwhere((ca != cb).any(axis=1)).ravel()
    print(tag, "blocks differing:", len(bad), list(bad[:12]))
    for blk in bad[:4]:
        k = np.argwhere(ca[blk] != cb[blk]).ravel()
        print("    block", blk, "idx", list(k[:8]), "alone", list(ca[blk][k[:8]]), "batch", list(cb[blk][k[:8]]))
    b.close()
cmp([big] * 4 + [v], 4, "4 x file81 + victim")
cmp([big] * 4 + [v], 4, "again")
cmp([gold] * 8 + [v], 8, "8 x yellowcat + victim")
cmp([gold2] * 16 + [v], 16, "16 x progress + victim")
cmp([v] * 8, 7, "8 x victim (last)")
cmp([v] * 300, 299, "300 x victim (last)")
cmp([v] * 300, 0, "300 x victim (first)")
