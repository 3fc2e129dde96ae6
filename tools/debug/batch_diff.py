import sys, os, glob, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
here = os.path.dirname(os.path.abspath(__file__))
files = sorted(glob.glob(os.path.join(here, sys.argv[1], "*.jpg")), key=lambda f: int(re.findall(r"(\d+)\.jpg", f)[0]))
datas = [open(f, "rb").read() for f in files]
refs = []
for d in datas:
    try:
        refs.append(po.decode_8bit(d)[0])
    except po.OracleError as e:
        refs.append(e.kind)
for rnd in range(3):
    outs, results = jl.decode_batch(datas, jl.FMT_INTERLEAVED_U8)
    bad = []
    for i, (ref, out, res) in enumerate(zip(refs, outs, results)):
        if isinstance(ref, str):
            continue
        if res.status != 0:
            continue
        o = np.asarray(out)
        if not np.array_equal(o, ref):
            diff = np.argwhere((o != ref).any(axis=2))
            bad.append((i, os.path.basename(files[i]), o.shape, len(diff), tuple(diff[0]), tuple(diff[-1])))
    print("round", rnd, "bad", bad)
