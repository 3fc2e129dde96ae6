import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po
here = os.path.dirname(os.path.abspath(__file__))
names = {0: "OK", 1: "InvalidDataException", 2: "InvalidOperationException", 3: "NotSupportedException"}
for f in sys.argv[1:]:
    d = open(os.path.join(here, f), "rb").read()
    try:
        ref, kind = po.decode_8bit(d)[0], "OK"
    except po.OracleError as e:
        ref, kind = None, e.kind
    outs, res = jl.decode_batch([d] * 8, jl.FMT_INTERLEAVED_U8)
    line = []
    for o, r in zip(outs, res):
        mine = names.get(r.status, r.status)
        if mine != kind:
            line.append(f"{mine}/{r.detail}")
        elif ref is not None:
            line.append(int((np.asarray(o) != ref).sum()))
        else:
            line.append("=")
    print(f, "oracle", kind, "->", line)
