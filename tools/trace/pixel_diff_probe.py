"""Where the GPU's pixels differ from the checker's for given files (debugging aid): python3 tools/trace/pixel_diff_probe.py a.jpg ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from oracle import pyoracle as po

for f in sys.argv[1:]:
    d = open(f, "rb").read()
    try:
        ref = po.decode_8bit(d)[0]
        err = None
    except po.OracleError as e:
        ref, err = None, e.kind
    outs, results = jl.decode_batch([d], jl.FMT_INTERLEAVED_U8)
    out = None if outs[0] is None else np.asarray(outs[0])
    print(os.path.basename(f), "oracle", err or "OK", "gpu status", results[0].status, "detail", results[0].detail)
    if ref is None or out is None:
        continue
    ref = np.asarray(ref)
    if ref.ndim == 2:
        ref = ref[..., None]
    out = out.reshape(ref.shape)
    for c in range(ref.shape[2]):
        bad = np.argwhere(out[..., c] != ref[..., c])
        if len(bad):
            ys, xs = bad[:, 0], bad[:, 1]
            print("  component", c, "differs at", len(bad), "pixels; y", ys.min(), "..", ys.max(), "x", xs.min(), "..", xs.max(),
                  "first", bad[0], "gpu", out[bad[0][0], bad[0][1], c], "ref", ref[bad[0][0], bad[0][1], c],
                  "blocks", sorted({(int(y) // 8, int(x) // 8) for y, x in bad[:4000]})[:6])
        else:
            print("  component", c, "equal")
