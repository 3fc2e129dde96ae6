"""Where the fused E1 kernel's blocks differ from the checker's (debugging aid): python3 tools/trace/e1f_probe.py W H [rgb]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import jpeglibrary_amd as jl
from oracle import pyoracle as po
from test_gpu_parity import _enc_image

w, h = int(sys.argv[1]), int(sys.argv[2])
rgb = _enc_image(w, h, w + h)
ycc = po.rgb_to_ycbcr8(rgb)
ref, ref_coefs = po.encode_8bit(ycc, 2, 2, 75, want_coefficients=True)
src = rgb if len(sys.argv) > 3 else ycc
b = jl.EncodeBatch().upload([src], (2, 2), 75, rgb=len(sys.argv) > 3).encode() if len(sys.argv) > 3 else jl.EncodeBatch().upload([src], (2, 2), 75).encode()
got = np.asarray(b.coefficients(0)).reshape(-1, 64)
exp = np.asarray(ref_coefs).reshape(-1, 64)
bad = np.nonzero((got != exp).any(axis=1))[0]
mpl = (w + 15) // 16
print("blocks", len(exp), "differ", len(bad))
for blk in bad[:12]:
    m, bi = divmod(int(blk), 6)
    pos = np.nonzero(got[blk] != exp[blk])[0]
    print("mcu", m, "(x", m % mpl, "y", m // mpl, ") block", bi, "positions", pos[:10], "got", got[blk][pos[:6]], "exp", exp[blk][pos[:6]])
import collections
print(collections.Counter(int(x) % 6 for x in bad), collections.Counter((int(x) // 6) % mpl for x in bad).most_common(5))
