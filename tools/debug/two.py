import sys, os, glob, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
here = os.path.dirname(os.path.abspath(__file__))
v = open(os.path.join(here, "p220", "p19.jpg"), "rb").read()
a = jl.Batch().upload([v], jl.FMT_PLANAR_I16).run_entropy().sync()
ca = a.coefficients(0).copy()
for n in (2, 3, 5, 8):
    b = jl.Batch().upload([v] * n, jl.FMT_PLANAR_I16).run_entropy().sync()
    line = []
    for k in range(n):
        cb = b.coefficients(k)
        bad = np.argwhere((ca != cb).any(axis=1)).ravel()
        line.append(len(bad))
    print(n, "copies: differing blocks per copy", line)
    b.close()
