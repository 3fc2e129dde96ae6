import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from oracle import pyoracle as po
d = open(os.path.join(os.path.dirname(__file__), sys.argv[1]), "rb").read()
for strip in (True, False):
    ref = po.optimize(d, strip)
    b = jl.OptimizeBatch().upload([d], strip).run()
    got = b.output(0)
    b.close()
    print("strip", strip, "len", len(got), len(ref), "equal", got == ref)
    if got != ref:
        n = min(len(got), len(ref))
        k = next((i for i in range(n) if got[i] != ref[i]), n)
        print(" first diff at", k, "got", got[max(0, k - 8):k + 16].hex(" "), "\n ref", ref[max(0, k - 8):k + 16].hex(" "))
        def markers(x):
            return [(i, hex(x[i + 1])) for i in range(len(x) - 1) if x[i] == 0xFF and x[i + 1] not in (0, 0xFF)]
        print(" got markers", markers(got)[:30])
        print(" ref markers", markers(ref)[:30])
