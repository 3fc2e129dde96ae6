import sys, os, glob, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
here = os.path.dirname(os.path.abspath(__file__))
files = sorted(glob.glob(os.path.join(here, sys.argv[1], "*.jpg")), key=lambda f: int(re.findall(r"(\d+)\.jpg", f)[0]))
sel = list(range(128))
datas = [open(files[i], "rb").read() for i in sel]
v = 19
def stores(tag):
    a = jl.Batch().upload([datas[v]], jl.FMT_PLANAR_I16).run_entropy().sync()
    ca = a.coefficients(0).copy()
    b = jl.Batch().upload(datas, jl.FMT_PLANAR_I16).run_entropy().sync()
    cb = b.coefficients(v).copy()
    bad = np.argwhere((ca != cb).any(axis=1)).ravel()
    print(tag, "blocks differing alone vs batch:", len(bad), bad[:10])
    if len(bad):
        k = np.argwhere(ca[bad[0]] != cb[bad[0]]).ravel()
        print("   block", bad[0], "idx", k[:10], "alone", ca[bad[0]][k[:8]], "batch", cb[bad[0]][k[:8]])
    a.close(); b.close()
stores("default")
os.environ["JPGPU_PROG_NO_PIPELINE"] = "1"
stores("no pipeline")
del os.environ["JPGPU_PROG_NO_PIPELINE"]
for n in range(1, 8):
    os.environ["JPGPU_DEBUG_MAX_PROGRESSIVE_SCANS"] = str(n)
    stores(f"levels {n}")
