import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import jpeglibrary_amd as jl
from golden_util import read_jpeg
data = read_jpeg("progress.jpg")
# list scans
i = 2; scans = []
while i < len(data) - 1:
    if data[i] == 0xFF and data[i+1] == 0xDA:
        L = (data[i+2] << 8) | data[i+3]; ns = data[i+4]
        comps = [data[i+5+2*k] for k in range(ns)]
        ss, se, ahal = data[i+5+2*ns], data[i+6+2*ns], data[i+7+2*ns]
        scans.append((comps, ss, se, ahal >> 4, ahal & 15)); i += 2 + L
    else:
        i += 1
print(scans)
prev = None
for n in range(1, len(scans) + 1):
    os.environ["JPGPU_DEBUG_MAX_PROGRESSIVE_SCANS"] = str(n)
    b = jl.Batch().upload([data]).decode().sync()
    c = b.coefficients(0).astype(np.int32)
    nz = (c != 0).sum(axis=0)
    ch = None if prev is None else int((c != prev).sum())
    print(n, scans[n-1], "status", b.result(0).status, b.result(0).detail, "nonzero dc/ac", int(nz[0]), int(nz[1:].sum()), "changed", ch)
    prev = c
