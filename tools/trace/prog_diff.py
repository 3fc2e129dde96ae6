"""Debug aid: decode one progressive file on the GPU and report where it differs from the oracle (per component)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl
from oracle import pyoracle as po

path = sys.argv[1]
data = open(path, "rb").read()
# list the scans
i = 2
while i < len(data):
    if data[i] != 0xFF:
        i += 1
        continue
    m = data[i + 1]
    if m in (0x00, 0xFF) or 0xD0 <= m <= 0xD9:
        i += 2
        continue
    ln = (data[i + 2] << 8) | data[i + 3]
    if m == 0xDA:
        ns = data[i + 4]
        comps = [data[i + 5 + 2 * k] for k in range(ns)]
        ss, se, a = data[i + 5 + 2 * ns], data[i + 6 + 2 * ns], data[i + 7 + 2 * ns]
        print(f"SOS comps={comps} ss={ss} se={se} ah={a >> 4} al={a & 15}")
    if m == 0xC2 or m == 0xC0:
        print("SOF", [(data[i + 10 + 3 * k], data[i + 11 + 3 * k] >> 4, data[i + 11 + 3 * k] & 15) for k in range(data[i + 9])])
    if m == 0xDD:
        print("DRI", (data[i + 4] << 8) | data[i + 5])
    i += 2 + ln
ref, _ = po.decode_8bit(data)
out = jl.decode_batch([data])[0]
out = np.asarray(out).reshape(ref.shape)
for c in range(ref.shape[-1]):
    bad = np.argwhere(out[..., c] != ref[..., c])
    print("component", c, "mismatches", len(bad), "first", bad[:3].tolist())
