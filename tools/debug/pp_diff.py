"""Progressive file: GPU coefficient store vs the restatement's, whole file and per scan prefix (first diverging scan)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeglibrary_amd as jl
from oracle import pyoracle as po

d = open(os.path.join(os.path.dirname(__file__), sys.argv[1]), "rb").read()


def store_diff(data, tag):
    try:
        info, blocks, _ = po.decode_progressive_store(data)
    except po.OracleError as e:
        print(tag, "oracle:", e)
        blocks = None
    b = jl.Batch().upload([data]).decode().sync()
    r = b.result(0)
    print(tag, "gpu status", r.status, r.detail, "interval", r.error_interval)
    if blocks is None or r.status != 0:
        return None
    coefs = b.coefficients(0)
    comps = [(info.comp[i].h, info.comp[i].v) for i in range(info.ncomp)]
    max_h, max_v = max(c[0] for c in comps), max(c[1] for c in comps)
    mcus_x = -(-info.width // (8 * max_h))
    bpm = sum(c[0] * c[1] for c in comps)
    base, bad = 0, []
    for ci, (ch, cv) in enumerate(comps):
        for (bx, by), blk in sorted(blocks[ci].items(), key=lambda kv: (kv[0][1], kv[0][0])):
            idx = ((by // cv) * mcus_x + bx // ch) * bpm + base + (by % cv) * ch + bx % ch
            if not np.array_equal(coefs[idx], blk):
                bad.append((ci, bx, by, idx))
        base += ch * cv
    print(tag, "bad blocks", len(bad), bad[:6])
    for ci, bx, by, idx in bad[:2]:
        k = np.argwhere(coefs[idx] != blocks[ci][(bx, by)]).ravel()
        print("   comp", ci, "block", (bx, by), "coef idx", k[:12], "ref", blocks[ci][(bx, by)][k[:8]], "got", coefs[idx][k[:8]])
    return len(bad)


store_diff(d, "whole")
sos = [i for i in range(len(d) - 1) if d[i] == 0xFF and d[i + 1] == 0xDA]
for n in range(1, len(sos)):
    # the file up to (not including) the DHT/SOS group of scan n, closed with EOI
    cut = sos[n]
    j = d.rfind(b"\xff\xc4", sos[n - 1], sos[n])
    if j > 0:
        cut = j
    nb = store_diff(d[:cut] + b"\xff\xd9", f"first {n} scans")
    if nb:
        print("   scan header", d[sos[n - 1]:sos[n - 1] + 14].hex(" "))
        break
