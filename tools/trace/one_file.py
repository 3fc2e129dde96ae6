#!/usr/bin/env python3
"""Decode one file on the GPU (batch API) and print status / detail next to the oracle's verdict (debugging aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

d = open(sys.argv[1], "rb").read()
try:
    ref = po.decode_8bit(d)[0]
    print("oracle: OK", ref.shape)
except po.OracleError as e:
    ref = None
    print("oracle:", e.kind, e.message)
b = jl.Batch().upload([d]).decode().sync()
r = b.result(0)
print("gpu: status", r.status, "detail", r.detail, "interval", r.error_interval, "decoded_mcus", r.decoded_mcus, "terminator", hex(r.terminator), "rounds", b.subseq_rounds(),
      "ingest", b.ingest_stats()["n_header_only"])
if ref is not None and r.status == 0:
    print("samples equal:", np.array_equal(b.output(0), ref))
