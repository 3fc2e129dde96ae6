"""Per-scan session vs whole-file batch vs the restatement on the corrupted progressive corpus (debugging aid)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpeglibrary_amd as jl  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_partial_flush_gpu import _corrupted_progressive  # noqa: E402
from test_per_scan_gpu import Walk  # noqa: E402

if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    files = [open(sys.argv[1], "rb").read()]
else:
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    files = _corrupted_progressive(int(sys.argv[2]) if len(sys.argv) > 2 else 30, seed)
for k, data in enumerate(files):
    try:
        px, info, err = po.decode_8bit_partial(data)
    except po.OracleError:
        continue
    w = Walk(data)
    st = {"dec": None, "err": None, "fh": None, "n": 0, "scans": []}

    def on_frame(marker, fh):
        st["fh"] = fh
        st["dec"] = jl.JpegGpuProgressiveScanDecoder(fh)

    def on_scan(entropy, sh):
        if st["err"] is not None:
            return 0
        st["scans"].append((sh.NumberOfComponents, sh.StartOfSpectralSelection, sh.EndOfSpectralSelection, sh.SuccessiveApproximationBitPositionHigh, sh.SuccessiveApproximationBitPositionLow, w.dri))
        try:
            r = st["dec"].ProcessScan(entropy, sh, w.quantization_tables(), w.huffman_tables(), w.dri)
            print("   scan", len(st["scans"]) - 1, st["scans"][-1], "OK", len(entropy))
            return r
        except jl.JpegError as e:
            print("   scan", len(st["scans"]) - 1, st["scans"][-1], type(e).__name__, e)
            st["err"] = e
            return 0
    try:
        w.run(on_frame, on_scan)
    except Exception as e:
        print(k, "walk", e)
        continue
    fh = st["fh"]
    out = st["dec"].Dispose(fmt=jl.FMT_INTERLEAVED_U8).reshape(fh.NumberOfLines, fh.SamplesPerLine, fh.NumberOfComponents)
    st["dec"].close()
    whole, res = jl.decode_batch([data])
    d_or = int((out != px).sum()) if out.shape == px.shape else -1
    d_wh = int((np.asarray(whole[0]) != px).sum()) if whole[0] is not None and np.asarray(whole[0]).shape == px.shape else -1
    if out.shape == px.shape and d_or:
        bad = np.argwhere((out != px).any(axis=2))
        print("  first / last differing pixel:", bad[0].tolist(), bad[-1].tolist(), "rows", sorted(set(bad[:, 0].tolist()))[:12])
    print(k, "oracle:", None if err is None else err.kind, "session:", None if st["err"] is None else type(st["err"]).__name__, "failing scan", len(st["scans"]) - 1,
          st["scans"][-1] if st["scans"] else None, "| session != oracle:", d_or, "whole-file != oracle:", d_wh)
