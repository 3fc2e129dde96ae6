"""Hunt for a failing-writer mismatch over many seeds (debugging aid): tools/trace/failing_writer_hunt.py <sub> <dri> <non> <n_seeds>"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpeglibrary_amd as jl  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_failing_writer_gpu import NAMES, _corrupt  # noqa: E402
from tools import jpegsynth  # noqa: E402

sub, dri, non, n = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1", int(sys.argv[4])
os.makedirs(os.path.join(ROOT, "gpurun_out", "hunt"), exist_ok=True)
bad = 0
for seed in range(100, 100 + n):
    rng = np.random.default_rng(zlib.crc32(repr((sub, dri, non, seed)).encode()))
    base = [jpegsynth.encode(int(rng.integers(40, 300)), int(rng.integers(40, 220)), sub, int(rng.integers(30, 95)), dri, seed=int(rng.integers(1, 1 << 20)),
                             noninterleaved=non) for _ in range(6)]
    files = [_corrupt(base[k % len(base)], rng) for k in range(60)] + base[:2]
    b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8).decode().sync()
    for i, f in enumerate(files):
        try:
            px, _, err = po.decode_8bit_partial(f)
        except po.OracleError as e:
            if NAMES.get(b.result(i).status) != e.kind:
                print("seed", seed, "file", i, "identify status differs", e.kind, b.result(i).status)
            continue
        res = b.result(i)
        kind = "OK" if err is None else err.kind
        ok = NAMES.get(res.status) == kind
        if ok and b.image_info(i).status == 0:
            got = b.output(i)
            ok = np.array_equal(got, px)
        if not ok:
            bad += 1
            alone = jl.Batch().upload([f], jl.FMT_INTERLEAVED_U8).decode().sync()
            ok_alone = NAMES.get(alone.result(0).status) == kind and (alone.image_info(0).status != 0 or np.array_equal(alone.output(0), px))
            print("seed", seed, "file", i, "oracle", kind, "gpu", NAMES.get(res.status), res.detail, "alone ok:", ok_alone, "len", len(f))
            open(os.path.join(ROOT, "gpurun_out", "hunt", f"{sub}_{dri}_{int(non)}_s{seed}_f{i}.jpg"), "wb").write(f)
            alone.close()
    b.close()
print("mismatches:", bad)
