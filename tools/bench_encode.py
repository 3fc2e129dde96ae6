#!/usr/bin/env python3
"""tools/bench_encode.py -- Mpixels/s of the GPU encoder on synthetic 4K RGB images, next to the oracle
restatement on the host cores.  Not the headline benchmark (that is bench.py); prints one JSON line.

    python tools/bench_encode.py [--images 64] [--steps 5] [--width 3840 --height 2160] [--quality 75]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def image(w, h, seed):
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 4)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([128 + 70 * np.sin(x / 37 + ph[0]) * np.cos(y / 53 + ph[1]), 128 + 60 * np.cos(x / 91 + y / 29 + ph[2]),
                    128 + 90 * np.sin((x + y) / 67 + ph[3])], axis=-1)
    img += rng.normal(0, 8, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--quality", type=int, default=75)
    ap.add_argument("--dri", type=int, default=0, help="restart interval in MCUs (0 = none, the reference encoder's only mode)")
    args = ap.parse_args()
    import jpeglibrary_amd as jl
    from oracle import pyoracle as po

    distinct = min(args.images, 16)
    with ThreadPoolExecutor(16) as ex:
        base = list(ex.map(lambda s: image(args.width, args.height, s), range(distinct)))
    imgs = [base[i % distinct] for i in range(args.images)]
    b = jl.EncodeBatch().upload(imgs, (2, 2), args.quality, rgb=True, restart_interval=args.dri)
    b.encode()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.encode()
    dt = (time.perf_counter() - t0) / args.steps
    px = args.images * args.width * args.height
    out0 = b.output(0)
    ref = po.encode_8bit(po.rgb_to_ycbcr8(imgs[0]), 2, 2, args.quality, restart_interval=args.dri)
    t1 = time.perf_counter()
    n_cpu = min(distinct, 4)
    for i in range(n_cpu):
        po.encode_8bit(po.rgb_to_ycbcr8(imgs[i]), 2, 2, args.quality)
    cpu = n_cpu * args.width * args.height / (time.perf_counter() - t1) / 1e6
    print(json.dumps({"metric": "Mpixels/s encoded (RGB 4:2:0 baseline, standard tables)", "value": round(px / dt / 1e6, 1),
                      "ms_per_step": round(dt * 1e3, 2), "images": args.images, "restart_interval": args.dri, "bytes_per_image": len(out0),
                      "byte_exact_vs_oracle": out0 == ref, "cpu_oracle_single_core_Mpx_s": round(cpu, 1)}))


if __name__ == "__main__":
    main()
