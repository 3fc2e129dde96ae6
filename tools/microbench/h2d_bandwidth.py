#!/usr/bin/env python3
"""Host link and host CPU budget of the box: pinned / pageable H2D and D2H rates (torch copies, HIP events) and the CPU
quota the container is granted.  Context for bench.py's host.parse_upload_s and value_ingest_inclusive."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import host_cpu_budget  # noqa: E402


def rate(src, dst, reps=5):
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, src.numel() / (time.perf_counter() - t0) / 1e9)
    return round(best, 1)


def main():
    n = 1 << 30
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
    pageable = torch.empty(n, dtype=torch.uint8)
    pageable.fill_(1)
    pinned.fill_(2)
    out = {"h2d_pinned_GBps": rate(pinned, dev), "h2d_pageable_GBps": rate(pageable, dev), "d2h_pinned_GBps": rate(dev, pinned),
           "d2h_pageable_GBps": rate(dev, pageable)}
    t0 = time.perf_counter()
    pinned.copy_(pageable)
    out["host_memcpy_1thread_GBps"] = round(n / (time.perf_counter() - t0) / 1e9, 1)
    out["host_cpu_budget"] = host_cpu_budget()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
