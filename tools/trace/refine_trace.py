#!/usr/bin/env python3
"""Where does a frame of the forced, oversubscribed progressive launch go wrong -- in what a luma refinement READS or in what it makes
of it?  Needs a diagnostic build (-DJPGPU_PS_TRACE, see tools/trace/refine_trace.sh): every luma AC refinement block of the last
`images` frames records which of its 64 coefficients it found non-zero, its bit position, the bits it consumed and the end-of-band
run behind it.  The batch is n copies of 16 sources, so a failing frame is held against a good copy of its source, block by
block, scan by scan."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeglibrary_amd as jl  # noqa: E402
from jpeglibrary_amd import _capi  # noqa: E402
from bench import progressive_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
images = int(sys.argv[2]) if len(sys.argv) > 2 else 128
units = 129600
distinct = 16
src = progressive_batch(distinct, 3840, 2160, 75, 1, 16)
files = [src[i % distinct] for i in range(n)]
import torch  # noqa: E402  (device memory for the trace: the library and torch share one HIP runtime)

lib = C.CDLL(_capi.LIB_PATH)
b = jl.Batch().upload(files, jl.FMT_INTERLEAVED_U8)
nbytes = images * 2 * units * 16
trace = torch.zeros(nbytes // 4, dtype=torch.int32, device="cuda:0")
buf = C.c_void_p(trace.data_ptr())
first = n - images
lib.jpgpu_debug_ps_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
for attempt in range(int(os.environ.get("ATTEMPTS", "6"))):
    trace.zero_()
    torch.cuda.synchronize()
    assert lib.jpgpu_debug_ps_trace(buf, first, images, units) == 0
    b.decode().sync()
    bad = [i for i in range(n) if b.result(i).status != 0]
    print(f"attempt {attempt}: failing frames {bad} fallbacks {b.progressive_fallbacks()}", flush=True)
    if not bad:
        continue
    host = trace.cpu().numpy().view(np.uint32)
    t = host.reshape(images, 2, units, 4)
    for i in bad:
        if i < first:
            continue
        good = next((j for j in range(i - distinct, first - 1, -distinct) if j not in bad), None)
        if good is None:
            good = next((j for j in range(i + distinct, n, distinct) if j not in bad), None)
        if good is None:
            print(f"  frame {i}: no good copy of its source in the traced range")
            continue
        for kind, name in ((0, "Y refinement Ah=2 Al=1"), (1, "Y refinement Ah=1 Al=0")):
            a, g = t[i - first, kind], t[good - first, kind]
            ran = int(np.count_nonzero(a[:, 0] | a[:, 1] | a[:, 2]))

            def f(x):
                return int(x[0]) if len(x) else None

            d_in = f(np.flatnonzero((a[:, 0] != g[:, 0]) | (a[:, 1] != g[:, 1])))
            d_out = None
            d_pos = f(np.flatnonzero(a[:, 2] != g[:, 2]))
            d_eob = f(np.flatnonzero(a[:, 3] != g[:, 3]))
            print(f"  frame {i} vs {good}, {name}: blocks traced {ran}; first block whose non-zero HISTORY differs {d_in}, "
                  f"whose bit position in front of it differs {d_pos}, whose consumed bits / end-of-band run differ {d_eob}", flush=True)
            k = min([x for x in (d_in, d_out, d_pos, d_eob) if x is not None], default=None)
            if k is not None:
                for u in range(max(0, k - 2), min(units, k + 3)):
                    print(f"      block {u} (block row {u // 480}, x {u % 480}): failing nz {a[u, 1]:08x}{a[u, 0]:08x} pos {a[u, 2]} bits {a[u, 3] & 0xFFFF} eob {a[u, 3] >> 16} | "
                          f"good nz {g[u, 1]:08x}{g[u, 0]:08x} pos {g[u, 2]} bits {g[u, 3] & 0xFFFF} eob {g[u, 3] >> 16}")
    break
b.close()
