#!/bin/bash
# tools/trace/progressive_cycles.sh -- where the last refinement scans of a progressive batch spend their cycles: a diagnostic
# build of libjpgpu.so (-DJPGPU_PS_PROFILE, in /tmp, the tree's library is not touched) and one 64-frame decode.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/psprof && cp -r $R /tmp/psprof && cd /tmp/psprof/jpeglibrary_amd/csrc
touch k*.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math -DJPGPU_PS_PROFILE=${PSP:-1} ${EXTRA:-}" > /tmp/psprof/build.log 2>&1 || { tail -5 /tmp/psprof/build.log; exit 1; }
cd /tmp/psprof && python3 - <<'PY'
import ctypes as C, sys
sys.path.insert(0, "/tmp/psprof")
import jpeglibrary_amd as jl
from jpeglibrary_amd import _capi
from bench import progressive_batch
files = progressive_batch(16, 3840, 2160, 75, 1, 16)
files = [files[i % 16] for i in range(64)]
b = jl.Batch().upload(files).decode().sync()
for rep in range(int(__import__("os").environ.get("REPS", "1"))):
    print("failing:", [(i, b.result(i).status, b.result(i).detail) for i in range(64) if b.result(i).status != 0][:5], "fallbacks", b.progressive_fallbacks(), flush=True)
    b.decode().sync()
out = (C.c_ulonglong * 16)()
lib = C.CDLL(_capi.LIB_PATH)
lib.jpgpu_debug_ps_profile(out, 1)
b.decode().sync()
lib.jpgpu_debug_ps_profile(out, 0)
n, units, wait, stage, blocks, total = [out[i] for i in range(6)]
print("symbols (refinement fast path, all scans)", out[6], "window refreshes (all stream scans)", out[7])
print(f"inside the parse-only block decoder, cycles per block: prologue {out[8]/units:.0f}, prologue + symbol loop {out[9]/units:.0f}, window rebuilds {out[11]/units:.0f}, epilogue {out[10]/units:.0f}; loop exits per block {out[12]/units:.2f}, loop trips per block (PSP=2 only) {out[13]/units:.2f}")
print(f"last refinement scans: {n} streams, {units} blocks; cycles per block: follow/wait {wait/units:.0f}, staging {stage/units:.0f}, block loop {blocks/units:.0f}, all {total/units:.0f}")
PY
