#!/bin/bash
# tools/trace/refine_trace.sh -- diagnostic build (-DJPGPU_PS_TRACE, progress of the refinements published every 256 blocks: the
# setting that provokes the failures) in /tmp and tools/trace/refine_trace.py on the forced, oversubscribed launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/rt && cp -r $R /tmp/rt && rm -rf /tmp/rt/gpurun_out
( cd /tmp/rt/jpeglibrary_amd/csrc && touch k*.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math -DJPGPU_PS_TRACE -DJPGPU_PS_PUBLISH_REFINE=${REFINE:-256} ${EXTRA:-}" > /tmp/rt/build.log 2>&1 ) || { tail -5 /tmp/rt/build.log; exit 1; }
export JPGPU_PS_RING=4096 JPGPU_PS_CHUNK=32 JPGPU_PROG_FORCE_PIPELINE=1 JPGPU_PROG_SPIN_BUDGET=4194304
cd /tmp/rt && timeout 600 python tools/trace/refine_trace.py ${1:-1024} ${2:-128}
