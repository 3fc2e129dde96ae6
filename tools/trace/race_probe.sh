#!/bin/bash
# tools/trace/race_probe.sh -- the oversubscribed forced-pipeline probe (1024 x 4K progressive frames, 14 workgroups per CU) run
# several times on diagnostic builds (VARIANTS="-DA=1;-DB=2 -DC=3": one build per ;-separated entry, in /tmp); prints each pass's failures.
R=${GRAFT_REPO_ROOT:-/root/repo}
export JPGPU_PS_RING=4096 JPGPU_PS_CHUNK=32 JPGPU_PROG_FORCE_PIPELINE=1
IFS=";" read -ra VS <<< "${VARIANTS:--DJPGPU_BASELINE=1}"
for v in "${VS[@]}"; do
  rm -rf /tmp/rp && cp -r $R /tmp/rp && rm -rf /tmp/rp/gpurun_out
  ( cd /tmp/rp/jpeglibrary_amd/csrc && touch k*.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math $v" > /tmp/rp/build.log 2>&1 ) || { tail -5 /tmp/rp/build.log; exit 1; }
  echo "== build: $v"
  for rep in 1 2 3 4; do ( cd /tmp/rp && timeout 300 python tools/trace/progressive_oversubscribed.py 1024 2>&1 | grep "^n=1024" | cut -c46-140 ); done
done
