#!/bin/bash
# tools/trace/progressive_ablation.sh -- what the parts of the AC-refinement parser cost, by leaving them out: diagnostic builds
# of libjpgpu.so in /tmp (the tree's library is not touched), each timed scan by scan (JPGPU_PROG_BY_SCAN) on 64 frames.
# The builds decode WRONGLY by design; only the two luma refinement launches' durations are read.
#   VARIANTS="-DA=1;-DB=2 -DC=3" (one build per ;-separated entry); the ablation switches live in the fourth form of the block decoder
#   (-DJPGPU_PS_EARLIER_FORMS -DJPGPU_PS_REFINE4 with -DJPGPU_PS_ABLATE_EPILOGUE: no correction bits, -DJPGPU_PS_ABLATE_LOOP: no symbol loop);
#   -DJPGPU_PS_PUBLISH_EVERY=n: progress published every n units.  PIPELINED=1: time the pipelined launch (bench.py) instead.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-64}
IFS=";" read -ra VS <<< "${VARIANTS:--DJPGPU_BASELINE=1;-DJPGPU_PS_EARLIER_FORMS -DJPGPU_PS_REFINE4;-DJPGPU_PS_EARLIER_FORMS -DJPGPU_PS_REFINE4 -DJPGPU_PS_ABLATE_EPILOGUE}"
for v in "${VS[@]}"; do
  rm -rf /tmp/abl && cp -r $R /tmp/abl && rm -rf /tmp/abl/gpurun_out
  ( cd /tmp/abl/jpeglibrary_amd/csrc && touch k*.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math $v" > /tmp/abl/build.log 2>&1 ) || { tail -5 /tmp/abl/build.log; exit 1; }
  echo "== build: $v"
  if [ -n "${PIPELINED:-}" ]; then
    ( cd /tmp/abl && timeout 300 python3 bench.py --workload 4k_progressive --images $N --distinct 64 --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --no-planar-pass 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined launch, ms per step:', d['ms_per_step'])" )
  else
    GRAFT_REPO_ROOT=/tmp/abl timeout 300 bash /tmp/abl/tools/trace/progressive_by_scan.sh $N /tmp/abl/by_scan.txt | grep -E "scan  6|scan 10|total"
  fi
done
