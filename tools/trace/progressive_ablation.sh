#!/bin/bash
# tools/trace/progressive_ablation.sh -- what the parts of the AC-refinement parser cost, by leaving them out: diagnostic builds
# of libjpgpu.so in /tmp (the tree's library is not touched), each timed scan by scan (JPGPU_PROG_BY_SCAN) on 64 frames.
# The builds decode WRONGLY by design; only the two luma refinement launches' durations are read.
#   baseline | no correction bits (-DJPGPU_PS_ABLATE_EPILOGUE) | no symbol loop (-DJPGPU_PS_ABLATE_LOOP: every block skipped
#   as if inside an end-of-band run: what is left is the per-block frame around the parse)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-64}
for v in "" "-DJPGPU_PS_ABLATE_EPILOGUE" "-DJPGPU_PS_ABLATE_LOOP" ${EXTRA_VARIANTS:-}; do
  rm -rf /tmp/abl && cp -r $R /tmp/abl && rm -rf /tmp/abl/gpurun_out
  ( cd /tmp/abl/jpeglibrary_amd/csrc && touch kernels.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math $v" > /tmp/abl/build.log 2>&1 ) || { tail -5 /tmp/abl/build.log; exit 1; }
  echo "== build: ${v:-baseline}"
  GRAFT_REPO_ROOT=/tmp/abl timeout 300 bash /tmp/abl/tools/trace/progressive_by_scan.sh $N /tmp/abl/by_scan.txt | grep -E "scan  6|scan 10|total"
done
