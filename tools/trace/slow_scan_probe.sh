#!/bin/bash
# tools/trace/slow_scan_probe.sh -- the slow-scan test (JPGPU_DEBUG_DELAY_SCAN) several times over, on the shipped build and on
# diagnostic builds (VARIANTS="-DA=1;-DB=2", built in /tmp).
R=${GRAFT_REPO_ROOT:-/root/repo}
T=tests/test_gpu_parity.py::test_a_scan_inside_an_end_of_band_run_still_follows_its_producers
REPS=${REPS:-4}
echo "== shipped build"
for r in $(seq $REPS); do ( cd $R && timeout 300 python -m pytest $T -q -p no:cacheprovider 2>&1 | grep -E "^E  +AssertionError|passed|failed" | cut -c1-240 ); done
IFS=";" read -ra VS <<< "${VARIANTS:-}"
for v in "${VS[@]}"; do
  rm -rf /tmp/ssp && cp -r $R /tmp/ssp && rm -rf /tmp/ssp/gpurun_out
  ( cd /tmp/ssp/jpeglibrary_amd/csrc && touch k*.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math $v" > /tmp/ssp/build.log 2>&1 ) || { tail -5 /tmp/ssp/build.log; exit 1; }
  echo "== build: $v"
  for r in $(seq $REPS); do ( cd /tmp/ssp && timeout 300 python -m pytest $T -q -p no:cacheprovider 2>&1 | grep -E "^E  +AssertionError|passed|failed" | cut -c1-240 ); done
done
