#!/bin/bash
# A/B of the encoder's one-pass entropy stage without its ticket (round 6) on ONE box: the tree as it is against -DJPGPU_ENC_TICKETS.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
for flags in none -DJPGPU_ENC_TICKETS none -DJPGPU_ENC_TICKETS; do
  [ "$flags" = none ] && flags=""
  echo "== build flags: [$flags]"
  bash tools/trace/ab_build.sh "$flags" bash -c 'python3 tools/bench_encode.py --images 256 2>/dev/null | cut -c1-420; python3 tools/bench_encode.py --workload het_8192 2>/dev/null | cut -c1-420; python3 tools/bench_encode.py --workload het_8192 --optimize-coding 2>/dev/null | cut -c1-420'
done
