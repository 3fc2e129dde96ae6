#!/bin/bash
# tools/trace/evidence.sh TAG -- the measurements README / DESIGN quote, with the binary in the tree, each as bench JSON +
# rocprofv3 --kernel-trace --stats CSV (separate runs: profiled passes run slower than un-profiled ones).
# Writes gpurun_out/evidence_TAG/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/evidence_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  python3 $R/bench.py "$@" --no-cpu-baseline --no-ingest > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  rm -rf /tmp/ev_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_$name -- python3 $R/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > $OUT/prof_$name.log 2>&1
  f=$(find /tmp/ev_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$name.csv
  echo "== $name"; cut -c1-400 $OUT/bench_$name.json; [ -n "$f" ] && head -8 $f | cut -c1-160
}
run 4k_dri0 --workload 4k_dri0
run 1080p_q90 --workload 1080p_q90
run planar_u8 --format planar_u8
run rgb_u8 --format rgb_u8
run rgba_u8 --format rgba_u8
sha256sum $R/jpeglibrary_amd/libjpgpu.so > $OUT/library.sha256
