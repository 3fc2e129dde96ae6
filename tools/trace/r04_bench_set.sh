#!/bin/bash
# tools/trace/r04_bench_set.sh OUTDIR [workload ...] -- bench line + rocprofv3 kernel statistics of each workload (default: the
# headline, DRI = 0, 1080p Q90), with the library of the tree the script runs in.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
WL=${@:-4k_dri4 4k_dri0 1080p_q90}
mkdir -p $OUT
cd $R
for w in $WL; do
  python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-planar-pass > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['stage_ms'])" $OUT/bench_$w.json $w
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$w && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --no-planar-pass > /tmp/ks_$w.log 2>&1
    cp $(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$w.csv
    python3 -c "import sys,csv; [print('   ', r[0][:60], r[1], r[3]) for r in list(csv.reader(open(sys.argv[1])))[1:7]]" $OUT/kernel_stats_$w.csv )
done
sha256sum jpeglibrary_amd/libjpgpu.so > $OUT/library.sha256
