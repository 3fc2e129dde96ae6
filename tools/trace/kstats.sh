#!/bin/bash
# Per-kernel statistics (timeout 600 rocprofv3 --kernel-trace --stats) of one bench.py workload:  kstats.sh <tag> <workload> [images] [bench args...]
# Environment switches (JPGPU_*) are inherited.  Output: gpurun_out/kstats_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WL=$2; IMAGES=${3:-0}; shift; shift; [ $# -gt 0 ] && shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ks_$TAG
ARGS="--workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-ingest"
[ "$IMAGES" != "0" ] && ARGS="$ARGS --images $IMAGES"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -- python3 $R/bench.py $ARGS "$@" > /tmp/ks_$TAG.log 2>&1
mkdir -p $R/gpurun_out
CSV=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -z "$CSV" ] && { echo "no kernel_stats.csv:"; tail -5 /tmp/ks_$TAG.log; exit 1; }
{ tail -1 /tmp/ks_$TAG.log | cut -c1-300; cat "$CSV" | python3 -c "
import sys,csv
rows=list(csv.reader(sys.stdin))
print('%-86s %8s %12s %10s %7s' % ('kernel','calls','total_ms','avg_ms','%'))
for r in rows[1:]:
    print('%-86s %8s %12.3f %10.4f %7s' % (r[0][:86], r[1], float(r[2])/1e6, float(r[3])/1e6, r[4]))
"; } > $R/gpurun_out/kstats_$TAG.txt
cat $R/gpurun_out/kstats_$TAG.txt
