#!/bin/bash
# Kernel timeline of the LAST step of one bench.py run (rocprofv3 --kernel-trace): start offset, duration, name.
#   timeline.sh <tag> <workload> <images> [bench args...]      -> gpurun_out/timeline_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WL=$2; IMAGES=$3; shift; shift; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -- python3 $R/bench.py --workload $WL --images $IMAGES --steps 2 --warmup 2 --no-ingest --no-cpu-baseline --no-planar-pass --no-configs "$@" > /tmp/tl_$TAG.log 2>&1
mkdir -p $R/gpurun_out
python3 - "$TAG" > $R/gpurun_out/timeline_$TAG.txt <<'PY'
import csv, glob, sys
fs = glob.glob("/tmp/tl_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)
if not fs:
    print("no kernel trace"); sys.exit(1)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "marker_count" in r["Kernel_Name"]]
first = idx[-1] if idx else 0
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
for r in rows[first:]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f ms  +%7.3f  gap %7.3f  %s" % ((a - t0) / 1e6, (b - a) / 1e6, (a - prev_end) / 1e6, r["Kernel_Name"][:70]))
    prev_end = max(prev_end, b)
print("total %9.3f ms" % ((prev_end - t0) / 1e6))
PY
cat $R/gpurun_out/timeline_$TAG.txt
