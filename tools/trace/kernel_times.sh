#!/bin/bash
# tools/trace/kernel_times.sh [bench args] -- per-kernel average durations of one bench.py run (rocprofv3 --kernel-trace --stats),
# printed as "calls avg_us name".  Scratch output under /tmp.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f'{int(r["Calls"]):5d} {float(r["AverageNs"]) / 1e3:10.1f} us  {r["Name"][:90]}')
PY
