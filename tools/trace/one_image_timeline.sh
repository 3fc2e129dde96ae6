#!/bin/bash
# Kernel timeline of ONE decode of a one-image batch (the reference's call pattern): tools/trace/one_image_timeline.sh WORKLOAD
# WORKLOAD = 512_444 | 1080p_q90 | 4k_dri4 | 4k_dri0.  The last decode of five, times relative to its first kernel.
W=${1:-512_444}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp1
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp1 -- python3 $GRAFT_REPO_ROOT/tools/trace/dri0_decodes.py $W 1 3 > /tmp/pp1.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp1/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "marker_onepass" in r["Kernel_Name"] or "marker_count" in r["Kernel_Name"]][-1]
t0=int(rows[idx]["Start_Timestamp"])
prev=t0
print("# $W: one decode of a one-image batch: start (ms), duration (ms), gap to the previous kernel's end (ms), kernel")
for r in rows[idx:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%8.3f %8.3f %8.3f  %s" % ((s-t0)/1e6, (e-s)/1e6, (s-prev)/1e6, r["Kernel_Name"][:70]))
    prev=e
print("# total %.3f ms" % ((prev-t0)/1e6))
PY
