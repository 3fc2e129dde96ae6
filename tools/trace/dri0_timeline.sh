cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --workload 4k_dri0 --images ${IMAGES:-1024} --steps 1 --warmup 1 --no-ingest --no-cpu-baseline > /tmp/pp.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: take the rows after the last marker_count kernel
idx=[i for i,r in enumerate(rows) if "marker_count" in r["Kernel_Name"]][-1]
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%8.3f %8.3f  %s" % ((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r["Kernel_Name"][:60]))
PY
