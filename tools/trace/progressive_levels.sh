cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --workload 4k_progressive --images 16 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pp.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "progressive" in r["Kernel_Name"]]
n=len(rows)//2
for r in rows[-n:]:
    print(r["Kernel_Name"][:28], r.get("Grid_Size_X",""), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms")
PY
