# Per-dispatch SQ instruction counters of the progressive kernels (16 images, one step): instructions per level.
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ppm
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d /tmp/ppm/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload 4k_progressive --images 16 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/ppm$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("/tmp/ppm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "progressive" not in r["Kernel_Name"]: continue
        rows[(int(r["Dispatch_Id"]), r["Kernel_Name"][:30], r["Grid_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    print(k, {c: int(v) for c, v in sorted(rows[k].items())})
PY
