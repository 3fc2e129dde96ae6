#!/bin/bash
# tools/trace/progressive_pmc.sh [frames] -- instruction counts of the progressive stream kernel scan by scan (JPGPU_PROG_BY_SCAN=1:
# one launch per scan of libjpeg's script): rocprofv3 --pmc passes over bench.py's config 5, the last ten launches of the
# kernel, per restart unit of the scan (blocks, or MCUs for the interleaved DC scans).  Counters only, no tracing besides.
N=${1:-64}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${2:-$R/gpurun_out/progressive_pmc.txt}
cd /tmp && export TMPDIR=/tmp
export JPGPU_PROG_BY_SCAN=1
rm -rf /tmp/ppmc; i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d /tmp/ppmc/p$i -- python3 $R/bench.py --workload 4k_progressive --images $N --distinct 64 --steps 1 --warmup 1 --no-cpu-baseline --no-ingest --no-planar-pass > /tmp/ppmc$i.log 2>&1
done
python3 - "$N" > "$OUT" <<'PY'
import csv, glob, sys, collections
n = int(sys.argv[1])
per = collections.defaultdict(dict)   # dispatch id -> counter -> value
for f in glob.glob("/tmp/ppmc/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "progressive_stream" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-10:]
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d in ids:
            per[ids.index(d)][r["Counter_Name"]] = per[ids.index(d)].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
units = [32400 * n] + [129600 * n, 32400 * n, 32400 * n, 129600 * n, 129600 * n] + [32400 * n] + [32400 * n, 32400 * n, 129600 * n]
print(f"# {n} x 4K 4:2:0 progressive frames, per launch (= scan of libjpeg's script): counters per restart unit (DC scans: MCU; AC scans: block)")
names = sorted({c for d in per.values() for c in d})
print("scan " + " ".join(f"{c.replace('SQ_', ''):>14s}" for c in names))
for k in range(10):
    print(f"{k + 1:4d} " + " ".join(f"{per[k].get(c, 0.0) / units[k]:14.2f}" for c in names))
PY
cat "$OUT"
