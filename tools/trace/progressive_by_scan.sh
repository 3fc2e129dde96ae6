#!/bin/bash
# Per-scan-kind time table of BASELINE config 5: every scan of libjpeg's 10-scan script in a launch of its
# own (JPGPU_PROG_BY_SCAN=1: scan k of all frames, file order), kernel durations from rocprofv3's kernel trace.  With n frames
# a launch holds n one-wave workgroups: at 2048 frames that is two waves per SIMD, so duration x 1024 SIMDs ~ SIMD-seconds.
# usage: progressive_by_scan.sh <frames> <out.txt> [workload: 4k_progressive | het_progressive]
N=${1:-2048}
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/progressive_by_scan.txt}
WL=${3:-4k_progressive}
case "$OUT" in /*) ;; *) OUT="$GRAFT_REPO_ROOT/$OUT" ;; esac
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pbs
JPGPU_PROG_BY_SCAN=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pbs -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --images $N --distinct 64 --steps 1 --warmup 1 --no-cpu-baseline --no-ingest --no-planar-pass > /tmp/pbs.log 2>&1
python3 - "$N" "$WL" > "$OUT" <<'PY'
import csv, glob, sys
n = int(sys.argv[1])
wl = sys.argv[2]
f = glob.glob("/tmp/pbs/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "progressive_stream" in r["Kernel_Name"]]
rows = rows[-10:]  # the last step's ten launches
kinds = ["DC first, Y+Cb+Cr interleaved (Al=1)", "Y AC 1-5 first (Al=2)", "Cr AC 1-63 first (Al=1)", "Cb AC 1-63 first (Al=1)",
         "Y AC 6-63 first (Al=2)", "Y AC 1-63 refine (Ah=2, Al=1)", "DC refine (Ah=1, Al=0)", "Cr AC refine (Ah=1, Al=0)",
         "Cb AC refine (Ah=1, Al=0)", "Y AC refine (Ah=1, Al=0)"]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
print(f"# {wl}: {n} x 4K 4:2:0 progressive frames (64 distinct), one launch per scan of libjpeg's script, ms per launch and share")
for k, r in enumerate(rows):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"scan {k + 1:2d}  {kinds[k] if k < len(kinds) else '?':42s} {ms:9.2f} ms  {100 * ms / tot:5.1f} %")
print(f"total {tot:9.2f} ms")
PY
cat "$OUT"
