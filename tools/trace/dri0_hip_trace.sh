#!/bin/bash
# HIP API calls of ONE jpgpu_batch_decode of a DRI = 0 batch ("zero stream syncs inside jpgpu_batch_decode"):
# rocprofv3 --hip-trace --stats over tools/trace/dri0_decodes.py with 1 and with 11 decodes behind the warm-up; the difference / 10.
#   dri0_hip_trace.sh [workload=4k_dri0] [images=256]      -> gpurun_out/dri0_hip_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-4k_dri0}; N=${2:-256}
cd /tmp && export TMPDIR=/tmp
for D in 1 11; do
  rm -rf /tmp/ht_$D
  timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/ht_$D -- python3 $R/tools/trace/dri0_decodes.py $WL $N $D > /tmp/ht_$D.log 2>&1
  tail -1 /tmp/ht_$D.log
done
mkdir -p $R/gpurun_out
python3 - "$WL" "$N" > $R/gpurun_out/dri0_hip_trace.txt <<'PY'
import csv, glob, sys
def counts(d):
    fs = glob.glob("/tmp/ht_%d/**/*hip_api_stats.csv" % d, recursive=True)
    if not fs:
        print("no hip_api_stats.csv for", d); sys.exit(1)
    return {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(fs[0]))}
a, b = counts(1), counts(11)
print("# HIP API calls per jpgpu_batch_decode of %s x %s (DRI = 0): (calls with 11 decodes - calls with 1 decode) / 10" % (sys.argv[2], sys.argv[1]))
print("# rocprofv3 --hip-trace --stats -- python3 tools/trace/dri0_decodes.py; both runs: upload, two decodes with a wait each, then N decodes and ONE wait")
print("%-40s %10s %10s %12s" % ("call", "1 decode", "11 decodes", "per decode"))
blocking = 0.0
for k in sorted(set(a) | set(b)):
    per = (b.get(k, 0) - a.get(k, 0)) / 10.0
    print("%-40s %10d %10d %12.1f" % (k, a.get(k, 0), b.get(k, 0), per))
    if k in ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy", "hipMemcpyDtoH", "hipMemcpyHtoD", "hipStreamWaitEvent") and k != "hipStreamWaitEvent":
        blocking += per
print("# calls that make the host wait, per decode: %.1f" % blocking)
PY
cat $R/gpurun_out/dri0_hip_trace.txt
