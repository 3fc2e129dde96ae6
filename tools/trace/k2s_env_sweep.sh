#!/bin/bash
# Sweep of the K2S environment switches on config 3 (prints value, ms per step, stage times, rounds per setting).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -f "$R/bench.py" ] || { echo "bench.py not found under $R" >&2; exit 1; }
run() { echo "== $*"; ( cd $R && env "$@" timeout 300 python3 bench.py --workload ${WORKLOAD:-4k_dri0} --steps 8 --warmup 3 --no-cpu-baseline --no-ingest --no-planar-pass 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['stage_ms'], j.get('subseq_rounds'))" ); }
run A=1
for w in 512 1024 1536; do run JPGPU_SUBSEQ_WARM_BITS=$w; done
for s in 11 13 14; do run JPGPU_SUBSEQ_SHIFT=$s; run JPGPU_SUBSEQ_SHIFT=$s JPGPU_SUBSEQ_WARM_BITS=1024; done
run JPGPU_SUBSEQ_SHIFT=13 JPGPU_SUBSEQ_WARM_BITS=4096
run JPGPU_SUBSEQ_SHIFT=14 JPGPU_SUBSEQ_WARM_BITS=4096
