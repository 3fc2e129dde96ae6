#!/bin/bash
# A/B of the round-5 K2S changes on config 3 (1024 x 4K DRI = 0) and the reference's benchmark canvas: gathered late rounds
# (JPGPU_SUBSEQ_NO_GATHER), pooled final pass (JPGPU_SF_NO_POOL), host-checked rounds (JPGPU_SUBSEQ_HOST_CHECK).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python3 bench.py --workload ${WORKLOAD:-4k_dri0} --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-ingest > gpurun_out/r05_k2s_$name.json 2> gpurun_out/r05_k2s_$name.log
  python3 - "$name" <<PY
import json,sys
d=json.loads(open("gpurun_out/r05_k2s_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %9.0f Mpx/s  %7.3f ms  stages %s  rounds %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("subseq_rounds")))
PY
}
run default JPGPU_NOP=1
run no_gather JPGPU_SUBSEQ_NO_GATHER=1
run no_pool JPGPU_SF_NO_POOL=1
run no_order JPGPU_SF_NO_ORDER=1
run none_of_them JPGPU_SUBSEQ_NO_GATHER=1 JPGPU_SF_NO_POOL=1 JPGPU_SF_NO_ORDER=1
run host_checked JPGPU_SUBSEQ_HOST_CHECK=1
