#!/bin/bash
# A/B of the round-5 K2S changes on config 3 (1024 x 4K DRI = 0) and the reference's benchmark canvas: gathered late rounds,
# pooled final pass, ordered lanes -- BUILD variants since round 6 (-DJPGPU_SUBSEQ_NO_GATHER, -DJPGPU_SF_NO_POOL, -DJPGPU_SF_NO_ORDER:
# each line below rebuilds a copy of the tree through tools/trace/ab_build.sh) -- and host-checked rounds (JPGPU_SUBSEQ_HOST_CHECK,
# still a run-time switch: the fallback path the tests force).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
mkdir -p gpurun_out
run() {  # name, build flags, env...
  local name=$1 flags=$2; shift 2
  env "$@" bash tools/trace/ab_build.sh "$flags" timeout 300 python3 bench.py --workload ${WORKLOAD:-4k_dri0} --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-ingest > $R/gpurun_out/r05_k2s_$name.json 2> $R/gpurun_out/r05_k2s_$name.log
  python3 - "$name" <<PY
import json,sys
d=json.loads(open("$R/gpurun_out/r05_k2s_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %9.0f Mpx/s  %7.3f ms  stages %s  rounds %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("stage_ms"), d.get("subseq_rounds")))
PY
}
run default "" JPGPU_NOP=1
run no_gather "-DJPGPU_SUBSEQ_NO_GATHER" JPGPU_NOP=1
run no_pool "-DJPGPU_SF_NO_POOL" JPGPU_NOP=1
run no_order "-DJPGPU_SF_NO_ORDER" JPGPU_NOP=1
run none_of_them "-DJPGPU_SUBSEQ_NO_GATHER -DJPGPU_SF_NO_POOL -DJPGPU_SF_NO_ORDER" JPGPU_NOP=1
run host_checked "" JPGPU_SUBSEQ_HOST_CHECK=1
