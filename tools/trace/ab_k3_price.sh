#!/bin/bash
# K3 priced by omission on ONE box (round 6): the tree as it is against a copy built with -DJPGPU_K3_PRICE (the 8 x 8 transform left out:
# wrong samples on purpose, only `stage_ms.idct` of its lines means anything).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
for flags in -DJPGPU_K3_PRICE none -DJPGPU_K3_PRICE none; do
  [ "$flags" = none ] && flags=""
  echo "== build flags: [$flags]"
  bash tools/trace/ab_build.sh "$flags" python3 - <<'PY'
import json, os, subprocess, sys
for w, extra in (("4k_dri4", ["--steps", "15", "--warmup", "3"]), ("4k_dri4", ["--steps", "15", "--warmup", "3", "--format", "planar_u8"]), ("1080p_q90", ["--steps", "15", "--warmup", "3"])):
    out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--no-cpu-baseline", "--no-ingest", "--no-configs", "--no-planar-pass"] + extra,
                         capture_output=True, text=True, env=dict(os.environ, JPGPU_BENCH_EXPERIMENT="1"))
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(w, extra[4:] , d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(w, "failed", e, out.stdout[-300:], out.stderr[-600:])
PY
done
