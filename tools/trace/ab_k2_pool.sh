#!/bin/bash
# A/B of the pooled K2 (round 6) on ONE box: the tree as it is against a copy built with -DJPGPU_K2_NO_POOL (tools/trace/ab_build.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
for flags in "" "-DJPGPU_K2_NO_POOL"; do
  echo "== build flags: [$flags]"
  bash tools/trace/ab_build.sh "$flags" python3 - <<'PY'
import json, subprocess, sys
for w, extra in (("4k_dri4", ["--steps", "15", "--warmup", "3"]), ("1080p_q90", ["--steps", "15", "--warmup", "3"])):
    out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--no-cpu-baseline", "--no-ingest", "--no-configs", "--no-planar-pass"] + extra,
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print(w, d["value"], d["ms_per_step"], d["stage_ms"])
PY
done
