#!/bin/bash
# A/B of the pooled Huffman kernels drawing their next ticket ahead (round 6) on ONE box: -DJPGPU_K2_TICKET_AHEAD against the tree as it is (measured: the tree wins).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
for flags in -DJPGPU_K2_TICKET_AHEAD none -DJPGPU_K2_TICKET_AHEAD none; do
  [ "$flags" = none ] && flags=""
  echo "== build flags: [$flags]"
  bash tools/trace/ab_build.sh "$flags" python3 - <<'PY'
import json, subprocess, sys
for w, extra in (("4k_dri4", ["--steps", "20", "--warmup", "3"]), ("1080p_q90", ["--steps", "20", "--warmup", "3"]), ("4k_dri0", ["--steps", "8", "--warmup", "2"]), ("het_8192", ["--steps", "5", "--warmup", "2"])):
    out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--no-cpu-baseline", "--no-ingest", "--no-configs", "--no-planar-pass"] + extra,
                         capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], d["stage_ms"], d.get("parity_spot_check"))
    except Exception as e:
        print(w, "failed", e, out.stdout[-300:], out.stderr[-600:])
PY
done
