#!/bin/bash
# Pricing of K1's phases on ONE box (round 6): the tree as it is against copies built with -DJPGPU_K1_PRICE=1 (no compaction loop) and
# =2 (no classification either), =3 (the copy-out stores aligned), =4 (nobody waits in the look-back); K1_PRICE_VARIANTS="3 4 0" picks.  The priced builds produce WRONG output: only `stage_ms.marker_index` of their lines means anything.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
for v in ${K1_PRICE_VARIANTS:-1 2 0}; do
  flags=""; [ "$v" != 0 ] && flags="-DJPGPU_K1_PRICE=$v"
  echo "== build flags: [$flags]"
  bash tools/trace/ab_build.sh "$flags" python3 - <<'PY'
import json, subprocess, sys
for w, extra in (("4k_dri4", ["--steps", "15", "--warmup", "3"]), ("1080p_q90", ["--steps", "15", "--warmup", "3"]), ("4k_dri0", ["--steps", "6", "--warmup", "2"])):
    out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--no-cpu-baseline", "--no-ingest", "--no-configs", "--no-planar-pass"] + extra, env=dict(__import__("os").environ, JPGPU_BENCH_EXPERIMENT="1"),
                         capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], d["stage_ms"])
    except Exception as e:
        print(w, "failed", e, out.stdout[-300:], out.stderr[-600:])
PY
done
