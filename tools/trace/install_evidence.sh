#!/bin/bash
# tools/trace/install_evidence.sh TAG [OLDTAG] -- copy what `TAG=... final_evidence.sh` (and `... headline`) left under gpurun_out/
# into profiles/ as TAG_*, replacing the OLDTAG_* set (run here, in the container, after the gpurun calls).
set -e
TAG=$1; OLD=${2:-}
R=$(cd "$(dirname "$0")/../.." && pwd)
S=$R/gpurun_out/all_$TAG
[ -d "$S" ] || { echo "no $S"; exit 1; }
if [ -n "$OLD" ]; then for f in $R/profiles/${OLD}_*; do [ -e "$f" ] && mv "$f" "${f/${OLD}_/${TAG}_}"; done; fi
cp $S/idct_traffic_entry.json $R/profiles/idct_traffic.json
for f in bench_4k_dri0.json bench_1080p_q90.json bench_planar_u8.json bench_rgb_u8.json bench_rgba_u8.json bench_4k_progressive.json \
         bench_4k_progressive_1024.json bench_4k_progressive_2048.json bench_encode_256.json bench_encode_256_dri4.json bench_optimize_256.json \
         kernel_stats_1080p_q90.csv kernel_stats_4k_dri0.csv kernel_stats_4k_progressive.csv kernel_stats_planar_u8.csv pmc_summary_128img.txt \
         progressive_by_scan_256.txt progressive_by_scan_2048.txt progressive_pmc_64.txt symbol_loop.txt fetch_rate.txt issue_latency.txt \
         kernel_stats_rgb_u8.csv kernel_stats_rgba_u8.csv multi_slots.jsonl multi_slots.txt encoder_pmc_summary_64img.txt library.sha256 \
         bench_het_progressive.json progressive_by_scan_256_het.txt bench_encode_het_8192_420.json bench_encode_het_8192_444.json \
         bench_encode_het_8192_420_optimize.json timeline_het_8192_one_canvas.txt k2s_ab_4k_dri0.txt latency.json one_image_timeline.txt k2_phases.txt line_order.txt; do
  [ -e $S/$f ] && cp $S/$f $R/profiles/${TAG}_$f
done
# the headline line and its kernel statistics inside the long evidence call (a box under load for minutes) ...
cp $S/bench.json $R/profiles/${TAG}_bench_evidence_call.json
cp $S/kernel_stats_1024img.csv $R/profiles/${TAG}_kernel_stats_1024img_evidence_call.csv
# ... and in a call of their own, with the reference's benchmark input and the stress sweeps (final_evidence.sh headline)
if [ -d $R/gpurun_out/${TAG}_headline ]; then
  H=$R/gpurun_out/${TAG}_headline
  cp $H/bench.json $R/profiles/${TAG}_bench.json
  cp $H/kernel_stats_1024img.csv $R/profiles/${TAG}_kernel_stats_1024img.csv
  for f in bench_het_8192.json kernel_stats_het_8192.csv stress.txt; do [ -e $H/$f ] && cp $H/$f $R/profiles/${TAG}_$f; done
fi
( cd $R && bash tools/trace/kernel_resources.sh > profiles/${TAG}_kernel_resources.txt 2>/dev/null )
python3 - $S $R/gpurun_out/${TAG}_headline <<'PY'
import json, sys, os
for d in sys.argv[1:]:
    for f in sorted(os.listdir(d)) if os.path.isdir(d) else []:
        if not (f.startswith("bench") and f.endswith(".json")): continue
        try:
            j = json.loads(open(os.path.join(d, f)).read().strip().splitlines()[-1])
        except Exception:
            continue
        r = j.get("roofline") or {}
        print(os.path.basename(d), f, j.get("value"), j.get("ms_per_step"), j.get("stage_ms"), r.get("frac"), r.get("traffic"), r.get("read_frac_planar"),
              (j.get("cpu_baseline") or {}).get("value"), j.get("value_ingest_inclusive"), j.get("value_ingest_inclusive_pinned"))
PY
sha256sum $R/jpeglibrary_amd/libjpgpu.so; cat $R/profiles/${TAG}_library.sha256
