#!/bin/bash
# tools/trace/final_evidence.sh -- everything profiles/ holds for the round's final binary, in TWO gpurun calls:
#   gpurun -- 'bash tools/trace/final_evidence.sh'            PMC passes (-> profiles/idct_traffic.json with this binary's hash),
#                                                             all configs / sinks (tools/trace/evidence_all.sh), encoder
#   gpurun -- 'bash tools/trace/final_evidence.sh headline'   the headline's kernel statistics and bench line back to back on a
#                                                             RESTED box (K3 drifts from 9.25 to 9.8-10 ms on a box that has
#                                                             been under load for minutes), after idct_traffic.json is in place
# Outputs under gpurun_out/ (pmc_$TAG/, all_$TAG/, ${TAG}_headline/); copy into profiles/ as ${TAG}_* (TAG from the environment).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03a}
cd $R
if [ "${1:-}" = "headline" ]; then
  mkdir -p gpurun_out/${TAG}_headline
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-configs > /tmp/ks.log 2>&1; cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_headline/kernel_stats_1024img.csv )
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_headline/bench.json 2>/dev/null
  cut -c1-200 gpurun_out/${TAG}_headline/bench.json
  # the reference's own benchmark input (DecoderBenchmark.cs): line + kernel statistics
  python3 bench.py --workload het_8192 --steps 5 --warmup 2 > gpurun_out/${TAG}_headline/bench_het_8192.json 2>/dev/null
  cut -c1-200 gpurun_out/${TAG}_headline/bench_het_8192.json
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kh && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kh -- python3 $R/bench.py --workload het_8192 --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > /tmp/kh.log 2>&1; cp $(find /tmp/kh -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_headline/kernel_stats_het_8192.csv )
  # random corpora through every GPU path against the checker (decode incl. failing writers / partial flushes / per-scan sessions,
  # optimizer, encoder): ONE summary, 3 seeds x 1500 files per corpus switch (tools/trace/round_stress.sh)
  TAG=$TAG bash tools/trace/round_stress.sh > /dev/null 2>&1; cp gpurun_out/${TAG}_stress.txt gpurun_out/${TAG}_headline/stress.txt
  cat gpurun_out/${TAG}_headline/stress.txt
  exit 0
fi
bash tools/trace/evidence_all.sh $TAG 2>&1 | tail -1   # (runs the PMC passes: tools/profile_pmc.sh)
cp gpurun_out/pmc_$TAG/idct_traffic_entry.json profiles/idct_traffic.json
python3 tools/bench_encode.py --images 256 > gpurun_out/all_$TAG/bench_encode_256.json 2>/dev/null
python3 tools/bench_encode.py --images 256 --dri 4 > gpurun_out/all_$TAG/bench_encode_256_dri4.json 2>/dev/null
python3 tools/bench_optimize.py --images 256 > gpurun_out/all_$TAG/bench_optimize_256.json 2>/dev/null
bash tools/trace/progressive_by_scan.sh 256 $R/gpurun_out/all_$TAG/progressive_by_scan_256.txt > /dev/null 2>&1
bash tools/trace/progressive_by_scan.sh 2048 $R/gpurun_out/all_$TAG/progressive_by_scan_2048.txt > /dev/null 2>&1
python3 tools/trace/multi_slots.py 256 3 $R/gpurun_out/all_$TAG/multi_slots.jsonl > gpurun_out/all_$TAG/multi_slots.txt 2>&1
( cd tools/microbench && ./issue_latency > $R/gpurun_out/all_$TAG/issue_latency.txt 2>&1; ./fetch_rate 1 > $R/gpurun_out/all_$TAG/fetch_rate.txt 2>&1; ./symbol_loop > $R/gpurun_out/all_$TAG/symbol_loop.txt 2>&1 )
bash tools/trace/encoder_pmc.sh ${TAG}enc --images 64 > /dev/null 2>&1; cp gpurun_out/pmc_${TAG}enc/summary.txt gpurun_out/all_$TAG/encoder_pmc_summary_64img.txt
timeout 600 bash tools/trace/progressive_pmc.sh 64 $R/gpurun_out/all_$TAG/progressive_pmc_64.txt > /dev/null 2>&1
# round 5: config 5 on real content, the reference's encoder benchmark, one canvas's kernel timeline, the config-3 A/B
python3 bench.py --workload het_progressive --steps 3 --warmup 1 --no-ingest > gpurun_out/all_$TAG/bench_het_progressive.json 2>/dev/null
bash tools/trace/progressive_by_scan.sh 256 $R/gpurun_out/all_$TAG/progressive_by_scan_256_het.txt het_progressive > /dev/null 2>&1
for a in "420" "444"; do python3 tools/bench_encode.py --workload het_8192 --subsampling $a > gpurun_out/all_$TAG/bench_encode_het_8192_$a.json 2>/dev/null; done
python3 tools/bench_encode.py --workload het_8192 --subsampling 420 --optimize-coding > gpurun_out/all_$TAG/bench_encode_het_8192_420_optimize.json 2>/dev/null
bash tools/trace/timeline.sh ${TAG}het1 het_8192 1 > /dev/null 2>&1; cp gpurun_out/timeline_${TAG}het1.txt gpurun_out/all_$TAG/timeline_het_8192_one_canvas.txt
bash tools/trace/r05_k2s_ab.sh > gpurun_out/all_$TAG/k2s_ab_4k_dri0.txt 2>&1
# round 6: the reference's call pattern (one image per call) and its kernel timelines; where a K2 wave spends its cycles; K2's stores alone
python3 bench.py --latency-only > gpurun_out/all_$TAG/latency.json 2>/dev/null
for w in 512_444 4k_dri4 4k_dri0; do bash tools/trace/one_image_timeline.sh $w; done > gpurun_out/all_$TAG/one_image_timeline.txt 2>&1
( bash tools/trace/ab_build.sh "-DJPGPU_K2_PROFILE" python3 tools/trace/k2_phases.py 128; bash tools/trace/ab_build.sh "-DJPGPU_K2_PROFILE" python3 tools/trace/k2_phases.py 256 q90 ) > gpurun_out/all_$TAG/k2_phases.txt 2>&1
( cd tools/microbench/k2_pairs && for d in 0 200; do ./line_order $d 512; done ) > gpurun_out/all_$TAG/line_order.txt 2>&1
sha256sum jpeglibrary_amd/libjpgpu.so
