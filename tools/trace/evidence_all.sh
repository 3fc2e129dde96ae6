#!/bin/bash
# tools/trace/evidence_all.sh TAG -- everything profiles/ holds for one state of the code: headline bench line (with CPU
# baseline and ingest-inclusive rate), headline kernel stats + PMC passes (-> roofline.traffic entry tied to the binary's
# hash), the other workloads / sinks (tools/trace/evidence.sh), config 5 at 256 / 1024 / 2048 frames.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/all_$TAG
mkdir -p $OUT
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-planar-pass --no-configs > $OUT/kernel_stats.log 2>&1; cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_1024img.csv )
bash tools/profile_pmc.sh $TAG --images 128 --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --no-planar-pass --no-configs > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary_128img.txt; cp gpurun_out/pmc_$TAG/idct_traffic_entry.json $OUT/ 2>/dev/null
bash tools/trace/evidence.sh $TAG > $OUT/evidence.log 2>&1
cp gpurun_out/evidence_$TAG/* $OUT/
python3 bench.py --workload 4k_progressive --steps 5 --warmup 1 > $OUT/bench_4k_progressive.json 2> $OUT/bench_4k_progressive.err
cut -c1-300 $OUT/bench_4k_progressive.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 $R/bench.py --workload 4k_progressive --steps 2 --warmup 1 --distinct 64 --no-cpu-baseline --no-ingest > $OUT/kernel_stats_prog.log 2>&1; cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_4k_progressive.csv )
for n in 1024 2048; do python3 bench.py --workload 4k_progressive --images $n --distinct 64 --steps 3 --warmup 1 --no-cpu-baseline --no-ingest > $OUT/bench_4k_progressive_$n.json 2>> $OUT/bench_4k_progressive.err; cut -c1-200 $OUT/bench_4k_progressive_$n.json; done
sha256sum jpeglibrary_amd/libjpgpu.so > $OUT/library.sha256
ls $OUT
