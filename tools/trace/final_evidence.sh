set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02h
# 1. headline: kernel stats first (cool box), then the bench line; traffic needs the PMC entry of THIS binary: PMC pass first of all
cd $R
bash tools/profile_pmc.sh r02d --images 128 --steps 2 --warmup 1 --no-cpu-baseline --no-ingest > gpurun_out/r02h/pmc.log 2>&1
cp gpurun_out/pmc_r02d/idct_traffic_entry.json profiles/idct_traffic.json
sleep 20
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > /tmp/ks.log 2>&1; cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02h/kernel_stats_1024img.csv )
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02h/bench.json 2>/dev/null
cut -c1-200 gpurun_out/r02h/bench.json
# 2. everything else
bash tools/trace/evidence_all.sh r02d 2>&1 | tail -1
python3 tools/bench_encode.py --images 256 > gpurun_out/all_r02d/bench_encode_256.json 2>/dev/null
python3 tools/bench_encode.py --images 256 --dri 4 > gpurun_out/all_r02d/bench_encode_256_dri4.json 2>/dev/null
sha256sum jpeglibrary_amd/libjpgpu.so
