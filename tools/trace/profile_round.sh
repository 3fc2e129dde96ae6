cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/pk.log 2>&1
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r01l_kernel_stats_1024img.csv
tail -1 /tmp/pk.log | cut -c1-300
cd $GRAFT_REPO_ROOT && bash tools/profile_pmc.sh r01l > /dev/null 2>&1; cp gpurun_out/pmc_r01l/summary.txt gpurun_out/r01l_pmc_summary_128img.txt; head -30 gpurun_out/r01l_pmc_summary_128img.txt
