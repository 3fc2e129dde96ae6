cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -- python3 $GRAFT_REPO_ROOT/bench.py --workload 4k_dri0 --images 1024 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pd.log 2>&1
tail -1 /tmp/pd.log | cut -c1-200
cat $(find /tmp/pd -name "*kernel_stats.csv") | python3 -c "import sys,csv; [print(r[0][:70], r[1], r[2], r[3]) for r in csv.reader(sys.stdin)]"
