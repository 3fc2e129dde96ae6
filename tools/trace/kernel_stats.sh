cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/bench.py --images 512 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pp.log 2>&1
cat $(find /tmp/pp -name "*kernel_stats.csv") | python3 -c "import sys,csv; [print(r[0][:70], r[1], r[3]) for r in csv.reader(sys.stdin)]"
