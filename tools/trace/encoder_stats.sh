cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pe
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python3 $GRAFT_REPO_ROOT/tools/bench_encode.py --images 256 > /tmp/pe.log 2>&1
tail -1 /tmp/pe.log | cut -c1-200
cat $(find /tmp/pe -name "*kernel_stats.csv") | python3 -c "import sys,csv; [print(r[0][:70], r[1], r[2], r[3]) for r in csv.reader(sys.stdin)]"
