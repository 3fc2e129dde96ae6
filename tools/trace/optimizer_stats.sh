# Per-kernel time of the optimizer path (K1 + the three KT passes) on 256 x 4K DRI = 7.
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/po
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -- python3 $GRAFT_REPO_ROOT/tools/bench_optimize.py --images 256 --steps 3 --cpu-images 8 > /tmp/po.log 2>&1
tail -1 /tmp/po.log | cut -c1-400
cat $(find /tmp/po -name "*kernel_stats.csv") | python3 -c "import sys,csv; [print(r[0][:80], r[1], r[3]) for r in csv.reader(sys.stdin)]"
