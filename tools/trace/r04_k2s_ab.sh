#!/bin/bash
# tools/trace/r04_k2s_ab.sh -- config 3 (DRI = 0) with the tree's library and with build variants of the K2S round kernel
# (lookup width, burst length): bench line + per-kernel times each.   usage: r04_k2s_ab.sh OUTDIR "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
mkdir -p $OUT
run() {  # name, root
  ( cd $2 && python3 bench.py --workload 4k_dri0 --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-planar-pass > $OUT/bench_$1.json 2> $OUT/bench_$1.err; cut -c1-260 $OUT/bench_$1.json | tail -1
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd_$1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_$1 -- python3 $2/bench.py --workload 4k_dri0 --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --no-planar-pass > /tmp/pd_$1.log 2>&1
    python3 -c "import sys,csv; [print(r[0][:48], r[1], r[3], r[5], r[6]) for r in csv.reader(open(sys.argv[1])) if 'subseq' in r[0] or 'Name' in r[0]]" $(find /tmp/pd_$1 -name "*kernel_stats.csv" | head -1) | tee $OUT/stats_$1.txt )
}
echo "== tree"; run tree $R
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  echo "== variant $i: $FLAGS"
  rm -rf /tmp/ab$i && cp -r $R /tmp/ab$i && ( cd /tmp/ab$i/jpeglibrary_amd/csrc && touch *.hip *.cpp && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-parameter -ffp-contract=off -fno-fast-math $FLAGS" > /tmp/ab$i/build.log 2>&1 ) || { tail -5 /tmp/ab$i/build.log; continue; }
  run v$i /tmp/ab$i
done
