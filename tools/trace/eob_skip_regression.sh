#!/bin/bash
# tools/trace/eob_skip_regression.sh -- does tests/test_gpu_parity.py::test_a_scan_inside_an_end_of_band_run_still_follows_its_producers
# catch what it is there for?  Runs it on the shipped build (expected: 3 passed), then on a copy of the tree in /tmp in which the
# AC first-pass loop skips a block of an end-of-band run BEFORE it follows its producers (the order up to round 3), expected:
# the scan-1-late case fails.  Output: gpurun_out/eob_skip_regression.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/eob_skip_regression.txt
mkdir -p $R/gpurun_out
T=tests/test_gpu_parity.py::test_a_scan_inside_an_end_of_band_run_still_follows_its_producers
{
echo "== shipped build"
( cd $R && timeout 600 python -m pytest $T -q -p no:cacheprovider 2>&1 | tail -4 )
rm -rf /tmp/eob && cp -r $R /tmp/eob && rm -rf /tmp/eob/gpurun_out
python3 - <<'EOF'
p = "/tmp/eob/jpeglibrary_amd/csrc/k2p_progressive.hip"
s = open(p).read()
follow = "                JPGPU_FOLLOW(w.my)\n                if (err != 0) break;  // gave up waiting (kDetailSpinTimeout)\n"
skip = "                if (eobrun != 0) {\n                    eobrun--;\n                    continue;\n                }\n"
assert s.count(follow + skip) == 1
open(p, "w").write(s.replace(follow + skip, skip + follow))
EOF
( cd /tmp/eob/jpeglibrary_amd/csrc && make -s -j8 > /tmp/eob/build.log 2>&1 ) || { tail -5 /tmp/eob/build.log; exit 1; }
echo "== the same test with the skip in front of the follow (the old order)"
( cd /tmp/eob && timeout 600 python -m pytest $T -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -12 | cut -c1-220 )
} > $OUT 2>&1
cat $OUT
