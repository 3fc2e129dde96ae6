#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pscheck && cp -r $R /tmp/pscheck && cd /tmp/pscheck/jpeglibrary_amd/csrc
touch k*.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math -DJPGPU_PS_EARLIER_FORMS -DJPGPU_PS_CHECK ${EXTRA:-}" > /tmp/pscheck/build.log 2>&1 || { tail -5 /tmp/pscheck/build.log; exit 1; }
cd /tmp/pscheck && python3 - <<'PY' 2>&1 | head -30
import sys
sys.path.insert(0, "/tmp/pscheck")
import jpeglibrary_amd as jl
data = open("/tmp/pscheck/tests/golden/progress.jpg","rb").read()
outs, res = jl.decode_batch([data])
print("status", res[0].status, res[0].detail)
PY
