#!/bin/bash
# A/B aid: copies the tree to /tmp/ab, rebuilds libjpgpu.so there with extra compiler flags (the tree's library is not touched),
# and runs a command inside the copy.   usage: ab_build.sh "<extra CXXFLAGS>" <command ...>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
FLAGS="$1"; shift
rm -rf /tmp/ab && cp -r $R /tmp/ab && cd /tmp/ab/jpeglibrary_amd/csrc
touch k*.hip encode_kernels.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -fno-fast-math $FLAGS" > /tmp/ab/build.log 2>&1 || { tail -5 /tmp/ab/build.log; exit 1; }
cd /tmp/ab && GRAFT_REPO_ROOT=/tmp/ab "$@"
