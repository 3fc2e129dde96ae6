#!/bin/bash
# One stress summary per round: tools/stress_parity.py, 3 seeds x 1500 files per corpus switch -- no more --
# on the final binary.  Output: gpurun_out/${TAG}_stress.txt (seed, switches, the tool's summary line; mismatching inputs under
# gpurun_out/stress/).  Round 6 added the STRESS_SAMPLING corpus (per-component sampling factors 1..4, three scans, progressive).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r06}_stress.txt
echo "# tools/stress_parity.py 1500 <seed>, library sha256 $(sha256sum jpeglibrary_amd/libjpgpu.so | cut -c1-16)" > $OUT
for sw in "" "STRESS_SYNTH=1" "STRESS_HEADER=1" "STRESS_CMYK=1 STRESS_SYNTH=1" "STRESS_SAMPLING=1"; do
  for seed in ${SEEDS:-51 52 53}; do
    line=$(env $sw timeout 900 python3 tools/stress_parity.py ${N:-1500} $seed 2>&1 | grep -v "amdgpu.ids" | tail -4 | tr '\n' ' ')
    echo "seed $seed [${sw:-default corpus}] $line" | tee -a $OUT
  done
done
grep -c "mismatches: 0" $OUT
