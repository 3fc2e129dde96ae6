#!/bin/bash
# Register / scratch / occupancy table of every kernel in the k*.hip files and encode_kernels.hip as the compiler reports it
# (-Rpass-analysis=kernel-resource-usage); no GPU needed.  usage: kernel_resources.sh > profiles/rNN_kernel_resources.txt
R=$(cd "$(dirname "$0")/../.." && pwd)
for f in k1_markers.hip k2_huffman.hip k2s_subseq.hip k2p_progressive.hip k3_idct.hip kt_transcode.hip encode_kernels.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -c --cuda-device-only -Rpass-analysis=kernel-resource-usage \
      -o /dev/null $R/jpeglibrary_amd/csrc/$f 2> /tmp/kres_$f.txt
  python3 - /tmp/kres_$f.txt <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for b in re.split(r'(?=remark: [^\n]*Function Name:)', txt):
    m = re.search(r'Function Name: (\S+)', b)
    if not m:
        continue
    def g(k):
        mm = re.search(k + r': (\d+)', b)
        return int(mm.group(1)) if mm else -1
    dem = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    short = re.sub(r'\(.*', '', dem).replace('jpgpu::', '').replace('void ', '')
    print("%-44s VGPRs %3d  SGPRs %3d  scratch %3d B/lane  VGPR spills %3d  SGPR spills %3d  waves/SIMD %d  LDS %6d B" % (
        short, g('VGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'), g('VGPRs Spill'), g('SGPRs Spill'), g(r'Occupancy \[waves/SIMD\]'),
        g(r'LDS Size \[bytes/block\]')))
PY
done
