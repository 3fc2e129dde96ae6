export JPGPU_PROG_FORCE_PIPELINE=1 JPGPU_PROG_SPIN_BUDGET=4194304
for cfg in "4096 32" "2048 16" "2048 8"; do
  set -- $cfg
  echo "== ring $1 chunk $2"
  for rep in 1 2 3 4; do JPGPU_PS_RING=$1 JPGPU_PS_CHUNK=$2 timeout 300 python tools/trace/progressive_oversubscribed.py 1024 2>&1 | grep "^n=1024" | cut -c46-140; done
done
