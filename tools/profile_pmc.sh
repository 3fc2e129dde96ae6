#!/bin/bash
# tools/profile_pmc.sh TAG [bench args...] -- rocprofv3 kernel-trace + PMC passes of bench.py on the GPU box.
# Counters go in their own runs (never combined with sys/hip tracing; see the gpurun rules). Writes gpurun_out/pmc_TAG/.
set -u
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="${@:---images 128 --steps 2 --warmup 1 --no-cpu-baseline}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LEVEL_WAVES" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
python3 - "$OUT" "$R" $ARGS <<'PY'
import csv, glob, sys, collections, os, json, hashlib
out, root, args = sys.argv[1], sys.argv[2], sys.argv[3:]
def opt(name, default):
    return args[args.index(name) + 1] if name in args else default
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in sorted(agg.items()):
        if "jpgpu" not in k: continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}\n")
    for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
        fo.write(open(f).read())
print(open(out + "/summary.txt").read())
# roofline.traffic of bench.py: HBM bytes per image of the IDCT kernel, tied to the binary that was profiled
sha = hashlib.sha256(open(os.path.join(root, "jpeglibrary_amd", "libjpgpu.so"), "rb").read()).hexdigest()
images = int(opt("--images", "0") or 0)
kfmt = {"interleaved_u8": 0, "planar_u8": 1, "rgb_u8": 3, "rgba_u8": 4}[opt("--format", "interleaved_u8")]
for k, d in agg.items():
    # the variant of the format that was asked for (bench.py's short planar pass runs idct_output_kernel<1, 0> besides)
    if f"idct_output_kernel<{kfmt}," in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d and images:
        fetch_kb = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
        write_kb = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        entry = {"kernel": k.split("jpgpu::")[-1], "hbm_bytes_per_image": int((2 * fetch_kb + write_kb) * 1024 / images),
                 "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb, "images_in_profiled_launch": images,
                 "library_sha256": sha,
                 "correction": "gfx950: FETCH_SIZE counts wide coalesced reads (incl. global_load_lds_dwordx4) at 1/2 -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B/lane stores",
                 "source": f"profiles/{os.path.basename(out).replace('pmc_', '')}_pmc_summary_{images}img.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, tools/profile_pmc.sh)"}
        key = opt("--workload", "4k_dri4") + ":" + opt("--format", "interleaved_u8")
        json.dump({key: entry}, open(out + "/idct_traffic_entry.json", "w"), indent=1)
        print("traffic entry", key, entry["hbm_bytes_per_image"])
PY
