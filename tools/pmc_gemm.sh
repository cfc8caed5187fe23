#!/bin/bash
# usage: tools/pmc_gemm.sh <tag>  -- SQ counters of tools/gemm_bench.py (M=128), summarised per kernel+grid
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
   --output-format csv -d $root/gpurun_out/$tag -- python3 $root/tools/gemm_bench.py > $root/gpurun_out/$tag.log 2>&1 || exit 1
cd $root
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if "xs_kernel" not in r["Kernel_Name"] and "xr_kernel" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"][:44], r["Grid_Size"])
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
for key, c in agg.items():
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(key, {k: round(v / wc, 3) for k, v in c.items() if k != "SQ_WAVE_CYCLES"}, "wave_cycles", wc)
PY
