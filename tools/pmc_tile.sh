#!/bin/bash
# usage: tools/pmc_tile.sh <tag> -- L2 hit rate of the prefill tile GEMM (tools/prefill_bench.py, ATTN=0)
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ATTN=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $root/gpurun_out/$tag -- python3 $root/tools/prefill_bench.py > $root/gpurun_out/$tag.log 2>&1 || exit 1
cd $root
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "tile_kernel" not in r["Kernel_Name"]: continue
    agg[r["Grid_Size"]][r["Counter_Name"]] += float(r["Counter_Value"])
for g, c in agg.items():
    h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print("grid", g, "hit rate", round(h / max(h + m, 1), 3), {k: int(v) for k, v in c.items()})
PY
