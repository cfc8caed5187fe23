#!/bin/bash
# usage (GPU box, repo root): tools/pmc_attn.sh <round-tag>
# HBM traffic of one decode-attention call (split kernel + merge) at the BASELINE shape, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2), kernel-trace only; FETCH_SIZE (KB) is DOUBLED (gfx950 reports half of
# the bytes of 16-B/lane coalesced reads), WRITE_SIZE (KB) taken as is.  Writes profiles/decode_attn_traffic.json
# and copies the two counter CSVs to profiles/<tag>_decode_attn_pmc_{fetch,write}.csv.
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
export SPLITS=${SPLITS:-2} NPOOL=4
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/gpurun_out/pmc_attn_$c -- \
      python3 $root/tools/attn_bench.py > $root/gpurun_out/pmc_attn_$c.log 2>&1 || exit 1
done
cd $root
python3 - "$tag" <<'PY'
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1]
splits = int(os.environ.get("SPLITS", "2"))
raw = {}
for c, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(f"gpurun_out/pmc_attn_{c}/**/*counter_collection.csv", recursive=True)[0]
    shutil.copy(f, f"profiles/{tag}_decode_attn_pmc_{name}.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = "decode_attn_kernel" if "decode_attn_kernel" in r["Kernel_Name"] else "decode_merge_kernel" if "decode_merge" in r["Kernel_Name"] else None
        if k: agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        raw.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
B, S, Hq, Hkv, D = 128, 2048, 32, 8, 128
total = sum((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 for v in raw.values())
out = {"kernel": "decode_attn_kernel<bf16,128,4,8> + decode_merge_kernel<bf16,128>", "batch": B, "seq": S, "tp": 1,
       "kv_splits": splits, "hbm_bytes_per_launch": int(total),
       "algorithmic_bytes_per_launch": 2 * B * S * Hkv * D * 2 + 2 * B * Hq * D * 2 + 4 * B * S,
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE on tools/attn_bench.py "
                 f"(SPLITS={splits} NPOOL=4, tools/pmc_attn.sh); per-dispatch averages; FETCH_SIZE (KB) doubled per the gfx950 "
                 "correction for 16-B-per-lane coalesced reads, WRITE_SIZE (KB) taken as is; bytes = (2*FETCH + WRITE) * 1024 "
                 "summed over the two kernels of one call",
       "raw_counters_kb": raw}
json.dump(out, open("profiles/decode_attn_traffic.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
