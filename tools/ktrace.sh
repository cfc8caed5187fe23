#!/bin/bash
# usage (GPU box, repo root): tools/ktrace.sh <tag> <script.py> [ENV=VAL ...] -- per-kernel durations (rocprofv3 --kernel-trace)
set -o pipefail
tag=$1; script=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
rm -rf $root/gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/$tag -- python3 $root/$script > $root/gpurun_out/$tag.log 2>&1 || { tail -5 $root/gpurun_out/$tag.log; exit 1; }
cd $root
f=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
python3 tools/prof_summary.py "$f" ${TOP:-25}
