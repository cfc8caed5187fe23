#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of bench.py -> gpurun_out/<tag>/ and gpurun_out/<tag>_by_grid.txt
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -- \
    python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --prefill-batch 0 "$@" > $root/gpurun_out/$tag.log 2>&1
rc=$?
cd $root
tail -1 gpurun_out/$tag.log | cut -c1-240
f=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
python3 tools/prof_summary.py "$f" 45 > gpurun_out/${tag}_by_grid.txt
cat gpurun_out/${tag}_by_grid.txt
exit $rc
