#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <round-tag>   e.g. r01
# rocprofv3 --kernel-trace --stats of the default bench command -> profiles/<tag>_bench_kernel_stats.csv and
# profiles/<tag>_bench_kernel_by_grid.txt (per kernel+grid: calls, avg/min us, share), then the PMC traffic pass.
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
tools/profile_bench.sh prof_$tag > gpurun_out/prof_$tag.summary 2>&1 || { tail -5 gpurun_out/prof_$tag.summary; exit 1; }
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" profiles/${tag}_bench_kernel_stats.csv
cp gpurun_out/prof_${tag}_by_grid.txt profiles/${tag}_bench_kernel_by_grid.txt
grep '^{"metric' gpurun_out/prof_$tag.log > profiles/${tag}_bench_under_rocprof.json
head -12 profiles/${tag}_bench_kernel_by_grid.txt | cut -c1-140
tools/pmc_attn.sh $tag
