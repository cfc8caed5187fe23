#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <round-tag>   e.g. r01
# rocprofv3 --kernel-trace --stats of the default bench command -> profiles/<tag>_bench_kernel_stats.csv and
# profiles/<tag>_bench_kernel_by_grid.txt (per kernel+grid: calls, avg/min us, share), then the PMC traffic pass.
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf gpurun_out/prof_$tag
tools/profile_bench.sh prof_$tag --no-plugin-surface > gpurun_out/prof_$tag.summary 2>&1 || { tail -5 gpurun_out/prof_$tag.summary; exit 1; }
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
# profiles/ does not travel back from the GPU box: the judged copies are staged under gpurun_out/profiles_<tag>/
out=gpurun_out/profiles_$tag
mkdir -p $out
cp "$f" $out/${tag}_bench_kernel_stats.csv
cp gpurun_out/prof_${tag}_by_grid.txt $out/${tag}_bench_kernel_by_grid.txt
grep '^{"metric' gpurun_out/prof_$tag.log > $out/${tag}_bench_under_rocprof.json
head -14 $out/${tag}_bench_kernel_by_grid.txt | cut -c1-140
tools/pmc_attn.sh $tag
cp profiles/decode_attn_traffic.json profiles/${tag}_decode_attn_pmc_fetch.csv profiles/${tag}_decode_attn_pmc_write.csv $out/ 2>/dev/null
# the same command with one 16-sequence prefill chunk: per-kernel prefill times
rm -rf gpurun_out/prof_${tag}_prefill
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_${tag}_prefill -- \
    python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plugin-surface --prefill-batch 16 > $root/gpurun_out/prof_${tag}_prefill.log 2>&1
cd $root
f=$(find gpurun_out/prof_${tag}_prefill -name "*kernel_trace.csv" | head -1)
python3 tools/prof_summary.py "$f" 45 > $out/${tag}_bench_with_prefill_kernel_by_grid.txt
head -24 $out/${tag}_bench_with_prefill_kernel_by_grid.txt | cut -c1-140
