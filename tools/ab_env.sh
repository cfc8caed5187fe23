#!/bin/bash
# usage (GPU box, repo root; library built with `make TUNING=1`): tools/ab_env.sh <ENVNAME> <v1> <v2> ... [-- bench args]
# A/B of one tuning variable on the captured decode step: per value, rocprofv3 kernel trace of bench.py -> the lines of
# the GEMM / glue kernels and ms_per_step in gpurun_out/ab_<ENVNAME>.txt
set -o pipefail
name=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
out=gpurun_out/ab_$name.txt
: > $out
for v in "${vals[@]}"; do
  export $name=$v
  tools/profile_bench.sh ab_${name}_$v --no-plugin-surface "$@" > gpurun_out/ab_${name}_$v.summary 2>&1 || { tail -5 gpurun_out/ab_${name}_$v.summary; exit 1; }
  echo "== $name=$v  $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_${name}_$v.log | head -1)" | tee -a $out
  grep -E "fp8_gemm|slab_|decode_merge|w4a16|w4_reduce|glue" gpurun_out/ab_${name}_${v}_by_grid.txt | cut -c1-60,78-140 | tee -a $out
  rm -rf gpurun_out/ab_${name}_$v
done
