#!/bin/bash
# usage (GPU box, tuning build): tools/w4_plan_sweep.sh -- kernel + reduce time of the C4 shapes (M = 64) per (NT, S)
for nt in 1 2 4; do for s in 1 2 3 4 6 8; do
  out=$(MI_W4_NT=$nt MI_W4_S=$s TOP=12 tools/ktrace.sh w4sw tools/w4_bench.py 2>/dev/null | grep "w4a16_xw\|w4_reduce" | awk '{print $2, $5}' | tr '\n' ';')
  echo "NT=$nt S=$s: $out"
done; done
