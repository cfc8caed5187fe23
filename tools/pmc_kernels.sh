#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernels.sh <out-name> <kernel-substring>[,<substring>...] <script.py> [ENV=VAL ...]
#   e.g. tools/pmc_kernels.sh r02_pmc_fp8_decode xd_kernel tools/gemm_bench.py
# Per-kernel SQ / TCC counters of one micro-benchmark, summarised per (kernel, grid) into profiles/<out-name>.txt.
# Three rocprofv3 passes, --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ has 8 slots per pass,
# FETCH_SIZE takes 3 of the 4 TCC slots), the program directly after `--`:
#   pass A  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
#           SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES  + GRBM_GUI_ACTIVE
#   pass B  SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES
#   pass C  FETCH_SIZE (KB; DOUBLED in the summary: gfx950 reports half of the bytes of 16-B/lane reads)
# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs).
set -o pipefail
out=$1; filt=$2; script=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
run() {  # <pass> <counters...>
  local p=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $root/gpurun_out/$out.$p -- \
      python3 $root/$script > $root/gpurun_out/$out.$p.log 2>&1 || { tail -5 $root/gpurun_out/$out.$p.log; return 1; }
}
run A SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1
run B SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES || echo "pass B failed (counter set not available?)"
run C FETCH_SIZE || echo "pass C failed"
cd $root
python3 tools/pmc_summary.py "$out" "$filt" "$script $*" | tee gpurun_out/$out.txt   # travels back with gpurun_out/; copy it to profiles/
