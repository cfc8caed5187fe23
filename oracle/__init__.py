"""CPU oracle for the MI355X hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (plain torch on the host) of the reference's
torch-native arithmetic for the hot path: paged-KV decode/extend attention,
KV-slot index gathers, FP8 / AWQ / GPTQ quantized linears and the lse merge.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``iaas_sglang_amd``) never imports ``oracle`` and fails loudly when the HIP
library is missing; there is no CPU fallback in the product.

Pinning: every function here is checked in ``tests/test_oracle_golden.py``
against ``tests/golden/*.pt`` -- vectors produced in the build container by
running the reference's own files (``tests/golden/make_golden.py``; recipe in
SURVEY.md section 8c).  GPTQ has no reference implementation under
/root/reference (it lives in vllm) and is marked "parity unpinned" where used.
"""
