import sys, torch
sys.path.insert(0, ".")
from iaas_sglang_amd import ops
FP8 = torch.float8_e4m3fn
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(M, K, device="cuda").to(FP8); w = torch.randn(N, K, device="cuda").to(FP8)
sa = torch.ones(1, device="cuda"); sb = torch.ones(1, device="cuda")
print("launch", M, N, K, flush=True)
out = ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16)
torch.cuda.synchronize()
ref = (x.float() @ w.float().t())
print("max err", float((out.float() - ref).abs().max()), "ref max", float(ref.abs().max()), flush=True)
