"""CPU oracle: RMSNorm / NeoX RoPE / SiLU-and-mul in the reference's *native* torch forms.

TEST INFRASTRUCTURE ONLY.  These are SURVEY section 8f "next" rows; the bodies restate the
reference's forward_native methods line by line (citations relative to /root/reference).
PINNED: tests/test_oracle_golden.py asserts every function here bit-identical to the outputs of the
reference's own forward_native methods (layernorm.py:128, activation.py:56, rotary_embedding.py:138)
held in tests/golden/elementwise.pt (tests/golden/make_golden_elementwise.py runs them), and within the
stated tolerance of the sgl-kernel tests' torch forms (test_norm.py, test_rotary_embedding.py).
"""
import torch
import torch.nn.functional as F


def rmsnorm(x, weight, eps, residual=None):
    """python/sglang/srt/layers/layernorm.py:128-146."""
    orig = x.dtype
    x = x.to(torch.float32)
    if residual is not None:
        x = x + residual.to(torch.float32)
        residual = x.to(orig)
    var = x.pow(2).mean(dim=-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    x = (x * weight).to(orig)
    return x if residual is None else (x, residual)


def rope_cos_sin_cache(head_dim, max_pos, base=10000.0):
    """rotary_embedding.py:108-125 (_compute_inv_freq/_compute_cos_sin_cache), fp32."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_pos, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def rope_neox(positions, q, k, cos_sin_cache, head_dim):
    """rotary_embedding.py:49-74,138-166 (forward_native, neox style, rotary_dim == head_dim)."""
    cos, sin = cos_sin_cache.index_select(0, positions.flatten()).chunk(2, dim=-1)

    def rot(x):
        shape = x.shape
        x = x.view(positions.numel(), -1, head_dim)
        c = cos.unsqueeze(-2).to(x.dtype)
        s = sin.unsqueeze(-2).to(x.dtype)
        x1, x2 = torch.chunk(x, 2, dim=-1)
        return torch.cat((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1).reshape(shape)

    return rot(q), rot(k)


def silu_and_mul(x):
    """layers/activation.py:56-58."""
    d = x.shape[-1] // 2
    return F.silu(x[..., :d]) * x[..., d:]
