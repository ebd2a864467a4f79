"""CPU restatement of one layer of the feature-matching transformer.

TEST INFRASTRUCTURE ONLY: the product (`uforecon_amd.fmt`) runs every layer as the HIP kernel pair behind `ufr_fmt_layer`
and has no CPU path.  This is the torch expression of the same layer -- what the kernel is checked against
(tests/test_fmt.py) and what the CPU test-suite swaps in to check the mirror's parameter tree and its three walks of the
stack against the reference's own outputs (tests/golden/fmt_small3.npz: pinned).

  layer               FMT.py:99-113  (post-norm, dropout 0)
  linear_attention    FMT.py:25-38   (feature map elu(.) + 1, 8 heads of 4, eps 1e-6)
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch
import torch.nn.functional as F

HEADS = 8


def _split_heads(t: torch.Tensor) -> torch.Tensor:
    """(N, T, C) -> (N*HEADS, T, C/HEADS)"""
    n, t_, c = t.shape
    return t.view(n, t_, HEADS, c // HEADS).permute(0, 2, 1, 3).reshape(n * HEADS, t_, c // HEADS)


def linear_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """out_l = phi(q_l) (sum_s phi(k_s)^T v_s) / (phi(q_l) . sum_s phi(k_s) + eps), phi = elu + 1 (FMT.py:25-38)."""
    n, tq, c = q.shape
    qh, kh, vh = _split_heads(F.elu(q) + 1), _split_heads(F.elu(k) + 1), _split_heads(v)
    state = torch.bmm(kh.transpose(1, 2), vh)                      # (N*H, 4, 4)   sum_s phi(k)^T v
    norm = torch.bmm(qh, kh.sum(dim=1, keepdim=True).transpose(1, 2)) + eps   # (N*H, T, 1)
    out = torch.bmm(qh, state) / norm
    return out.view(n, HEADS, tq, c // HEADS).permute(0, 2, 1, 3).reshape(n, tq, c)


def layer(p, x: torch.Tensor, src: Optional[torch.Tensor]) -> torch.Tensor:
    """One post-norm layer (FMT.py:99-113); ``p``: a `uforecon_amd.fmt._LayerParams`; ``src is None``: self-attention."""
    a = p.attention
    q = F.linear(x, a.query_projection.weight, a.query_projection.bias)
    kv = x if src is None else src
    k = F.linear(kv, a.key_projection.weight, a.key_projection.bias)
    v = F.linear(kv, a.value_projection.weight, a.value_projection.bias)
    x = F.layer_norm(x + F.linear(linear_attention(q, k, v), a.out_projection.weight, a.out_projection.bias),
                     x.shape[-1:], p.norm1.weight, p.norm1.bias)
    y = F.linear(F.relu(F.linear(x, p.linear1.weight, p.linear1.bias)), p.linear2.weight, p.linear2.bias)
    return F.layer_norm(x + y, x.shape[-1:], p.norm2.weight, p.norm2.bias)


@contextlib.contextmanager
def cpu_layers():
    """Inside the block `uforecon_amd.fmt` evaluates its layers with the CPU restatement above."""
    from uforecon_amd import fmt

    saved = fmt._layer
    fmt._layer = layer
    try:
        yield
    finally:
        fmt._layer = saved
