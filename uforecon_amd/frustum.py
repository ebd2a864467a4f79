"""Host side of the correlation-volume construction step (first part).

Mirrors step 2 of the reference's ``DepthNet.forward`` (code1/encoder_utils/fmt/TransMVSNet.py:66-97) and
``homo_warping_trans`` (code1/encoder_utils/fmt/module.py:329-367) for one frame (B = 1): the per-view similarity
volumes and their pixel-wise weighted aggregate come out of ONE fused HIP kernel (csrc/frustum.hip); the
(C, D, H, W) warped volume of the reference is never built.  There is no CPU fallback: CPU tensors or a missing
libufr.so raise ``UfrError``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from .ops import _dev, _opt, _stream


def fold_projection(proj_pair: torch.Tensor) -> torch.Tensor:
    """(2,4,4) [extrinsic, intrinsic] -> 4x4 whose top 3x4 is K[:3,:3] @ E[:3,:4]  (TransMVSNet.py:73-76)."""
    pp = proj_pair[None]          # batched (B=1) matmul like the reference: the 2-D code path rounds differently
    out = pp[:, 0].clone()
    out[:, :3, :4] = torch.matmul(pp[:, 1, :3, :3], pp[:, 0, :3, :4])
    return out[0]


def relative_projections(ref_proj_pair: torch.Tensor, src_proj_pairs: Sequence[torch.Tensor]) -> torch.Tensor:
    """Host (CPU, fp32) array (NS,12): rows of src_proj_new @ inverse(ref_proj_new) (module.py:340-342).  Twelve
    numbers per view: computed with the same torch calls as the reference, then passed to the kernel by value."""
    ref_new = fold_projection(ref_proj_pair.detach().float().cpu())
    inv = torch.inverse(ref_new[None])
    rows = [torch.matmul(fold_projection(pp.detach().float().cpu())[None], inv)[0, :3, :4].reshape(12)
            for pp in src_proj_pairs]
    return torch.stack(rows).contiguous()


def correlate(ref_fea: torch.Tensor, src_feas: torch.Tensor, ref_proj_pair: torch.Tensor,
              src_proj_pairs: Sequence[torch.Tensor], depth_values: torch.Tensor,
              view_weights: Optional[torch.Tensor] = None, want_similarity: bool = True,
              rel_proj: Optional[torch.Tensor] = None):
    """ref_fea (C,H,W), src_feas (NS,C,H,W), depth_values (D,H,W), view_weights (NS,H,W) or None: device fp32.
    rel_proj: optional precomputed (NS,12) host array replacing relative_projections(...) -- a 4x4 inverse in fp32
    comes out differently on different CPUs (LAPACK code paths) and white-noise test features amplify that to 1e-5
    in the similarity, so the golden fixtures carry the reference host's matrices.
    Returns (similarity (NS,D,H,W) or None, aggregated (D,H,W) or None)."""
    lib = _lib.load()
    Cc, H, W = ref_fea.shape
    NS = src_feas.shape[0]
    D = depth_values.shape[0]
    if tuple(src_feas.shape[1:]) != (Cc, H, W) or tuple(depth_values.shape[1:]) != (H, W) or len(src_proj_pairs) != NS:
        raise _lib.UfrError("correlate: inconsistent shapes")
    if view_weights is not None and tuple(view_weights.shape) != (NS, H, W):
        raise _lib.UfrError("correlate: view_weights must be (NS,H,W)")
    dev = ref_fea.device
    rel = (relative_projections(ref_proj_pair, src_proj_pairs) if rel_proj is None
           else rel_proj.detach().float().cpu().reshape(NS, 12).contiguous())
    sim = torch.empty(NS, D, H, W, dtype=torch.float32, device=dev) if want_similarity else None
    agg = torch.empty(D, H, W, dtype=torch.float32, device=dev) if view_weights is not None else None
    nbytes = lib.ufr_correlate_workspace_bytes(Cc, H, W, NS)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    _lib.check(lib.ufr_frustum_correlate(
        _dev(ref_fea, "ref_fea"), _dev(src_feas, "src_feas"), rel.numpy().ctypes.data_as(C.POINTER(C.c_float)),
        _dev(depth_values, "depth_values"), _opt(view_weights, "view_weights"), Cc, H, W, D, NS,
        _opt(sim, "similarity"), _opt(agg, "aggregated"), ws.data_ptr(), nbytes, _stream()), "ufr_frustum_correlate")
    return sim, agg


def pixelwise_params(net) -> torch.Tensor:
    """The 185 floats ufr_pixelwise_view_weights reads, from a PixelwiseNet (cascade.py; eval-mode BatchNorms folded), cached
    on the module until one of its tensors changes."""
    key = tuple((t.data_ptr(), t._version) for t in list(net.parameters()) + list(net.buffers()))
    cached = getattr(net, "_ufr_params", None)
    if cached is not None and cached[0] == key:
        return cached[1]

    def fold(bn):
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        return scale, bn.bias.detach() - bn.running_mean * scale

    s0, h0 = fold(net.conv0.bn)
    s1, h1 = fold(net.conv1.bn)
    p = torch.cat([net.conv0.conv.weight.detach().reshape(16) * s0, h0, net.conv1.conv.weight.detach().reshape(128), s1, h1,
                   net.conv2.weight.detach().reshape(8), net.conv2.bias.detach().reshape(1)]).float().contiguous()
    net._ufr_params = (key, p)
    return p


def view_weights(net, similarity: torch.Tensor, want_aggregate: bool = True):
    """PixelwiseNet on every source view's similarity volume and the weighted aggregate (TransMVSNet.py:80-97) in one pass:
    ``similarity`` (NS,D,H,W) -> ``view_weights (NS,H,W)``, ``aggregated (D,H,W)`` (or None)."""
    if net.training:
        raise _lib.UfrError("view_weights: inference only (BatchNorm in eval mode)")
    NS, D, H, W = similarity.shape
    dev = similarity.device
    vw = torch.empty(NS, H, W, dtype=torch.float32, device=dev)
    agg = torch.empty(D, H, W, dtype=torch.float32, device=dev) if want_aggregate else None
    p = pixelwise_params(net)
    _lib.check(_lib.load().ufr_pixelwise_view_weights(_dev(similarity, "similarity"), _dev(p, "params"), vw.data_ptr(),
                                                      _opt(agg, "aggregated"), NS, D, H, W, _stream()), "ufr_pixelwise_view_weights")
    return vw, agg
