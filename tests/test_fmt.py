"""FMT + get_match_feat mirror vs the outputs of the reference's own FMT_with_pathway."""
import os

import numpy as np
import pytest
import torch

from uforecon_amd import cascade
from uforecon_amd.scene import FMT_CASES, fill_state_dict, make_fmt_case

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(dev):
    if dev == "cpu":                # the product has no CPU path: the layers run through the CPU restatement (oracle/)
        from oracle import fmt_oracle

        with fmt_oracle.cpu_layers():
            return _run_on(dev)
    return _run_on(dev)


def _run_on(dev):
    c = make_fmt_case("small3")
    m = fill_state_dict(cascade.FrustumBuilder(), c["weight_seed"]).eval().to(dev)
    feats = [{k: v.to(dev) for k, v in f.items()} for f in c["features"]]
    with torch.no_grad():
        enc = m.transmvsnet.encode(feats, ref_idx=0)
        got = {f"view{v}.{st}": f[st].cpu() for v, f in enumerate(enc) for st in ("stage1", "stage2", "stage3")}
        for f in enc:
            f["stage1"] = f["stage1"][0:1]                      # model.py:782-783
        got["match_feature"] = m.transmvsnet.get_match_feat(enc, cur_n_src_views=c["NV"])[0].cpu()
    return got


def test_state_dict_keys_include_the_references_fmt():
    keys = set(cascade.FrustumBuilder().state_dict())
    for k in ("transmvsnet.FMT_with_pathway.FMT.layers.7.attention.out_projection.bias",
              "transmvsnet.FMT_with_pathway.FMT.layers.0.norm2.weight", "transmvsnet.FMT_with_pathway.smooth_2.weight",
              "transmvsnet.FMT_with_pathway.dim_reduction_1.weight"):
        assert k in keys, k
    assert not any("pos_encoding" in k for k in keys)           # non-persistent buffer, as in the reference


def test_fmt_mirror_matches_reference_golden_cpu():
    g = np.load(os.path.join(HERE, "golden", "fmt_small3.npz"))
    got = _run("cpu")
    assert set(got) == set(g.files)
    for k in g.files:                          # the mirror's walks of the stack around the CPU restatement of a layer
        ref = torch.from_numpy(g[k])
        assert float((got[k] - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), k
    assert got["match_feature"].shape == (1, 3, 64, 8, 16)      # (B, V, 32 (V-1), h, w): ufr_frame_prepare's match_feature


def test_fmt_product_has_no_cpu_path():
    from uforecon_amd import fmt
    from uforecon_amd.ops import UfrError

    with pytest.raises(UfrError), torch.no_grad():
        fmt._layer(fmt._LayerParams(32), torch.randn(1, 40, 32), None)


@pytest.mark.gpu
def test_fmt_mirror_matches_reference_golden_gpu():
    g = np.load(os.path.join(HERE, "golden", "fmt_small3.npz"))
    got = _run("cuda:0")
    for k in g.files:
        ref = torch.from_numpy(g[k])
        assert float((got[k] - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), k


@pytest.mark.gpu
@pytest.mark.parametrize("cross", [False, True])
def test_fmt_layer_kernel_matches_the_torch_layer(cross):
    """ufr_fmt_layer (csrc/fmt.hip) against the same layer written with torch ops on the CPU (oracle/fmt_oracle.py), self-
    and cross-attention, token counts that are not multiples of any tile."""
    from oracle import fmt_oracle
    from uforecon_amd import fmt

    torch.manual_seed(3)
    p = fmt._LayerParams(32)
    with torch.no_grad():
        for q in p.parameters():
            q.copy_(torch.randn(q.shape) * (0.3 if q.dim() > 1 else 0.1))
        p.norm1.weight.add_(1.0)
        p.norm2.weight.add_(1.0)
    x = torch.randn(3, 1237, 32)
    src = torch.randn(3, 2049, 32) if cross else None
    with torch.no_grad():
        want = fmt_oracle.layer(p, x, src)
        got = fmt._layer(p.to("cuda:0"), x.to("cuda:0"), None if src is None else src.to("cuda:0"))
    assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
