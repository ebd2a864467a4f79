"""The oracle against the LIVE reference (build container only: skipped wherever /root/reference is absent, e.g. on the
GPU box, where the committed golden vectors made from the same reference stand in -- tests/test_oracle_golden.py)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from ref_harness import build_reference_model, ray_path_state_dict, reference_available  # noqa: E402

from helpers import grad_rel_err, rel_err  # noqa: E402
from oracle import ufo_oracle as O  # noqa: E402
from uforecon_amd.scene import make_frame, sampler_uniforms  # noqa: E402

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present on this host")


def test_infer_matches_the_imported_reference():
    model = build_reference_model(3, test_sample_coarse=32, test_sample_fine=32, coarse_sample=32, fine_sample=32, test_n_view=3)
    P = ray_path_state_dict(model)
    fr = make_frame(32, 48, 3, seed=21)
    idx = (torch.arange(12) * 100 + 57)[None]
    with torch.no_grad():
        torch.manual_seed(4)
        srdf, pts, depth, rgb = model.infer(batch=fr.batch, ray_idx=idx, source_imgs_feat=fr.source_imgs_feat,
                                            feature_volume=fr.feature_volume, match_feature=fr.match_feature,
                                            extract_geometry=True, is_train=False)
        U1, U2 = sampler_uniforms(4, 32, 32, 12)
        s2, p2, d2, c2 = O.infer(P, fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2)
    assert rel_err(p2, pts[0]) < 2e-6 and rel_err(d2, depth[0]) < 2e-6 and rel_err(c2, rgb[0]) < 2e-6
    assert rel_err(s2, srdf[0]) < 1e-5


def test_training_gradients_match_the_imported_reference():
    model = build_reference_model(3, test_sample_coarse=32, test_sample_fine=32, coarse_sample=32, fine_sample=32,
                                  test_n_view=3, extract_geometry=False)
    model.train()
    fr = make_frame(32, 48, 3, seed=22, train_layout=True)
    idx = (torch.arange(10) * 120 + 250)[None]
    torch.manual_seed(5)
    r = model.infer(batch=fr.batch, ray_idx=idx, source_imgs_feat=fr.source_imgs_feat, feature_volume=fr.feature_volume,
                    match_feature=fr.match_feature)
    d = dict(rgb=r[1][0], depth=r[2][0], rgb_2=r[8][0], depth_2=r[9][0])
    loss_ref = O.training_loss(d, fr.batch, idx)
    loss_ref.backward()
    P = {k: v.clone().requires_grad_("depthcode" not in k) for k, v in ray_path_state_dict(model).items()}
    U1, U2 = sampler_uniforms(5, 32, 32, 10)
    ro = O.infer(P, fr.batch, idx, fr.source_imgs_feat, fr.feature_volume, fr.match_feature, U1, U2, extract_geometry=False)
    loss = O.training_loss(ro, fr.batch, idx)
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    loss.backward()
    ref = dict(model.named_parameters())
    for k, p in P.items():
        if p.requires_grad:
            assert grad_rel_err(p.grad, ref[k].grad) < 2e-3, k      # a ReLU unit within rounding of 0 may flip (see make_golden)
