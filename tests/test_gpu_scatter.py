"""The frustum scatter's workspace contract (ufr_project_gather_bwd, include/ufr.h): the call leaves the record volume
zero, so a kept workspace is zero-filled once; results with a kept workspace equal those with a library-zeroed one, call
after call, in both the overwrite and the accumulate mode."""
import pytest
import torch

from uforecon_amd import ops
from uforecon_amd.scene import make_frame

from helpers import load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(fh, seed, RN=96, SN=64):
    g = torch.Generator().manual_seed(seed)
    ray_o = torch.tensor([0.05, -0.02, -1.1])
    d = torch.randn(RN, 3, generator=g) * 0.25 + torch.tensor([0.0, 0.0, 1.0])
    ray_d = d / d.norm(dim=1, keepdim=True)
    z = torch.sort(0.5 + 1.4 * torch.rand(RN, SN, generator=g), dim=1).values
    return ray_o.to(DEV), ray_d.to(DEV).contiguous(), z.to(DEV).contiguous(), (torch.rand(RN * SN, 40, generator=g) - 0.5).to(DEV), \
        torch.rand(RN * SN, 8, generator=g).to(DEV)


def test_kept_workspace_is_left_zero_and_gives_the_same_gradients():
    NV = 3
    fr = make_frame(64, 80, NV, seed=3, train_layout=True).to(DEV)
    fh = ops.FrameHandle(fr.batch, fr.source_imgs_feat, fr.feature_volume, fr.match_feature)
    W = ops.PackedWeights({k: v.to(DEV) for k, v in load_weights().items()})
    shapes_f = [tuple(fr.feature_volume[st]["feature_volume"].shape) for st in ("stage1", "stage2", "stage3")]
    shapes_w = [tuple(fr.feature_volume[st]["weight_volume"].shape) for st in ("stage1", "stage2", "stage3")]
    kept = torch.zeros(ops.project_gather_bwd_workspace_floats(fh), device=DEV)       # zero-filled ONCE
    acc_f = [torch.zeros(s, device=DEV) for s in shapes_f]
    acc_w = [torch.zeros(s, device=DEV) for s in shapes_w]
    sum_f = [torch.zeros(s, device=DEV) for s in shapes_f]
    sum_w = [torch.zeros(s, device=DEV) for s in shapes_w]
    touched = []
    for seed in (1, 2, 3):
        ray_o, ray_d, z, d_pv, sim8 = _inputs(fh, seed)
        # library-zeroed private workspace, outputs overwritten
        ref_f = [torch.full(s, 7.0, device=DEV) for s in shapes_f]
        ref_w = [torch.full(s, 7.0, device=DEV) for s in shapes_w]
        ops.project_gather_bwd(fh, W, ops.GradBuffer(DEV), ray_o, ray_d, z, sim8, d_pv, ref_f, ref_w, accumulate=False)
        # kept workspace, never zeroed again
        got_f = [torch.full(s, -3.0, device=DEV) for s in shapes_f]
        got_w = [torch.full(s, -3.0, device=DEV) for s in shapes_w]
        ops.project_gather_bwd(fh, W, ops.GradBuffer(DEV), ray_o, ray_d, z, sim8, d_pv, got_f, got_w, accumulate=False,
                               zeroed_workspace=kept)
        assert int(torch.count_nonzero(kept)) == 0                                  # left zero: records and marks
        for a, b in zip(got_f + got_w, ref_f + ref_w):      # (float atomics: the order of a voxel's additions varies from run to run)
            assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-12)
            assert torch.equal(a == 0, b == 0)
        touched.append(float(sum(int(torch.count_nonzero(t)) for t in ref_w)) / sum(t.numel() for t in ref_w))
        # accumulate mode on the kept workspace: += of every call
        ops.project_gather_bwd(fh, W, ops.GradBuffer(DEV), ray_o, ray_d, z, sim8, d_pv, acc_f, acc_w, accumulate=True,
                               zeroed_workspace=kept)
        assert int(torch.count_nonzero(kept)) == 0
        for t, r in zip(sum_f + sum_w, ref_f + ref_w):
            t += r
    for a, b in zip(acc_f + acc_w, sum_f + sum_w):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-12)
    assert 0.0 < min(touched) and max(touched) < 0.9        # the rays reach part of the volume: both branches of the unpack ran
