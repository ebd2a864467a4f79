"""Import the upstream UFORecon per-ray path on CPU (THIS container only).

Used only by ``make_golden.py`` and by ``tests/test_oracle_vs_reference.py`` (which is
skipped wherever /root/reference is absent, e.g. on the GPU box).  Nothing of the
reference travels: only the numeric vectors this produces are committed.

The reference needs pytorch_lightning / torchvision / kornia / cv2 / piq / mcubes /
easydict at *import* time only; none of them is touched by the per-ray path, so empty
stand-in modules are registered for the import.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "code1"))


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def import_reference():
    if not reference_available():
        raise RuntimeError("reference tree not present")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "pytorch_lightning" not in sys.modules:
        _stub("easydict", EasyDict=dict)
        tv = _stub("torchvision")
        tv.ops = _stub("torchvision.ops", DeformConv2d=object, deform_conv2d=None)
        k = _stub("kornia")
        k.utils = _stub("kornia.utils", create_meshgrid=None)
        _stub("piq", psnr=None)
        _stub("cv2")
        _stub("mcubes")

        class LightningModule(nn.Module):
            def log(self, *a, **k):
                pass

        _stub("pytorch_lightning", LightningModule=LightningModule)
    from code1.model import UFORecon  # noqa: E402

    return UFORecon


def reference_args(**over):
    a = dict(
        patch_size=48, sW=1, sH=1, train_ray_num=1024, extract_geometry=True,
        test_sample_coarse=64, test_sample_fine=64, coarse_sample=64, fine_sample=64,
        ndepths="48,32,8", depth_inter_r="4,2,1", share_cr=False, cr_base_chs="8,8,8",
        grad_method="detach", volume_type="correlation", volume_reso=96, mvs_depth_guide=1,
        depth_pos_encoding=True, use_dir_srdf=False, explicit_similarity=True,
        test_coarse_only=False, test_ray_num=800, test_n_view=3, train_n_view=5,
        uforecon_lr=1e-4, weight_rgb=1.0, weight_depth=1.0, logdir=".", out_dir=".")
    a.update(over)
    return argparse.Namespace(**a)


def build_reference_model(weight_seed: int = 0, **over):
    UFORecon = import_reference()
    torch.manual_seed(weight_seed)
    model = UFORecon(reference_args(**over)).eval()
    return model


RAY_PATH_PREFIXES = ("ray_transformer.", "deviation_network.")


def ray_path_state_dict(model) -> dict:
    """The per-ray parameters (148 947 floats) under their reference key names."""
    return {k: v.detach().clone() for k, v in model.state_dict().items()
            if k.startswith(RAY_PATH_PREFIXES)}
