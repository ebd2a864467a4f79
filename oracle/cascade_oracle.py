"""CPU check harness for the frustum cascade (second part).

TEST INFRASTRUCTURE ONLY.  The product (`uforecon_amd.cascade`) computes step 2 of every cascade stage and the two 3-D
U-Nets with HIP kernels and has no CPU path; for the CPU test-suite this module swaps those calls for CPU restatements
(`oracle.frustum_oracle.correlate`; the torch expressions of the two U-Nets below), so that the mirror's parameter
tree, state_dict keys and stage loop can be checked against the reference's outputs (tests/golden/cascade_*.npz,
produced by the reference's own modules) without a GPU -- and so that the GPU tests have a same-host checker of the
convolution kernel.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

from oracle import frustum_oracle as FO
from uforecon_amd import cascade


def _correlate_cpu(ref_fea, src_feas, ref_proj_pair, src_proj_pairs, depth_values, view_weights=None,
                   want_similarity=True, rel_proj=None):
    sims, agg = FO.correlate(ref_fea, list(src_feas), ref_proj_pair, list(src_proj_pairs), depth_values, view_weights)
    return (sims if want_similarity else None), agg


def _block(blk, x):
    """Conv3d / Deconv3d of the reference: convolution, BatchNorm (eval), ReLU (module.py:134-143, 176-187)."""
    return F.relu(blk.bn(blk.conv(x)))


def cost_reg_net(m, x):
    """CostRegNet.forward, code1/encoder_utils/fmt/module.py:490-500."""
    c0 = _block(m.conv0, x)
    c2 = _block(m.conv2, _block(m.conv1, c0))
    c4 = _block(m.conv4, _block(m.conv3, c2))
    y = _block(m.conv6, _block(m.conv5, c4))
    y = c4 + _block(m.conv7, y)
    y = c2 + _block(m.conv9, y)
    y = c0 + _block(m.conv11, y)
    return m.prob(y)


def cost_reg_net_weight(m, x):
    """CostRegNetWeight.forward, code1/encoder_utils/fmt/module.py:530-543 (no activation between the convolutions)."""
    c0 = m.conv0(x)
    c2 = m.conv2(m.conv1(c0))
    c4 = m.conv4(m.conv3(c2))
    y = m.conv6(m.conv5(c4))
    y = c4 + m.conv7(y)
    y = c2 + m.conv9(y)
    y = c0 + m.conv11(y)
    return m.features(y), torch.sigmoid(m.weights(y))


@contextlib.contextmanager
def cpu_correlate():
    """Inside the block `uforecon_amd.cascade` uses the CPU restatements of the correlate step and of the two U-Nets."""
    saved = cascade.frustum.correlate, cascade.unet3d.cost_reg_net, cascade.unet3d.cost_reg_net_weight
    cascade.frustum.correlate = _correlate_cpu
    cascade.unet3d.cost_reg_net, cascade.unet3d.cost_reg_net_weight = cost_reg_net, cost_reg_net_weight
    try:
        yield
    finally:
        cascade.frustum.correlate, cascade.unet3d.cost_reg_net, cascade.unet3d.cost_reg_net_weight = saved


def run_cascade_cpu(builder: "cascade.FrustumBuilder", case) -> tuple:
    with cpu_correlate(), torch.no_grad():
        return builder(case["features"], case["proj_matrices"], case["depth_values"], case["img_hw"])
