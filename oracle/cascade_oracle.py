"""CPU check harness for the frustum cascade (second part).

TEST INFRASTRUCTURE ONLY.  The product (`uforecon_amd.cascade`) computes step 2 of every cascade stage with the HIP
kernel and has no CPU path; for the CPU test-suite this module swaps that one call for the CPU restatement
`oracle.frustum_oracle.correlate`, so that the mirror's modules, state_dict keys and stage loop can be checked against
the reference's outputs (tests/golden/cascade_*.npz, produced by the reference's own modules) without a GPU.
"""
from __future__ import annotations

import contextlib

import torch

from oracle import frustum_oracle as FO
from uforecon_amd import cascade


def _correlate_cpu(ref_fea, src_feas, ref_proj_pair, src_proj_pairs, depth_values, view_weights=None,
                   want_similarity=True, rel_proj=None):
    sims, agg = FO.correlate(ref_fea, list(src_feas), ref_proj_pair, list(src_proj_pairs), depth_values, view_weights)
    return (sims if want_similarity else None), agg


@contextlib.contextmanager
def cpu_correlate():
    """Inside the block `uforecon_amd.cascade` uses the CPU restatement of the correlate step."""
    saved = cascade.frustum.correlate
    cascade.frustum.correlate = _correlate_cpu
    try:
        yield
    finally:
        cascade.frustum.correlate = saved


def run_cascade_cpu(builder: "cascade.FrustumBuilder", case) -> tuple:
    with cpu_correlate(), torch.no_grad():
        return builder(case["features"], case["proj_matrices"], case["depth_values"], case["img_hw"])
