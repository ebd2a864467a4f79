"""TSDF fusion: oracle vs the reference's CPU-mode outputs (CPU), HIP kernel vs oracle (GPU)."""
import os

import numpy as np
import pytest

from oracle import tsdf_oracle as T
from uforecon_amd.scene import TSDF_CASES, make_tsdf_case

HERE = os.path.dirname(os.path.abspath(__file__))


def _oracle_volumes(c, integrate_color=False):
    b = T.volume_bounds(c["depths"], c["intrinsics"], c["poses"])
    dim, org = T.volume_layout(b, c["voxel_size"])
    tsdf, w, col = np.ones(dim, np.float32), np.zeros(dim, np.float32), np.zeros(dim, np.float32)
    for d, rgb, K, P in zip(c["depths"], c["colors"], c["intrinsics"], c["poses"]):
        T.integrate(tsdf, w, col, org, c["voxel_size"], c["margin"] * c["voxel_size"], K, P, d,
                    color_folded=T.fold_color(rgb), integrate_color=integrate_color)
    return b, dim, org, tsdf, w, col


@pytest.mark.parametrize("name", list(TSDF_CASES))
def test_oracle_matches_reference_cpu_mode(name):
    c = make_tsdf_case(name)
    g = np.load(os.path.join(HERE, "golden", f"tsdf_{name}.npz"))
    assert abs(sum(float(np.abs(d).sum()) for d in c["depths"]) - float(g["input_digest"])) < 1e-6 * float(g["input_digest"])
    b, dim, org, tsdf, w, col = _oracle_volumes(c)
    assert np.array_equal(b, g["vol_bnds"]) and np.array_equal(dim, g["vol_dim"]) and np.array_equal(org, g["vol_origin"])
    # the reference's GPU kernel (restated by the oracle, fp32) and its CPU mode (the golden: fp64 transform, np.round)
    # may disagree on voxels that sit on a pixel-rounding or truncation boundary: at most a handful
    disagree = w != g["weight"]
    assert disagree.sum() <= max(2, int(2e-4 * w.size)), int(disagree.sum())
    assert np.abs(tsdf - g["tsdf"])[~disagree].max() < 1e-5
    assert not col.any() and not g["color"].any()          # neither reference path integrates colour


def test_oracle_edge_cases():
    dim = np.array([4, 3, 5])
    tsdf, w, col = np.ones(dim, np.float32), np.zeros(dim, np.float32), np.zeros(dim, np.float32)
    K = np.array([[10, 0, 4], [0, 10, 3], [0, 0, 1]], np.float32)
    P = np.eye(4, dtype=np.float32)
    # all-zero depth image: nothing observed; camera inside the volume with voxels at z = 0 and behind it
    assert T.integrate(tsdf, w, col, np.array([-0.2, -0.1, -0.2], np.float32), 0.1, 0.3, K, P, np.zeros((6, 8), np.float32)) == 0
    n = T.integrate(tsdf, w, col, np.array([-0.2, -0.1, -0.2], np.float32), 0.1, 0.3, K, P, np.full((6, 8), 0.15, np.float32))
    assert n > 0 and np.isfinite(tsdf).all() and (w[:, :, :2] == 0).all()     # z <= 0 planes never updated


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TSDF_CASES))
@pytest.mark.parametrize("color", [False, True])
def test_hip_integrate_matches_oracle_bit_for_bit(name, color):
    from uforecon_amd import tsdf as H

    c = make_tsdf_case(name)
    b, dim, org, tsdf, w, col = _oracle_volumes(c, integrate_color=color)
    vol = H.TSDFVolume(b.copy(), voxel_size=c["voxel_size"], margin=c["margin"], integrate_color=color)
    assert np.array_equal(vol._vol_dim, dim) and np.array_equal(vol._vol_origin, org)
    for d, rgb, K, P in zip(c["depths"], c["colors"], c["intrinsics"], c["poses"]):
        vol.integrate(rgb, d, K, P, obs_weight=1.0)
    t, cc, ww = vol.get_volume()
    assert np.array_equal(ww, w)
    assert np.array_equal(t, tsdf)
    assert np.array_equal(cc, col) and (cc.any() == color)


@pytest.mark.gpu
def test_fuse_depth_maps_matches_reference_golden():
    from uforecon_amd import tsdf as H

    c = make_tsdf_case("sphere3")
    g = np.load(os.path.join(HERE, "golden", "tsdf_sphere3.npz"))
    vol = H.fuse_depth_maps(c["depths"], c["intrinsics"], [np.linalg.inv(P) for P in c["poses"]],
                            voxel_size=c["voxel_size"], margin=c["margin"])
    t, cc, ww = vol.get_volume()
    disagree = ww != g["weight"]
    assert disagree.sum() <= max(2, int(2e-4 * ww.size))
    assert np.abs(t - g["tsdf"])[~disagree].max() < 1e-5
    # the fused surface is the sphere: zero crossing of the TSDF along +x through the centre lies at radius 0.8
    assert (ww > 0).sum() > 1000
