"""Golden vectors of TSDF fusion by RUNNING THE REFERENCE's TSDFVolume (build container only):
``python tests/golden/make_golden_tsdf.py`` -> tests/golden/tsdf_*.npz.

pycuda and numba are absent here, so the reference's CPU mode is used (tsdf_fusion.py:280-310, `use_gpu=False`) with
`numba.njit` / `prange` replaced by the identity / `range`: its three jitted helpers are plain scalar loops and run
unchanged as Python.  The reference's GPU kernel (tsdf_fusion.py:77-152, what production runs) differs from its own CPU
mode in four documented ways -- fp32 instead of fp64 camera transform, `roundf` instead of `np.round`, `z < 0` instead
of `z <= 0` rejected (colour is integrated by neither: early `return` there, IndexError here) -- so the fixtures pin the algorithm, and the
voxels on which the two reference paths may legitimately disagree are bounded in tests/test_tsdf.py.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")


def import_reference_tsdf():
    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    stub("numba", njit=njit, prange=range)
    sk = stub("skimage")
    sk.measure = stub("skimage.measure")
    stub("cv2")
    pc = stub("pycuda")
    pc.driver = stub("pycuda.driver")
    pc.autoinit = stub("pycuda.autoinit")
    pc.compiler = stub("pycuda.compiler", SourceModule=None)
    sys.path.insert(0, "/root/reference")
    import tsdf_fusion

    return tsdf_fusion


def run(name):
    from uforecon_amd.scene import make_tsdf_case

    T = import_reference_tsdf()
    c = make_tsdf_case(name)
    vol_bnds = np.zeros((3, 2))
    for depth, K, pose in zip(c["depths"], c["intrinsics"], c["poses"]):     # save_tsdf:459-472
        pts = T.get_view_frustum(depth, K, pose)
        vol_bnds[:, 0] = np.minimum(vol_bnds[:, 0], np.amin(pts, axis=1))
        vol_bnds[:, 1] = np.maximum(vol_bnds[:, 1], np.amax(pts, axis=1))
    vol = T.TSDFVolume(vol_bnds.copy(), voxel_size=c["voxel_size"], use_gpu=False, margin=c["margin"])
    for depth, col, K, pose in zip(c["depths"], c["colors"], c["intrinsics"], c["poses"]):
        try:
            vol.integrate(col, depth, K, pose, obs_weight=1.0)                # save_tsdf:496
        except IndexError:
            # upstream bug: CPU mode flattens color_im (tsdf_fusion.py:237) and then indexes it [y, x] (:303).  The TSDF
            # and weight volumes are already updated in place at that point (:290-296); the colour volume is not --
            # exactly like the GPU kernel, whose colour block sits behind an early `return` (:139).
            pass
    tsdf, color, weight = vol.get_volume()
    np.savez_compressed(os.path.join(HERE, f"tsdf_{name}.npz"), vol_bnds=vol_bnds, vol_dim=vol._vol_dim,
                        vol_origin=vol._vol_origin, tsdf=tsdf.astype(np.float32), weight=weight.astype(np.float32),
                        color=color.astype(np.float32),
                        input_digest=np.float64(sum(float(np.abs(d).sum()) for d in c["depths"])))
    print(name, "dim", vol._vol_dim, "observed voxels", int((weight > 0).sum()), "of", weight.size)


if __name__ == "__main__":
    from uforecon_amd.scene import TSDF_CASES

    for n in TSDF_CASES:
        run(n)
