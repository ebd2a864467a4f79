"""Development probe: encode_frame alone, a few repetitions (for rocprofv3 --kernel-trace --stats)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import pipeline  # noqa: E402
from uforecon_amd.evalset import make_eval_batch  # noqa: E402
from uforecon_amd.scene import fill_state_dict  # noqa: E402

dev = "cuda:0"
args = argparse.Namespace(extract_geometry=True, test_sample_coarse=64, test_sample_fine=64, coarse_sample=64, fine_sample=64,
                          volume_type="correlation", volume_reso=96, mvs_depth_guide=1, depth_pos_encoding=True,
                          use_dir_srdf=False, explicit_similarity=True, test_coarse_only=False, test_ray_num=800,
                          test_n_view=3, out_dir=None)
net = fill_state_dict(pipeline.UFOReconInference(args, tune_convolutions=False), 21).eval().to(dev)
batch = make_eval_batch(512, 640, 3, 0, dev)
with torch.no_grad():
    for _ in range(int(os.environ.get("REPS", 6))):
        net.encode_frame(batch)
torch.cuda.synchronize()
