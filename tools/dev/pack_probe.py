"""Time of ufr_weights_pack: back to back (GPU busy) and after idle gaps (what a training step sees)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import ops
DEV = "cuda:0"
wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
W = ops.PackedWeights({k: torch.from_numpy(wz[k]).to(DEV) for k in wz.files})
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200):
    W.repack()
b.record(); torch.cuda.synchronize()
print(f"back to back: {a.elapsed_time(b) / 200 * 1e3:.1f} us per pack")
ts = []
for _ in range(50):
    time.sleep(0.002)
    a.record(); W.repack(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
print(f"after 2 ms idle: median {sorted(ts)[25]:.1f} us per pack")
