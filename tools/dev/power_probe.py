"""Is the view transformer power-limited?  Loops ufr_aggregate for ~10 s per setting while a thread samples rocm-smi
(package power, sclk); prints kernel time next to the samples.  UFR_LIB / UFR_VT_PAD_LDS select the variant."""
import os, subprocess, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uforecon_amd import ops
from uforecon_amd.scene import make_frame
DEV = "cuda:0"
RN, SN, NV = 4096, 128, 3
wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
W = ops.PackedWeights({k: torch.from_numpy(wz[k]).to(DEV) for k in wz.files})
fr = make_frame(512, 640, NV, 0).to(DEV)
fh = ops.FrameHandle(fr.batch, fr.source_imgs_feat, fr.feature_volume, fr.match_feature)
idx = torch.arange(RN, device=DEV) + 200 * 640
ray_d = fr.batch["ray_d"][0][:, idx].t().contiguous()
ray_o = fr.batch["ray_o"][0].contiguous()
cz = fr.batch["cam_ray_d"][0][2, idx]
near = (fr.batch["near_fars"][0, 0, 0] / cz).contiguous()
far = (fr.batch["near_fars"][0, 0, 1] / cz).contiguous()
z = ops.sample_fixed(near, far, torch.rand(SN, RN, device=DEV))
x, rgbm, dirs, _ = ops.project_gather(fh, W, ray_o, ray_d, z)
lib = ops._lib.load()
P = RN * SN
radiance = torch.empty(P, 3, device=DEV)
srdf = torch.empty(RN, SN, device=DEV)
ws = torch.empty(lib.ufr_aggregate_workspace_bytes(RN, SN, NV) // 4, device=DEV)
samples = []
stop = False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        pw = [l.split(":")[-1].strip() for l in r.splitlines() if "Power (W)" in l]
        sc = [l.split("(")[-1].strip(")") for l in r.splitlines() if "sclk" in l]
        samples.append((time.time(), pw, sc))
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
def loop(seconds, what):
    t0 = time.time(); n = 0
    ops.profile_enable(True)
    while time.time() - t0 < seconds:
        for _ in range(20):
            lib.ufr_aggregate(W.packed.data_ptr(), x.data_ptr(), rgbm.data_ptr(), dirs.data_ptr(), RN, SN, NV,
                              radiance.data_ptr(), srdf.data_ptr(), ws.data_ptr(), None, None, -1, ops._stream())
        torch.cuda.synchronize(); n += 20
    prof = ops.profile_read(); ops.profile_enable(False)
    vt = prof["view_transformer"]["ms"] / prof["view_transformer"]["launches"]
    rt = prof["ray_transformer"]["ms"] / prof["ray_transformer"]["launches"]
    s = [x for x in samples if x[0] >= t0 + 2]
    print(f"{what}: view {vt:.3f} ms ray {rt:.3f} ms over {n} launches; samples: {[(a[1], a[2]) for a in s[-6:]]}", flush=True)
time.sleep(3)
print("idle samples:", [(a[1], a[2]) for a in samples[-3:]], flush=True)
loop(6, os.environ.get("UFR_LIB", "default").split("/")[-1] + " pad=" + os.environ.get("UFR_VT_PAD_LDS", "0"))
if os.environ.get("ZERO_TEST"):
    xs = x.clone()
    x.zero_()
    loop(6, "tokens = 0")
    x.copy_(xs)
    W.packed.zero_()
    loop(6, "weights = 0 (whole packed blob)")
stop = True; th.join()
