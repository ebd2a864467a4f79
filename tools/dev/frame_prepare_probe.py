import sys, time, torch
sys.path.insert(0,'/root/repo')
from uforecon_amd import ops
from uforecon_amd.scene import make_frame
f = make_frame(512, 640, 3, seed=0).to("cuda:0")
def prep():
    return ops.FrameHandle(f.batch, f.source_imgs_feat, f.feature_volume, f.match_feature)
prep(); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): h=prep()
torch.cuda.synchronize()
print("frame_prepare %.3f ms" % ((time.perf_counter()-t0)/10*1e3))
