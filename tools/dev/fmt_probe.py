import sys, time, torch
sys.path.insert(0, '.')
from uforecon_amd import fmt, ops
dev='cuda:0'
torch.manual_seed(0)
m = fmt.FMT_with_pathway().to(dev).eval()
feats=[{"stage1": torch.randn(3,32,128,160,device=dev), "stage2": torch.randn(3,16,256,320,device=dev), "stage3": torch.randn(3,8,512,640,device=dev)} for _ in range(3)]
with torch.no_grad():
    for i in range(2):
        out = m([dict(f) for f in feats], ref_idx=0)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    t=time.perf_counter()
    out = m([dict(f) for f in feats], ref_idx=0)
    torch.cuda.synchronize()
    print("FMT wall ms", (time.perf_counter()-t)*1e3, ops.profile_read())
    ops.profile_enable(False)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        out = m([dict(f) for f in feats], ref_idx=0)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12))
