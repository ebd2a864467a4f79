"""Development micro-benchmark: time the three heavy kernels alone on configs[1] shapes.
UFR_LIB=<path> selects a variant build.  Prints one line per kernel (median of reps)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uforecon_amd import ops  # noqa: E402
from uforecon_amd.scene import make_frame  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    if os.environ.get("UFR_PRECISION") == "16":   # the reduced-precision (one 16-bit plane) mode of include/ufr.h
        ops.set_matrix_precision(ops.PRECISION_16BIT)
    RN = int(os.environ.get("RN", 4096))
    SN = int(os.environ.get("SN", 128))
    NV = int(os.environ.get("NV", 3))
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
    P = RN * SN
    lib = ops._lib.load()
    import ctypes as C
    radiance = torch.empty(P, 3, device=DEV)
    srdf = torch.empty(RN, SN, device=DEV)
    ws = torch.empty(lib.ufr_aggregate_workspace_bytes(RN, SN, NV) // 4, device=DEV)
    tag = os.environ.get("UFR_LIB", "default").split("/")[-1]
    t = timeit(lambda: ops.project_gather(fh, W, ray_o, ray_d, z))
    print(f"[{tag}] gather            {t:8.3f} ms  {P / t / 1e6:8.2f} Gpt/s... ({P} points)")
    ops.profile_enable(True)
    for _ in range(5):
        lib.ufr_aggregate(W.packed.data_ptr(), x.data_ptr(), rgbm.data_ptr(), dirs.data_ptr(), RN, SN, NV,
                          radiance.data_ptr(), srdf.data_ptr(), ws.data_ptr(), None, None, -1, ops._stream())
    torch.cuda.synchronize()
    prof = ops.profile_read()
    ops.profile_enable(False)
    vt = prof["view_transformer"]["ms"] / prof["view_transformer"]["launches"]
    rt = prof["ray_transformer"]["ms"] / prof["ray_transformer"]["launches"]
    print(f"[{tag}] view_transformer  {vt:8.3f} ms  {534224 * P / vt / 1e9:8.2f} TFLOP/s algorithmic")
    print(f"[{tag}] ray_transformer   {rt:8.3f} ms  {(61952 + 92928 + 4048 + 6688) * P / rt / 1e9:8.2f} TFLOP/s algorithmic")
    if hasattr(lib, "ufr_debug_rt_phases"):  # -DUFR_PHASE_TIMING development build: ray transformer, wave of ray 5
        rbuf = (C.c_ulonglong * 32)()
        lib.ufr_debug_rt_phases(rbuf, 32, 1)
        rnames = ["s1 load + K,V gemm", "s1 KV mfma", "s2 load + Q gemm", "message", "merge gemm", "LN1", "MLP0", "relu+MLP2",
                  "LN2", "stores + DensityMLP"]
        rtot = sum(rbuf[:10])
        for i, nm in enumerate(rnames):
            print(f"   rt phase {nm:20s} {rbuf[i] / 5:12.0f} cyc/launch  {100.0 * rbuf[i] / max(rtot, 1):5.1f} %")
        print(f"   rt total {rtot / 5:.0f} cycle-counter ticks per launch (one ray, {SN // 16} tiles)")
    if hasattr(lib, "ufr_debug_vt_phases"):  # -DUFR_PHASE_TIMING development build
        buf = (C.c_ulonglong * 32)()
        lib.ufr_debug_vt_phases(buf, 32, 1)
        names = ["load", "qk gemm", "scores", "v gemm", "message", "merge gemm", "LN1", "MLP0", "relu+MLP2",
                 "LN2+stores", "radiance MLP", "softmax"]
        tot = sum(buf[:12])
        for i, nm in enumerate(names):
            print(f"   phase {nm:14s} {buf[i] / 5:12.0f} cyc/launch  {100.0 * buf[i] / tot:5.1f} %")
        print(f"   total {tot / 5:.0f} cycle-counter ticks per launch (wave 0)")
        if buf[20]:
            print(f"   wave 5: {buf[20] / 5 / 100:.1f} us by the 100 MHz clock, {buf[21] / 5:.0f} cycle-counter ticks -> {buf[21] / buf[20] * 100:.0f} MHz")
        wb = (C.c_ulonglong * 4096)()
        lib.ufr_debug_vt_waves(wb, 4096)
        a = np.array(wb[:], dtype=np.uint64).reshape(-1, 2)
        start = a[:, 0].astype(np.int64)
        dur = (a[:, 1] & np.uint64((1 << 40) - 1)).astype(np.int64)
        xcc = (a[:, 1] >> np.uint64(60)).astype(np.int64)
        hwid = ((a[:, 1] >> np.uint64(40)) & np.uint64(0xffff)).astype(np.int64)
        ok = dur > 0
        print("   waves recorded", ok.sum(), "dur min/med/max", dur[ok].min(), int(np.median(dur[ok])), dur[ok].max())
        for x in range(8):
            m = ok & (xcc == x)
            if m.any():
                st = start[m] - start[m].min()
                print(f"   xcc {x}: waves {m.sum()} start spread {st.max()} dur med {int(np.median(dur[m]))} max {dur[m].max()} end-span {(start[m] + dur[m]).max() - start[m].min()}")
        cu = (hwid >> 8) & 0xf
        se = (hwid >> 13) & 0x7
        key = xcc * 1000 + se * 16 + cu
        u, cnt = np.unique(key[ok], return_counts=True)
        print("   distinct (xcc,se,cu):", len(u), "waves per cu histogram:", np.bincount(cnt))
        slow = ok & (dur > 1.2 * np.median(dur[ok]))
        print("   slow waves:", slow.sum(), "on cus with wave counts", np.unique(cnt[np.searchsorted(u, key[slow])], return_counts=True))


if __name__ == "__main__":
    main()
