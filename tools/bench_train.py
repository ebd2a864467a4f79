#!/usr/bin/env python
"""Training-step benchmark (BASELINE.json configs[4]): 3 source views + GT reference view, 1024 random rays of a
512x640 frame, 64+64 samples, forward + backward through the HIP kernels + Adam step (+ one flat gradient all-reduce when
launched with N ranks: data parallel, every rank its own 1024 rays -- weak scaling).

A step = what the reference's training_step does after its encoder (code1/model.py:537-566): sample rays, infer, the two
colour MSE + two depth L1 losses, backward, optimizer step.  Inputs synthetic (uforecon_amd.scene), resident in HBM; the
six frustums require gradients (through them feature_volume.cost_reg_2.* trains), so the scatter-add and the 0.67 GB of
zeroed gradient volumes are inside the timed region.  Rank 0 prints one JSON line with the per-kernel split.

    python tools/bench_train.py --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_train.py --gpus N
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--coarse", type=int, default=64)
    ap.add_argument("--fine", type=int, default=64)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--precision", choices=("fp32", "16bit"), default="fp32",
                    help="16bit: ufr_set_matrix_precision(UFR_PRECISION_16BIT), the mixed-precision mode of configs[4]")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("UFR_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist

    from uforecon_amd import model as M
    from uforecon_amd import ops
    from uforecon_amd.dist import allreduce_gradients
    from uforecon_amd.scene import make_frame

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
    args = argparse.Namespace(extract_geometry=False, test_sample_coarse=a.coarse, test_sample_fine=a.fine,
                              coarse_sample=a.coarse, fine_sample=a.fine, volume_type="correlation", volume_reso=96,
                              mvs_depth_guide=1, depth_pos_encoding=True, use_dir_srdf=False, explicit_similarity=True,
                              test_coarse_only=False, test_ray_num=800)
    torch.manual_seed(0)
    m = M.UFORecon(args).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)                       # model.py:72-87 (uforecon_lr)
    f = make_frame(a.height, a.width, a.views, seed=rank, train_layout=True).to(dev)
    vols = [f.feature_volume[st][k] for st in f.feature_volume for k in f.feature_volume[st]]
    for v in vols:
        v.requires_grad_(True)
    HW = a.height * a.width
    gen = torch.Generator(device=dev).manual_seed(100 + rank)

    def step():
        idx = torch.randperm(HW, device=dev, generator=gen)[: a.rays][None]              # model.py:537
        U1 = torch.rand(a.coarse, a.rays, device=dev, generator=gen)
        U2 = torch.rand(a.fine, a.rays, device=dev, generator=gen)
        for v in vols:
            v.grad = None
        opt.zero_grad(set_to_none=True)
        r = m.infer(f.batch, idx, f.source_imgs_feat, f.feature_volume, match_feature=f.match_feature, uniforms=(U1, U2))
        rgb_gt, rgb, depth, depth_gt, rgb2, depth2 = r[0], r[1], r[2], r[3], r[8], r[9]
        nf = f.batch["near_fars"]
        mask = (depth_gt != 0) & (depth_gt >= nf[:, 0, 0:1]) & (depth_gt <= nf[:, 0, 1:2])
        loss = (torch.nn.functional.mse_loss(rgb, rgb_gt) + torch.nn.functional.mse_loss(rgb2, rgb_gt)
                + torch.nn.functional.l1_loss(depth[mask], depth_gt[mask])
                + torch.nn.functional.l1_loss(depth2[mask], depth_gt[mask]))           # model.py:552-566
        loss.backward()
        if world > 1:
            allreduce_gradients(list(m.parameters()), world)
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.precision == "16bit":
        ops.set_matrix_precision(ops.PRECISION_16BIT)
    for _ in range(a.warmup):
        step()
    fence()
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof = ops.profile_read()
    ops.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        S = a.coarse + a.fine
        pts = a.rays * (a.coarse + S)
        print(json.dumps(dict(
            metric="training rays/s (fwd + bwd + Adam through the HIP ray path)", value=world * a.rays * a.steps / dt,
            unit="rays/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=dt / a.steps * 1e3, scaling="weak",
            dtype="f32" if a.precision == "fp32" else "bf16/fp16 operands, f32 accumulate", data="synthetic",
            config=dict(workload=f"configs[4]: {a.views} source views + GT view, {a.rays} random rays per rank of a "
                                 f"{a.height}x{a.width} frame, {a.coarse}+{a.fine} samples ({pts} point evaluations per rank and "
                                 f"step), frustum gradients on", loss=float(loss),
                        kernel_ms_per_step_rank0={k: v["ms"] / a.steps for k, v in prof.items()},
                        kernel_launches_per_step={k: v["launches"] / a.steps for k, v in prof.items()}))), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
