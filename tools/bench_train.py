#!/usr/bin/env python
"""Training-step benchmark (BASELINE.json configs[4]): 3 source views + GT reference view, 1024 random rays of a
512x640 frame, 64+64 samples, forward + backward through the HIP kernels + Adam step (+ one flat gradient all-reduce when
launched with N ranks: data parallel, every rank its own 1024 rays -- weak scaling).

A step = what the reference's training_step does after its encoder (code1/model.py:537-566): sample rays, infer, the two
colour MSE + two depth L1 losses, backward, optimizer step.  Inputs synthetic (uforecon_amd.scene), resident in HBM; the
six frustums require gradients (through them feature_volume.cost_reg_2.* trains), so the scatter-add and the 0.67 GB of
zeroed gradient volumes are inside the timed region.  Rank 0 prints one JSON line with the per-kernel split, the roofline
of the dominant kernel (view_bwd_kernel: recompute + data gradients + weight gradients = 3 x the forward's
view-transformer flop per point) and the CPU baseline (autograd through the oracle on the host cores, bounded sample).

    python tools/bench_train.py --steps 10 --warmup 2 [--precision 16bit]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_train.py --gpus N

`run(...)` is also what bench.py calls for its `secondary` configs[4] entries.
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

# view-transformer flop per point (q/k/v/merge 204 800 + MLP 307 200 + attention 13 440 + radiance MLP 8 784
# at NV = 3); the backward kernel recomputes the layer, forms the data gradients and the weight gradients: 3 x
VIEWT_FLOP_PER_POINT = {3: 534_224, 5: (204_800 + 307_200) * 6 // 4 + 20_160 + 14_640}
PEAK_FP32_MFMA_TFLOPS = 157.3     # v_mfma_f32_16x16x4_f32 (side field only: no kernel of the step issues it for its dense layers)
PEAK_16BIT_MFMA_TFLOPS = 2516.6   # v_mfma_f32_16x16x32_{f16,bf16}: the 16-bit mode issues one per product ...
# ... and the fp32 mode three (fp16 planes forwards, bf16 hi / lo planes in the backward chains and contractions): the same
# basis as bench.py's headline
PEAK_F32_VIA_3_PLANE_PRODUCTS_TFLOPS = PEAK_16BIT_MFMA_TFLOPS / 3.0


def parse(argv=None):
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
                    help="16bit: UFR_PRECISION_16BIT, the mixed-precision mode BASELINE configs[4] names")
    ap.add_argument("--cost-reg", action="store_true",
                    help="the step WITH the one producer the reference trains in front of the ray path (model.py:72-87, 517-524): "
                         "seeded cost volumes -> MVSVolume (feature_volume.cost_reg_2, ufr_conv3d) -> frustums -> infer -> loss -> "
                         "backward through ufr_project_gather_bwd and ufr_conv3d_bwd_* -> Adam over both parameter sets")
    ap.add_argument("--graph", action="store_true",
                    help="capture one whole step (forward, loss, backward on its three streams, re-pack, Adam) in a HIP graph after "
                         "the warm-up and REPLAY it in the timed region: the ~60 few-microsecond launches of the caller's side "
                         "(loss arithmetic, Adam, re-pack) stop bounding the step.  Single rank, without --cost-reg; the ray "
                         "indices / uniforms of a step are drawn outside the graph into static buffers")
    ap.add_argument("--torch-loss", action="store_true", help="the loss of model.py:552-566 as torch expressions instead of "
                    "UFORecon.training_loss (ufr_render_loss): the A/B of the fused loss node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--dump-grads", default="", help="rank 0 writes the post-all-reduce parameter gradients of the LAST step "
                                                     "here (.pt) -- the 2-rank data-parallel test compares them")
    ap.add_argument("--data-rank", type=int, default=-1, help="single-process run on the data (frame, rays, uniforms) of this "
                                                               "rank of a multi-rank run (the data-parallel test's reference legs)")
    ap.add_argument("--fixed-seed", type=int, default=-1, help=">= 0: ray indices / uniforms of a step depend only on "
                                                                "(seed, rank, step), frames on the rank: reproducible")
    return ap.parse_args(argv)


def cpu_baseline(a, frame_cpu, weights_cpu):
    """One training step (forward, loss, backward) of `cpu_rays` rays through autograd over the oracle -- the CPU port of
    the reference path -- on this host's cores; the six volumes require grad, as in the GPU step."""
    from oracle import ufo_oracle as O
    from uforecon_amd.scene import sampler_uniforms

    RN, HW = a.cpu_rays, a.height * a.width
    idx = (torch.arange(RN) * (HW // RN) + (HW // RN) // 3)[None]
    U1, U2 = sampler_uniforms(2, a.coarse, a.fine, RN)
    P = {k: v.clone().requires_grad_("depthcode" not in k) for k, v in weights_cpu.items()}
    vols = [frame_cpu.feature_volume[st][k] for st in frame_cpu.feature_volume for k in frame_cpu.feature_volume[st]]
    times = []
    for i in range(a.cpu_steps + 1):
        for v in vols:
            v.requires_grad_(True)
            v.grad = None
        for p in P.values():
            p.grad = None
        t = time.perf_counter()
        r = O.infer(P, frame_cpu.batch, idx, frame_cpu.source_imgs_feat, frame_cpu.feature_volume, frame_cpu.match_feature,
                    U1, U2, extract_geometry=False)
        O.training_loss(r, frame_cpu.batch, idx).backward()
        if i:
            times.append(time.perf_counter() - t)
    for v in vols:
        v.requires_grad_(False)
        v.grad = None
    times.sort()
    med = times[len(times) // 2]
    return dict(value=RN / med, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle.infer(extract_geometry=False) + training loss + autograd backward on {RN} rays of the same "
                       f"frame (volume gradients on), {a.coarse}+{a.fine} samples, fp32, median of {a.cpu_steps} steps after "
                       f"1 warm-up ({med * 1e3:.0f} ms/step)")


def cpu_baseline_standalone(a):
    """cpu_baseline on freshly generated inputs (for callers that time the GPU legs first)."""
    import numpy as np

    from uforecon_amd.scene import make_frame

    wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
    return cpu_baseline(a, make_frame(a.height, a.width, a.views, seed=0, train_layout=True),
                        {k: torch.from_numpy(wz[k]) for k in wz.files})


def run(a, dev, world=1, rank=0):
    """The measurement; returns the JSON-able result dict on rank 0 (None elsewhere)."""
    import numpy as np
    import torch.distributed as dist

    from uforecon_amd import model as M
    from uforecon_amd import ops
    from uforecon_amd.dist import allreduce_gradients
    from uforecon_amd.scene import make_frame

    args = argparse.Namespace(extract_geometry=False, test_sample_coarse=a.coarse, test_sample_fine=a.fine,
                              coarse_sample=a.coarse, fine_sample=a.fine, volume_type="correlation", volume_reso=96,
                              mvs_depth_guide=1, depth_pos_encoding=True, use_dir_srdf=False, explicit_similarity=True,
                              test_coarse_only=False, test_ray_num=800)
    precision = ops.PRECISION_16BIT if a.precision == "16bit" else ops.PRECISION_FP32
    wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
    weights_cpu = {k: torch.from_numpy(wz[k]) for k in wz.files}
    m = M.UFORecon(args, precision=precision).to(dev).train()     # the mode travels with the model, not with the process
    m.load_state_dict(weights_cpu, strict=True)
    drank = rank if a.data_rank < 0 else a.data_rank
    frame_cpu = make_frame(a.height, a.width, a.views, seed=drank, train_layout=True)
    f = frame_cpu.to(dev)
    mvs, cost = None, None
    if a.cost_reg:
        from uforecon_amd import cascade
        from uforecon_amd.scene import fill_state_dict, make_cost_volumes

        mvs = cascade.MVSVolume(1, 8)
        fill_state_dict(mvs, 31)
        mvs = mvs.to(dev).train()
        cost = {st: v.to(dev) for st, v in make_cost_volumes(a.height, a.width, a.views, drank).items()}
        vols = []
    else:
        vols = [f.feature_volume[st][k] for st in f.feature_volume for k in f.feature_volume[st]]
        for v in vols:
            v.requires_grad_(True)
    use_graph = bool(a.graph) and world == 1 and mvs is None
    opt = torch.optim.Adam(list(m.parameters()) + (list(mvs.parameters()) if mvs is not None else []), lr=1e-4,
                           capturable=use_graph)   # model.py:72-87
    HW = a.height * a.width
    gen = torch.Generator(device=dev).manual_seed(100 + drank if a.fixed_seed < 0 else a.fixed_seed * 1000 + drank)
    ar_events = []

    # --graph: the step's random inputs live in static buffers that a replayed graph reads
    # Two sets: while a step runs on set k, the NEXT step's rays and uniforms are drawn into set 1 - k on a side stream --
    # they depend on nothing (the reference draws them inline, model.py:537; a data loader would hand them over), and as
    # the first launches of a step they cost 0.2 ms of an otherwise idle GPU.  (--graph: one set, drawn between replays.)
    bufs = [dict(idx=torch.zeros(1, a.rays, dtype=torch.int64, device=dev), U1=torch.zeros(a.coarse, a.rays, device=dev),
                 U2=torch.zeros(a.fine, a.rays, device=dev), ready=None) for _ in range(2)]
    idx_buf, U1_buf, U2_buf = bufs[0]["idx"], bufs[0]["U1"], bufs[0]["U2"]
    draw_stream = torch.cuda.Stream(dev)
    turn = [0]

    def draw(b=None):
        b = bufs[0] if b is None else b
        # model.py:537 takes the first train_ray_num entries of argsort(rand(H W)): a uniform sample without replacement in
        # random order -- as is the top-k of the same uniforms (a radix select instead of the full sort: ~20 launches fewer)
        b["idx"].copy_(torch.rand(HW, device=dev, generator=gen).topk(a.rays, sorted=False).indices[None])
        b["U1"].copy_(torch.rand(a.coarse, a.rays, device=dev, generator=gen))
        b["U2"].copy_(torch.rand(a.fine, a.rays, device=dev, generator=gen))

    def draw_ahead(b):
        draw_stream.wait_stream(torch.cuda.current_stream(dev))      # the set's previous readers are done
        with torch.cuda.stream(draw_stream):
            draw(b)
            b["ready"] = torch.cuda.Event()
            b["ready"].record()

    phase_marks = []      # UFR_BT_PHASES=1 (development): events on the main stream at the phase boundaries of every step
    want_phases = os.environ.get("UFR_BT_PHASES") == "1"

    def mark(row):
        if want_phases:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            row.append(e)

    def step(drawn: bool = False):
        row = []
        phase_marks.append(row)
        mark(row)
        if drawn:
            idx, U1, U2 = idx_buf, U1_buf, U2_buf
        else:
            cur, nxt = bufs[turn[0]], bufs[1 - turn[0]]
            turn[0] = 1 - turn[0]
            if cur["ready"] is None:
                draw_ahead(cur)                       # the very first step
            torch.cuda.current_stream(dev).wait_event(cur["ready"])
            draw_ahead(nxt)
            idx, U1, U2 = cur["idx"], cur["U1"], cur["U2"]
        for v in vols:
            v.grad = None
        opt.zero_grad(set_to_none=True)
        fv = f.feature_volume
        if mvs is not None:      # model.py:517-524: the three stages' frustums from the trainable U-Net
            fv = {}
            for st in ("stage1", "stage2", "stage3"):
                vf, vw = mvs(f.batch, cost[st])
                fv[st] = {"feature_volume": vf, "weight_volume": vw}
        mark(row)
        r = m.infer(f.batch, idx, f.source_imgs_feat, fv, match_feature=f.match_feature, uniforms=(U1, U2))
        mark(row)
        if a.torch_loss:
            # model.py:552-566 as torch expressions (the A/B of --torch-loss).  The masked L1 means are written without
            # boolean indexing: `depth[mask]` makes the host wait for the forward (it needs the count) before it can enqueue
            # the backward -- same value, no pipeline bubble
            rgb_gt, rgb, depth, depth_gt, rgb2, depth2 = r[0], r[1], r[2], r[3], r[8], r[9]
            nf = f.batch["near_fars"]
            mask = (depth_gt != 0) & (depth_gt >= nf[:, 0, 0:1]) & (depth_gt <= nf[:, 0, 1:2])
            n_valid = mask.sum().clamp_min(1)
            l1 = lambda d: ((d - depth_gt).abs() * mask).sum() / n_valid
            loss = torch.nn.functional.mse_loss(rgb, rgb_gt) + torch.nn.functional.mse_loss(rgb2, rgb_gt) + l1(depth) + l1(depth2)
        else:
            loss, _ = m.training_loss(r, f.batch)      # the same loss as one node on one kernel (ufr_render_loss)
        mark(row)
        loss.backward()
        mark(row)
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            # the per-ray parameters; the volumes are per-rank frames (every rank trains on its own scene), so their
            # gradients stay local -- in the real pipeline they flow on into feature_volume.cost_reg_2.*, whose own
            # gradients would join this bucket
            allreduce_gradients(list(m.parameters()) + (list(mvs.parameters()) if mvs is not None else []), world)
            e1.record()
            ar_events.append((e0, e1))
        if a.dump_grads:
            step.last_grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        opt.step()
        mark(row)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("UFR_BT_OVERLAP") == "0":      # development / profiling: the whole backward on one stream
        m.overlap = False
    for _ in range(a.warmup):
        step()
    fence()
    ar_events.clear()
    if os.environ.get("UFR_BT_PROF") == "1":
        ops.profile_enable(True)
    graph, graph_note = None, "eager launches"
    if use_graph:
        try:
            # capture on a side stream (torch.cuda.graph does that), after the warm-up made every lazily created object
            # (workspaces, side streams, kernel attributes, the frame handle) exist
            draw()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                g_loss = step(drawn=True)
            graph.replay()
            fence()
            graph_note = "one HIP graph per step (captured after the warm-up), replayed"
        except Exception as e:  # noqa: BLE001 -- a box whose runtime cannot capture this step still measures it, and says so
            graph, graph_note = None, f"eager launches (HIP graph capture failed: {type(e).__name__}: {str(e)[:160]})"
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if graph is not None:
            draw()
            graph.replay()
            loss = g_loss
        else:
            loss = step()
    fence()
    dt = time.perf_counter() - t0
    if want_phases and rank == 0:
        rows = [r for r in phase_marks[-a.steps:] if len(r) == 6]
        names = ("draw+zero_grad", "infer", "loss", "backward", "optimizer")
        avg = [sum(r[i].elapsed_time(r[i + 1]) for r in rows) / len(rows) for i in range(5)]
        gaps = sum(rows[i][5].elapsed_time(rows[i + 1][0]) for i in range(len(rows) - 1)) / max(1, len(rows) - 1)
        print("phases_ms", {n: round(v, 3) for n, v in zip(names, avg)}, "between steps", round(gaps, 3), file=sys.stderr)
    phase_marks.clear()
    want_phases = False
    timed_grads = getattr(step, "last_grads", None)
    # per-kernel durations from the same number of extra, UNTIMED steps with the backward's stream overlap switched off
    # (uforecon_amd/autograd.py runs independent stages side by side: overlapped, the HIP-event intervals of the kernels
    # include their neighbours -- like bench.py's extra single-stream frame)
    m.overlap = False
    ops.profile_enable(True)
    for _ in range(a.steps):
        step()
    fence()
    prof = ops.profile_read()
    ops.profile_enable(False)
    m.overlap = os.environ.get("UFR_BT_OVERLAP") != "0"
    ops.status_poll(True)        # an activation / weight outside the split-precision planes' range fails the run loudly
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_local, dt = dt, float(t.item())
        mine = dict(rank=rank, wall_ms_per_step=dt_local / a.steps * 1e3,
                    kernel_ms_per_step={k: v["ms"] / a.steps for k, v in prof.items()},
                    all_reduce_ms_per_step=sum(e0.elapsed_time(e1) for e0, e1 in ar_events) / max(len(ar_events), 1))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if a.dump_grads and rank == 0:
        torch.save({k: v.cpu() for k, v in timed_grads.items()}, a.dump_grads)
    if rank != 0:
        return None
    S = a.coarse + a.fine
    pts = a.rays * (a.coarse + S)
    # the view transformer's backward = three kernels per launch (csrc/bwd_tape.h): forward with a tape, data gradients,
    # weight-gradient contraction
    parts = [prof[k] for k in ("view_tape", "view_dgrad", "view_wgrad") if k in prof]
    # (a launch GROUP = one data-gradient launch; the tape build of a group is two launches since the training forward
    # records it: the coarse rows and the new rows of the sample pool)
    vb = dict(ms=sum(p["ms"] for p in parts), launches=prof.get("view_dgrad", dict(launches=1))["launches"] or 1)
    vb_ms = vb["ms"] / max(vb["launches"], 1)
    # the fine pass evaluates only its new samples at the point level: coarse + fine points per ray over the launches
    vb_pts = a.rays * S * a.steps / max(vb["launches"], 1)
    vb_flop = 3.0 * VIEWT_FLOP_PER_POINT.get(a.views, 0) * vb_pts
    achieved = vb_flop / (vb_ms * 1e-3) / 1e12 if vb_ms > 0 else 0.0
    peak = PEAK_16BIT_MFMA_TFLOPS if a.precision == "16bit" else PEAK_F32_VIA_3_PLANE_PRODUCTS_TFLOPS
    traffic, traffic_src = None, None
    try:
        import glob
        pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_pmc.json")))[-1]
        tot = 0.0
        # EVERY launch of the group: the training forward records the tape in two launches (coarse rows, new rows) per
        # data-gradient launch
        tape_per_group = prof.get("view_tape", dict(launches=0))["launches"] / max(vb["launches"], 1)
        for k, v in json.load(open(pj)).items():
            # the three kernels of the view transformer's backward (the TAPE instantiation has four template arguments)
            if not v.get("hbm_bytes_per_launch"):
                continue
            if "view_dgrad_kernel" in k or "view_wgrad_kernel" in k:
                tot += v["hbm_bytes_per_launch"]
            elif "view_transformer_kernel" in k and k.count(",") >= 3:
                tot += v["hbm_bytes_per_launch"] * max(tape_per_group, 1.0)
        if tot > 0 and a.precision == "fp32":     # the committed counter pass ran the fp32 mode (the 16-bit mode stores bf16 tiles)
            traffic, traffic_src = tot, os.path.relpath(pj, ROOT)
    except Exception:  # noqa: BLE001
        pass
    line = dict(
        metric="training rays/s (fwd + bwd + Adam through the HIP ray path)", value=world * a.rays * a.steps / dt,
        unit="rays/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=dt / a.steps * 1e3,
        higher_is_better=True, scaling="weak",
        dtype="f32" if a.precision == "fp32" else "bf16", data="synthetic",
        config=dict(workload=f"configs[4]: {a.views} source views + GT view, {a.rays} random rays per rank of a "
                             f"{a.height}x{a.width} frame, {a.coarse}+{a.fine} samples ({pts} point evaluations per rank and "
                             f"step), frustum gradients on, Adam step included"
                             + (" -- WITH feature_volume.cost_reg_2 (the 3-D U-Net the reference trains, module.py:502-543) in "
                                "front: three stages' frustums rebuilt from seeded cost volumes every step (ufr_conv3d), its "
                                "backward through ufr_conv3d_bwd_data / ufr_conv3d_bwd_weight" if a.cost_reg else ""),
                    arithmetic=("fp32 mode: forward dense layers as three fp16 plane products (fp32-grade); backward data-gradient "
                                "chains and weight-gradient contractions as three bf16 plane products (16 significand bits per "
                                "operand, fp32 accumulate; every gradient tensor within 3e-5 of the reference's autograd)"
                                if a.precision == "fp32" else
                                "16-bit mode: one fp16 plane per operand in the forward, one bf16 plane per operand in the backward "
                                "chains and weight gradients (tiles stored as bf16), fp32 accumulation; LayerNorm / attention / "
                                "softmax / compositor fp32"),
                    backward_ms_per_step=dict(
                        view_transformer=vb_ms * vb["launches"] / a.steps,
                        ray_transformer=sum(prof[k]["ms"] for k in ("ray_tape", "ray_dgrad", "ray_wgrad") if k in prof) / a.steps,
                        frustum_scatter=prof.get("gather_bwd", dict(ms=0.0))["ms"] / a.steps),
                    kernel_ms_measured=("extra untimed steps after the timed region with the backward's stream overlap off; in "
                                        "the timed steps the coarse / fine ray backwards run side by side, and the weight "
                                        "gradients beside the view data gradients / the frustum scatter "
                                        "(uforecon_amd/autograd.py), so ms_per_step is less than the kernels' sum.  view_tape / "
                                        "ray_tape ARE the training forward of the two transformers (they record the "
                                        "backward's tape): there is no separate view_transformer / ray_transformer launch"),
                    loss=float(loss.detach()),
                    launch=graph_note,
                    kernel_ms_per_step_rank0={k: v["ms"] / a.steps for k, v in prof.items()},
                    kernel_launches_per_step={k: v["launches"] / a.steps for k, v in prof.items()},
                    per_rank=per_rank),
        roofline=dict(bound="mfma", achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak, traffic=traffic,
                      traffic_source=traffic_src, kernel="view_transformer_kernel<TAPE> + view_dgrad_kernel + view_wgrad_kernel", avg_launch_ms=vb_ms, launches=vb["launches"],
                      algorithmic_flop_per_launch=vb_flop,
                      peak_basis=("dense 16-bit MFMA peak / 3 plane products per fp32 product (the fp32 mode issues bf16 hi / lo "
                                  "planes: the headline's basis)" if a.precision == "fp32" else "dense bf16 MFMA peak")
                                 + "; algorithmic flop = 3 x the forward view-transformer flop per point (recompute + data "
                                   "gradients + weight gradients); the three kernels are bound by the tile traffic between them "
                                   "(`traffic` bytes per launch group at `hbm_gbps`), not by the matrix cores",
                      hbm_gbps=(traffic / (vb_ms * 1e-3) / 1e9) if (traffic and vb_ms > 0) else None,
                      frac_of_fp32_mfma_peak=(achieved / PEAK_FP32_MFMA_TFLOPS) if a.precision == "fp32" else None,
                      bwd_operand_dtype="bf16x3 (hi + lo planes, 16 significand bits per operand)" if a.precision == "fp32" else "bf16"))
    if not a.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(a, frame_cpu, weights_cpu)
    return line


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("UFR_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
    line = run(a, dev, world, rank)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
