#!/usr/bin/env python
"""Headline benchmark: rays/s (+ depth-map ms/frame) of the per-ray path, DTU-shaped 3-view 512x640.

One "step" = one full frame through the hot path (BASELINE.json configs[1]: 64+64 hierarchical
samples, every one of the 327 680 pixel rays): channel-last re-layout of the per-frame tensors,
then ufr_render_rays over this rank's row tile, then (N>1) an RCCL all-gather of the depth / RGB
tiles.  Inputs are synthetic (uforecon_amd.scene, seed 0; random-init weights seed 0) and resident
in HBM before the timed region.  Rank 0 prints ONE JSON line.

`roofline` describes the dominant kernel (view_transformer_kernel): algorithmic flop of the reference
layer chain per launch / its launch duration from HIP events.  With --streams > 1 the chunks of a
frame overlap on side streams and a kernel's event interval includes its neighbours, so the
per-kernel durations are then taken from extra, untimed, single-stream frames (the median of three) right after the
timed region (same inputs, same launches); with --streams 1 they come from the timed region itself.

After the timed configs[1] region, a single-GPU run also measures the other single-GPU configurations of BASELINE.json
and writes them to `bench_secondary.json` beside this file (and echoes that document on STDERR) as `secondary` (each entry a
complete line of its own: value, ms_per_step, kernel split, `roofline`): the per-frame producers, the configs[4] training step (1024 rays, forward + backward + Adam) in both
matrix precisions (tools/bench_train.py's loop), the configs[2] evaluation loop on this one GPU and -- last, so that it
survives a truncated log -- configs[3] (5 views, 800x600, 128+128 samples; two frames).  STDOUT carries exactly ONE line, the
last thing printed: the headline record, numbers only, under 4 KB (`headline()` below; round 5's single 20.9 KB line was
not parseable by the driver), closed by a short `digest` of the secondary entries.  `--full-secondary` adds the CPU / GPU-eager baselines of those entries and the evaluation loop without
the producer overlap (minutes of host time); `--no-secondary` skips them all.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per point evaluation (NV=3 / NV=5), flop
FLOP_PER_POINT = {3: 708_018, 5: 983_150}
# ... of which the view-transformer launch (q/k/v/merge 204 800 + MLP 307 200 + attention 13 440 +
# radiance MLP 8 784 at NV=3): the dominant kernel the roofline object describes
VIEWT_FLOP_PER_POINT = {3: 204_800 + 307_200 + 13_440 + 8_784, 5: (204_800 + 307_200) * 6 // 4 + 20_160 + 14_640}
RAYT_FLOP_PER_POINT = 61_952 + 92_928 + 4_048 + 6_688   # ray transformer + DensityMLP (d = 88, 8 heads)
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2516.6  # dense fp16 / bf16 MFMA peak: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz
# the transformer kernels compute every fp32 product as three fp16 plane products (two-plane split of both
# operands, fp32 accumulate: ufr_layout_f16.h), so their matrix-core bound in fp32-equivalent flop is peak/3
PEAK_F32_VIA_F16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
SIDE_FILE = os.path.join(ROOT, "bench_secondary.json")


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--views", type=int, default=3)
    p.add_argument("--coarse", type=int, default=64)
    p.add_argument("--fine", type=int, default=64)
    p.add_argument("--chunk", type=int, default=0, help="rays per launch group (0 = library default)")
    p.add_argument("--streams", type=int, default=3, help="side streams the chunks are spread over")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-gpu-eager-baseline", action="store_true")
    p.add_argument("--eager-chunks", type=int, default=6, help="800-ray chunks of the GPU-eager 1x reference leg")
    p.add_argument("--cpu-rays", type=int, default=256)
    p.add_argument("--cpu-calls", type=int, default=5)
    p.add_argument("--fixed-uniforms", type=int, default=-1, metavar="SEED",
                   help=">= 0: sampler uniforms are one seeded draw for the whole frame, sliced per rank (so an N-rank run "
                        "renders exactly the frame a 1-rank run renders); default: fresh GPU draws every step")
    p.add_argument("--dump-depth", default="", help="rank 0 writes the (gathered) H x W depth map of the last step here (.npy)")
    p.add_argument("--no-secondary", action="store_true", help="skip the configs[3] / configs[4] measurements attached as "
                                                                "`secondary` to a single-GPU line")
    p.add_argument("--secondary-train-steps", type=int, default=10)
    p.add_argument("--details", default=SIDE_FILE, help="where the whole record (per-kernel split, per-rank tables, secondary "
                                                        "entries, projection, prose) is written; it is echoed on stderr too")
    p.add_argument("--stub-secondary", action="store_true",
                   help="tests: attach worst-case-sized placeholder secondary entries instead of measuring them, so that the "
                        "real configs[1] command path (digest, projection, headline size guard) runs in seconds")
    p.add_argument("--full-secondary", action="store_true",
                   help="also: the evaluation loop a second time without the producer overlap, and the CPU / GPU-eager baselines "
                        "of configs[3] and of the training step (minutes of host time; the default run carries the headline's "
                        "own two baselines only and finishes within ~5 minutes on a cold box)")
    p.add_argument("--config", default="", help="c3: BASELINE configs[2], the evaluation loop over 15 scenes x 3 render views "
                                                 "(per-frame producers + sharded ray path + all-gather, uforecon_amd/evalset.py) "
                                                 "instead of one pre-encoded frame")
    p.add_argument("--frames", type=int, default=45, help="--config c3: frames of the evaluation set")
    p.add_argument("--producers", default=None, choices=["replicated", "sharded"],
                   help="--config c3, N > 1: every rank encodes every frame, or the frames' producers are dealt over the "
                        "ranks and broadcast (default: sharded when N > 1 -- replicated producers bound the frame at the "
                        "encode time whatever N is)")
    p.add_argument("--no-overlap", action="store_true", help="--config c3: encode frame k+1 only after frame k's rays")
    p.add_argument("--dump-depths", default="", help="--config c3: rank 0 writes all depth maps here (.npy, frames x H x W)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                                                      "exercise the multi-rank path on a box with fewer GPUs than ranks)")
    return p.parse_args(argv)


def cpu_baseline(frame_cpu, weights_cpu, a):
    """The oracle (CPU port of the reference path) on this host's cores, bounded sample."""
    from oracle import ufo_oracle as O
    from uforecon_amd.scene import sampler_uniforms

    RN = a.cpu_rays
    HW = a.height * a.width
    idx = (torch.arange(RN) * (HW // RN) + (HW // RN) // 3)[None]
    U1, U2 = sampler_uniforms(1, a.coarse, a.fine, RN)
    times = []
    with torch.no_grad():
        for i in range(a.cpu_calls + 1):
            t = time.perf_counter()
            O.infer(weights_cpu, frame_cpu.batch, idx, frame_cpu.source_imgs_feat, frame_cpu.feature_volume,
                    frame_cpu.match_feature, U1, U2)
            if i:
                times.append(time.perf_counter() - t)
    times.sort()
    med = times[len(times) // 2]
    return dict(value=RN / med, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle.infer on {RN} rays of the same frame, {a.coarse}+{a.fine} samples, "
                       f"median of {a.cpu_calls} calls after 1 warm-up ({med * 1e3:.0f} ms/call)")


def gpu_eager_baseline(frame_cpu, weights_cpu, a, dev):
    """The 1x denominator of the >= 20x target (last bullet): the eager-PyTorch restatement of the
    reference path (the oracle, op for op the reference's aten sequence) on this GPU with the reference's chunking
    (test_ray_num = 800 rays per infer call, code1/model.py:814-823), sampler uniforms drawn on the CPU generator and moved
    over per chunk exactly like the reference does (sampler.py:42, 86).  Bounded sample: a few chunks of the same frame."""
    from oracle import ufo_oracle as O

    RN = 800
    HW = a.height * a.width
    frame = frame_cpu.to(dev)
    P = {k: v.to(dev) for k, v in weights_cpu.items()}
    times = []
    with torch.no_grad():
        for c in range(a.eager_chunks + 1):
            idx = (torch.arange(RN) + c * (HW // (a.eager_chunks + 1)))[None].to(dev)          # model.py:814: consecutive pixels
            torch.cuda.synchronize()
            t = time.perf_counter()
            U1, U2 = torch.rand(a.coarse, RN), torch.rand(a.fine, RN)                           # CPU generator, like upstream
            O.infer(P, frame.batch, idx, frame.source_imgs_feat, frame.feature_volume, frame.match_feature, U1.to(dev),
                    U2.to(dev))
            torch.cuda.synchronize()
            if c:
                times.append(time.perf_counter() - t)
    times.sort()
    med = times[len(times) // 2]
    return dict(value=RN / med, unit="rays/s", kind="port-gpu-eager", device=torch.cuda.get_device_name(dev),
                sample=f"oracle.infer (eager PyTorch-ROCm) on cuda, {a.eager_chunks} chunks of {RN} rays of the same frame after "
                       f"1 warm-up chunk, {a.coarse}+{a.fine} samples, median {med * 1e3:.0f} ms per chunk "
                       f"(= {HW / RN * med:.1f} s per {a.height}x{a.width} frame)")


def config_name(a):
    if (a.views, a.height, a.width, a.coarse, a.fine) == (3, 512, 640, 64, 64):
        return "configs[1]"
    if (a.views, a.height, a.width, a.coarse, a.fine) == (5, 600, 800, 128, 128):
        return "configs[3]"
    return "custom"


def measure_frames(a, dev, world, rank):
    """Time a.steps frames of the configuration in `a` (after a.warmup); returns the JSON line (rank 0) or None."""
    import numpy as np
    import torch.distributed as dist

    from uforecon_amd import ops
    from uforecon_amd.dist import RayShard, all_gather_tiles
    from uforecon_amd.scene import make_frame

    wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
    weights_cpu = {k: torch.from_numpy(wz[k]) for k in wz.files}
    frame_cpu = make_frame(a.height, a.width, a.views, seed=0)
    frame = frame_cpu.to(dev)
    W = ops.PackedWeights({k: v.to(dev) for k, v in weights_cpu.items()})

    HW = a.height * a.width
    shard = RayShard(a.height, a.width, world, rank)        # contiguous row tile of this rank
    ray_idx = shard.ray_indices(dev)
    RN = ray_idx.numel()
    ws = ops.RenderWorkspace(dev, a.coarse, a.fine, a.views, chunk_rays=a.chunk, n_streams=a.streams)
    out = dict(depth=torch.empty(RN, device=dev), depth_z=torch.empty(RN, device=dev), rgb=torch.empty(RN, 3, device=dev))
    gathered = None
    fixed = None
    if a.fixed_uniforms >= 0:
        g = torch.Generator().manual_seed(a.fixed_uniforms)
        fixed = (torch.rand(a.coarse, HW, generator=g).to(dev)[:, ray_idx].contiguous(),
                 torch.rand(a.fine, HW, generator=g).to(dev)[:, ray_idx].contiguous())
    ag_events = []

    def step():
        nonlocal gathered
        fh = ops.FrameHandle(frame.batch, frame.source_imgs_feat, frame.feature_volume, frame.match_feature)
        U1 = fixed[0] if fixed else torch.rand(a.coarse, RN, device=dev)
        U2 = fixed[1] if fixed else torch.rand(a.fine, RN, device=dev)
        ops.render_rays(fh, W, ray_idx, U1, U2, workspace=ws, want_srdf=False, out=out)
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gathered = all_gather_tiles(out["depth_z"], out["rgb"], shard)
            e1.record()
            ag_events.append((e0, e1))
        else:
            gathered = (out["depth_z"].view(a.height, a.width), out["rgb"].view(a.height, a.width, 3))
        return fh

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    ag_events.clear()
    ops.profile_enable(a.streams <= 1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof_steps = a.steps
    if a.streams > 1:  # per-kernel durations: untimed single-stream frames (see the module docstring)
        ws1 = ops.RenderWorkspace(dev, a.coarse, a.fine, a.views, chunk_rays=a.chunk, n_streams=1)
        ws, prof_steps = ws1, 1
        step()
        fence()
        # three profiled frames, per kernel the MEDIAN frame: one frame's 160 launches are 60 ms, and a single frame now and
        # then catches a clock dip of the package (a view transformer at 0.51 instead of 0.375 ms per launch in one run of
        # round 5, with the rocprofv3 average of the same box at 0.380) -- the roofline line must not hang on that
        frames = []
        for _ in range(3):
            ops.profile_enable(True)
            step()
            fence()
            frames.append(ops.profile_read())
        prof = {}
        for k in frames[0]:
            cand = sorted((f[k] for f in frames if k in f), key=lambda v: v["ms"])
            prof[k] = cand[len(cand) // 2]
    else:
        prof = ops.profile_read()
    ops.profile_enable(False)
    ops.status_poll(True)        # an activation / weight outside the split-precision planes' range fails the run loudly
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_local, dt = dt, float(t.item())
        # what makes a scaling number interpretable: every rank's kernel split, its all-gather time (which includes waiting
        # for the slowest rank) and its own wall time
        mine = dict(rank=rank, rays=RN, wall_ms_per_step=dt_local / a.steps * 1e3,
                    kernel_ms_per_frame={k: v["ms"] / prof_steps for k, v in prof.items()},
                    all_gather_ms_per_step=sum(e0.elapsed_time(e1) for e0, e1 in ag_events[:a.steps]) / max(a.steps, 1))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if a.dump_depth and rank == 0:
        np.save(a.dump_depth, gathered[0].detach().cpu().numpy())

    line = None
    if rank == 0:
        S = a.coarse + a.fine
        cfg = config_name(a)
        # the reference evaluates coarse + (coarse + fine) samples per ray; this path keeps the coarse per-point
        # results and evaluates coarse + fine points (gathers, view transformer), coarse + (coarse + fine) ray-level
        ref_pts_per_ray = a.coarse + S
        point_evals_per_ray = a.coarse + a.fine
        ray_evals_per_ray = a.coarse + S
        rays_per_s = HW * a.steps / dt
        flop_pt = FLOP_PER_POINT.get(a.views)
        exec_flop_per_ray = None
        if flop_pt:
            exec_flop_per_ray = (flop_pt - RAYT_FLOP_PER_POINT) * point_evals_per_ray + RAYT_FLOP_PER_POINT * ray_evals_per_ray
        vt = prof.get("view_transformer", dict(ms=0.0, launches=1))
        chunk = ws.chunk
        vt_pts_per_launch = (RN * point_evals_per_ray * prof_steps) / max(vt["launches"], 1)
        vt_ms = vt["ms"] / max(vt["launches"], 1)
        vt_flop = VIEWT_FLOP_PER_POINT.get(a.views, 0) * vt_pts_per_launch
        achieved = vt_flop / (vt_ms * 1e-3) / 1e12 if vt_ms > 0 else 0.0
        rt = prof.get("ray_transformer", dict(ms=0.0, launches=1))
        rt_ms = rt["ms"] / max(rt["launches"], 1)
        rt_flop = RAYT_FLOP_PER_POINT * (RN * ray_evals_per_ray * prof_steps) / max(rt["launches"], 1)
        rt_achieved = rt_flop / (rt_ms * 1e-3) / 1e12 if rt_ms > 0 else 0.0
        traffic, traffic_src = None, None
        pmc_vt, pmc_rt, pmc_src = None, None, None
        try:  # HBM bytes per view-transformer launch from the latest committed PMC pass OF THIS CONFIGURATION
            # (profiles/rN_pmc.json: configs[1]; profiles/rN_c4_pmc.json: configs[3], the L = 6 instantiation)
            import glob
            pat = {"configs[1]": "r*[0-9]_pmc.json", "configs[3]": "r*_c4_pmc.json"}.get(cfg)
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))[-1]
            for k, v in json.load(open(pj)).items():
                if f"view_transformer_kernel<{a.views + 1}," in k and v.get("hbm_bytes_per_launch"):
                    traffic, traffic_src = v["hbm_bytes_per_launch"], os.path.relpath(pj, ROOT)
                if f"view_transformer_kernel<{a.views + 1}," in k and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                    pmc_vt, pmc_src = v, os.path.relpath(pj, ROOT)
                if k.startswith("ray_transformer_kernel") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                    pmc_rt = v
        except Exception:  # noqa: BLE001
            pass

        def pmc_fracs(v, algo_flop_per_launch_in_that_run):
            """From the committed SQ counter pass: the fraction of the launch the matrix pipes were busy (busy cycles summed
            over the 1024 SIMDs / launch duration at the 2.4 GHz the peak assumes) and issued / algorithmic matrix work
            (SQ_INSTS_MFMA wave instructions x 16 384 flop of a 16x16x32 MFMA / 3 plane products / the algorithmic flop)."""
            if not v or not v.get("avg_ns[pmc_sq]"):
                return None, None, None
            cyc = v["avg_ns[pmc_sq]"] * 2.4
            busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
            valu = v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / cyc        # quad-cycles summed over the 1024 SIMDs
            issued = v.get("SQ_INSTS_MFMA", 0.0) * 16384.0 / 3.0
            return busy, (issued / algo_flop_per_launch_in_that_run if algo_flop_per_launch_in_that_run else None), valu
        line = dict(
            metric=f"rays/s (per-ray volume-rendering path, {a.coarse}+{a.fine} hierarchical samples, DTU-shaped "
                   f"{a.views}-view {a.height}x{a.width})",
            value=rays_per_s, unit="rays/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
            ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None,
            dtype="f32", mfma_operand_dtype="f16x3", data="synthetic",
            config=dict(workload=f"{cfg}: full {a.height}x{a.width} frame = {HW} rays, {a.views} source views, "
                                 f"{a.coarse}+{a.fine} samples, rays sharded by row tiles over {world} GPU(s), depth/RGB "
                                 f"tiles all-gathered",
                        rays_per_frame=HW, chunk_rays=chunk, side_streams=a.streams, depth_map_ms_per_frame=dt / a.steps * 1e3,
                        reference_point_evaluations_per_ray=ref_pts_per_ray,
                        executed_evaluations_per_ray=dict(gather_and_view_transformer=point_evals_per_ray,
                                                          ray_transformer_and_compositor=ray_evals_per_ray,
                                                          note="coarse per-point results are reused by the fine pass (bit-identical)"),
                        arithmetic="fp32 in/out; dense layers as three fp16 plane products per fp32 product on the fp16 MFMA "
                                   "(hi/lo split of both operands, 22+ significand bits, fp32 accumulate; same measured error "
                                   "as an fp32 GEMM); gathers, attention, norms, compositor in fp32 VALU",
                        executed_tflops=(rays_per_s * exec_flop_per_ray / 1e12) if exec_flop_per_ray else None,
                        reference_equivalent_tflops=(rays_per_s * ref_pts_per_ray * flop_pt / 1e12) if flop_pt else None,
                        kernel_ms_per_frame_rank0={k: v["ms"] / prof_steps for k, v in prof.items()},
                        kernel_ms_measured="timed region" if a.streams <= 1 else "median of three extra single-stream frames after the timed region",
                        per_rank=per_rank),
            roofline=dict(bound="mfma", achieved=achieved, peak=PEAK_F32_VIA_F16X3_TFLOPS, unit="TFLOP/s",
                          frac=achieved / PEAK_F32_VIA_F16X3_TFLOPS, traffic=traffic, traffic_source=traffic_src,
                          kernel="view_transformer_kernel", avg_launch_ms=vt_ms, launches=vt["launches"],
                          algorithmic_flop_per_launch=vt_flop,
                          peak_basis="dense fp16 MFMA peak 2516.6 TFLOP/s / 3 plane products per fp32 product (fp16x3); "
                                     "the fp32 MFMA peak is 157.3 TFLOP/s (round 1-2 kernels issued 6 bf16 plane products: "
                                     "their peak basis was 419.4)",
                          power_note="the package runs these kernels at ~1.31 kW of its 1.4 kW cap and the firmware lowers the "
                                     "shader clock to ~2.0 GHz (peak assumes 2.4 GHz; the same instruction stream on a zeroed "
                                     "weight blob keeps 2.39 GHz and is 18.5 % faster): profiles/r3_power_probe.md",
                          frac_of_fp32_mfma_peak=achieved / PEAK_FP32_MFMA_TFLOPS,
                          # the same achieved rate against the roof the round-1 / round-2 reviews priced this kernel on
                          # (six bf16 plane products per fp32 product); the ray transformer on both bases beside it
                          frac_on_bf16x6_basis=achieved / (PEAK_F16_MFMA_TFLOPS / 6.0),
                          ray_transformer=dict(achieved=rt_achieved, frac=rt_achieved / PEAK_F32_VIA_F16X3_TFLOPS,
                                               frac_on_bf16x6_basis=rt_achieved / (PEAK_F16_MFMA_TFLOPS / 6.0),
                                               avg_launch_ms=rt_ms),
                          # 840 v_mfma_f32_16x16x32_f16 (16 384 flop each) per 8 points at NV = 3, K / row padding included
                          issued_f16_tflops=achieved * (840 * 16384 / 8) / VIEWT_FLOP_PER_POINT[3] if a.views == 3 else None),
        )
        # reproducible from profiles/ without trusting the peak basis: matrix-pipe busy fraction and issued / algorithmic work
        # of the same kernels in the committed counter pass (which ran the same chunking: points per launch = vt_pts_per_launch)
        busy, over, valu = pmc_fracs(pmc_vt, vt_flop)
        line["roofline"].update(mfma_busy_frac=busy, valu_busy_frac=valu, issued_over_algorithmic=over, pmc_source=pmc_src)
        rbusy, rover, rvalu = pmc_fracs(pmc_rt, rt_flop)
        line["roofline"]["ray_transformer"].update(mfma_busy_frac=rbusy, valu_busy_frac=rvalu, issued_over_algorithmic=rover,
                                                   traffic=(pmc_rt or {}).get("hbm_bytes_per_launch"))
        if not a.no_gpu_eager_baseline and world == 1:
            # the >= 20x target's denominator, measured in the same run on the same GPU (BASELINE.md has no published
            # number: vs_baseline is relative to this leg, not to a figure from the reference's authors)
            eager = gpu_eager_baseline(frame_cpu, weights_cpu, a, dev)
            line["gpu_eager_baseline"] = eager
            line["vs_baseline"] = rays_per_s / eager["value"]
            line["config"]["vs_baseline_denominator"] = ("gpu_eager_baseline.value: eager PyTorch restatement of the reference "
                                                         "path on this GPU, 800-ray chunks (BASELINE.md section 3)")
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(frame_cpu, weights_cpu, a)
    del frame, out, ws, W, gathered, fixed
    torch.cuda.empty_cache()
    return line


def measure_evalset(a, dev, world, rank):
    """BASELINE configs[2]: the evaluation loop (main.py:229-230 -> model.py:760-842) over a.frames synthetic frames."""
    import numpy as np

    from uforecon_amd import pipeline
    from uforecon_amd.dist import RayShard
    from uforecon_amd.evalset import EvalLoop, make_eval_batch
    from uforecon_amd.scene import fill_state_dict

    args = argparse.Namespace(extract_geometry=True, test_sample_coarse=a.coarse, test_sample_fine=a.fine, coarse_sample=a.coarse,
                              fine_sample=a.fine, volume_type="correlation", volume_reso=96, mvs_depth_guide=1,
                              depth_pos_encoding=True, use_dir_srdf=False, explicit_similarity=True, test_coarse_only=False,
                              test_ray_num=800, test_n_view=a.views, out_dir=None)
    # library convolutions of the backbone: the default algorithms (no per-process search), so that every process
    # computes the same bits
    net = fill_state_dict(pipeline.UFOReconInference(args, tune_convolutions=False), 21).eval().to(dev)
    wz = np.load(os.path.join(ROOT, "tests", "golden", "ray_path_weights_seed0.npz"))
    net.load_state_dict({k: torch.from_numpy(wz[k]) for k in wz.files}, strict=False)     # the per-ray path's seed-0 weights
    batches = [make_eval_batch(a.height, a.width, a.views, k, dev) for k in range(a.frames)]
    shard = RayShard(a.height, a.width, world, rank)
    if a.producers is None:      # N > 1: deal the producers over the ranks (replicated ones bound the frame at the encode time)
        a.producers = "sharded" if world > 1 else "replicated"
    loop = EvalLoop(net, shard, n_streams=a.streams, overlap=not a.no_overlap, producers=a.producers, chunk_rays=a.chunk)
    uni = None
    if a.fixed_uniforms >= 0:
        HW = a.height * a.width

        def uni(k):
            g = torch.Generator().manual_seed(a.fixed_uniforms * 1000 + k)
            return torch.rand(a.coarse, HW, generator=g), torch.rand(a.fine, HW, generator=g)
    warm = min(2, a.frames)
    loop.run(batches[:warm], uni)                       # warm-up: allocator, first-use costs
    st = loop.run(batches, uni, keep_depth=bool(a.dump_depths))
    ops_mod = __import__("uforecon_amd.ops", fromlist=["status_poll"])
    ops_mod.status_poll(True)
    per_rank = None
    mine = dict(rank=rank, encode_ms_mean=sum(st["encode_ms"]) / a.frames, ray_path_ms_mean=sum(st["ray_path_ms"]) / a.frames,
                all_gather_ms_mean=sum(st["all_gather_ms"]) / a.frames, encodes=st["encodes_on_this_rank"],
                ray_path_rays_per_s=st["ray_path_rays_per_s_this_rank"])
    if world > 1:
        import torch.distributed as dist

        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([st["wall_s"]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        st["wall_s"] = float(t.item())
    if a.dump_depths and rank == 0:
        np.save(a.dump_depths, torch.stack(st["depths"]).cpu().numpy())
    if rank != 0:
        return None
    HW = a.height * a.width
    wall = st["wall_s"]
    ray_only = sum(r["ray_path_rays_per_s"] for r in (per_rank or [mine]))
    return dict(
        metric=f"rays/s and depth-map ms/frame over the evaluation set ({a.frames} frames = 15 scenes x 3 render views, "
               f"{a.views}-view {a.height}x{a.width}, {a.coarse}+{a.fine} samples), producers included",
        value=HW * a.frames / wall, unit="rays/s", n_gpus=world, steps=a.frames, warmup=warm, ms_per_step=wall / a.frames * 1e3,
        higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f32", mfma_operand_dtype="f16x3", data="synthetic",
        config=dict(workload=f"configs[2]: per frame encode_frame (backbone, FMT, frustum cascade) -> this rank's row tile "
                             f"through ufr_render_rays -> all-gather of the depth / RGB tiles; frame k+1's producers run on "
                             f"a side stream beside frame k's rays" + ("" if not a.no_overlap else " (overlap OFF)"),
                    frames=a.frames, rays_per_frame=HW, producers=a.producers if world > 1 else "local",
                    depth_map_ms_per_frame_inclusive=wall / a.frames * 1e3,
                    ray_path_only_rays_per_s=ray_only,
                    encode_frame_ms=mine["encode_ms_mean"], ray_path_ms_per_frame_rank0=mine["ray_path_ms_mean"],
                    all_gather_ms_per_frame_rank0=mine["all_gather_ms_mean"],
                    note="encode_frame_ms and ray_path_ms are HIP-event intervals on their own streams; with the overlap on they "
                         "run concurrently and each is stretched by the other's share of the GPU" if not a.no_overlap else
                         "no overlap: the intervals add up to the frame time",
                    scaling_note="unmeasured beyond this run's n_gpus: no multi-GPU box in the build environment" if world == 1 else None,
                    per_rank=per_rank))


def secondary_measurements(a, dev):
    """The other single-GPU configurations of BASELINE.json, measured in the same process after the headline run.
    Insertion order = print order: the driver keeps only the tail of a long stdout line, so the entries it has asked for by
    name come LAST (configs[4], configs[2]@1gpu, configs[3]) and a few-hundred-byte `digest` of all of them closes the line."""
    sec = {}
    full = a.full_secondary
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_correlate
    import bench_encoder
    import bench_train
    import bench_tsdf

    # the per-frame producers and the consumer of the depth maps: driver-timed numbers with achieved GB/s
    sec["encode_frame"] = bench_encoder.measure(512, 640, reps=3)
    sec["correlate"] = bench_correlate.measure(reps=20)
    sec["tsdf"] = bench_tsdf.measure(384, reps=10)
    torch.cuda.empty_cache()

    # both GPU measurements first, the CPU leg (--full-secondary) after them (its worker threads keep spinning for a while
    # and slow the host side of a step that follows)
    # (a throwaway run first: the first training steps of a process allocate the kept backward workspaces -- a few GB -- and
    # compile nothing but do fault memory in; measured in front of the fp32-mode entry they made it 5.8 instead of 4.2 ms)
    bench_train.run(bench_train.parse(["--steps", "2", "--warmup", "2", "--precision", "fp32", "--no-cpu-baseline"]), dev, 1, 0)
    for prec in ("fp32", "16bit"):
        ta = bench_train.parse(["--steps", str(a.secondary_train_steps), "--warmup", "3", "--precision", prec, "--no-cpu-baseline"])
        sec[f"configs[4]_{prec}"] = bench_train.run(ta, dev, 1, 0)
        torch.cuda.empty_cache()
    # ... and the step with the producer the reference trains in front of it (feature_volume.cost_reg_2 on ufr_conv3d / _bwd)
    ta = bench_train.parse(["--steps", "3", "--warmup", "1", "--precision", "fp32", "--no-cpu-baseline", "--cost-reg"])
    sec["configs[4]+cost_reg_2"] = bench_train.run(ta, dev, 1, 0)
    torch.cuda.empty_cache()
    if full and not a.no_cpu_baseline:
        base = bench_train.cpu_baseline_standalone(bench_train.parse([]))
        sec["configs[4]_fp32"]["cpu_baseline"] = base
        # the CPU leg has one arithmetic (fp32): the 16-bit entry points at the same measurement
        sec["configs[4]_16bit"]["cpu_baseline"] = dict(base, note="same measurement as configs[4]_fp32 (the oracle is fp32)")
    # configs[2] on this one GPU: the whole evaluation loop, producers included (--full-secondary: again without the overlap)
    ev = parse(["--config", "c3", "--frames", "45", "--streams", str(a.streams)])
    sec["configs[2]@1gpu"] = measure_evalset(ev, dev, 1, 0)
    if full:
        ev_serial = parse(["--config", "c3", "--frames", "9", "--streams", str(a.streams), "--no-overlap"])
        serial = measure_evalset(ev_serial, dev, 1, 0)
        sec["configs[2]@1gpu"]["config"]["without_overlap"] = dict(
            frames=9, depth_map_ms_per_frame_inclusive=serial["ms_per_step"], encode_frame_ms=serial["config"]["encode_frame_ms"],
            ray_path_ms_per_frame=serial["config"]["ray_path_ms_per_frame_rank0"])
    torch.cuda.empty_cache()
    c3 = parse(["--views", "5", "--height", "600", "--width", "800", "--coarse", "128", "--fine", "128", "--steps", "2",
                "--warmup", "1", "--streams", str(a.streams), "--cpu-rays", "64", "--cpu-calls", "2", "--eager-chunks", "2"])
    c3.no_cpu_baseline = a.no_cpu_baseline or not full
    c3.no_gpu_eager_baseline = a.no_gpu_eager_baseline or not full
    sec["configs[3]"] = measure_frames(c3, dev, 1, 0)
    return sec


def stub_secondary():
    """Placeholder secondary entries with the fields digest() / projection() read and long filler where the real entries
    carry prose (tests/test_sharding.py::test_bench_headline_size: the real configs[1] command path in seconds)."""
    filler = "x" * 3000
    ent = lambda **kw: dict(dict(value=1.0, ms_per_step=1.0, roofline=dict(frac=0.123456789, note=filler), note=filler), **kw)
    return {"encode_frame": dict(encode_frame_ms=20.123456789, note=filler), "correlate": ent(), "tsdf": ent(),
            "configs[4]_fp32": ent(), "configs[4]_16bit": ent(), "configs[4]+cost_reg_2": ent(),
            "configs[2]@1gpu": ent(ms_per_step=140.123456789, config=dict(encode_frame_ms=20.123456789, ray_path_ms_per_frame_rank0=122.0)),
            "configs[3]": ent(value=123456.789, ms_per_step=650.123456789, config=dict(kernel_ms_per_frame_rank0=dict(gather=200.123456789)))}


def projection(line):
    """What the multi-GPU legs should measure, written down BEFORE any multi-GPU run exists (no such box in the build
    environment): per-rank ray-path time, the producers' share under both modes, the bytes on the wire.  Two models per
    entry: `frame_ms_sum` (producers and rays take turns: what ONE GPU measures -- `hidden_ms_1gpu` below is how much of
    encode_frame the side stream actually hid behind the rays of the previous frame, ~0 on a power-capped package) and
    `frame_ms_max` (perfect overlap, the upper bound); `frame_ms` interpolates with the MEASURED 1-GPU overlap efficiency
    (hidden / encode), so a reader is never shown the optimistic bound alone."""
    from uforecon_amd.evalset import _frame_tensor_shapes

    sec = line.get("secondary") or {}
    ray_ms = line["ms_per_step"]
    enc = (sec.get("encode_frame") or {}).get("encode_frame_ms")
    c2 = sec.get("configs[2]@1gpu") or {}
    hidden, eff = None, 0.0
    if c2 and enc:
        # inclusive ms/frame of the evaluation loop against (its own ray path alone) + (encode alone)
        ray_c2 = c2["config"].get("ray_path_alone_ms") or ray_ms
        hidden = ray_c2 + enc - c2["ms_per_step"]
        eff = min(1.0, max(0.0, hidden / enc))
    bcast = 4 * sum(int(torch.Size(s).numel()) for _, s in _frame_tensor_shapes(512, 640, 3))
    gather = 512 * 640 * 16
    xgmi = 100.0      # GB/s assumed for a one-to-all broadcast (7 links x ~153 GB/s per GPU, a ring / tree keeps one busy)
    out = dict(assumptions=f"rays are independent: the ray path divides by N (row tiles of >= 2048-ray chunks); broadcast at "
                           f"{xgmi:g} GB/s effective over xGMI (assumed, unmeasured); fp32 on the wire -- fp16 frustums were "
                           f"checked and REJECTED: the oracle on c2_hier_512x640_interior moves depth by 3.3e-4 (bf16: 3.5e-3)",
               hidden_ms_1gpu=hidden, overlap_efficiency_1gpu=eff,
               broadcast_bytes_per_frame=bcast, all_gather_bytes_per_frame=gather, per_n={})

    def model(ray, prod):
        lo, hi = max(ray, prod), ray + prod
        return dict(frame_ms_sum=hi, frame_ms_max=lo, frame_ms=hi - eff * (hi - lo))
    for n in (2, 4, 8):
        e = dict(ray_path_ms_per_rank=ray_ms / n, ray_only_scaling=n)
        if enc:
            b_ms = bcast / (xgmi * 1e9) * 1e3
            one = ray_ms + enc - (hidden or 0.0)            # the 1-GPU inclusive frame this is a scaling OF
            e["replicated"] = dict(encode_ms_per_rank=enc, **model(ray_ms / n, enc))
            e["sharded"] = dict(encode_ms_per_rank=enc / n, broadcast_ms=b_ms, **model(ray_ms / n, enc / n + b_ms))
            for k in ("replicated", "sharded"):
                e[k]["frame_scaling"] = one / e[k]["frame_ms"]
                e[k]["frame_scaling_if_sum"] = one / e[k]["frame_ms_sum"]
                e[k]["frame_scaling_if_max"] = one / e[k]["frame_ms_max"]
        out["per_n"][str(n)] = e
    return out


def digest(line):
    """The numbers a reader of a TRUNCATED line needs, at its very end."""
    sec = line.get("secondary") or {}
    d = dict(frame_ms=round(line["ms_per_step"], 2), rays_per_s=round(line["value"]),
             view_t_ms=round(line["roofline"]["avg_launch_ms"], 4), view_t_frac=round(line["roofline"]["frac"], 3),
             ray_t_ms=round(line["roofline"]["ray_transformer"]["avg_launch_ms"], 4),
             ray_t_frac=round(line["roofline"]["ray_transformer"]["frac"], 3))
    c3 = sec.get("configs[3]")
    if c3:
        d["configs[3]"] = dict(rays_per_s=round(c3["value"]), frame_ms=round(c3["ms_per_step"], 1), view_t_frac=round(c3["roofline"]["frac"], 3),
                               gather_ms=round(c3["config"]["kernel_ms_per_frame_rank0"].get("gather", 0.0), 1))
    c2 = sec.get("configs[2]@1gpu")
    if c2:
        # (the loop's own encode interval is stretched by the low stream priority: the stand-alone encode_frame_ms is below;
        # hidden = ray path alone + encode alone - inclusive)
        d["configs[2]@1gpu"] = dict(frame_ms_inclusive=round(c2["ms_per_step"], 1))
        if line.get("projected") and line["projected"].get("hidden_ms_1gpu") is not None:
            d["configs[2]@1gpu"]["encode_hidden_ms"] = round(line["projected"]["hidden_ms_1gpu"], 1)
    for k in ("configs[4]_fp32", "configs[4]_16bit", "configs[4]+cost_reg_2"):
        if sec.get(k):
            d[k] = dict(step_ms=round(sec[k]["ms_per_step"], 2), frac=round(sec[k]["roofline"]["frac"], 3))
    if sec.get("encode_frame"):
        d["encode_frame_ms"] = round(sec["encode_frame"]["encode_frame_ms"], 1)
    return d


HEADLINE_MAX_BYTES = 4096


def _r(x, nd=8):
    """Numbers of the stdout line: enough digits to recompute every ratio, no 17-digit tails."""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    return x


def headline(full):
    """The ONE stdout line: the contract's fields, `roofline` / `cpu_baseline` / `gpu_eager_baseline` as numbers plus a
    one-sentence sample description, a digest of the secondary entries -- and nothing else.  Everything that explains
    (peak basis, power note, arithmetic, per-kernel split, per-rank tables, projection) lives in bench_secondary.json
    and on stderr.  Guaranteed < HEADLINE_MAX_BYTES: optional blocks are dropped, last first, until it fits."""
    cfg = full["config"]
    rf = full.get("roofline")
    h = dict(metric=full["metric"], value=_r(full["value"], 9), unit=full["unit"], n_gpus=full["n_gpus"], steps=full["steps"],
             warmup=full["warmup"], ms_per_step=_r(full["ms_per_step"], 9), higher_is_better=True, scaling=full["scaling"],
             vs_baseline=_r(full.get("vs_baseline")), dtype=full["dtype"], mfma_operand_dtype=full.get("mfma_operand_dtype"),
             data=full["data"])
    c = dict(workload=cfg["workload"])
    for k in ("rays_per_frame", "chunk_rays", "side_streams", "depth_map_ms_per_frame", "frames", "producers",
              "depth_map_ms_per_frame_inclusive", "encode_frame_ms", "ray_path_only_rays_per_s", "vs_baseline_denominator"):
        if cfg.get(k) is not None:
            c[k] = _r(cfg[k], 7)
    if cfg.get("per_rank"):       # N > 1: [rank, rays, wall ms per step, all-gather ms per step]; the kernel split is in the side file
        c["per_rank"] = [[r["rank"], r.get("rays"), _r(r.get("wall_ms_per_step", r.get("ray_path_ms_mean")), 5),
                          _r(r.get("all_gather_ms_per_step", r.get("all_gather_ms_mean")), 4)] for r in cfg["per_rank"]]
    h["config"] = c
    if rf:
        keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches",
                "algorithmic_flop_per_launch", "mfma_busy_frac", "valu_busy_frac", "issued_over_algorithmic")
        h["roofline"] = {k: _r(rf.get(k)) for k in keep}
        rt = rf.get("ray_transformer")
        if rt:
            h["roofline"]["ray_transformer"] = {k: _r(rt.get(k)) for k in ("achieved", "frac", "avg_launch_ms", "traffic",
                                                                           "mfma_busy_frac", "issued_over_algorithmic")}
    for k in ("cpu_baseline", "gpu_eager_baseline"):
        if full.get(k):
            h[k] = {kk: _r(vv) for kk, vv in full[k].items()}
    if full.get("secondary_error"):
        h["secondary_error"] = str(full["secondary_error"])[:200]
    h["details"] = os.path.basename(SIDE_FILE) + " (also on stderr)"
    if full.get("digest"):
        h["digest"] = full["digest"]
    for drop in ((), ("details",), ("config", "per_rank"), ("gpu_eager_baseline", "sample"), ("cpu_baseline", "sample"), ("digest",)):
        if drop:
            d = h
            for k in drop[:-1]:
                d = d.get(k, {})
            d.pop(drop[-1], None)
        text = json.dumps(h, separators=(",", ":"))
        if len(text) < HEADLINE_MAX_BYTES:
            return text
    raise RuntimeError(f"headline line is {len(text)} bytes")


def emit(full, path=None):
    """Side file + stderr get the whole record; stdout gets the headline as its ONLY and LAST line."""
    path = path or SIDE_FILE
    doc = json.dumps(full, indent=1)
    try:
        with open(path, "w") as f:
            f.write(doc + "\n")
    except OSError as e:          # a read-only checkout must not cost the run its headline
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    print(doc, file=sys.stderr, flush=True)
    sys.stderr.flush()
    print(headline(full), flush=True)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    if os.environ.get("UFR_BENCH_SHARE_GPU"):      # diagnostics only: all ranks on one device (with --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    if a.producers is None:
        a.producers = "sharded" if world > 1 else "replicated"
    if a.config == "c3":
        line = measure_evalset(a, dev, world, rank)
        if rank == 0:
            emit(line, a.details)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    line = measure_frames(a, dev, world, rank)
    if rank == 0:
        if world == 1 and not a.no_secondary and config_name(a) == "configs[1]":
            if a.stub_secondary:
                line["secondary"] = stub_secondary()
            else:
                try:
                    line["secondary"] = secondary_measurements(a, dev)
                except Exception as e:  # noqa: BLE001 -- the headline line must reach the driver whatever a secondary leg does
                    import traceback
                    line["secondary_error"] = f"{type(e).__name__}: {e}"
                    traceback.print_exc(file=sys.stderr)
        if world == 1 and config_name(a) == "configs[1]":
            line["projected"] = projection(line)
        line["digest"] = digest(line)
        emit(line, a.details)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
