"""bench.py's stdout contract: ONE line, the last one, under 4 KB, that json.loads round-trips -- whatever the secondary
entries weigh.  Round 5's line was 20.9 KB and the driver recorded `parsed: null`; the old contract test ran a 64x96 frame,
where no secondary entries are attached, and never saw the line the driver sees.  The per-ray loop being timed is the
reference's code1/model.py:814-823."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full(world):
    prose = "p" * 1500
    kern = {f"kernel_{i}": 1.23456789012345 * i for i in range(14)}
    per_rank = [dict(rank=r, rays=327680 // world, wall_ms_per_step=15.123456789, kernel_ms_per_frame=kern,
                     all_gather_ms_per_step=0.123456789) for r in range(world)] if world > 1 else None
    import bench
    full = dict(metric="rays/s (per-ray volume-rendering path, 64+64 hierarchical samples, DTU-shaped 3-view 512x640)",
                value=2669123.456789012, unit="rays/s", n_gpus=world, steps=20, warmup=5, ms_per_step=122.7654321098, higher_is_better=True,
                scaling="strong", vs_baseline=180.123456789, dtype="f32", mfma_operand_dtype="f16x3", data="synthetic",
                config=dict(workload="configs[1]: " + "w" * 200, rays_per_frame=327680, chunk_rays=2048, side_streams=3,
                            depth_map_ms_per_frame=122.7654321098, arithmetic=prose, kernel_ms_per_frame_rank0=kern, per_rank=per_rank,
                            vs_baseline_denominator="d" * 150),
                roofline=dict(bound="mfma", achieved=368.123456789, peak=838.8666666, unit="TFLOP/s", frac=368.123456789 / 838.8666666, traffic=287307079.2,
                              traffic_source="profiles/r5_pmc.json", kernel="view_transformer_kernel", avg_launch_ms=0.379912345,
                              launches=160, algorithmic_flop_per_launch=140043616256.0, peak_basis=prose, power_note=prose,
                              mfma_busy_frac=0.4701234, valu_busy_frac=0.5281234, issued_over_algorithmic=1.0731234,
                              ray_transformer=dict(achieved=264.123456, frac=0.3151234, avg_launch_ms=0.24631234, traffic=260650538.0,
                                                   mfma_busy_frac=0.451234, issued_over_algorithmic=1.311234, frac_on_bf16x6_basis=0.63)),
                cpu_baseline=dict(value=88.123456, unit="rays/s", cores=128, kind="port", sample="s" * 160),
                gpu_eager_baseline=dict(value=14800.123, unit="rays/s", kind="port-gpu-eager", device="AMD Instinct MI355X", sample="s" * 260),
                secondary=bench.stub_secondary())
    full["projected"] = dict(assumptions=prose, per_n={str(n): dict(x=prose) for n in (2, 4, 8)})
    full["digest"] = bench.digest(full)
    return full


@pytest.mark.parametrize("world", [1, 8])
def test_headline_is_small_and_complete(world):
    import bench

    full = _full(world)
    assert len(json.dumps(full)) > 20000            # the record that broke the driver's parser in round 5
    text = bench.headline(full)
    assert "\n" not in text and len(text) < 4096
    d = json.loads(text)
    for k in CONTRACT + ("gpu_eager_baseline", "digest"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-8) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-8)
    rf = d["roofline"]
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches",
                       "algorithmic_flop_per_launch", "mfma_busy_frac", "issued_over_algorithmic", "ray_transformer"}
    assert not any(isinstance(v, str) and len(v) > 400 for v in rf.values())           # numbers, not prose
    assert "secondary" not in d and "projected" not in d and "model" not in d["config"]
    if world > 1:
        assert len(d["config"]["per_rank"]) == world and d["config"]["per_rank"][3][0] == 3


def test_headline_sheds_optional_blocks_before_failing():
    import bench

    full = _full(8)
    full["digest"] = {f"k{i}": "v" * 50 for i in range(60)}          # a digest that would not fit by itself
    d = json.loads(bench.headline(full))
    assert "digest" not in d and all(k in d for k in CONTRACT)


@pytest.mark.gpu
def test_bench_headline_size(tmp_path):
    """The command the driver runs (configs[1], full 512x640 frame), with stub secondary entries of worst-case size."""
    side = str(tmp_path / "details.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--stub-secondary", "--details", side,
                        "--eager-chunks", "1", "--cpu-calls", "1", "--cpu-rays", "64"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 1, out[:3]                                    # nothing but the headline on stdout
    assert len(out[-1]) < 4096
    d = json.loads(out[-1])
    for k in CONTRACT + ("gpu_eager_baseline", "digest"):
        assert k in d, k
    assert d["config"]["workload"].startswith("configs[1]") and d["steps"] == 2 and d["warmup"] == 1
    assert abs(d["value"] - 512 * 640 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"] == "view_transformer_kernel" and d["roofline"]["avg_launch_ms"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    full = json.load(open(side))                                     # the whole record went to the side file (and stderr)
    assert "secondary" in full and "projected" in full and full["value"] == pytest.approx(d["value"], rel=1e-8)
    assert '"projected"' in r.stderr
