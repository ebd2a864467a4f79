#!/usr/bin/env python
"""BASELINE.json configs[3]: 5 source views, 800x600 frame, 128+128 samples (the frustum-lookup / HBM stress case) on one
MI355X.  Same measurement as bench.py (one JSON line; the roofline object describes the view transformer at L = 6 tokens
per point), with the configuration pinned:  python tools/bench_c4.py [--steps K --warmup W ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    pinned = ["--views", "5", "--height", "600", "--width", "800", "--coarse", "128", "--fine", "128"]
    defaults = [] if any(a.startswith("--steps") for a in sys.argv[1:]) else ["--steps", "2", "--warmup", "1"]
    sys.argv = [sys.argv[0], *pinned, *defaults, "--cpu-rays", "64", "--cpu-calls", "2", "--eager-chunks", "2", *sys.argv[1:]]
    bench.main()
