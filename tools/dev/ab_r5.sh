#!/bin/bash
# round-5 A/B: parity of the new library, then kernel times of the variants on the same box
# usage: tools/dev/ab_r5.sh <rounds> variant...   (variants: default r4 olddma ...)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
R=$1; shift
{
echo "== parity (default lib)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -4
echo "== A/B"
bash tools/dev/ab_kernels.sh $R "$@"
} > gpurun_out/ab_r5.txt 2>&1
cat gpurun_out/ab_r5.txt
