#!/bin/bash
# full GPU test suite with the failures' assertion lines kept (gpurun only returns the tail of stdout)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout ${1:-1500} python -m pytest tests -q -m gpu ${@:2} 2>&1 | grep -E "^(FAILED|ERROR|E  |tests/.*(Error|assert)|[0-9]+ (passed|failed)|>  )" | cut -c1-400 | tail -80 > gpurun_out/gpu_tests.txt
cat gpurun_out/gpu_tests.txt
