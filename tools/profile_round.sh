#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r2      -> gpurun_out/prof_r2/*, then  python tools/summarize_prof.py gpurun_out/prof_r2 profiles/r2
# Pass 1: kernel trace + stats of the single-stream bench (per-kernel durations the bench's roofline must agree with).
# Passes 2-4: PMC counters, each in its own run with --kernel-trace only (SQ block; FETCH_SIZE; WRITE_SIZE need separate passes).
TAG=${1:-rN}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-gpu-eager-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $BENCH > $OUT/bench_line.json 2> $OUT/stats_err.txt
PMC_BENCH="bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-gpu-eager-baseline"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d $OUT -o pmc_sq -- python3 $PMC_BENCH > /dev/null 2> $OUT/pmc_sq_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o pmc_fetch -- python3 $PMC_BENCH > /dev/null 2> $OUT/pmc_fetch_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT -o pmc_write -- python3 $PMC_BENCH > /dev/null 2> $OUT/pmc_write_err.txt
# training step (configs[4]): kernel trace + stats of tools/bench_train.py (forward + backward kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 tools/bench_train.py --steps 3 --warmup 1 > $OUT/train_line.json 2> $OUT/train_err.txt
# the raw per-dispatch tables are large: keep only what the summariser needs
rm -f $OUT/*_agent_info.csv $OUT/stats_kernel_trace.csv $OUT/train_kernel_trace.csv $OUT/pmc_*_kernel_trace.csv
ls -la $OUT
