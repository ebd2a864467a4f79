#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r3      -> gpurun_out/prof_r3/*, then  python tools/summarize_prof.py gpurun_out/prof_r3 profiles/r3
# Pass 1: kernel trace + stats of the single-stream bench (per-kernel durations the bench's roofline must agree with).
# Passes 2-4: PMC counters, each in its own run with --kernel-trace only (SQ block; FETCH_SIZE; WRITE_SIZE need separate passes).
# Then the same for the training step (configs[4], tools/bench_train.py: stats + SQ / FETCH / WRITE passes, fp32 mode; stats
# of the 16-bit mode) and FETCH / WRITE passes of configs[3] (tools/bench_c4.py: the L = 6 view transformer's HBM traffic).
TAG=${1:-rN}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
BENCH="bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-gpu-eager-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $BENCH > $OUT/bench_line.json 2> $OUT/stats_err.txt
PMC_BENCH="bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-gpu-eager-baseline --no-secondary"
rocprofv3 --kernel-trace --output-format csv --pmc $SQ -d $OUT -o pmc_sq -- python3 $PMC_BENCH > /dev/null 2> $OUT/pmc_sq_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o pmc_fetch -- python3 $PMC_BENCH > /dev/null 2> $OUT/pmc_fetch_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT -o pmc_write -- python3 $PMC_BENCH > /dev/null 2> $OUT/pmc_write_err.txt
# training step (configs[4]): kernel trace + stats (forward + backward kernels), both precisions; counters in the fp32 mode.
# UFR_BT_OVERLAP=0: every stage of the backward on one stream, so that a kernel's duration is its own (the timed steps of
# tools/bench_train.py overlap independent stages on three streams; its per-kernel numbers come from non-overlapped steps too)
export UFR_BT_OVERLAP=0
TRAIN="tools/bench_train.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 $TRAIN > $OUT/train_line.json 2> $OUT/train_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train16 -- python3 $TRAIN --precision 16bit > $OUT/train16_line.json 2> $OUT/train16_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o traincr -- python3 $TRAIN --cost-reg > $OUT/traincr_line.json 2> $OUT/traincr_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o encoder -- python3 tools/bench_encoder.py > $OUT/encoder_out.txt 2> $OUT/encoder_err.txt
TRAIN_PMC="tools/bench_train.py --steps 2 --warmup 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv --pmc $SQ -d $OUT -o train_pmc_sq -- python3 $TRAIN_PMC > /dev/null 2> $OUT/train_pmc_sq_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o train_pmc_fetch -- python3 $TRAIN_PMC > /dev/null 2> $OUT/train_pmc_fetch_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT -o train_pmc_write -- python3 $TRAIN_PMC > /dev/null 2> $OUT/train_pmc_write_err.txt
unset UFR_BT_OVERLAP
# configs[3] (5 views, 800x600, 128+128): HBM traffic of the L = 6 kernels
C4="tools/bench_c4.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-gpu-eager-baseline"
rocprofv3 --kernel-trace --output-format csv --pmc $SQ -d $OUT -o c4_pmc_sq -- python3 $C4 > /dev/null 2> $OUT/c4_pmc_sq_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o c4_pmc_fetch -- python3 $C4 > /dev/null 2> $OUT/c4_pmc_fetch_err.txt
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT -o c4_pmc_write -- python3 $C4 > /dev/null 2> $OUT/c4_pmc_write_err.txt
# the raw per-dispatch tables are large: keep only what the summariser needs
rm -f $OUT/*_agent_info.csv $OUT/*_kernel_trace.csv
ls -la $OUT
