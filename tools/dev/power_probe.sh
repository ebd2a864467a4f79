#!/bin/bash
# samples rocm-smi power / clocks while a kernel microbenchmark loops (is the view transformer power-limited?)
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30
echo "---- under load"
(for i in 1 2 3; do python tools/bench_kernels.py > /dev/null 2>&1; done) &
BG=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" ; sleep 1.5; done
wait $BG
