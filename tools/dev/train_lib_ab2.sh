#!/bin/bash
# round-6 training-step A/B over library variants on one box: step ms + per-kernel ms (non-overlapped steps)
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then L=$PWD/uforecon_amd/lib/libufr.so; else L=$PWD/uforecon_amd/lib/libufr_$v.so; fi
  for prec in fp32 16bit; do
  UFR_LIB=$L python tools/bench_train.py --steps 40 --warmup 5 --precision $prec --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['config']['kernel_ms_per_step_rank0']
        print('$v', '$prec', 'ms_per_step', round(d['ms_per_step'], 3), {n: round(k[n], 3) for n in ('view_tape','view_dgrad','view_wgrad','ray_wgrad','ray_dgrad','gather_bwd') if n in k})"
  done
done; done
