#!/bin/bash
# training-step A/B on one box: tools/dev/train_ab.sh [extra bench_train flags...]  -> step ms of both precisions, fused vs torch loss
cd "$(dirname "$0")/../.."
for prec in fp32 16bit; do
  for extra in "" "--torch-loss"; do
    python tools/bench_train.py --steps 40 --warmup 5 --precision $prec --no-cpu-baseline $extra "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$prec', '$extra', 'ms_per_step', round(d['ms_per_step'], 3))"
  done
done
