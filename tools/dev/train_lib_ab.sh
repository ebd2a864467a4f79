#!/bin/bash
# same-box A/B of library variants on the training step: tools/dev/train_lib_ab.sh rounds variant...  ("default" = lib/libufr.so)
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do for v in "$@"; do
  lib=uforecon_amd/lib/libufr_$v.so; [ "$v" = default ] && lib=uforecon_amd/lib/libufr.so
  for prec in fp32 16bit; do
    UFR_LIB=$PWD/$lib python tools/bench_train.py --steps 40 --warmup 5 --precision $prec --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', '$prec', round(d['ms_per_step'], 3))"
  done
done; done
