#!/bin/bash
# same-box A/B of library variants on the headline: tools/dev/bench_lib_ab.sh rounds variant...  ("default" = lib/libufr.so)
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do for v in "$@"; do
  lib=uforecon_amd/lib/libufr_$v.so; [ "$v" = default ] && lib=uforecon_amd/lib/libufr.so
  UFR_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-gpu-eager-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'frame', round(d['ms_per_step'],2), 'view', round(r['avg_launch_ms'],4), 'ray', round(r['ray_transformer']['avg_launch_ms'],4))"
done; done
