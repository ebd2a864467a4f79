#!/bin/bash
# round-6 A/B: does a low-priority producer stream beside high-priority ray streams hide encode_frame? (configs[2] on one GPU)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
for v in prio flat serial; do
  case $v in
    prio) env -u X python bench.py --config c3 --frames 24 --details gpurun_out/prio_$v.json 2>/dev/null ;;
    flat) UFR_SIDE_PRIORITY_OFF=1 UFR_ENC_PRIORITY_OFF=1 python bench.py --config c3 --frames 24 --details gpurun_out/prio_$v.json 2>/dev/null ;;
    serial) python bench.py --config c3 --frames 24 --no-overlap --details gpurun_out/prio_$v.json 2>/dev/null ;;
  esac | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$v', 'frame_ms', round(d['ms_per_step'],2), 'encode_ms', round(c['encode_frame_ms'],2), 'ray_only_rays/s', round(c['ray_path_only_rays_per_s']))"
done; done > gpurun_out/prio_ab.txt 2>&1
cat gpurun_out/prio_ab.txt
