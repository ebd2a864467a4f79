#!/bin/bash
# round-6 A/B of the view transformer variants: parity of each variant on the golden fixture, then kernel times (same box)
cd "$(dirname "$0")/../.."
{
for v in "$@"; do
  if [ "$v" != default ]; then
    echo "== parity $v"; UFR_LIB=$PWD/uforecon_amd/lib/libufr_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
  fi
done
echo "== kernels (4096 x 128 points back to back)"
bash tools/dev/ab_kernels.sh 4 "$@"
echo "== frame (configs[1], 3 side streams)"
for r in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then L=$PWD/uforecon_amd/lib/libufr.so; else L=$PWD/uforecon_amd/lib/libufr_$v.so; fi
  UFR_LIB=$L python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-gpu-eager-baseline --details /tmp/ab_$v.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v frame_ms', round(d['ms_per_step'],2), 'view_t_ms', d['roofline']['avg_launch_ms'], 'ray_t_ms', d['roofline']['ray_transformer']['avg_launch_ms'])"
done; done
} > gpurun_out/ab_r6_vt.txt 2>&1
cat gpurun_out/ab_r6_vt.txt
