run() { python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-gpu-eager-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config']['kernel_ms_per_frame_rank0'])"; }
run NEW
(cd _old_tree && run OLD)
run NEW2
(cd _old_tree && python tools/bench_kernels.py 2>/dev/null | grep transformer)
python tools/bench_kernels.py 2>/dev/null | grep transformer
