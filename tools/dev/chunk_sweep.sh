for c in 2048 4096 6144 8192; do for s in 2 3 4; do
python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-gpu-eager-baseline --chunk $c --streams $s 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $c streams $s', round(d['ms_per_step'],1))"
done; done
