run() { python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-gpu-eager-baseline $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernel_ms_per_frame_rank0']; print('$1', round(d['ms_per_step'],1), round(k['view_transformer'],1), round(k['ray_transformer'],1))"; }
ARGS="--height 600 --width 800 --views 5 --coarse 128 --fine 128"
run NEW
for v in nf nr nfr; do UFR_LIB=$PWD/uforecon_amd/lib/libufr_$v.so run $v; done
(cd _old_tree && run OLD)
run NEW
for v in nf nr nfr; do UFR_LIB=$PWD/uforecon_amd/lib/libufr_$v.so run $v; done
