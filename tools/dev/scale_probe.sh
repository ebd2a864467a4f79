for c in 384 256 192 128; do
UFR_RW_CHUNKS=$c python tools/bench_train.py 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ray chunks $c', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['config']['kernel_ms_per_step_rank0'].items() if 'wgrad' in k})"
done
