for r in 1 2; do
python tools/bench_train.py 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('BARRIER', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['config']['kernel_ms_per_step_rank0'].items() if 'wgrad' in k})"
UFR_LIB=$PWD/uforecon_amd/lib/libufr_rnb.so python tools/bench_train.py 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('FREE', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['config']['kernel_ms_per_step_rank0'].items() if 'wgrad' in k})"
done
python tools/bench_train.py --precision 16bit 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('BARRIER16', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['config']['kernel_ms_per_step_rank0'].items() if 'wgrad' in k})"
UFR_LIB=$PWD/uforecon_amd/lib/libufr_rnb.so python tools/bench_train.py --precision 16bit 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('FREE16', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['config']['kernel_ms_per_step_rank0'].items() if 'wgrad' in k})"
python -m pytest tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -n 1
