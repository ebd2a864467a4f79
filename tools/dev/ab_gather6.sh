cd "$(dirname "$0")/../.."
for r in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then L=$PWD/uforecon_amd/lib/libufr.so; else L=$PWD/uforecon_amd/lib/libufr_$v.so; fi
  UFR_LIB=$L python tools/bench_kernels.py 2>&1 | grep -E "gather" | awk -v v=$v '{print v, $2, $3}'
done; done
