#!/bin/bash
# usage: tools/dev/ab_gather.sh rounds variant...   (gather time of tools/bench_kernels.py; "default" = lib/libufr.so)
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do for v in "$@"; do
  if [ "$v" = default ]; then L=$PWD/uforecon_amd/lib/libufr.so; else L=$PWD/uforecon_amd/lib/libufr_$v.so; fi
  UFR_LIB=$L python tools/bench_kernels.py 2>&1 | grep -E "gather " | awk -v v=$v '{print v, $2, $3}'
done; done | sort | awk '{k=$1" "$2; if(!(k in m)||$3<m[k])m[k]=$3; s[k]+=$3; n[k]++} END{for(k in m) printf "%-28s min %.3f mean %.3f\n", k, m[k], s[k]/n[k]}' | sort
