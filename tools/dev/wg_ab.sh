#!/bin/bash
# A/B of conv3d weight-gradient variants: tools/dev/wg_ab.sh variant...  ("default" = lib/libufr.so)
cd "$(dirname "$0")/../.."
for v in "$@"; do
  lib=uforecon_amd/lib/libufr_$v.so; [ "$v" = default ] && lib=uforecon_amd/lib/libufr.so
  echo "== $v"
  UFR_LIB=$PWD/$lib python tools/dev/conv3d_bwd_probe.py 2>&1 | grep -E "^stage|totals" | awk '{printf "%s %s %s%s %s | ", $1, $2, $3, $4, $(NF-1)} /features|totals/{print ""}'
done
