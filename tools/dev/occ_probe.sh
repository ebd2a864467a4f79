for lib in occ2 occ3; do for pad in 0 20000 45000 90000; do
  echo "$lib pad=$pad: $(UFR_VT_PAD_LDS=$pad UFR_LIB=$PWD/uforecon_amd/lib/libufr_$lib.so python tools/bench_kernels.py 2>&1 | grep view_)"
done; done
