for v in "" _abl1 _abl2 _abl3 _abl4; do echo "== $v"; UFR_LIB=$PWD/uforecon_amd/lib/libufr$v.so python tools/dev/conv3d_planes_probe.py 2>&1 | grep "stage3"; done > gpurun_out/r6_planes_abl.txt
