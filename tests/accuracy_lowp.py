"""development aid: aggregate-stage error of the 16-bit matrix mode vs the fp32 oracle (run on a GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import accuracy_report as A
from uforecon_amd import ops
for mode in (ops.PRECISION_FP32, ops.PRECISION_16BIT):
    ops.set_matrix_precision(mode)
    print("=== mode", mode)
    A.main()
