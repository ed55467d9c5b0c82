#!/usr/bin/env python3
"""Run only dag_best_alignment a few times (for rocprofv3 counter passes).  usage: run_align.py B T L TR path [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import inputs

B, T, L, TR, path = [int(v) for v in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
m, k, ol, tl = inputs(B, T, L, TR)
_lib.set_option("dp_path", path)
with torch.no_grad():
    for _ in range(n):
        ops.dag_best_alignment(m, k, ol, tl)
torch.cuda.synchronize()
print("status", _lib.last_launch_status())
