#!/usr/bin/env python3
"""One split-precision conv / GEMM shape, a few launches (for rocprofv3 --pmc): gemm_one.py B T Cin Cout K [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.decode_ops import SplitConv1d
B, T, Cin, Cout, K = (int(a) for a in sys.argv[1:6])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) // 2).cuda()
x = torch.randn(B, T, Cin, device="cuda")
sc = SplitConv1d(conv.weight, conv.bias)
with torch.no_grad():
    for _ in range(n):
        y = sc(x)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
