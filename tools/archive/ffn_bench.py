#!/usr/bin/env python3
"""The Conformer feed-forward module (LN - 256->2048 - SiLU - 2048->256 - 0.5x + residual) at the encoder's batch shape: fused launch vs
LayerNorm + two split GEMMs.  GPU box; run under rocprofv3 --kernel-trace for kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import decode_ops
dev = torch.device("cuda")
for B, T in ((32, 197), (32, 137), (64, 197)):
    C, H = 256, 2048
    ln = torch.nn.LayerNorm(C).to(dev).eval(); l1 = torch.nn.Linear(C, H).to(dev).eval(); l2 = torch.nn.Linear(H, C).to(dev).eval()
    x = torch.randn(B, T, C, device=dev)

    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    with torch.no_grad():
        tf = timeit(lambda: decode_ops.ffn_fused(x, ln, l1, l2, "silu", residual=x, alpha=0.5))
        tt = timeit(lambda: decode_ops.linear(decode_ops.linear(decode_ops.layer_norm(x, ln), l1, act="silu"), l2, residual=x, alpha=0.5))
    print(f"B={B} T={T}: fused {tf:6.1f} us   LN + two GEMMs {tt:6.1f} us")
