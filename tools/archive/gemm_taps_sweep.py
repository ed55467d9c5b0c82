#!/usr/bin/env python3
"""Fixed cost vs per-step cost of the split-precision conv / GEMM kernel: the same layer shape at 1, 3, 5, 9 taps.  The slope is the matrix-core
loop, the intercept everything a workgroup does once (tile staging, first weight fetch, epilogue).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.decode_ops import SplitConv1d

SHAPES = [("conformer ffn1 256->2048", 32, 197, 256, 2048), ("conformer ffn2 2048->256", 32, 197, 2048, 256), ("decoder proj 512->512", 32, 394, 512, 512),
          ("decoder fc1 512->2048", 32, 394, 512, 2048), ("decoder fc2 2048->512", 32, 394, 2048, 512), ("tts proj 256->256", 32, 483, 256, 256),
          ("tts conv1 256->1024", 32, 483, 256, 1024), ("tts conv2 1024->256", 32, 483, 1024, 256)]


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for what, B, T, Cin, Cout in SHAPES:
    x = torch.randn(B, T, Cin, device="cuda")
    ts = {}
    for K in (1, 3, 5, 9):
        conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) // 2).cuda()
        sc = SplitConv1d(conv.weight, conv.bias)
        with torch.no_grad():
            ts[K] = timeit(lambda: sc(x))
    slope = (ts[9] - ts[1]) / 8
    fl = 2.0 * B * T * Cin * Cout
    print(f"{what:28s} T={T}: " + "  ".join(f"k={k}: {v:6.1f}" for k, v in ts.items()) + f"   per tap {slope:6.1f} us ({fl / slope / 1e6:5.0f} TF/s), fixed {ts[1] - slope:6.1f} us")
