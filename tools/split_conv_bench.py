#!/usr/bin/env python3
"""SplitConv1d (fp32-accurate conv / GEMM on the fp16 matrix cores) at the S2ST pipeline's shapes: time and result checksum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.decode_ops import SplitConv1d
torch.manual_seed(0)
for (pos, cin, cout, k) in [(35072, 512, 2048, 1), (35072, 2048, 512, 1), (10560, 256, 1024, 9), (10560, 1024, 256, 9), (10560, 1024, 256, 1), (4384, 256, 2048, 1), (4384, 2048, 256, 1)]:
    B = 32; T = pos // B
    x = torch.randn(B, T, cin, device="cuda")
    w = torch.randn(cout, cin, k, device="cuda") / (cin * k) ** 0.5; b = torch.randn(cout, device="cuda") * 0.1
    conv = SplitConv1d(w, b)
    for _ in range(3): y = conv(x, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = conv(x, relu=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ref = torch.relu(torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), b.double(), padding=(k - 1) // 2)).transpose(1, 2)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"{pos} positions {cin} -> {cout} k={k}: {ms * 1e3:7.1f} us  {2 * pos * cin * cout * k / ms / 1e9:7.1f} TFLOP/s fp32-equivalent ({3 * 2 * pos * cin * cout * k / ms / 1e9:7.1f} of fp16 MFMA work)  max rel err {err:.2e}")
