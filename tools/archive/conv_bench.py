#!/usr/bin/env python3
"""Split-precision Conv1d / Linear kernel (conv1d_split.hip) per layer shape of the acoustic model: us per call and TFLOP/s of convolution
(the matrix cores issue 3x that).  (r04's K-streaming experiment kernel this tool compared against lives only at commit 072a6ad:
profiles/r04_conv_stream_experiment.txt.)  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import decode_ops
from daspeech_amd.decode_ops import SplitConv1d

SHAPES = [  # (what, B, T, Cin, Cout, K)
    ("decoder q/k/v/out 512->512", 32, 400, 512, 512, 1), ("decoder fc1 512->2048", 32, 400, 512, 2048, 1), ("decoder fc2 2048->512", 32, 400, 2048, 512, 1),
    ("links q/k 1024->512", 32, 400, 1024, 512, 1), ("conformer 256->256", 32, 137, 256, 256, 1), ("conformer ffn 256->2048", 32, 137, 256, 2048, 1),
    ("conformer ffn 2048->256", 32, 137, 2048, 256, 1), ("tts dec fft conv1 256->1024 k9", 32, 330, 256, 1024, 9), ("tts dec fft conv2 1024->256 k9", 32, 330, 1024, 256, 9),
    ("tts dec attn proj 256->256", 32, 330, 256, 256, 1), ("tts enc fft conv1 k9", 32, 45, 256, 1024, 9), ("var pred 256->256 k3", 32, 45, 256, 256, 3),
    ("adaptor 512->1024", 32, 45, 512, 1024, 1), ("subsample 2 512(256pairs)->512 k3", 32, 150, 1024, 512, 3),
]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = {"slice": 0.0}
for what, B, T, Cin, Cout, K in SHAPES:
    conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) // 2).cuda()
    x = torch.randn(B, T, Cin, device="cuda")
    res = {}
    for name in ("slice",):
        sc = SplitConv1d(conv.weight, conv.bias)
        with torch.no_grad():
            res[name] = timeit(lambda: sc(x, relu=True))
        tot[name] += res[name]
    fl = 2.0 * B * T * Cin * Cout * K
    print(f"{what:36s} B={B} T={T}: {res['slice']:7.1f} us ({fl / res['slice'] / 1e6:6.1f} TF/s)")
print("sum", tot)
