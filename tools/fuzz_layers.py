#!/usr/bin/env python3
"""Randomised sweep of the inference layer kernels against fp64 / torch references: SplitConv1d (every tile family incl. residual /
activations / multi-slice inputs), layer_norm, dwconv_bn_silu.   usage: fuzz_layers.py [n_cases] [seed]   (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from daspeech_amd import decode_ops
from daspeech_amd.decode_ops import SplitConv1d
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g))
bad = 0
for case in range(n):
    try:
        B, T = ri(1, 6), ri(1, 400)
        Cin = [128, 256, 512, 1024, 2048][ri(0, 5)]; Cout = [4, 36, 80, 128, 256, 512, 1024][ri(0, 7)]; K = [1, 3, 5, 9][ri(0, 4)]
        if Cin >= 1024 and K > 3 and Cout > 256: Cout = 256
        tag = f"case {case}: conv B={B} T={T} {Cin}->{Cout} k={K}"
        conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) // 2).cuda()
        x = torch.randn(B, T, Cin, device="cuda") * torch.exp2(torch.randint(-6, 4, (B, T, 1), generator=g).float().cuda())
        sc = SplitConv1d(conv.weight, conv.bias)
        act = [None, "relu", "silu", "gelu"][ri(0, 4)]
        res = torch.randn(B, T, Cout, device="cuda") if ri(0, 2) else None
        alpha = [1.0, 0.5][ri(0, 2)]
        with torch.no_grad():
            got = sc(x, act=act, residual=res, alpha=alpha)
            y = F.conv1d(x.double().transpose(1, 2), conv.weight.double(), conv.bias.double(), padding=(K - 1) // 2).transpose(1, 2)
            if act: y = {"relu": torch.relu, "silu": F.silu, "gelu": F.gelu}[act](y)
            y = alpha * y
            if res is not None: y = res.double() + y
        scale = float(y.abs().max()) + 1e-30
        err = float((got.double() - y).abs().max()) / scale
        assert got.shape == y.shape and torch.isfinite(got).all() and err < 6e-6, f"conv err {err:.3e}"
        C = [36, 80, 256, 512, 1024, 2048][ri(0, 6)]
        tag = f"case {case}: layer_norm rows={B * T} C={C}"
        ln = torch.nn.LayerNorm(C).cuda().eval()
        with torch.no_grad():
            ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
            xx = torch.randn(B, T, C, device="cuda") * 3 + 1.5
            a, b = decode_ops.layer_norm(xx, ln), F.layer_norm(xx.double(), (C,), ln.weight.double(), ln.bias.double(), ln.eps)
        assert float((a.double() - b).abs().max()) < 5e-6 * (float(b.abs().max()) + 1), "layer_norm"
        Cd = [8, 32, 64, 128, 256][ri(0, 5)]; Kd = [3, 7, 15, 31][ri(0, 4)]
        tag = f"case {case}: dwconv B={B} T={T} C={Cd} K={Kd}"
        dw = torch.nn.Conv1d(Cd, Cd, Kd, padding=(Kd - 1) // 2, groups=Cd, bias=False).cuda(); bn = torch.nn.BatchNorm1d(Cd).cuda().eval()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.3, 2.0)
            xd = torch.randn(B, T, Cd, device="cuda")
            want = F.silu(F.batch_norm(F.conv1d(xd.double().transpose(1, 2), dw.weight.double(), None, padding=(Kd - 1) // 2, groups=Cd),
                                       bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(), False, 0.0, bn.eps)).transpose(1, 2)
            gotd = decode_ops.dwconv_bn_silu(xd, dw.weight, bn)
        assert float((gotd.double() - want).abs().max()) < 2e-5, "dwconv"
    except Exception as e:   # noqa
        bad += 1; print("FAIL", tag, "->", repr(e)[:300])
print(f"{n} cases, {bad} failures")
