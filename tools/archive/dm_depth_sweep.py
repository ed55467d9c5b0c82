#!/usr/bin/env python3
"""dense matrix-core DP: pipeline depth x chunk height sweep (dm_depth, dm_mt) at the shapes of the r02 tables."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
lib = _lib.load(); st = _lib.current_stream_handle()
for (B, T, L) in [(4, 256, 2048), (32, 100, 400), (16, 150, 1024), (32, 512, 4096)]:
    TR = L - 1
    g = torch.Generator(device="cuda").manual_seed(0)
    match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6
    ol = torch.full((B,), L, device="cuda") - torch.arange(B, device="cuda") % 5; tl = torch.full((B,), T, device="cuda") - torch.arange(B, device="cuda") % 4
    links = torch.empty(B, L, TR, device="cuda")
    i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
    for b0 in range(0, B, 2):
        raw = torch.randn(min(2, B - b0), L, TR, device="cuda", generator=g)
        valid = (i + d + 1) < ol[b0:b0 + 2].view(-1, 1, 1)
        links[b0:b0 + 2] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
        del raw, valid
    alpha = torch.empty_like(match); beta = torch.empty_like(match)
    def run():
        assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _lib.ptr(beta), None, B, T, L, TR, None, 0, st) == 0
    out = []
    for mt, dep in ((1, 1), (1, 2), (2, 1), (2, 2)):
        _lib.set_option("dp_path", 9); _lib.set_option("dm_mt", mt); _lib.set_option("dm_depth", dep)
        run(); run(); torch.cuda.synchronize()
        n = 3 if L >= 4096 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        out.append(f"mt={mt} depth={dep}: {e0.elapsed_time(e1) / n:.3f} ms")
    _lib.set_option("dp_path", 0); _lib.set_option("dm_mt", 0); _lib.set_option("dm_depth", 0)
    print(f"B={B} T={T} L={L} TR={TR}: " + " | ".join(out), flush=True)
    del links, match, alpha, beta
