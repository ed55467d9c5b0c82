#!/usr/bin/env python3
"""dense K5 (block products on the matrix cores) vs the fp64 oracle and vs the tiled log-space kernel (k5_path 1); timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daspeech_amd import custom_ops as ops, _lib
from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs
allok = True
for (B, T, L, TR, masked) in [(3, 24, 200, 199, False), (2, 40, 256, 255, True), (4, 33, 130, 129, False), (2, 20, 500, 100, False), (2, 70, 400, 399, True), (1, 9, 1024, 1023, False), (3, 18, 192, 191, False)]:
    match, links, ol, tl = make_dag_inputs(17 + L, B, T, L, TR)
    if masked:
        rng = np.random.default_rng(L); match[rng.random(match.shape) < 0.1] = -np.inf
    t = lambda a: torch.from_numpy(a).cuda()
    res = {}
    for k5 in (0, 1):
        _lib.set_option("k5_path", k5)
        m, k, o, tt = t(match).requires_grad_(), t(links).requires_grad_(), t(ol), t(tl)
        loss = ops.dag_loss(m, k, o, tt)
        fin = torch.isfinite(loss)
        gm, gk = torch.autograd.grad(loss[fin].sum(), [m, k])
        res[k5] = gk.cpu().numpy()
    _lib.set_option("k5_path", 0)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    _, gl64 = orc.dag_grad(fin.cpu().numpy().astype(np.float64), a64, b64, match, links, ol, tl, np.float64)
    for k5 in (0, 1):
        err = np.abs(res[k5] - gl64); rel = err / (np.abs(gl64) + 1e-7 / 2e-3)
        good = np.allclose(res[k5], gl64, rtol=2e-3, atol=1e-7) and np.isfinite(res[k5]).all()
        allok &= good
        print(f"B={B} T={T} L={L} TR={TR} masked={masked} k5_path={k5}: max abs err {err.max():.2e} max rel {rel.max():.2e} sum {res[k5].sum():.4f} (oracle {gl64.sum():.4f}) -> {'ok' if good else 'FAIL'}", flush=True)
print("ALL OK" if allok else "FAILURES", flush=True)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(4, 256, 2048, 2047), (32, 100, 400, 399), (16, 150, 1024, 1023)]
if len(sys.argv) > 1 and sys.argv[1] == "big": shapes.append((32, 512, 4096, 4095))
for (B, T, L, TR) in shapes:
    g = torch.Generator(device="cuda").manual_seed(0)
    match = (torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6).requires_grad_()
    ol = torch.full((B,), L, device="cuda") - torch.arange(B, device="cuda") % 5; tl = torch.full((B,), T, device="cuda") - torch.arange(B, device="cuda") % 4
    links = torch.empty(B, L, TR, device="cuda")
    for b0 in range(0, B, 2):
        raw = torch.randn(min(2, B - b0), L, TR, device="cuda", generator=g)
        i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
        valid = (i + d + 1) < ol[b0:b0 + 2].view(-1, 1, 1)
        links[b0:b0 + 2] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
        del raw, valid
    links.requires_grad_()
    loss = ops.dag_loss(match, links, ol, tl)
    out = {}
    for name, k5 in (("dense block products", 0), ("tiled log-space", 1)):
        _lib.set_option("k5_path", k5)
        ms = timeit(lambda: torch.autograd.grad(loss.sum(), [match, links], retain_graph=True), n=3)
        out[name] = torch.autograd.grad(loss.sum(), [match, links], retain_graph=True)[1]
        print(f"B={B} T={T} L={L} TR={TR} backward (K4 + K5 {name}): {ms:.3f} ms", flush=True)
    _lib.set_option("k5_path", 0)
    x, y = out["dense block products"], out["tiled log-space"]
    print(f"   max abs diff {float((x - y).abs().max()):.3e}, rel-to-max {float((x - y).abs().max() / y.abs().max()):.3e}, sums {float(x.sum()):.3f} / {float(y.sum()):.3f}", flush=True)
    del links, match, loss, out, x, y; torch.cuda.empty_cache()
