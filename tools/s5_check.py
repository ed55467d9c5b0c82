#!/usr/bin/env python3
"""strip5 (dp_path 8) vs the fp64 oracle and vs strip4g (dp_path 5): correctness on small shapes, timing at C2.  usage: s5_check.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daspeech_amd import custom_ops as ops, _lib
from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs

def check(B, T, L, TR, seed, cpl, peaked=0.0):
    match, links, ol, tl = make_dag_inputs(seed, B, T, L, TR)
    if peaked:
        jj = np.arange(L, dtype=np.float32)[None, None, :]
        centre = (np.arange(T, dtype=np.float32) * (L - 1) / (T - 1))[None, :, None]
        match = (match * 0.1 - peaked * np.abs(jj - centre)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    m, k, o, tt = t(match), t(links), t(ol), t(tl)
    m.requires_grad_()
    _lib.set_option("dp_path", 8); _lib.set_option("s5_cpl", cpl)
    loss, (a, b) = ops.dag_loss_with_alpha_beta(m, k, o, tt)
    st = _lib.last_launch_status(); fb = _lib.last_fallback_count()
    _lib.set_option("dp_path", 0)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    ok = np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64))
    fa, fb_ = np.isfinite(a64) & np.isfinite(a), np.isfinite(b64) & np.isfinite(b)
    ea = np.abs(a[fa] - a64[fa]).max() if fa.any() else 0; eb = np.abs(b[fb_] - b64[fb_]).max() if fb_.any() else 0
    print(f"B={B} T={T} L={L} TR={TR} cpl={cpl} peaked={peaked}: status {st} exact-cells {fb} inf-pattern {'ok' if ok else 'MISMATCH'} max|da| {ea:.2e} max|db| {eb:.2e}", flush=True)
    if not ok:
        bad = np.argwhere(np.isneginf(a) != np.isneginf(a64))[:5]; print("   alpha mismatches (b,t,j):", bad.tolist(), [(a[tuple(i)], a64[tuple(i)]) for i in bad])
        bad = np.argwhere(np.isneginf(b) != np.isneginf(b64))[:5]; print("   beta mismatches (b,t,j):", bad.tolist(), [(b[tuple(i)], b64[tuple(i)]) for i in bad])
    return ok and ea < 3e-5 * T + 1e-3 and eb < 3e-5 * T + 1e-3 and st == 0

allok = True
for cpl in (2, 4):
    for shape in [(2, 12, 64, 5), (3, 20, 512, 32), (2, 33, 2304, 20), (3, 40, 1028, 32), (40, 9, 1024, 32), (2, 24, 4096, 32), (33, 16, 4096, 32), (2, 64, 1024, 7)]:
        allok &= check(*shape, seed=21 + shape[2], cpl=cpl)
    for slope in (2.0, 12.0, 40.0):
        allok &= check(2, 40, 1024, 32, 123, cpl, peaked=slope)
print("ALL OK" if allok else "FAILURES")
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0 if allok else 1)

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B, T, L, TR = 32, 512, 4096, 32
g = torch.Generator(device="cuda").manual_seed(0)
match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 9
raw = torch.randn(B, L, TR, device="cuda", generator=g)
ol = torch.full((B,), L, device="cuda") - torch.arange(B, device="cuda") % 5; tl = torch.full((B,), T, device="cuda") - torch.arange(B, device="cuda") % 4
i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
valid = (i + d + 1) < ol.view(B, 1, 1)
links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
lib = _lib.load(); st = _lib.current_stream_handle()
alpha = torch.empty_like(match); beta = torch.empty_like(match)
def run(a, b):
    rc = lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(a), _lib.ptr(b), None, B, T, L, TR, None, 0, st)
    assert rc == 0
res = {}
for name, path, cpl in (("strip4g", 5, 0), ("strip5 cpl2", 8, 2), ("strip5 cpl4", 8, 4)):
    _lib.set_option("dp_path", path); _lib.set_option("s5_cpl", cpl)
    tb = timeit(lambda: run(alpha, beta)); ta = timeit(lambda: run(alpha, None))
    res[name] = (alpha.clone(), beta.clone())
    print(f"{name}: alpha||beta {tb:.3f} ms ({1.107296256 / tb * 1e3 / 8000:.3f} of 8 TB/s), alpha only {ta:.3f} ms, status {_lib.last_launch_status()}", flush=True)
_lib.set_option("dp_path", 0); _lib.set_option("s5_cpl", 0)
for name in ("strip5 cpl2", "strip5 cpl4"):
    for w, x, y in (("alpha", res[name][0], res["strip4g"][0]), ("beta", res[name][1], res["strip4g"][1])):
        f = torch.isfinite(y)
        print(f"  {name} vs strip4g {w}: inf pattern equal {bool(torch.equal(torch.isneginf(x), torch.isneginf(y)))}, max diff {float((x[f] - y[f]).abs().max()):.3e}")
