#!/usr/bin/env python3
"""dense-window MFMA DP (dp_path 9) vs the fp64 oracle and the log-space dense kernel (dp_path 1); timings.  usage: dm_check.py [quick|big]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daspeech_amd import custom_ops as ops, _lib
from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs

def check(B, T, L, TR, seed, mt=0, masked=False):
    match, links, ol, tl = make_dag_inputs(seed, B, T, L, TR)
    if masked:      # force-emit style rows: -inf everywhere except one column, plus scattered -inf
        rng = np.random.default_rng(seed)
        match[0, min(3, T - 1), :] = -np.inf; match[0, min(3, T - 1), min(L - 1, 10)] = 0.0
        match[rng.random(match.shape) < 0.1] = -np.inf
    t = lambda a: torch.from_numpy(a).cuda()
    m, k, o, tt = t(match), t(links), t(ol), t(tl)
    m.requires_grad_()
    _lib.set_option("dp_path", 9); _lib.set_option("dm_mt", mt)
    loss, (a, b) = ops.dag_loss_with_alpha_beta(m, k, o, tt)
    st = _lib.last_launch_status(); fb = _lib.last_fallback_count()
    _lib.set_option("dp_path", 0); _lib.set_option("dm_mt", 0)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    ok = np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64)) and not np.isnan(a).any() and not np.isnan(b).any()
    fa, fb_ = np.isfinite(a64) & np.isfinite(a), np.isfinite(b64) & np.isfinite(b)
    ea = np.abs(a[fa] - a64[fa]).max() if fa.any() else 0; eb = np.abs(b[fb_] - b64[fb_]).max() if fb_.any() else 0
    scale = max(np.abs(a64[fa]).max() if fa.any() else 1, 1.0)
    good = ok and ea < 3e-6 * scale + 3e-5 * T + 1e-3 and eb < 3e-6 * scale + 3e-5 * T + 1e-3 and st == 0
    print(f"B={B} T={T} L={L} TR={TR} mt={mt} masked={masked}: status {st} exact-cells {fb} inf-pattern {'ok' if ok else 'MISMATCH'} max|da| {ea:.2e} max|db| {eb:.2e} -> {'ok' if good else 'FAIL'}", flush=True)
    if not ok:
        for nm, x, y in (("alpha", a, a64), ("beta", b, b64)):
            bad = np.argwhere((np.isneginf(x) != np.isneginf(y)) | np.isnan(x))[:6]
            if len(bad): print(f"   {nm} mismatches (b,t,j):", bad.tolist(), [(float(x[tuple(i)]), float(y[tuple(i)])) for i in bad])
    return good

allok = True
for shape in [(3, 24, 200, 199), (2, 40, 256, 255), (4, 33, 130, 129), (2, 20, 500, 100), (2, 70, 400, 399), (1, 9, 1024, 1023), (3, 18, 192, 191)]:
    for mt in (1, 2):
        allok &= check(*shape, seed=7 + shape[2], mt=mt)
allok &= check(3, 30, 256, 255, 5, 1, masked=True)
allok &= check(2, 30, 333, 332, 6, 2, masked=True)
print("ALL OK" if allok else "FAILURES", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0 if allok else 1)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes = [(4, 256, 2048, 2047), (32, 100, 400, 399), (16, 150, 1024, 1023), (64, 60, 256, 255)]
if len(sys.argv) > 1 and sys.argv[1] == "big": shapes.append((32, 512, 4096, 4095))
lib = _lib.load(); st = _lib.current_stream_handle()
for (B, T, L, TR) in shapes:
    g = torch.Generator(device="cuda").manual_seed(0)
    match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6
    ol = torch.full((B,), L, device="cuda") - torch.arange(B, device="cuda") % 5; tl = torch.full((B,), T, device="cuda") - torch.arange(B, device="cuda") % 4
    links = torch.empty(B, L, TR, device="cuda")
    for b0 in range(0, B, 4):        # build in slices (memory)
        raw = torch.randn(min(4, B - b0), L, TR, device="cuda", generator=g)
        i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
        valid = (i + d + 1) < ol[b0:b0 + 4].view(-1, 1, 1)
        links[b0:b0 + 4] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
        del raw, valid
    alpha = torch.empty_like(match); beta = torch.empty_like(match)
    def run(a, b):
        assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(a), _lib.ptr(b), None, B, T, L, TR, None, 0, st) == 0
    res = {}
    for name, path, mt in (("log-space dense", 1, 0), ("mfma mt=1", 9, 1), ("mfma mt=2", 9, 2)):
        if path == 1 and L >= 4096: _lib.set_option("dp_path", 1)
        _lib.set_option("dp_path", path); _lib.set_option("dm_mt", mt)
        tb = timeit(lambda: run(alpha, beta), n=3 if L >= 4096 else 5)
        res[name] = (alpha.clone(), beta.clone()) if L < 4096 else None
        print(f"B={B} T={T} L={L} TR={TR} {name}: alpha||beta {tb:.3f} ms, status {_lib.last_launch_status()} exact-cells {_lib.last_fallback_count()}", flush=True)
    _lib.set_option("dp_path", 0); _lib.set_option("dm_mt", 0)
    if res["mfma mt=1"] is not None:
        for w in (0, 1):
            x, y = res["mfma mt=1"][w], res["log-space dense"][w]
            f = torch.isfinite(y)
            print(f"   mfma vs log-space {'alpha' if w == 0 else 'beta'}: inf pattern equal {bool(torch.equal(torch.isneginf(x), torch.isneginf(y)))}, max diff {float((x[f] - y[f]).abs().max()):.3e}", flush=True)
    del links, match, alpha, beta
    torch.cuda.empty_cache()
