#!/usr/bin/env python3
"""column-sweep DP (dp_path 10 / 11) vs the fp64 oracle and strip4g (dp_path 5); timings at C2.  usage: cs_check.py [quick|time|peaked]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daspeech_amd import custom_ops as ops, _lib
from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs

lib = _lib.load()


def run_path(path, m, k, o, tt):
    _lib.set_option("dp_path", path)
    try:
        loss, (a, b) = ops.dag_loss_with_alpha_beta(m, k, o, tt)
        st = _lib.last_launch_status()
        w = lib.dsp_dag_debug_words()
        info = (st, int(w[1]), int(w[5]))
        if os.environ.get("DSP_DEBUG") == "cs" and path == 10:
            recs = [(hex(int(w[14 + 4 * i])), int(w[15 + 4 * i]), int(w[16 + 4 * i]), hex(int(w[17 + 4 * i]))) for i in range(4)]
            print("   abort records (b|beta<<8|lossage<<16, row u, column v, S bits):", recs, "count", int(w[6]))
    finally:
        _lib.set_option("dp_path", 0)
    return loss, a, b, info


def check(B, T, L, TR, seed, masked=False, path=10):
    match, links, ol, tl = make_dag_inputs(seed, B, T, L, TR)
    if masked:
        rng = np.random.default_rng(seed)
        match[0, min(3, T - 1), :] = -np.inf; match[0, min(3, T - 1), min(L - 1, 10)] = 0.0
        match[rng.random(match.shape) < 0.1] = -np.inf
    t = lambda a: torch.from_numpy(a).cuda()
    m, k, o, tt = t(match), t(links), t(ol), t(tl)
    m.requires_grad_()
    loss, a, b, info = run_path(path, m, k, o, tt)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    ok = np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64)) and not np.isnan(a).any() and not np.isnan(b).any()
    fa, fb_ = np.isfinite(a64) & np.isfinite(a), np.isfinite(b64) & np.isfinite(b)
    ea = np.abs(a[fa] - a64[fa]).max() if fa.any() else 0; eb = np.abs(b[fb_] - b64[fb_]).max() if fb_.any() else 0
    scale = max(np.abs(a64[fa]).max() if fa.any() else 1, 1.0)
    good = ok and ea < 3e-6 * scale + 3e-5 * T + 1e-3 and eb < 3e-6 * scale + 3e-5 * T + 1e-3 and info[0] == 0
    print(f"B={B} T={T} L={L} TR={TR} masked={masked} path={path}: status {info[0]} standby-cells {info[1]} aborted-samples {info[2]} inf-pattern {'ok' if ok else 'MISMATCH'} "
          f"max|da| {ea:.2e} max|db| {eb:.2e} -> {'ok' if good else 'FAIL'}", flush=True)
    if not ok:
        for nm, x, y in (("alpha", a, a64), ("beta", b, b64)):
            bad = np.argwhere((np.isneginf(x) != np.isneginf(y)) | np.isnan(x))[:6]
            if len(bad): print(f"   {nm} mismatches (b,t,j):", bad.tolist(), [(float(x[tuple(i)]), float(y[tuple(i)])) for i in bad])
    elif not good:
        for nm, x, y, f in (("alpha", a, a64, fa), ("beta", b, b64, fb_)):
            d = np.where(f, np.abs(x - y), 0); i = np.unravel_index(np.argmax(d), d.shape)
            print(f"   {nm} worst at {i}: {x[i]} vs {y[i]}")
    return good


mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
SHAPES = [(2, 8, 64, 32), (3, 24, 200, 32), (2, 40, 256, 32), (4, 33, 132, 16), (2, 70, 400, 32), (1, 9, 1024, 8), (3, 100, 512, 32), (2, 130, 640, 24)]
if len(sys.argv) > 2: SHAPES = [tuple(int(x) for x in sys.argv[2].split(","))]
allok = True
if mode in ("quick", "all"):
    for path in ((10,) if len(sys.argv) > 2 else (11, 10)):
        for shape in SHAPES:
            allok &= check(*shape, seed=7 + shape[2], path=path)
        if len(sys.argv) <= 2:
            allok &= check(3, 30, 256, 32, 5, masked=True, path=path)
            allok &= check(2, 66, 332, 32, 6, masked=True, path=path)
    print("ALL OK" if allok else "FAILURES", flush=True)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if mode in ("time", "all", "prof"):
    st = _lib.current_stream_handle()
    for (B, T, L, TR) in ([(32, 512, 4096, 32)] if mode == "prof" else [(32, 512, 4096, 32), (32, 128, 1024, 32), (64, 256, 2048, 32)]):
        g = torch.Generator(device="cuda").manual_seed(0)
        match = torch.log_softmax(torch.randn(B, T, L, device="cuda", generator=g) * 2, -1) - 6
        ol = torch.full((B,), L, device="cuda") - (torch.arange(B, device="cuda") % 5) * 4; tl = torch.full((B,), T, device="cuda") - torch.arange(B, device="cuda") % 4
        raw = torch.randn(B, L, TR, device="cuda", generator=g)
        i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
        valid = (i + d + 1) < ol.view(-1, 1, 1)
        links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
        del raw, valid
        alpha = torch.empty_like(match); beta = torch.empty_like(match)
        wsb = lib.dsp_dag_workspace_bytes(B, T, L, TR)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        def run(a, b):
            assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(a), _lib.ptr(b), None, B, T, L, TR, _lib.ptr(ws), wsb, st) == 0
        res = {}
        for name, path in ((("colsweep alone", 11),) if mode == "prof" else (("strip4g", 5), ("colsweep+standby", 10), ("colsweep alone", 11))):
            _lib.set_option("dp_path", path)
            tb = timeit(lambda: run(alpha, beta))
            res[name] = (alpha.clone(), beta.clone())
            w = lib.dsp_dag_debug_words(); stt = _lib.last_launch_status()
            print(f"B={B} T={T} L={L} TR={TR} {name}: alpha||beta {tb:.3f} ms, status {stt} aborted {int(w[5])}", flush=True)
        _lib.set_option("dp_path", 0)
        for w_ in (() if mode == "prof" else (0, 1)):
            x, y = res["colsweep+standby"][w_], res["strip4g"][w_]
            f = torch.isfinite(y)
            print(f"   colsweep vs strip4g {'alpha' if w_ == 0 else 'beta'}: inf pattern equal {bool(torch.equal(torch.isneginf(x), torch.isneginf(y)))}, max diff {float((x[f] - y[f]).abs().max()):.3e}", flush=True)
        del links, match, alpha, beta, ws
        torch.cuda.empty_cache()
sys.exit(0 if allok else 1)
