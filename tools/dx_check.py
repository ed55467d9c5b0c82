#!/usr/bin/env python3
"""dense max-DP alignment (auto / dp_path 9) vs the f32 oracle (bit-exact paths, ties included) and vs the log-space-era kernels (dp_path 1); timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from daspeech_amd import custom_ops as ops, _lib
from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs

allok = True
for (B, T, L, TR, quant) in [(3, 24, 200, 199, False), (2, 40, 256, 255, True), (4, 33, 130, 129, True), (2, 20, 500, 100, False), (2, 70, 400, 399, True), (1, 9, 1024, 1023, False), (3, 18, 192, 191, True)]:
    match, links, ol, tl = make_dag_inputs(9 + L, B, T, L, TR)
    if quant:       # quantised scores: ties on most rows
        match = np.round(match * 2) / 2; links = np.where(np.isfinite(links), np.round(links * 2) / 2, links).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=a.dtype)).cuda()
    m, k, o, tt = t(match.astype(np.float32)), t(links), t(ol), t(tl)
    ref = orc.dag_best_alignment(match.astype(np.float32), links, ol, tl, np.float32)
    res = {}
    for path in (0, 1):
        _lib.set_option("dp_path", path)
        res[path] = ops.dag_best_alignment(m, k, o, tt).cpu().numpy()
        st = _lib.last_launch_status()
    _lib.set_option("dp_path", 0)
    ok = np.array_equal(res[0], ref) and np.array_equal(res[1], ref)
    allok &= ok
    print(f"B={B} T={T} L={L} TR={TR} quant={quant}: new==oracle {np.array_equal(res[0], ref)} old==oracle {np.array_equal(res[1], ref)} status {st}", flush=True)
    if not np.array_equal(res[0], ref):
        bad = np.argwhere(res[0] != ref)[:8]; print("   first mismatches (b, j):", bad.tolist(), [(int(res[0][tuple(i)]), int(ref[tuple(i)])) for i in bad])
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
    match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6
    ol = torch.full((B,), L, device="cuda") - torch.arange(B, device="cuda") % 5; tl = torch.full((B,), T, device="cuda") - torch.arange(B, device="cuda") % 4
    links = torch.empty(B, L, TR, device="cuda")
    for b0 in range(0, B, 2):
        raw = torch.randn(min(2, B - b0), L, TR, device="cuda", generator=g)
        i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
        valid = (i + d + 1) < ol[b0:b0 + 2].view(-1, 1, 1)
        links[b0:b0 + 2] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
        del raw, valid
    out = {}
    for name, path in (("dense max-plus (new)", 0), ("row-sequential (old)", 1)):
        _lib.set_option("dp_path", path)
        ms = timeit(lambda: ops.dag_best_alignment(match, links, ol, tl), n=3)
        out[name] = ops.dag_best_alignment(match, links, ol, tl)
        print(f"B={B} T={T} L={L} TR={TR} {name}: {ms:.3f} ms status {_lib.last_launch_status()}", flush=True)
    _lib.set_option("dp_path", 0)
    print("   same paths:", bool(torch.equal(out["dense max-plus (new)"], out["row-sequential (old)"])), flush=True)
    del links, match; torch.cuda.empty_cache()
