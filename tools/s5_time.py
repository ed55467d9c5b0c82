#!/usr/bin/env python3
"""timing of dp paths at a given shape: s5_time.py B T L TR [paths...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
B, T, L, TR = [int(v) for v in sys.argv[1:5]]
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
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, path, cpl, w in (("strip4g", 5, 0, 0), ("strip5 cpl2 w1024", 8, 2, 1024), ("strip5 cpl4 w1024", 8, 4, 1024), ("strip5 cpl2 w512", 8, 2, 512), ("strip5 cpl4 w512", 8, 4, 512)):
    _lib.set_option("dp_path", path); _lib.set_option("s5_cpl", cpl); _lib.set_option("s5_w", w)
    tb = timeit(lambda: run(alpha, beta)); ta = timeit(lambda: run(alpha, None)); tbb = timeit(lambda: run(None, beta))
    print(f"B={B} T={T} L={L} TR={TR} {name}: alpha||beta {tb:.3f} ms, alpha only {ta:.3f} ms, beta only {tbb:.3f} ms, status {_lib.last_launch_status()}", flush=True)
