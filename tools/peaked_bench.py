#!/usr/bin/env python3
"""DP forward on 'trained-model-like' scores: emissions near 0 on a band around the alignment, a bounded floor elsewhere,
peaked transitions.  Reports time and exactness-guard counters for the exp-space kernel.  usage: peaked_bench.py [floor_nats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import timeit

floor = float(sys.argv[1]) if len(sys.argv) > 1 else -20.0
B, T, L, TR = 32, 512, 4096, 32
d = torch.device("cuda"); g = torch.Generator(device=d).manual_seed(0)
ol = torch.full((B,), L, device=d); tl = torch.full((B,), T, device=d)
j = torch.arange(L, device=d).view(1, 1, L).float(); c = (torch.arange(T, device=d).float() * (L - 1) / (T - 1)).view(1, T, 1)
match = torch.where((j - c).abs() < 6, -0.5 + 0.3 * torch.randn(B, T, L, device=d, generator=g), floor + 3.0 * torch.randn(B, T, L, device=d, generator=g))
raw = 4.0 * torch.randn(B, L, TR, device=d, generator=g)                 # peaked transition distributions
i = torch.arange(L, device=d).view(1, L, 1); dd = torch.arange(TR, device=d).view(1, 1, TR)
valid = (i + dd + 1) < L
links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
mg = match.clone().requires_grad_()
for path in (5, 3):
    _lib.set_option("dp_path", path)
    t = timeit(lambda: ops.dag_loss(mg, links, ol, tl))
    loss = ops.dag_loss(mg, links, ol, tl)
    st = _lib.last_launch_status(); w = _lib.load().dsp_dag_debug_words()
    if os.environ.get("DSP_DEBUG") == "medium" and path == 5:
        import struct
        n = min(14, int(w[2]))
        print("   first medium cells (sample|0x100=beta, t, vertex, S):", [(int(w[7 + 4 * i]), int(w[8 + 4 * i]), int(w[9 + 4 * i]), struct.unpack('f', struct.pack('I', w[10 + 4 * i]))[0]) for i in range(n)])
    print(f"path {path}: alpha||beta {t[0]:.3f} ms, status {st}, exact-path cells {w[1]}, medium lane-rows {w[2]}, loss[0] {loss[0].item():.3f}")
_lib.set_option("dp_path", 0)

# direction-separated timings through the C ABI
lib = _lib.load(); st = _lib.current_stream_handle()
alpha = torch.empty_like(match); beta = torch.empty_like(match)
def run(a, b):
    rc = lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(a), _lib.ptr(b), None, B, T, L, TR, None, 0, st)
    assert rc == 0
_lib.set_option("dp_path", 5)
print("path 5 alpha-only %.3f ms | beta-only %.3f ms | both %.3f ms" % (timeit(lambda: run(alpha, None))[0], timeit(lambda: run(None, beta))[0], timeit(lambda: run(alpha, beta))[0]))
_lib.set_option("dp_path", 0)

# backward and alignment on the same data
import ctypes
kg = links.clone().requires_grad_()
loss = ops.dag_loss(mg, kg, ol, tl)
diag = (ctypes.c_uint * 4)(); lib.dsp_dag_debug_k5(diag)
tb = timeit(lambda: torch.autograd.grad(loss.sum(), [mg, kg], retain_graph=True))
lib.dsp_dag_debug_k5(diag)
with torch.no_grad():
    ta = timeit(lambda: ops.dag_best_alignment(match, links, ol, tl))
print("dag_loss backward %.3f ms (exp-space K5 lanes redone exactly: %d) | best_alignment %.3f ms" % (tb[0], diag[0], ta[0]))
