"""dag_loss forward + backward and dag_best_alignment at dense-window training shapes (README --max-transition-length 99999).
usage: dense_step_bench.py B T L"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import inputs
B, T, L = [int(v) for v in sys.argv[1:4]]
m, k, ol, tl = inputs(B, T, L, L - 1)
m.requires_grad_(); k.requires_grad_()
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
fwd = t(lambda: ops.dag_loss(m.detach(), k.detach(), ol, tl))
def fb():
    m.grad = None; k.grad = None
    ops.dag_loss(m, k, ol, tl).sum().backward()
both = t(fb)
al = t(lambda: ops.dag_best_alignment(m.detach(), k.detach(), ol, tl))
print(f"B={B} T={T} L={L} TR={L-1}: dag_loss fwd {fwd:.2f} ms, fwd+bwd {both:.2f} ms, best_alignment {al:.2f} ms, status {_lib.last_launch_status()}")
