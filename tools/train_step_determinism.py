#!/usr/bin/env python3
"""Run-to-run spread of the training step's gradients (GPU box only): the step of tests/test_gpu_distributed_step.py N times on the same
batch with the same seed; reports, per parameter, the largest deviation from the first run relative to the gradient's own maximum."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_distributed_step import _model_and_halves, _step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
m, halves = _model_and_halves(dev)
worst = {}
first = None
for it in range(n):
    loss, _ = _step(m, halves[it % 2 if len(sys.argv) > 2 else 0], 100)
    g = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
    if first is None:
        first, l0 = g, loss
        continue
    if len(sys.argv) > 2:
        continue
    for k in g:
        d = float((g[k] - first[k]).abs().max()) / max(1e-30, float(first[k].abs().max()))
        if d > worst.get(k, (0, 0))[0]:
            worst[k] = (d, it)
    if abs(loss - l0) > 1e-6 * abs(l0):
        print(f"iter {it}: loss {loss} vs {l0}")
top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:12]
for k, (d, it) in top:
    print(f"{d:10.3e}  (iter {it})  {k}")
print("max relative deviation over all parameters:", max([v[0] for v in worst.values()] + [0.0]))
