#!/usr/bin/env python3
"""Two host threads, each on its own HIP stream, hammer the DAG ops at different shapes at the same time: every result must equal
the one computed alone.  (Per-stream library scratch, thread-local caller workspaces and options, ticket counters.)  GPU box only."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util_inputs import make_dag_inputs
from daspeech_amd import custom_ops as ops, _lib
shapes = [(4, 40, 1024, 32), (3, 30, 330, 329), (2, 24, 600, 599), (4, 20, 512, 16), (2, 50, 200, 64), (3, 33, 257, 256)]
def inputs(i):
    B, T, L, TR = shapes[i % len(shapes)]
    m, k, o, t = make_dag_inputs(100 + i, B, T, L, TR)
    return [torch.from_numpy(x).cuda() for x in (m, k, o, t)]
def run(m, k, o, t):
    mm = m.clone().requires_grad_(); kk = k.clone().requires_grad_()
    loss, (a, b) = ops.dag_loss_with_alpha_beta(mm, kk, o, t)
    gm, gk = torch.autograd.grad(loss.nan_to_num(neginf=0).sum(), [mm, kk])
    return loss.detach(), a, b, gm, gk, ops.dag_best_alignment(m, k, o, t)
data = [inputs(i) for i in range(len(shapes))]
ref = [run(*d) for d in data]
torch.cuda.synchronize()
errors = []
def worker(tid):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for it in range(40):
            i = (it * (tid + 1) + tid) % len(shapes)
            out = run(*data[i])
            s.synchronize()
            for x, y in zip(out, ref[i]):
                if not (torch.equal(x, y) or torch.allclose(x.float().nan_to_num(neginf=-1e30), y.float().nan_to_num(neginf=-1e30), rtol=1e-5, atol=1e-6)):
                    errors.append((tid, it, i)); break
ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
[t.start() for t in ths]; [t.join() for t in ths]
print("mismatches:", errors[:10], "total", len(errors))
