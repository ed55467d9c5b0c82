#!/usr/bin/env python3
"""Run-to-run determinism of the DAG training ops under GPU contention (GPU box only): the same gather -> dag_loss -> backward -> gather
backward on the same inputs N times on one stream while a second host thread keeps the chip busy with GEMMs on another stream; every
output must be BIT-identical to the first iteration's.  usage: determinism_stress.py [iters]"""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops
from tools.dp_microbench import inputs

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
stop = False


def noise():
    s = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device=dev); b = torch.randn(2048, 2048, device=dev)
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(20):
                a = (a @ b).tanh()
            s.synchronize()


th = threading.Thread(target=noise); th.start()
bad = 0
try:
    for (B, T, L, TR, V) in [(6, 23, 518, 32, 300), (5, 40, 1023, 32, 200), (8, 64, 1024, 32, 200), (4, 17, 261, 9, 100), (32, 120, 800, 32, 64)]:
        _, k0, ol, tl = inputs(B, T, L, TR, seed=L)
        g = torch.Generator(device=dev).manual_seed(L)
        logits0 = torch.randn(B, L, V, device=dev, generator=g)
        tgt = torch.randint(0, V, (B, T), device=dev, generator=g)
        first = None
        for it in range(iters):
            x = logits0.clone().requires_grad_()
            k = k0.clone().requires_grad_()
            _, sel = ops.dag_logsoftmax_gather_inplace(x.clone(), tgt.unsqueeze(1).expand(-1, L, -1))
            loss = ops.dag_loss(sel.transpose(1, 2), k, ol, tl)
            fin = torch.isfinite(loss)
            gx, gk = torch.autograd.grad(loss[fin].sum(), [x, k])
            path = ops.dag_best_alignment(sel.transpose(1, 2).detach(), k.detach(), ol, tl)
            out = (loss.detach().clone(), gx.clone(), gk.clone(), path.clone())
            if first is None:
                first = out
            else:
                for name, a, b in zip(("loss", "grad_logits", "grad_links", "path"), out, first):
                    if not torch.equal(a, b):
                        bad += 1
                        d = (a.float() - b.float()).abs().max().item()
                        print(f"MISMATCH shape {(B, T, L, TR)} iter {it} {name}: max diff {d:.3g}", flush=True)
        print(f"shape {(B, T, L, TR)}: {iters} iterations done, mismatches so far {bad}", flush=True)
finally:
    stop = True
    th.join()
print("DETERMINISM", "OK" if bad == 0 else f"FAILED ({bad})")
