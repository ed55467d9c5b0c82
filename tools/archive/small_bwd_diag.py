"""dag_loss backward at small shapes: time + the exact-redo diagnostics of the exp-space kernel (lanes redone, unsafe factor, weak link)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from daspeech_amd import custom_ops as ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
for (B, T, L, TR) in ((32, 64, 130, 4), (32, 64, 130, 32), (32, 64, 400, 16), (32, 64, 1024, 32), (32, 64, 4096, 32), (32, 100, 400, 32), (8, 50, 260, 32)):
    for k5 in (0, 1):
        _lib.set_option("k5_path", k5)
        _, links, ol, tl, _ = bench.make_dag_inputs(torch, dev, B, L, T, 16, TR, 5)
        match = torch.log_softmax(torch.randn(B, T, L, device=dev) * 2, -1).contiguous().requires_grad_()
        k = links.requires_grad_()
        ts = []
        diag = (ctypes.c_uint * 4)()
        for it in range(6):
            loss = ops.dag_loss(match, k, ol, tl)
            fin = torch.isfinite(loss)
            torch.cuda.synchronize(); lib.dsp_dag_debug_k5(diag)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); g = torch.autograd.grad(loss[fin].sum(), [match, k]); b.record(); torch.cuda.synchronize()
            lib.dsp_dag_debug_k5(diag)
            ts.append(a.elapsed_time(b))
        print(f"B={B} T={T} L={L} TR={TR} k5_path={k5}: bwd {min(ts):.3f} ms (autograd incl.), finite {int(fin.sum())}/{B}, diag redo/unsafe/weak/family = {list(diag)}", flush=True)
_lib.set_option("k5_path", 0)
