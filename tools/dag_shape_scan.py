"""Timing scan of the DAG operators over window sizes and graph lengths: looks for shapes that fall off the fast paths (r05: TR 33..64 ran the
row-sequential generic kernels — 112 ms where its neighbours take 0.5 and 18 ms)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from daspeech_amd import custom_ops as ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
print(f"B={B} T={T}: ms fwd(alpha||beta) / bwd / alignment")
for L in (130, 400, 1022, 1024, 2050, 4096):
    row = []
    for TR in (4, 16, 31, 32, 33, 64, 65, 128, L - 1):
        if TR > L - 1: row.append("   -   "); continue
        _, links, ol, tl, _ = bench.make_dag_inputs(torch, dev, B, L, T, 16, TR, 5)
        match = torch.log_softmax(torch.randn(B, T, L, device=dev) * 2, -1).contiguous().requires_grad_()
        k = links.requires_grad_()
        def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
        for it in range(3):
            e0 = ev(); loss = ops.dag_loss(match, k, ol, tl); e1 = ev()
            g = torch.autograd.grad(loss.sum(), [match, k]); e2 = ev()
            with torch.no_grad(): p = ops.dag_best_alignment(match.detach(), k.detach(), ol, tl)
            e3 = ev()
        torch.cuda.synchronize()
        row.append(f"{e0.elapsed_time(e1):.2f}/{e1.elapsed_time(e2):.2f}/{e2.elapsed_time(e3):.2f}")
    print(f"L={L:5d} " + "  ".join(r.rjust(16) for r in row), flush=True)
print("columns: TR = 4, 16, 31, 32, 33, 64, 65, 128, L-1")
