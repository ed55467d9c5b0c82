import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from daspeech_amd import custom_ops as ops, _lib
dev = torch.device("cuda:0")
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
print("B T L TR dtype: fwd / bwd / align ms | per-cell ns (fwd)")
for (B, T, L, TR, dt) in [(32, 512, 4096, 32, torch.float32), (32, 512, 4096, 32, torch.float16), (32, 512, 4096, 32, torch.bfloat16), (1, 512, 4096, 32, torch.float32),
                          (2, 512, 8192, 32, torch.float32), (2, 512, 16384, 32, torch.float32), (8, 1024, 8192, 32, torch.float32), (8, 2000, 4096, 32, torch.float32),
                          (64, 100, 400, 32, torch.float32), (256, 64, 256, 32, torch.float32), (8, 200, 8192, 8191, torch.float32), (1, 50, 20000, 32, torch.float32),
                          (32, 512, 4098, 32, torch.float32), (32, 512, 4097, 32, torch.float32), (32, 60, 398, 32, torch.float32), (32, 60, 400, 32, torch.float32)]:
    try:
        _, links, ol, tl, _ = bench.make_dag_inputs(torch, dev, B, L, T, 16, TR, 5)
        match = torch.log_softmax(torch.randn(B, T, L, device=dev) * 2, -1).to(dt).contiguous().requires_grad_()
        k = links.to(dt).requires_grad_()
        for it in range(3):
            e0 = ev(); loss = ops.dag_loss(match, k, ol, tl); e1 = ev()
            g = torch.autograd.grad(loss.float().sum(), [match, k]); e2 = ev()
            with torch.no_grad(): p = ops.dag_best_alignment(match.detach(), k.detach(), ol, tl)
            e3 = ev()
        torch.cuda.synchronize()
        f = e0.elapsed_time(e1)
        print(f"{B} {T} {L} {TR} {str(dt)[6:]}: {f:.2f} / {e1.elapsed_time(e2):.2f} / {e2.elapsed_time(e3):.2f} | {f * 1e6 / (2.0 * B * T * L):.3f}  status {_lib.last_launch_status()} finite {int(torch.isfinite(loss).sum())}/{B}", flush=True)
    except Exception as e:
        print(f"{B} {T} {L} {TR} {str(dt)[6:]}: ERROR {repr(e)[:200]}", flush=True)
    torch.cuda.empty_cache()
