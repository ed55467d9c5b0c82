"""Bandwidth scan of dag_logsoftmax_gather_inplace (forward with the in-place soft-max, backward) over vocabulary sizes, dtypes and gather widths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops
dev = torch.device("cuda:0")
print("B L V S dtype: fwd ms (TB/s)  bwd ms (TB/s)")
for (B, L, V, S) in [(32, 4096, 8192, 512), (32, 4096, 512, 512), (32, 4096, 1000, 512), (32, 4096, 6004, 512), (32, 4096, 10001, 512), (32, 1024, 8192, 128), (64, 400, 6004, 60), (4, 2048, 512, 256)]:
    for dt in (torch.float32, torch.float16):
        x = torch.randn(B, L, V, device=dev, dtype=dt)
        idx = torch.randint(0, V, (B, 1, S), device=dev).expand(-1, L, -1)
        es = x.element_size()
        res = []
        for it in range(3):
            xx = x.clone().requires_grad_()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
            y = xx * 1.0
            e0.record(); _, m = ops.dag_logsoftmax_gather_inplace(y, idx); e1.record()
            g = torch.ones_like(m)
            e1.record(); m.backward(g); e2.record()
            torch.cuda.synchronize()
            res = (e0.elapsed_time(e1), e1.elapsed_time(e2))
        bytes_f = 2.0 * B * L * V * es + B * L * S * 4.0
        print(f"{B} {L} {V} {S} {str(dt)[6:]}: fwd {res[0]:.3f} ({bytes_f / res[0] / 1e9:.2f})  bwd(+autograd mul) {res[1]:.3f}", flush=True)
        del x, xx, y, m
