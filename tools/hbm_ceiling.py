#!/usr/bin/env python3
"""Practical HBM ceilings on this box with torch's own streaming kernels (fp32, 4.3 GB tensors)."""
import torch, time
d = torch.device("cuda")
n = 32 * 4096 * 8192
x = torch.empty(n, device=d); y = torch.empty(n, device=d)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    es = []
    for _ in range(it):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); es.append(a.elapsed_time(b))
    return min(es)
gb = n * 4 / 1e9
ms = t(lambda: x.add_(1.0)); print(f"in-place add_: {ms:.3f} ms  {2 * gb / ms:.2f} TB/s (read+write)")
ms = t(lambda: y.copy_(x)); print(f"copy_:         {ms:.3f} ms  {2 * gb / ms:.2f} TB/s (read+write)")
ms = t(lambda: x.sum()); print(f"sum (read):    {ms:.3f} ms  {gb / ms:.2f} TB/s")
ms = t(lambda: x.fill_(0.5)); print(f"fill (write):  {ms:.3f} ms  {gb / ms:.2f} TB/s")
xr = x.view(32 * 4096, 8192)
ms = t(lambda: torch.log_softmax(xr, -1, out=None)); print(f"torch log_softmax (out of place): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
