#!/usr/bin/env python3
"""Dense-window DP on 'trained-model-like' scores (emissions near 0 on a band around the alignment, a -20-nat floor elsewhere, 4-sigma
transition logits over the whole window): time, exact-redo counters, give-up flag, against the random-score time of the same shape.
usage: peaked_dense.py [B T L]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
B, T, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 256, 2048); TR = L - 1
d = torch.device("cuda"); g = torch.Generator(device=d).manual_seed(0)
ol = torch.full((B,), L, device=d); tl = torch.full((B,), T, device=d)
i = torch.arange(L, device=d).view(1, L, 1); dd = torch.arange(TR, device=d).view(1, 1, TR); valid = (i + dd + 1) < L
def links_of(sigma, jump=None):
    raw = sigma * torch.randn(B, L, TR, device=d, generator=g)
    if jump is not None: raw = raw - 0.5 * ((dd.float() + 1 - jump) / 2.0) ** 2          # a distance prior: mass on jumps of ~`jump` vertices
    return torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
j = torch.arange(L, device=d).view(1, 1, L).float(); c = (torch.arange(T, device=d).float() * (L - 1) / max(T - 1, 1)).view(1, T, 1)
cases = {"random scores": (torch.randn(B, T, L, device=d, generator=g) * 2 - 6, links_of(1.0)),
         "peaked emissions, 4-sigma links": (torch.where((j - c).abs() < 6, -0.5 + 0.3 * torch.randn(B, T, L, device=d, generator=g), -20.0 + 3.0 * torch.randn(B, T, L, device=d, generator=g)), links_of(4.0)),
         "peaked emissions, distance prior": (torch.where((j - c).abs() < 6, -0.5 + 0.3 * torch.randn(B, T, L, device=d, generator=g), -20.0 + 3.0 * torch.randn(B, T, L, device=d, generator=g)), links_of(1.0, jump=(L - 1) / max(T - 1, 1)))}
if os.environ.get("DM_BUDGET"): _lib.set_option("dm_budget", int(os.environ["DM_BUDGET"]))
for name, (m, k) in cases.items():
    mg = m.clone().requires_grad_(); kg = k.clone().requires_grad_()
    def fwd(): return ops.dag_loss(mg, kg, ol, tl)
    for _ in range(2): loss = fwd(); torch.autograd.grad(loss.sum(), [mg, kg], retain_graph=True)
    st = _lib.last_launch_status(); cells = _lib.last_fallback_count(); gave = _lib.last_dense_gave_up()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): loss = fwd()
    e1.record(); torch.cuda.synchronize(); tf = e0.elapsed_time(e1) / 5
    e0.record()
    for _ in range(5): torch.autograd.grad(loss.sum(), [mg, kg], retain_graph=True)
    e1.record(); torch.cuda.synchronize(); tb = e0.elapsed_time(e1) / 5
    with torch.no_grad():
        e0.record()
        for _ in range(5): ops.dag_best_alignment(m, k, ol, tl)
        e1.record(); torch.cuda.synchronize(); ta = e0.elapsed_time(e1) / 5
    if os.environ.get("CELLS") and cells: print("   first flagged cells (sample | 0x100 = beta, step, column, distrusted sum):", _lib.debug_fallback_cells()[:12], "band centre = step *", (L - 1) / max(T - 1, 1))
    print(f"{name}: forward {tf:.3f} ms (status {st}, exact cells {cells}, gave up {gave}, finite losses {int(torch.isfinite(loss).sum())}/{B}) | backward {tb:.3f} ms | alignment {ta:.3f} ms")
