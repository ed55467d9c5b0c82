#!/usr/bin/env python3
"""C1 (B=4, T=256, L=2048, TR=2047) fwd + bwd + alignment in a loop, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops
B, T, L, TR = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (4, 256, 2048, 2047)))
g = torch.Generator(device="cuda").manual_seed(0)
match = (torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6).requires_grad_()
ol = torch.full((B,), L, device="cuda"); tl = torch.full((B,), T, device="cuda")
i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
links = torch.empty(B, L, TR, device="cuda")
for b in range(B):
    raw = torch.randn(1, L, TR, device="cuda", generator=g)
    valid = (i + d + 1) < L
    links[b:b + 1] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
links.requires_grad_()
from daspeech_amd import _lib
if os.environ.get("DSP_SO"): _lib.SO_PATH = os.path.abspath(os.environ["DSP_SO"])
if os.environ.get("DM_BUDGET"): _lib.set_option("dm_budget", int(os.environ["DM_BUDGET"]))
def step():
    loss = ops.dag_loss(match, links, ol, tl)
    return torch.autograd.grad(loss.sum(), [match, links])
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print(f"fwd+bwd {e0.elapsed_time(e1) / 10:.3f} ms/step")
