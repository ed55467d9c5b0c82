import sys, os
sys.path.insert(0, "/root/repo")
import torch
from daspeech_amd import decode_ops, _lib
torch.manual_seed(0)
B, L, H, CK = 2, 1500, 8, 64
TR = L - 1
olen = torch.tensor([1500, 1203], device="cuda")
q = (torch.randn(B, L, H, CK, device="cuda") * 0.5).requires_grad_()
k = (torch.randn(B, L, H, CK, device="cuda") * 0.5).requires_grad_()
lg = torch.log_softmax(torch.randn(B, L, H, device="cuda"), -1).requires_grad_()
with torch.no_grad():
    l0 = decode_ops.extract_links(q, k, lg, olen, TR)
torch.cuda.synchronize(); print("inference ok", l0.shape, float(l0[torch.isfinite(l0)].mean()))
links = decode_ops.extract_links_autograd(q, k, lg, olen, TR)
torch.cuda.synchronize(); print("train fwd ok", float((links - l0)[torch.isfinite(l0)].abs().max()))
fin = torch.isfinite(links)
links.masked_fill(~fin, 0.0).sum().backward()
torch.cuda.synchronize(); print("bwd ok", float(q.grad.abs().max()), float(k.grad.abs().max()), float(lg.grad.abs().max()))
