#!/usr/bin/env python3
"""DAG ops on tensors that are VIEWS with odd storage offsets (4-byte aligned only) or non-contiguous layouts: same results as on
fresh contiguous copies.  (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from util_inputs import make_dag_inputs
from daspeech_amd import custom_ops as ops, _lib
bad = 0
def off(t, k):                     # same values in a buffer whose data pointer is offset by k elements
    buf = torch.empty(t.numel() + k, dtype=t.dtype, device=t.device)
    v = buf[k:].view(t.shape); v.copy_(t); return v
for case, (B, T, L, TR) in enumerate([(2, 10, 128, 32), (3, 12, 130, 7), (2, 20, 300, 299), (2, 9, 96, 95), (1, 30, 1028, 32), (2, 15, 257, 64), (2, 16, 512, 32)]):
    match, links, ol, tl = make_dag_inputs(5 + case, B, T, L, TR)
    m0 = torch.from_numpy(match).cuda(); k0 = torch.from_numpy(links).cuda(); o = torch.from_numpy(ol).cuda(); t = torch.from_numpy(tl).cuda()
    ref_loss, (ra, rb) = ops.dag_loss_with_alpha_beta(m0.clone().requires_grad_(), k0, o, t)
    ref_path = ops.dag_best_alignment(m0, k0, o, t)
    mm = m0.clone().requires_grad_(); kk = k0.clone().requires_grad_()
    gm0, gk0 = torch.autograd.grad(ops.dag_loss(mm, kk, o, t).nan_to_num(neginf=0).sum(), [mm, kk])
    for km, kk_ in ((1, 0), (0, 1), (3, 1), (2, 2)):
        variants = {"offset": (off(m0, km), off(k0, kk_)),
                    "transposed": (m0.transpose(1, 2).contiguous().transpose(1, 2), k0.transpose(1, 2).contiguous().transpose(1, 2))}
        for name, (mv, kv) in variants.items():
            tag = f"case {case} B={B} T={T} L={L} TR={TR} {name} offsets=({km},{kk_})"
            try:
                loss, (a, b) = ops.dag_loss_with_alpha_beta(mv.detach().requires_grad_(), kv, o, t)
                assert _lib.last_launch_status() == 0
                assert torch.equal(loss, ref_loss) or torch.allclose(loss, ref_loss, rtol=1e-6, atol=1e-5), "loss"
                assert torch.allclose(a.nan_to_num(neginf=-1e30), ra.nan_to_num(neginf=-1e30), rtol=1e-6, atol=1e-4), "alpha"
                assert torch.equal(ops.dag_best_alignment(mv, kv, o, t), ref_path), "path"
                m2 = mv.detach().requires_grad_(); k2 = kv.detach().requires_grad_()
                gm, gk = torch.autograd.grad(ops.dag_loss(m2, k2, o, t).nan_to_num(neginf=0).sum(), [m2, k2])
                assert torch.allclose(gm, gm0, rtol=1e-4, atol=1e-7) and torch.allclose(gk, gk0, rtol=1e-4, atol=1e-7), "grads"
            except Exception as e:   # noqa
                bad += 1; print("FAIL", tag, "->", repr(e)[:300])
print("failures:", bad)
