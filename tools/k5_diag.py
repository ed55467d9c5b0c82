#!/usr/bin/env python3
"""fwd+bwd once at the given shape, then the exp-space grad_links diagnostics.  usage: k5_diag.py B T L TR"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import inputs, timeit
B, T, L, TR = [int(v) for v in sys.argv[1:5]]
m, k, ol, tl = inputs(B, T, L, TR)
mg = m.clone().requires_grad_(); kg = k.clone().requires_grad_()
out = (ctypes.c_uint * 4)()
_lib.load().dsp_dag_debug_k5(out)
loss = ops.dag_loss(mg, kg, ol, tl)
gm, gk = torch.autograd.grad(loss.sum(), [mg, kg])
torch.cuda.synchronize()
_lib.load().dsp_dag_debug_k5(out)
print("lanes redone", out[0], "unsafe factor", out[1], "weak link", out[2], "edge-group rows", out[3], "of", B * L // 4 * 4, "lane-waves")
for path in (2, 1):
    _lib.set_option("k5_path", path)
    loss = ops.dag_loss(mg, kg, ol, tl)
    tt = timeit(lambda: torch.autograd.grad(loss.sum(), [kg], retain_graph=True))
    print("k5_path", path, "grad_links-only backward", f"{tt[0]:.3f} ms")
_lib.set_option("k5_path", 0)
