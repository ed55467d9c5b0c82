#!/usr/bin/env python3
"""Cycle accounting of one compute wave of the alignment's max-DP (dag_maxstrip_kernel; GPU box, DSP_DEBUG=prof; needs a library built
with -DDSP_MX_PROF: add `// HIPCC_FLAGS: -DDSP_MX_PROF` to the header comment of csrc/dag_dp_maxstrip.hip and rebuild).  DSP_MX_ABLATE=4|8|16
(bits) times the kernel without its alpha store / max trees / adds.
usage: DSP_DEBUG=prof python tools/prof_maxstrip.py [B T L TR]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
B, T, L, TR = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (32, 512, 4096, 32)
d = torch.device("cuda"); g = torch.Generator(device=d).manual_seed(0)
ol = torch.full((B,), L, device=d); tl = torch.full((B,), T, device=d)
i = torch.arange(L, device=d).view(1, L, 1); dd = torch.arange(TR, device=d).view(1, 1, TR); valid = (i + dd + 1) < L
k = torch.log_softmax(torch.randn(B, L, TR, device=d, generator=g).masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
m = torch.randn(B, T, L, device=d, generator=g) - 9.0
for _ in range(3): ops.dag_best_alignment(m, k, ol, tl)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); ops.dag_best_alignment(m, k, ol, tl); b.record(); torch.cuda.synchronize()
print(f"alignment {a.elapsed_time(b):.3f} ms; status {_lib.last_launch_status()}")
w = _lib.load().dsp_dag_debug_words()
rows = max(1, int(w[11]))
ph = [int(w[7 + i]) * 16 for i in range(4)]
print("s_memtime ticks per row (100 MHz ticks x ?; relative shares matter): rows", rows)
for nm, v in zip(("barrier exit -> operands in registers", "adds + max tree", "stores issued, LDS write done", "barrier"), ph):
    print(f"  {nm:40s} {v / rows:8.1f} ticks/row  {100 * v / max(1, sum(ph)):5.1f} %")
