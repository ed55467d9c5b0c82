#!/usr/bin/env python3
"""Per-wave cycle accounting of the strip4g DP kernel (GPU box only; run with DSP_DEBUG=prof).
usage: DSP_DEBUG=prof python tools/prof_strip.py [B T L TR]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import inputs

B, T, L, TR = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (32, 512, 4096, 32)
m, k, ol, tl = inputs(B, T, L, TR)
mg = m.clone().requires_grad_()
_lib.set_option("dp_path", 5)
for _ in range(3):
    ops.dag_loss(mg, k, ol, tl)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); ops.dag_loss(mg, k, ol, tl); b.record(); torch.cuda.synchronize()
print(f"launch+pick {a.elapsed_time(b):.3f} ms; status {_lib.last_launch_status()}")
w = _lib.load().dsp_dag_debug_words()
names = ["compute0", "compute1", "compute2", "compute3", "loader", "fetch", "publish"]
for slot, label in ((0, "ticket 0 (alpha strip 0)"), (1, "ticket 2*per (alpha strip 2)")):
    print(label)
    for wv, nm in enumerate(names):
        work, wait, rd = (int(w[7 + slot * 21 + wv * 3 + i]) for i in range(3))
        tot = work + wait
        print(f"  {nm:9s} total {tot:9d} cyc  ({tot / max(1, int(tl[0])):7.1f}/row)  own work {work / max(1, tot) * 100:5.1f}%  barrier wait {wait / max(1, tot) * 100:5.1f}%  of work: LDS-read wait {rd / max(1, work) * 100:5.1f}%")
print("compute0 FMA-phase cycles/row: strip0 %.1f strip2 %.1f" % (int(w[55]) / int(tl[0]), int(w[56]) / int(tl[0])))
rt = [int(w[49 + i]) for i in range(6)]
base = rt[0]
print("realtime (100 MHz ticks -> us): strip0 start/row64/end", [(x - base) / 100 for x in rt[:3]], " strip2 start/row64/end", [(x - base) / 100 for x in rt[3:]])
_lib.set_option("dp_path", 0)
