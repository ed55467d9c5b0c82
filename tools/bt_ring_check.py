#!/usr/bin/env python3
"""Back-trace kernels of dag_best_alignment (TR = 32): ring kernel (r04) vs the r01-r03 one — same paths, time.  usage: bt_ring_check.py [B T L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import inputs, timeit
B, T, L = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (32, 512, 4096)
for seed in range(3):
    m, k, ol, tl = inputs(B, T, L, 32, seed=seed)
    res = {}
    for ring in (0, 1):
        _lib.set_option("bt_ring", ring)
        with torch.no_grad():
            tm = timeit(lambda: ops.dag_best_alignment(m, k, ol, tl), n=7)
            res[ring] = (ops.dag_best_alignment(m, k, ol, tl), tm)
    print(f"seed {seed} B={B} T={T} L={L}: old {res[0][1][0]:.4f} ms  ring {res[1][1][0]:.4f} ms  same_path={bool(torch.equal(res[0][0], res[1][0]))} status={_lib.last_launch_status()}")
_lib.set_option("bt_ring", 1)
