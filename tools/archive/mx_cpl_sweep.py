#!/usr/bin/env python3
"""Max-DP (K6) with 1 / 2 / 4 vertices per lane at a given batch: time of dag_best_alignment by events, paths compared across variants.
usage: mx_cpl_sweep.py B [T L TR]   (GPU box; run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import inputs, timeit

B = int(sys.argv[1]); T, L, TR = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else (512, 4096, 32)
m, k, ol, tl = inputs(B, T, L, TR)
ref = None
for cpl in (2, 4, 1, 0):
    _lib.set_option("mx_cpl", cpl)
    with torch.no_grad():
        t = timeit(lambda: ops.dag_best_alignment(m, k, ol, tl), n=7)
        p = ops.dag_best_alignment(m, k, ol, tl)
    ref = p if ref is None else ref
    print(f"B={B} cpl={cpl}: align min {t[0]:.4f} avg {t[1]:.4f} ms  same_path={bool(torch.equal(p, ref))} status={_lib.last_launch_status()}")
_lib.set_option("mx_cpl", 0)
