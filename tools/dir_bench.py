#!/usr/bin/env python3
"""alpha-only / beta-only / both timings of the DP launch through the C ABI.  usage: dir_bench.py B T L TR paths"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
from tools.dp_microbench import inputs, timeit

B, T, L, TR = [int(v) for v in sys.argv[1:5]]
paths = [int(v) for v in sys.argv[5].split(",")]
m, k, ol, tl = inputs(B, T, L, TR)
lib = _lib.load()
alpha = torch.empty_like(m); beta = torch.empty_like(m)
st = _lib.current_stream_handle()
def run(a, b):
    rc = lib.dsp_dag_loss_fwd(_lib.ptr(m), _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(a), _lib.ptr(b), None, B, T, L, TR, None, 0, st)
    assert rc == 0
for path in paths:
    _lib.set_option("dp_path", path)
    ta = timeit(lambda: run(alpha, None)); tb = timeit(lambda: run(None, beta)); tab = timeit(lambda: run(alpha, beta))
    print(f"path {path}: alpha-only {ta[0]:.3f} ms | beta-only {tb[0]:.3f} ms | both {tab[0]:.3f} ms | status {_lib.last_launch_status()}")
_lib.set_option("dp_path", 0)
