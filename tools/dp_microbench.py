#!/usr/bin/env python3
"""Micro-benchmark of the DP kernel families (GPU box only).  usage: dp_microbench.py B T L TR [paths]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib


def inputs(B, T, L, TR, seed=0):
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(seed)
    cg = torch.Generator().manual_seed(seed)
    out_len = (L - torch.randint(0, 5, (B,), generator=cg)).to(d)
    tgt_len = (T - torch.randint(0, 5, (B,), generator=cg)).to(d)
    raw = torch.randn(B, L, TR, device=d, generator=g)
    i = torch.arange(L, device=d).view(1, L, 1); dd = torch.arange(TR, device=d).view(1, 1, TR)
    valid = (i + dd + 1) < out_len.view(B, 1, 1)
    dead = ~valid.any(-1, keepdim=True)
    links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(dead, 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
    match = torch.randn(B, T, L, device=d, generator=g) - 9.0
    return match, links, out_len, tgt_len


def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    es = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); es.append(a.elapsed_time(b))
    return min(es), sum(es) / len(es)


def dbg_cells():
    import struct
    w = _lib.load().dsp_dag_debug_words()
    n = min(14, max(int(w[1]), int(w[2])))
    return [(int(w[7 + 4 * i]), int(w[8 + 4 * i]), int(w[9 + 4 * i]), struct.unpack('f', struct.pack('I', w[10 + 4 * i]))[0]) for i in range(n)]


def main():
    B, T, L, TR = [int(v) for v in sys.argv[1:5]]
    paths = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [3]
    m, k, ol, tl = inputs(B, T, L, TR)
    mg = m.clone().requires_grad_()
    for path in paths:
        _lib.set_option("dp_path", path)
        f = timeit(lambda: ops.dag_loss(mg, k, ol, tl))
        with torch.no_grad():
            a = timeit(lambda: ops.dag_loss(m, k, ol, tl))
            v = timeit(lambda: ops.dag_best_alignment(m, k, ol, tl))
        ops.dag_loss(mg, k, ol, tl)
        st = (_lib.last_launch_status(), _lib.last_fallback_count(), int(_lib.load().dsp_dag_debug_words()[2]), dbg_cells(), ol[:4].tolist(), tl[:4].tolist())
        gb = 2 * (B * T * L * 4 * 2 + B * L * TR * 4) / 1e9
        print(f"path {path} B={B} T={T} L={L} TR={TR}: fwd(a+b) min {f[0]:.3f} ms ({gb / f[0] * 1e3:.0f} GB/s) | alpha-only {a[0]:.3f} ms | align {v[0]:.3f} ms | status {st}")
    _lib.set_option("dp_path", 0)


if __name__ == "__main__":
    main()
