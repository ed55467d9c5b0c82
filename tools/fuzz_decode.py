#!/usr/bin/env python3
"""Randomised sweep of the viterbi / jointviterbi graph decode (HIP max-DP + back-trace) against the torch restatement of the
reference loop (s2s_conformer_dag_fastspeech2.py:244-304): identical tokens, lengths, masks, gathered features.
usage: fuzz_decode.py [n_cases] [seed]   (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import decode_ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    B = int(torch.randint(1, 6, (1,), generator=g)); L = int(torch.randint(3, 1400, (1,), generator=g))
    TR = int(torch.randint(1, L, (1,), generator=g)) if torch.rand(1, generator=g) < 0.5 else min(L - 1, int(torch.randint(1, 70, (1,), generator=g)))
    if torch.rand(1, generator=g) < 0.3: TR = L - 1
    V, D, pad = 13, 8, 1
    joint = bool(torch.rand(1, generator=g) < 0.5)
    logits = torch.randn(B, L, V, generator=g) * 2
    logits[:, ::4, pad] += 6
    raw = torch.randn(B, L, TR, generator=g) * float(torch.tensor([1.0, 4.0, 12.0])[int(torch.randint(0, 3, (1,), generator=g))])
    if torch.rand(1, generator=g) < 0.5: raw = torch.round(raw * 4) / 4                                  # ties
    out_len = torch.randint(max(3, L - 9), L + 1, (B,), generator=g).clamp(max=L); out_len[0] = L
    i = torch.arange(L).view(1, L, 1); d = torch.arange(TR).view(1, 1, TR)
    valid = (i + d + 1) < out_len.view(B, 1, 1)
    links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
    feats = torch.randn(B, L, D, generator=g)
    beta, vb = (1.0, 1.0) if torch.rand(1, generator=g) < 0.5 else (0.5, 1.3)
    tag = f"case {case}: B={B} L={L} TR={TR} joint={joint} beta={beta}"
    try:
        got = decode_ops.viterbi_decode(logits.cuda(), links.cuda(), feats.cuda(), out_len.cuda(), pad, beta, vb, joint, 0.5)
        ref = decode_ops.viterbi_decode_torch(logits.cuda(), links.cuda(), feats.cuda(), out_len.cuda(), pad, beta, vb, joint, 0.5)
        for k_, (a, b) in enumerate(zip(got[:4], ref[:4])):
            if not torch.equal(a, b):
                w = (a != b).nonzero()[:3].tolist() if a.shape == b.shape else "shape"
                raise AssertionError(f"output {k_} differs at {w}: shapes {tuple(a.shape)} / {tuple(b.shape)}; quantised={bool((raw * 4 == torch.round(raw * 4)).all())}; out_len={out_len.tolist()}")
        if L <= 260:                                  # lookahead / greedy against the numpy oracle (oracle/graph_oracle.py)
            import numpy as np
            from oracle import graph_oracle as gorc
            prev = np.full((B, L), 3, np.int64); prev[np.arange(L)[None] >= out_len.numpy()[:, None]] = pad
            for strat in ("lookahead", "greedy"):
                want = gorc.forward_decoder(logits.numpy(), links.numpy(), feats.numpy(), prev, strat, pad, beta, vb)
                got2 = decode_ops.graph_decode(logits.cuda(), links.cuda(), feats.cuda(), out_len.cuda(), pad, beta, strat)
                for k_, (a, b) in enumerate(zip(got2, want)):
                    assert np.array_equal(a.cpu().numpy(), b), f"{strat}: output {k_} differs"
    except Exception as e:   # noqa
        bad += 1; print("FAIL", tag, "->", str(e).splitlines()[0][:200] if str(e) else repr(e))
print(f"{n} cases, {bad} failures")
