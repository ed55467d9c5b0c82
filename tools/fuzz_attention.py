#!/usr/bin/env python3
"""Randomised sweep of the r04 matrix-core kernels against fp64 references: attention (head widths 64 / 128, self / cross lengths, key
padding masks with interior padding, row-strided q / k / v slices, ragged query lengths), relative-position attention, the one-launch
Conformer feed-forward module, ragged (tile-skipping) GEMMs / convolutions.   usage: fuzz_attention.py [n_cases] [seed]   (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from daspeech_amd import decode_ops
from daspeech_amd.decode_ops import SplitConv1d
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g))
dev = torch.device("cuda")
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
bad = 0


def rand_mask(B, M):
    """key padding: suffix padding, sometimes with a padded key in the middle; sample 0 keeps every key"""
    lens = torch.randint(1, M + 1, (B,), generator=g); lens[0] = M
    pad = torch.arange(M)[None, :] >= lens[:, None]
    if M > 3 and ri(0, 3) == 0:
        pad[ri(0, B), ri(0, max(1, int(lens.min()) - 1)) + 0] = True
        pad[:, 0] = False                                           # never a sample without keys
    return pad.to(dev), lens


for case in range(n):
    tag = "?"
    try:
        # ---- attention
        B, H, dk = ri(1, 5), ri(1, 9), [64, 128][ri(0, 2)]
        if dk == 128: H = min(H, 4)
        N, M = ri(1, 420), ri(1, 420)
        C = H * dk
        fused = ri(0, 2) == 0 and N == M
        tag = f"case {case}: attention B={B} N={N} M={M} H={H} dk={dk} fused={fused}"
        if fused:
            qkv = rn(B, N, 3 * C) * 1.3
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        else:
            q, k, v = rn(B, N, C) * 1.3, rn(B, M, C) * 1.3, rn(B, M, C) * 2 + 0.1
        pad, _ = rand_mask(B, M) if ri(0, 4) else (None, None)
        qlens = slack = None
        if ri(0, 2):
            qlens = torch.randint(1, N + 1, (B,), generator=g).to(torch.int32).to(dev); slack = [0, 8, 40][ri(0, 3)]
        with torch.no_grad():
            got = decode_ops.attention(q, k, v, pad, H, q_lens=qlens, q_slack=slack or 0)
            qd, kd, vd = (t.double().reshape(B, -1, H, dk).transpose(1, 2) for t in (q, k, v))
            s = qd @ kd.transpose(-1, -2) * dk ** -0.5
            if pad is not None: s = s.masked_fill(pad.view(B, 1, 1, M), float("-inf"))
            ref = (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(B, N, C)
        assert got is not None and got.shape == ref.shape
        scale = float(ref.abs().max()) + 1e-30
        for b in range(B):
            lim = N if qlens is None else min(N, int(qlens[b]) + slack)
            err = float((got[b, :lim].double() - ref[b, :lim]).abs().max()) / scale
            assert err < 3e-6, f"attention err {err:.3e} (sample {b})"
            assert torch.isfinite(got[b]).all(), "non-finite padding rows"
            assert (got[b, (lim + 31) // 32 * 32:] == 0).all(), "skipped query groups must be zero"
        # ---- relative-position attention
        B, H, T = ri(1, 5), ri(1, 6), ri(1, 330)
        C = H * 64
        tag = f"case {case}: relpos B={B} T={T} H={H}"
        q, k, v = rn(B, T, C) * 1.2, rn(B, T, C) * 1.2, rn(B, T, C) * 1.5
        pos, bu, bv = rn(1, 2 * T - 1, C), rn(H, 64) * 0.5, rn(H, 64) * 0.5
        pad, _ = rand_mask(B, T) if ri(0, 4) else (None, None)
        with torch.no_grad():
            got = decode_ops.relpos_attention(q, k, v, pos, bu, bv, pad, H)
            qd, kd, vd = (t.double().view(B, T, H, 64) for t in (q, k, v))
            ac = torch.einsum("bihd,bjhd->bhij", qd + bu.double(), kd)
            bdf = torch.einsum("bihd,rhd->bhir", qd + bv.double(), pos.double().view(2 * T - 1, H, 64))
            idx = (T - 1) - torch.arange(T, device=dev)[:, None] + torch.arange(T, device=dev)[None, :]
            s = (ac + torch.gather(bdf, 3, idx.expand(B, H, T, T))) / 8.0
            if pad is not None: s = s.masked_fill(pad.view(B, 1, 1, T), float("-inf"))
            ref = torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1), vd).reshape(B, T, C)
        err = float((got.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        assert got is not None and err < 3e-6, f"relpos err {err:.3e}"
        # ---- one-launch feed-forward module
        B, T, Hd, act = ri(1, 40), ri(4, 300), [512, 1024, 2048][ri(0, 3)], ["relu", "silu", "gelu"][ri(0, 3)]
        if B * T < 128: T = 128 // B + 1
        tag = f"case {case}: ffn B={B} T={T} H={Hd} {act}"
        ln = torch.nn.LayerNorm(256).to(dev).eval() if ri(0, 3) else None
        l1, l2 = torch.nn.Linear(256, Hd).to(dev).eval(), torch.nn.Linear(Hd, 256).to(dev).eval()
        x = rn(B, T, 256) * 1.5 + 0.3
        use_res, alpha = ri(0, 2), [1.0, 0.5][ri(0, 2)]
        with torch.no_grad():
            if ln is not None: ln.weight.normal_(1, 0.2, generator=None); ln.bias.normal_(0, 0.2)
            got = decode_ops.ffn_fused(x, ln, l1, l2, act, residual=x if use_res else None, alpha=alpha)
            xd = x.double()
            h = xd if ln is None else F.layer_norm(xd, (256,), ln.weight.double(), ln.bias.double(), ln.eps)
            h = {"relu": torch.relu, "silu": F.silu, "gelu": F.gelu}[act](F.linear(h, l1.weight.double(), l1.bias.double()))
            ref = alpha * F.linear(h, l2.weight.double(), l2.bias.double()) + (xd if use_res else 0)
        err = float((got.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        assert got is not None and err < 3e-6, f"ffn err {err:.3e}"
        # ---- ragged GEMM / convolution: valid rows keep the bits of the dense call, skipped tiles are zero
        B, T = ri(1, 6), ri(1, 400)
        Cin, Cout, K = [256, 512, 1024][ri(0, 3)], [128, 256, 512, 768][ri(0, 4)], [1, 1, 3, 9][ri(0, 4)]
        tag = f"case {case}: ragged conv B={B} T={T} {Cin}->{Cout} k={K}"
        conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) // 2).to(dev); sc = SplitConv1d(conv.weight, conv.bias)
        x = rn(B, T, Cin); lens = torch.randint(0, T + 1, (B,), generator=g).to(torch.int32).to(dev); slack = [0, 4, 32][ri(0, 3)]
        res = rn(B, T, Cout) if ri(0, 2) else None
        with torch.no_grad():
            SplitConv1d.KSPLIT = False                        # the dense call in the single-launch form, like the ragged one
            dense, rag = sc(x, act="relu", residual=res), sc(x, act="relu", residual=res, lens=lens, slack=slack)
            SplitConv1d.KSPLIT = True
        for b in range(B):
            lim = min(T, int(lens[b]) + slack)
            assert torch.equal(dense[b, :lim], rag[b, :lim]), "valid rows differ"
            assert (rag[b, (lim + 127) // 128 * 128:] == 0).all() and torch.isfinite(rag[b]).all(), "skipped tiles"
    except Exception as e:                                   # noqa: BLE001
        bad += 1
        print(f"FAIL {tag}: {type(e).__name__}: {e}")
print(f"{n} cases, {bad} failures")
sys.exit(1 if bad else 0)
