#!/usr/bin/env python3
"""Randomised sweep of the TTS glue kernels (csrc/decode_tts.hip) against the oracle: length regulator (bit-exact), durations
(within one on rounding boundaries), bucketize + embedding add (bit-exact), posterior / expected features.
usage: fuzz_tts_glue.py [n_cases] [seed]   (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from util_inputs import make_dag_inputs
from oracle import dag_oracle as orc
from daspeech_amd import decode_ops as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"; bad = 0
for case in range(n):
    B = int(rng.integers(1, 7)); N = int(rng.integers(1, 400)); C = int(rng.choice([1, 2, 7, 80, 256, 257, 512]))
    tag = f"case {case}: B={B} N={N} C={C}"
    try:
        dtype = [torch.float32, torch.float16][int(rng.integers(0, 2))]
        x = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(dtype)
        dur = rng.poisson(float(rng.choice([0.3, 3.0, 9.0])), (B, N)).astype(np.int64)
        if rng.random() < 0.3: dur[int(rng.integers(0, B))] = 0
        ref, lens_ref = orc.length_regulate(x.float().numpy(), dur)
        out, lens = D.length_regulate(x.to(dev), torch.from_numpy(dur).to(dev))
        assert np.array_equal(lens.cpu().numpy(), lens_ref) and np.array_equal(out.float().cpu().numpy(), ref), "length regulator"
        ld = (rng.standard_normal((B, N)) * 1.2 + 1.0).astype(np.float32); pm = rng.random((B, N)) < 0.2
        fac = float(rng.choice([1.0, 0.7, 1.5]))
        got = D.predicted_durations(torch.from_numpy(ld).to(dev), torch.from_numpy(pm).to(dev), fac).cpu().numpy()
        diff = np.abs(got - orc.durations(ld, pm, fac))
        assert diff.max() <= 1 and (diff > 0).mean() < 0.01 and np.all(got[pm] == 0), f"durations: max {diff.max()} frac {(diff > 0).mean():.4f}"
        nb = int(rng.choice([2, 17, 255])); bins = np.sort(rng.standard_normal(nb)).astype(np.float32)
        v = (rng.standard_normal(B * N) * 2).astype(np.float32); kk = min(3, v.size, nb); v[:kk] = bins[:kk]
        emb = rng.standard_normal((nb + 1, C)).astype(np.float32); xx = rng.standard_normal((B * N, C)).astype(np.float32)
        o2 = D.bucketize_embed_add(torch.from_numpy(xx).to(dev), torch.from_numpy(v).to(dev), torch.from_numpy(bins).to(dev), torch.from_numpy(emb).to(dev))
        assert np.array_equal(o2.cpu().numpy(), xx + emb[orc.bucketize(v, bins)]), "bucketize + embed"
        T = int(rng.integers(2, 14)); L = int(rng.integers(T + 1, 120)); TR = int(rng.integers(1, L))
        match, links, ol, tl = make_dag_inputs(int(rng.integers(1 << 30)), B, T, L, TR)
        a = orc.dag_alpha(match, links, ol, tl, np.float32); b = orc.dag_beta(match, links, ol, tl, np.float32)
        feats = rng.standard_normal((B, L, 16)).astype(np.float32)
        score_ref, ex_ref = orc.posterior_expect(a, b, feats)
        score = D.posterior(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
        assert np.allclose(score, score_ref, rtol=1e-4, atol=1e-6), "posterior"
        ex = D.expect_features(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(feats).to(dev)).cpu().numpy()
        assert np.allclose(ex, ex_ref[:, 1:], rtol=1e-3, atol=1e-4), "expected features"
        # r03: the fused posterior . features op (no [B,T,L] score tensor) forward and backward against fp64 torch autograd on the
        # two-step form (softmax over L of alpha + beta - match, masked rows -> 0, then @ features); the feature width varies
        Cf = int(rng.choice([8, 16, 40, 64]))
        f2 = rng.standard_normal((B, L, Cf)).astype(np.float32)
        ta, tb, tf = (torch.from_numpy(x).to(dev) for x in (a, b, f2))
        fa = tf.clone().requires_grad_()
        got = D.posterior_features(ta, tb, fa)
        cot = torch.from_numpy(rng.standard_normal(tuple(got.shape)).astype(np.float32)).to(dev)
        (got * cot).sum().backward()
        fd = tf.double().clone().requires_grad_()
        sc = (ta.double() + tb.double())
        sc = (sc - torch.logsumexp(sc, -1, keepdim=True)).exp()
        sc = sc.masked_fill(torch.isnan(sc), 0.0)
        want = torch.matmul(sc, fd)
        (want * cot.double()).sum().backward()
        assert float((got.detach().double() - want.detach()).abs().max()) <= 1e-4 * max(1.0, float(want.detach().abs().max())), "posterior_features forward"
        assert float((fa.grad.double() - fd.grad).abs().max()) <= 1e-4 * max(1.0, float(fd.grad.abs().max())), "posterior_features backward (features)"
    except Exception as e:   # noqa
        bad += 1; print("FAIL", tag, "->", repr(e)[:300])
print(f"{n} cases, {bad} failures")
