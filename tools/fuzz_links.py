#!/usr/bin/env python3
"""Randomised sweep of the fused extract_links kernel (csrc/extract_links.hip) against the numpy oracle (oracle/graph_oracle.py,
s2t_conformer_dag.py:171-212): head geometry, ragged graph sizes (down to 1-2 vertices), banded / full windows.
r03: the same cases also run the training path (decode_ops.extract_links_autograd: forward with saved soft-max statistics +
dsp_extract_links_bwd) and compare its gradients w.r.t. q, k and the gate log-probabilities with torch autograd through the band
formulation, in fp64.
usage: fuzz_links.py [n_cases] [seed]   (GPU box only)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import graph_oracle as gorc
from daspeech_amd import decode_ops, _lib
PAD = 1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    B = int(rng.integers(1, 5)); L = int(rng.integers(2, 180)) if rng.random() < 0.8 else int(rng.integers(180, 420))
    h, ck = [(8, 64), (8, 32), (8, 128)][int(rng.integers(0, 3))]          # (the fused kernel serves 8 heads of 32 / 64 / 128; other geometries raise and the model uses the torch form)
    d = h * ck
    TRmax = int(rng.choice([1, 3, 7, 32, 64, 99999, int(rng.integers(1, L + 5))]))
    scale = float(rng.choice([0.1, 0.3, 1.0]))
    tile = int(rng.choice([0, 0, 32, 64, 128]))                              # r05: the tiled kernels (windows beyond the one-image LDS bound), forced on small graphs
    _lib.set_option("xl_tile", tile)
    mfma = int(rng.choice([0, 1, 1]))                                        # r05: the matrix-core kernels (8 x 64 heads), forced on graphs of every size
    _lib.set_option("xl_mfma", mfma)
    _lib.set_option("xl_contract", int(rng.choice([0, 1])))                 # the backward's contractions: exact-fp32 MFMAs | bf16-triple products
    feats = rng.standard_normal((B, L, d)).astype(np.float32) * scale
    lens = rng.integers(1, L + 1, B); lens[0] = L
    prev = np.full((B, L), 3, np.int64); prev[np.arange(L)[None] >= lens[:, None]] = PAD
    pos_w = (rng.standard_normal((L + 2, d)) * 0.3).astype(np.float32)
    ws = {k_: (rng.standard_normal((o, 2 * d)) * (0.5 / math.sqrt(d))).astype(np.float32) for k_, o in (("q", d), ("k", d), ("g", h))}
    bs = {k_: (rng.standard_normal(o) * 0.1).astype(np.float32) for k_, o in (("q", d), ("k", d), ("g", h))}
    tag = f"case {case}: B={B} L={L} heads={h}x{ck} TRmax={TRmax} lens={lens.tolist()} scale={scale} xl_tile={tile} xl_mfma={mfma}"
    try:
        want = gorc.extract_links(feats, prev, pos_w, ws["q"], bs["q"], ws["k"], bs["k"], ws["g"], bs["g"], TRmax, h, PAD)
        t = lambda a: torch.from_numpy(a).cuda()
        fp = torch.cat([t(feats), t(pos_w)[torch.from_numpy(gorc.make_positions(prev, PAD)).cuda()]], -1)
        q = (fp.double() @ t(ws["q"]).double().T + t(bs["q"]).double()).float().view(B, L, h, ck)
        k = (fp.double() @ t(ws["k"]).double().T + t(bs["k"]).double()).float().view(B, L, h, ck)
        lg = torch.log_softmax(fp.double() @ t(ws["g"]).double().T + t(bs["g"]).double(), -1).float()
        got = decode_ops.extract_links(q, k, lg, t(lens), max(1, min(TRmax, L - 1))).cpu().numpy()
        TRg = got.shape[2]
        w = want[:, :, :TRg] if want.shape[2] >= TRg else np.pad(want, ((0, 0), (0, 0), (0, TRg - want.shape[2])), constant_values=-np.inf)
        assert np.array_equal(np.isneginf(got), np.isneginf(w)), "-inf pattern"
        f = np.isfinite(w)
        assert np.allclose(got[f], w[f], rtol=1e-4, atol=1e-4), f"max diff {np.abs(got[f] - w[f]).max():.3e}"
        # ---- training path: forward equal to the inference kernel, gradients against fp64 torch autograd on the band formulation
        TR = max(1, min(TRmax, L - 1)); olen = t(lens)
        qa, ka, ga = q.clone().requires_grad_(), k.clone().requires_grad_(), lg.clone().requires_grad_()
        la = decode_ops.extract_links_autograd(qa, ka, ga, olen, TR)
        assert np.array_equal(la.detach().cpu().numpy(), got), "training forward != inference forward"
        cot = torch.from_numpy(rng.standard_normal((B, L, TR)).astype(np.float32)).cuda()
        fin = torch.isfinite(la)
        (la.masked_fill(~fin, 0.0) * cot).sum().backward()
        qd, kd, gd = q.double().clone().requires_grad_(), k.double().clone().requires_grad_(), lg.double().clone().requires_grad_()
        content = torch.einsum("bicf,bjcf->bijc", qd, kd) / (ck ** 0.5)
        idx = torch.arange(L, device="cuda").unsqueeze(1) + torch.arange(TR, device="cuda").unsqueeze(0) + 1
        invalid = idx.unsqueeze(0) >= olen.view(B, 1, 1)
        band = content.gather(2, idx.unsqueeze(0).masked_fill(invalid, 0).unsqueeze(-1).expand(-1, -1, -1, h))
        nouse = invalid.all(-1)
        band = band.masked_fill(invalid.unsqueeze(-1), float("-inf")).masked_fill(nouse.view(B, L, 1, 1), 0.0)
        band = torch.log_softmax(band, 2).masked_fill(invalid.unsqueeze(-1), -1e30)
        ld = torch.logsumexp(band + gd.unsqueeze(2), -1).masked_fill(invalid, float("-inf"))
        (ld.masked_fill(~torch.isfinite(ld), 0.0) * cot.double()).sum().backward()
        for nm, x, y in (("dq", qa.grad, qd.grad), ("dk", ka.grad, kd.grad), ("dgates", ga.grad, gd.grad)):
            err = float((x.double() - y).abs().max()); ref = max(1.0, float(y.abs().max()))
            assert err <= 2e-5 * ref, f"{nm}: max diff {err:.3e} (scale {ref:.2e})"
    except Exception as e:   # noqa
        bad += 1; print("FAIL", tag, "->", repr(e)[:300])
_lib.set_option("xl_tile", 0)
_lib.set_option("xl_mfma", -1)
_lib.set_option("xl_contract", -1)
print(f"{n} cases, {bad} failures")
