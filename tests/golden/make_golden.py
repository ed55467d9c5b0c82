#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own implementations.

Run only in the authoring container (needs /root/reference); the outputs (*.npz: inputs + expected outputs,
data only) are committed, this script is committed, nothing from the reference is copied.

Reference entry points executed here (imported by file path):
  DASpeech/custom_ops/dag_loss.py : torch_dag_loss (:325-366), __torch_max_loss (:369-386),
      torch_dag_best_alignment (:388-419), torch_dag_logsoftmax_gather_inplace (:421-425),
      logsumexp_keepdim (:303-311)
Dense<->compact links adapter: restated from dag_loss.py:439-448 (it lives under `if __name__ == "__main__"`
there, so it cannot be imported).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_ref_dag():
    spec = importlib.util.spec_from_file_location("ref_dag", f"{REF}/DASpeech/custom_ops/dag_loss.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def restore_valid_links(links):
    B, L, TR = links.shape
    idx = torch.arange(L).unsqueeze(1) + torch.arange(TR).unsqueeze(0) + 1
    idx = idx.masked_fill(idx >= L, L)
    res = torch.full((B, L, L + 1), float("-inf"), dtype=links.dtype)
    res.scatter_(2, idx.unsqueeze(0).expand(B, -1, -1), links)
    return res[:, :, :L]


def compact_from_dense(dense, TR):
    B, L, _ = dense.shape
    out = torch.zeros(B, L, TR, dtype=dense.dtype)
    for d in range(TR):
        n = L - d - 1
        if n <= 0:
            break
        i = torch.arange(n)
        out[:, i, d] = dense[:, i, i + d + 1]
    return out


def make_links(rng, B, L, TR, out_len):
    """masked log_softmax exactly like the model: mask AFTER softmax (s2t_conformer_dag.py:197-201)."""
    raw = torch.from_numpy(rng.standard_normal((B, L, TR)).astype(np.float32))
    i = torch.arange(L).view(1, L, 1)
    d = torch.arange(TR).view(1, 1, TR)
    valid = (i + d + 1) < out_len.view(B, 1, 1)
    raw = raw.masked_fill(~valid, float("-inf"))
    allinf = ~valid.any(-1, keepdim=True)
    ls = torch.log_softmax(raw.masked_fill(allinf, 0.0), -1)
    return ls.masked_fill(~valid, float("-inf"))


def dag_case(ref, name, match, links, out_len, tgt_len, dtype=torch.float64):
    """Run the reference torch path in `dtype` and store inputs (fp32) + expected outputs."""
    m = match.to(dtype)
    lk = links.to(dtype)
    m_g = m.clone().requires_grad_()
    lk_g = lk.clone().requires_grad_()
    dense = restore_valid_links(lk_g)
    loss = ref.torch_dag_loss(m_g, dense, out_len, tgt_len)
    finite = torch.isfinite(loss)
    if finite.any():
        gm, gl = torch.autograd.grad(loss[finite].sum(), [m_g, lk_g], allow_unused=True)
    else:
        gm, gl = torch.zeros_like(m), torch.zeros_like(lk)
    gm = torch.nan_to_num(gm, nan=0.0)
    gl = torch.nan_to_num(gl, nan=0.0)
    # Viterbi in fp32 — the dtype the product runs; ties resolved by torch.max (first index)
    m32, lk32 = match.float(), links.float()
    dense32 = restore_valid_links(lk32)
    score = ref.__dict__["__torch_max_loss"](m32, dense32, out_len, tgt_len)
    ok = torch.isfinite(score)
    path = torch.full((match.shape[0], match.shape[2]), -1, dtype=torch.long)
    if ok.any():
        p = ref.torch_dag_best_alignment(m32[ok].clone(), dense32[ok], out_len[ok], tgt_len[ok])
        path[ok] = p
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        match=match.float().numpy(), links=links.float().numpy(),
        out_len=out_len.numpy(), tgt_len=tgt_len.numpy(),
        loss=loss.detach().double().numpy(), grad_match=gm.double().numpy(), grad_links=gl.double().numpy(),
        finite=finite.numpy(), max_score=score.double().numpy(), path=path.numpy(), path_valid=ok.numpy(),
    )
    print(name, "loss", loss.detach().numpy(), "viterbi", score.numpy())


def main():
    torch.manual_seed(0)
    ref = load_ref_dag()
    rng = np.random.default_rng(1234)

    # A: banded TR < L-1, ragged lengths
    B, T, L, TR = 3, 5, 14, 6
    out_len = torch.tensor([14, 12, 13]); tgt_len = torch.tensor([5, 4, 3])
    match = torch.from_numpy(rng.standard_normal((B, T, L)).astype(np.float32)) - 2.0
    dag_case(ref, "dag_banded", match, make_links(rng, B, L, TR, out_len), out_len, tgt_len)

    # B: full transitions TR = L-1 (README flag --max-transition-length 99999)
    B, T, L = 2, 6, 12; TR = L - 1
    out_len = torch.tensor([12, 9]); tgt_len = torch.tensor([6, 5])
    match = torch.from_numpy(rng.standard_normal((B, T, L)).astype(np.float32)) - 3.0
    dag_case(ref, "dag_full", match, make_links(rng, B, L, TR, out_len), out_len, tgt_len)

    # C: force-emit style match (-inf and exact 0 entries, nat_dag_loss.py:130-132) + an unreachable sample
    B, T, L, TR = 3, 4, 10, 3
    out_len = torch.tensor([10, 10, 10]); tgt_len = torch.tensor([4, 4, 3])
    match = torch.from_numpy(rng.standard_normal((B, T, L)).astype(np.float32)) - 1.0
    match[0, 1, :] = float("-inf"); match[0, 1, 3] = 0.0          # glanced vertex 3 must emit token 1
    match[1, 2, 5:] = float("-inf")
    # sample 2: (T_b-1)*TR+1 = 7 < L_b = 10 -> end unreachable -> loss -inf (criterion zeroes it)
    dag_case(ref, "dag_forceemit", match, make_links(rng, B, L, TR, out_len), out_len, tgt_len)

    # D: constructed ties — all-zero scores, and a two-level plateau (bit-exact tie-break pin, fp32)
    B, T, L, TR = 2, 3, 6, 5
    out_len = torch.tensor([6, 6]); tgt_len = torch.tensor([3, 3])
    match = torch.zeros(B, T, L)
    links = torch.zeros(B, L, TR)
    i = torch.arange(L).view(1, L, 1); d = torch.arange(TR).view(1, 1, TR)
    links = links.masked_fill((i + d + 1) >= 6, float("-inf"))
    match[1, 1, 2] = 1.0; match[1, 1, 4] = 1.0                   # two equal best middle vertices
    dag_case(ref, "dag_ties", match, links, out_len, tgt_len)

    # E: moderately larger random case in fp32-friendly range, many ragged samples
    B, T, L, TR = 6, 9, 40, 8
    out_len = torch.tensor([40, 39, 38, 37, 36, 40]); tgt_len = torch.tensor([9, 8, 7, 9, 5, 2])
    match = torch.from_numpy(rng.standard_normal((B, T, L)).astype(np.float32)) * 2 - 4.0
    dag_case(ref, "dag_ragged", match, make_links(rng, B, L, TR, out_len), out_len, tgt_len)

    # K1: logsoftmax_gather, fp32 and fp16 logits; stride-0 expanded targets like nat_dag_loss.py:127
    for nm, dt in (("lsg_f32", torch.float32), ("lsg_f16", torch.float16)):
        B, L, V, T = 2, 7, 37, 5
        logits = torch.from_numpy(rng.standard_normal((B, L, V)).astype(np.float32) * 3).to(dt)
        tgt = torch.from_numpy(rng.integers(0, V, (B, T))).long()
        tgt[0, 1] = tgt[0, 3]                                     # duplicate target -> scatter_add accumulates
        idx = tgt.unsqueeze(1).expand(-1, L, -1)
        x = logits.clone().requires_grad_()
        _, m = ref.torch_dag_logsoftmax_gather_inplace(x, idx)
        w = torch.from_numpy(rng.standard_normal((B, L, T)).astype(np.float32))
        (gx,) = torch.autograd.grad((m * w).sum(), [x])
        sm = torch.softmax(logits.float(), -1)
        np.savez_compressed(os.path.join(HERE, nm + ".npz"), logits=logits.float().numpy(), targets=tgt.numpy(),
                            match=m.detach().numpy(), grad_out=w.numpy(), grad_logits=gx.float().numpy(),
                            softmax=sm.numpy())
        print(nm, m.shape)

    # logsumexp_keepdim pin (all -inf column guard, dag_loss.py:303-311)
    x = torch.tensor([[[-1.0, float("-inf")], [0.5, float("-inf")], [2.0, float("-inf")]]])
    y = ref.logsumexp_keepdim(x.clone(), 1)
    np.savez_compressed(os.path.join(HERE, "lse_keepdim.npz"), x=x.numpy(), y=y.numpy())


if __name__ == "__main__":
    sys.exit(main())
