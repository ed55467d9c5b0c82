"""TEST INFRASTRUCTURE ONLY — CPU baseline leg.

Behavioural twin of the reference's CPU fallback path (DASpeech/custom_ops/dag_loss.py:303-425:
torch_dag_logsoftmax_gather_inplace -> torch_dag_loss (dense [B,L,L] links, T-1 torch steps, autograd backward)
and torch_dag_best_alignment), i.e. what a user of the reference runs without a GPU (`--torch-dag-*` flags).
The functions themselves are the torch_* variants of the operator surface (pinned to the golden vectors in
tests/test_torch_variants.py); this module only adds the bounded timing harness used by bench.py's
`cpu_baseline` ("kind": "port").
"""
import time

import torch

from daspeech_amd.custom_ops import (torch_dag_best_alignment, torch_dag_logsoftmax_gather_inplace,  # noqa: F401
                                     torch_dag_loss)


def _dense_links(links):
    B, L, TR = links.shape
    cols = (torch.arange(L).view(L, 1) + torch.arange(TR).view(1, TR) + 1).clamp(max=L)
    full = torch.full((B, L, L + 1), float("-inf"), dtype=links.dtype)
    return full.scatter(2, cols.unsqueeze(0).expand(B, -1, -1), links)[:, :, :L]


def time_cpu_dag_path(B, T, L, V, TR, seed=0, threads=None, with_alignment=True):
    """One pass of gather -> torch_dag_loss fwd+bwd (-> best alignment) on CPU tensors; returns seconds per phase."""
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, L, V, generator=g).requires_grad_()
    tgt = torch.randint(4, V, (B, T), generator=g)
    raw = torch.randn(B, L, TR, generator=g)
    i = torch.arange(L).view(1, L, 1)
    d = torch.arange(TR).view(1, 1, TR)
    valid = (i + d + 1) < L
    dead = ~valid.any(-1, keepdim=True)
    links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(dead, 0.0), -1).masked_fill(~valid, float("-inf"))
    links.requires_grad_()
    ol = torch.full((B,), L, dtype=torch.long)
    tl = torch.full((B,), T, dtype=torch.long)
    out = {}
    t0 = time.perf_counter()
    _, match = torch_dag_logsoftmax_gather_inplace(logits, tgt.unsqueeze(1).expand(-1, L, -1))
    match = match.transpose(1, 2)
    dense = _dense_links(links)
    loss = torch_dag_loss(match, dense, ol, tl)
    out["fwd_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    (-(loss / tl).mean()).backward()
    out["bwd_s"] = time.perf_counter() - t0
    if with_alignment:
        t0 = time.perf_counter()
        with torch.no_grad():
            torch_dag_best_alignment(match.detach().clone(), dense.detach(), ol, tl)
        out["align_s"] = time.perf_counter() - t0
    out["loss0"] = float(loss[0].detach())
    return out
