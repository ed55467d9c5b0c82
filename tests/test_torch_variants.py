"""The torch_* operator variants (device-agnostic part of the reference surface, dag_loss.py:303-425) against the
golden vectors made from the reference, on CPU."""
import os

import numpy as np
import pytest
import torch

from daspeech_amd.custom_ops import (logsumexp_keepdim, torch_dag_best_alignment, torch_dag_logsoftmax_gather_inplace,
                                     torch_dag_loss)
from oracle import dag_oracle as orc

DAG_CASES = ["dag_banded", "dag_full", "dag_forceemit", "dag_ties", "dag_ragged"]


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


@pytest.mark.parametrize("name", DAG_CASES)
def test_torch_dag_loss_and_grads(golden_dir, name):
    g = load(golden_dir, name)
    m = torch.from_numpy(g["match"]).double().requires_grad_()
    k = torch.from_numpy(g["links"]).double().requires_grad_()
    dense = torch.from_numpy(orc.restore_valid_links(np.zeros_like(g["links"]))).double()      # -inf mask
    # dense[b,i,j] = links[b,i,j-i-1] built differentiably
    B, L, TR = g["links"].shape
    idx = (torch.arange(L).view(L, 1) + torch.arange(TR).view(1, TR) + 1)
    cols = idx.clamp(max=L)
    full = torch.full((B, L, L + 1), float("-inf"), dtype=torch.float64)
    full = full.scatter(2, cols.unsqueeze(0).expand(B, -1, -1), k)[:, :, :L]
    ol = torch.from_numpy(g["out_len"]); tl = torch.from_numpy(g["tgt_len"])
    loss = torch_dag_loss(m, full, ol, tl)
    fin = torch.from_numpy(g["finite"])
    np.testing.assert_allclose(loss.detach().numpy()[g["finite"]], g["loss"][g["finite"]], rtol=1e-12, atol=1e-12)
    gm, gl = torch.autograd.grad(loss[fin].sum(), [m, k])
    np.testing.assert_allclose(torch.nan_to_num(gm).numpy(), g["grad_match"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(torch.nan_to_num(gl).numpy(), g["grad_links"], rtol=1e-9, atol=1e-12)
    assert dense.shape == full.shape


@pytest.mark.parametrize("name", DAG_CASES)
def test_torch_best_alignment(golden_dir, name):
    g = load(golden_dir, name)
    ok = g["path_valid"]
    m = torch.from_numpy(g["match"][ok]).clone()
    dense = torch.from_numpy(orc.restore_valid_links(g["links"][ok]))
    p = torch_dag_best_alignment(m, dense, torch.from_numpy(g["out_len"][ok]), torch.from_numpy(g["tgt_len"][ok]))
    np.testing.assert_array_equal(p.numpy(), g["path"][ok])


def test_torch_logsoftmax_gather(golden_dir):
    g = load(golden_dir, "lsg_f32")
    x = torch.from_numpy(g["logits"])
    B, L, V = x.shape
    idx = torch.from_numpy(g["targets"]).unsqueeze(1).expand(-1, L, -1)
    same, match = torch_dag_logsoftmax_gather_inplace(x, idx)
    assert same is x
    np.testing.assert_allclose(match.numpy(), g["match"], rtol=1e-6, atol=1e-6)


def test_logsumexp_keepdim(golden_dir):
    g = load(golden_dir, "lse_keepdim")
    y = logsumexp_keepdim(torch.from_numpy(g["x"]), 1)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-6)
    assert np.isneginf(y.numpy()[0, 0, 1])
