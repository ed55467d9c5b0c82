"""float64 inputs to the DAG operators (the reference dispatches a double instantiation, dag_loss.cu:160,294,415,499): the band DP of
custom_ops/dag_double.py against the fp64 C oracle (CPU — the functions are device-agnostic torch) and, on a GPU, through the operator
names themselves."""
import numpy as np
import pytest
import torch

from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs
from daspeech_amd.custom_ops import dag_double as dd

CASES = [(0, 3, 9, 40, 8), (1, 2, 12, 64, 63), (2, 4, 7, 33, 4), (3, 1, 2, 5, 3), (4, 2, 20, 20, 19)]


def _t(seed, B, T, L, TR):
    m, k, ol, tl = make_dag_inputs(seed, B, T, L, TR)
    return m, k, ol, tl, [torch.from_numpy(x) for x in (m.astype(np.float64), k.astype(np.float64), ol, tl)]


@pytest.mark.parametrize("seed,B,T,L,TR", CASES)
def test_double_tables_and_loss_match_the_fp64_oracle(seed, B, T, L, TR):
    m, k, ol, tl, (mt, kt, olt, tlt) = _t(seed, B, T, L, TR)
    a = dd.alpha_table(mt, kt, olt, tlt).numpy()
    b = dd.beta_table(mt, kt, olt, tlt).numpy()
    a64 = orc.dag_alpha(m, k, ol, tl, np.float64)
    b64 = orc.dag_beta(m, k, ol, tl, np.float64)
    assert np.array_equal(np.isinf(a), np.isinf(a64)) and np.array_equal(np.isinf(b), np.isinf(b64))
    fa, fb = np.isfinite(a64), np.isfinite(b64)
    np.testing.assert_allclose(a[fa], a64[fa], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(b[fb], b64[fb], rtol=1e-12, atol=1e-12)
    loss = dd.dag_loss(mt, kt, olt, tlt).numpy()
    np.testing.assert_allclose(loss, b64[:, 0, 0], rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("seed,B,T,L,TR", CASES)
def test_double_gradients_match_the_fp64_oracle(seed, B, T, L, TR):
    m, k, ol, tl, (mt, kt, olt, tlt) = _t(seed, B, T, L, TR)
    mt.requires_grad_(); kt.requires_grad_()
    loss, (a, b) = dd.dag_loss_with_alpha_beta(mt, kt, olt, tlt)
    w = torch.linspace(0.5, 1.5, B, dtype=torch.float64)
    gm, gk = torch.autograd.grad((loss * w).sum(), [mt, kt])
    a64 = orc.dag_alpha(m, k, ol, tl, np.float64); b64 = orc.dag_beta(m, k, ol, tl, np.float64)
    gm64, gl64 = orc.dag_grad(w.numpy(), a64, b64, m, k, ol, tl, np.float64)
    np.testing.assert_allclose(gm.numpy(), gm64, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(torch.nan_to_num(gk, nan=0.0).numpy(), gl64, rtol=1e-9, atol=1e-12)
    fb = np.isfinite(b64)
    np.testing.assert_allclose(b.numpy()[fb], b64[fb], rtol=1e-12, atol=1e-12)
    # no gradient required -> beta is the reference's zeros
    _, (_, b0) = dd.dag_loss_with_alpha_beta(mt.detach(), kt.detach(), olt, tlt)
    assert float(b0.abs().sum()) == 0.0


@pytest.mark.parametrize("seed,B,T,L,TR", CASES)
def test_double_best_alignment_is_the_oracle_path(seed, B, T, L, TR):
    m, k, ol, tl, (mt, kt, olt, tlt) = _t(seed, B, T, L, TR)
    path = dd.dag_best_alignment(mt, kt, olt, tlt).numpy()
    np.testing.assert_array_equal(path, orc.dag_best_alignment(m, k, ol, tl, np.float64))


def test_double_best_alignment_ties_take_the_smallest_predecessor():
    B, T, L, TR = 1, 4, 8, 7
    m = np.zeros((B, T, L), np.float32); k = np.full((B, L, TR), -1.0, np.float32)
    i = np.arange(L)[:, None]; d = np.arange(TR)[None, :]
    k[0][(i + d + 1) >= L] = -np.inf
    ol = np.array([L], np.int64); tl = np.array([T], np.int64)
    p = dd.dag_best_alignment(torch.from_numpy(m).double(), torch.from_numpy(k).double(), torch.from_numpy(ol), torch.from_numpy(tl)).numpy()
    np.testing.assert_array_equal(p, orc.dag_best_alignment(m, k, ol, tl, np.float64))


@pytest.mark.gpu
def test_operator_names_take_float64_on_the_gpu():
    from daspeech_amd import custom_ops as ops
    dev = torch.device("cuda:0")
    m, k, ol, tl, (mt, kt, olt, tlt) = _t(7, 3, 10, 48, 16)
    mt, kt, olt, tlt = (x.to(dev) for x in (mt, kt, olt, tlt))
    mt.requires_grad_(); kt.requires_grad_()
    loss = ops.dag_loss(mt, kt, olt, tlt)
    assert loss.dtype == torch.float64
    b64 = orc.dag_beta(m, k, ol, tl, np.float64)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), b64[:, 0, 0], rtol=1e-12, atol=1e-11)
    gm, gk = torch.autograd.grad(loss.sum(), [mt, kt])
    a64 = orc.dag_alpha(m, k, ol, tl, np.float64)
    gm64, gl64 = orc.dag_grad(np.ones(3), a64, b64, m, k, ol, tl, np.float64)
    np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(torch.nan_to_num(gk, nan=0.0).cpu().numpy(), gl64, rtol=1e-9, atol=1e-12)
    l2, (a, b) = ops.dag_loss_with_alpha_beta(mt, kt, olt, tlt)
    assert a.dtype == torch.float64 and b.dtype == torch.float64
    path = ops.dag_best_alignment(mt.detach(), kt.detach(), olt, tlt)
    np.testing.assert_array_equal(path.cpu().numpy(), orc.dag_best_alignment(m, k, ol, tl, np.float64))
    # and the fp32 result of the HIP kernels agrees with the double one to fp32 accuracy
    l32 = ops.dag_loss(mt.detach().float(), kt.detach().float(), olt, tlt)
    np.testing.assert_allclose(l32.cpu().numpy(), loss.detach().cpu().numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,B,T,L,TR", CASES + [(5, 3, 40, 700, 32), (6, 2, 25, 300, 299), (7, 2, 16, 1031, 64)])
def test_native_double_kernels_match_the_fp64_oracle(seed, B, T, L, TR):
    """r06: float64 CUDA tensors run csrc/dag_dp_f64.hip (the reference dispatches double too, dag_loss.cu:160,294,415,499), not a torch loop:
    alpha / beta / loss to 1e-12, both gradients to 1e-9, the Viterbi path exactly — banded, dense and ragged graphs, an unreachable sample."""
    from daspeech_amd import custom_ops as ops
    dev = torch.device("cuda:0")
    m, k, ol, tl, (mt, kt, olt, tlt) = _t(seed, B, T, L, TR)
    if B > 1 and (T - 1) * TR >= L:          # make the last sample's end unreachable: two target rows cannot span the graph
        tl = tl.copy(); tl[-1] = 2 if L > TR + 2 else tl[-1]
        tlt = torch.from_numpy(tl)
    mt, kt, olt, tlt = (x.to(dev) for x in (mt, kt, olt, tlt))
    mt.requires_grad_(); kt.requires_grad_()
    loss, (a, b) = ops.dag_loss_with_alpha_beta(mt, kt, olt, tlt)
    assert loss.dtype == a.dtype == b.dtype == torch.float64 and loss.grad_fn is not None and "F64" in type(loss.grad_fn).__name__
    a64 = orc.dag_alpha(m, k, ol, tl, np.float64); b64 = orc.dag_beta(m, k, ol, tl, np.float64)
    an, bn = a.cpu().numpy(), b.cpu().numpy()
    assert np.array_equal(np.isinf(an), np.isinf(a64)) and np.array_equal(np.isinf(bn), np.isinf(b64))
    fa, fb = np.isfinite(a64), np.isfinite(b64)
    np.testing.assert_allclose(an[fa], a64[fa], rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(bn[fb], b64[fb], rtol=1e-12, atol=1e-11)
    ln = loss.detach().cpu().numpy()
    assert np.array_equal(np.isinf(ln), np.isinf(b64[:, 0, 0]))
    fin = np.isfinite(ln)
    np.testing.assert_allclose(ln[fin], b64[:, 0, 0][fin], rtol=1e-12, atol=1e-11)
    w = torch.linspace(0.5, 1.5, B, dtype=torch.float64, device=dev)
    gm, gk = torch.autograd.grad((loss.nan_to_num(neginf=0.0) * w).sum(), [mt, kt])
    gm64, gl64 = orc.dag_grad(w.cpu().numpy() * fin, a64, b64, m, k, ol, tl, np.float64)
    np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gk.cpu().numpy(), gl64, rtol=1e-9, atol=1e-12)
    path = ops.dag_best_alignment(mt.detach(), kt.detach(), olt, tlt)
    np.testing.assert_array_equal(path.cpu().numpy(), orc.dag_best_alignment(m, k, ol, tl, np.float64))
    # the torch band DP (CPU form) agrees with the kernels
    a_t = dd.alpha_table(mt.detach().cpu(), kt.detach().cpu(), olt.cpu(), tlt.cpu()).numpy()
    np.testing.assert_allclose(a_t[fa], an[fa], rtol=1e-12, atol=1e-11)
    # without a gradient beta is the reference's zeros
    _, (_, b0) = ops.dag_loss_with_alpha_beta(mt.detach(), kt.detach(), olt, tlt)
    assert float(b0.abs().sum()) == 0.0
