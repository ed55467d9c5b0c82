"""Pin the CPU oracle (oracle/dag_oracle.c) to the golden vectors produced by the REFERENCE's torch
implementations (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import dag_oracle as orc

DAG_CASES = ["dag_banded", "dag_full", "dag_forceemit", "dag_ties", "dag_ragged"]


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


@pytest.mark.parametrize("name", DAG_CASES)
def test_loss_alpha_beta_f64(golden_dir, name):
    g = load(golden_dir, name)
    a = orc.dag_alpha(g["match"], g["links"], g["out_len"], g["tgt_len"], np.float64)
    b = orc.dag_beta(g["match"], g["links"], g["out_len"], g["tgt_len"], np.float64)
    B = a.shape[0]
    la = a[np.arange(B), g["tgt_len"] - 1, g["out_len"] - 1]
    lb = b[:, 0, 0]
    fin = g["finite"]
    np.testing.assert_allclose(la[fin], g["loss"][fin], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lb[fin], g["loss"][fin], rtol=1e-12, atol=1e-12)
    assert np.all(np.isneginf(la[~fin])) and np.all(np.isneginf(lb[~fin]))


@pytest.mark.parametrize("name", DAG_CASES)
def test_loss_f32_within_reference_tolerance(golden_dir, name):
    # reference's own pin: allclose(rtol=1e-3, atol=1e-4)  (DASpeech/custom_ops/dag_loss.py:478)
    g = load(golden_dir, name)
    l32 = orc.dag_loss(g["match"], g["links"], g["out_len"], g["tgt_len"], np.float32)
    fin = g["finite"]
    np.testing.assert_allclose(l32[fin], g["loss"][fin], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", DAG_CASES)
def test_grads_f64(golden_dir, name):
    g = load(golden_dir, name)
    args = (g["match"], g["links"], g["out_len"], g["tgt_len"])
    a = orc.dag_alpha(*args, np.float64)
    b = orc.dag_beta(*args, np.float64)
    go = g["finite"].astype(np.float64)          # d(sum of finite losses)
    gm, gl = orc.dag_grad(go, a, b, *args, np.float64)
    np.testing.assert_allclose(gm, g["grad_match"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gl, g["grad_links"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", DAG_CASES)
def test_viterbi_path_bit_exact(golden_dir, name):
    g = load(golden_dir, name)
    path, a, _ = orc.dag_best_alignment(g["match"], g["links"], g["out_len"], g["tgt_len"], np.float32,
                                        want_internals=True)
    ok = g["path_valid"]
    assert ok.any()
    np.testing.assert_array_equal(path[ok], g["path"][ok])
    B = a.shape[0]
    sc = a[np.arange(B), g["tgt_len"] - 1, g["out_len"] - 1]
    np.testing.assert_array_equal(sc[ok].astype(np.float32), g["max_score"][ok].astype(np.float32))
    assert np.all(np.isneginf(sc[~ok]))
    # structural invariants
    for b in np.nonzero(ok)[0]:
        p = path[b]
        sel = p[p >= 0]
        assert len(sel) == g["tgt_len"][b] and np.all(np.diff(sel) == 1)
        assert p[0] == 0 and p[g["out_len"][b] - 1] == g["tgt_len"][b] - 1


@pytest.mark.parametrize("name,tol", [("lsg_f32", 1e-6), ("lsg_f16", 1e-6)])
def test_logsoftmax_gather(golden_dir, name, tol):
    g = load(golden_dir, name)
    B, L, V = g["logits"].shape
    T = g["targets"].shape[1]
    idx = np.broadcast_to(g["targets"][:, None, :], (B, L, T))      # stride-0 view like the caller's expand
    out, sm = orc.logsoftmax_gather(g["logits"], idx, np.float32, want_softmax=True)
    np.testing.assert_allclose(out, g["match"], rtol=tol, atol=tol)
    np.testing.assert_allclose(sm, g["softmax"], rtol=1e-5, atol=1e-7)
    gx = orc.logsoftmax_gather_bwd(sm, idx, g["grad_out"], np.float32)
    # fp16 reference grads are rounded to half: tolerance of one half ulp on O(1) values
    gt = 2e-3 if name.endswith("f16") else 1e-5
    np.testing.assert_allclose(gx, g["grad_logits"], rtol=gt, atol=gt)


def test_restore_valid_links_roundtrip():
    rng = np.random.default_rng(0)
    links = rng.standard_normal((2, 7, 3)).astype(np.float32)
    dense = orc.restore_valid_links(links)
    assert dense.shape == (2, 7, 7)
    assert dense[0, 2, 3] == links[0, 2, 0] and dense[1, 3, 6] == links[1, 3, 2]
    assert np.isneginf(dense[0, 3, 3]) and np.isneginf(dense[0, 0, 4]) and np.isneginf(dense[0, 5, 2])


def test_length_regulator_survey_example():
    # SURVEY.md §9.3, verified there against the stub-imported reference LengthRegulator
    x = np.arange(6 * 2, dtype=np.float32).reshape(2, 3, 2) + 1
    dur = np.array([[2, 0, 1], [1, 1, 0]])
    out, lens = orc.length_regulate(x, dur)
    assert lens.tolist() == [3, 2]
    np.testing.assert_array_equal(out[0], np.stack([x[0, 0], x[0, 0], x[0, 2]]))
    np.testing.assert_array_equal(out[1], np.stack([x[1, 0], x[1, 1], np.zeros(2, np.float32)]))
