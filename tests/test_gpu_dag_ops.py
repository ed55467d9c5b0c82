"""GPU parity tests: the HIP path (through the C ABI, via the reference-shaped operator API) against
(1) the golden vectors made from the reference's torch code, (2) the CPU oracle on seeded inputs,
(3) size-independent properties at BASELINE.json's full C2 size."""
import os

import numpy as np
import pytest
import torch

from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs

pytestmark = pytest.mark.gpu

DAG_CASES = ["dag_banded", "dag_full", "dag_forceemit", "dag_ties", "dag_ragged"]


def dev():
    return torch.device("cuda:0")


def ops():
    from daspeech_amd import custom_ops
    return custom_ops


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def to_dev(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev()) for a in arrs]


def test_library_loaded_is_in_tree():
    from daspeech_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libdaspeech_hip.so" in maps


# ---------------------------------------------------------------------------------------------- golden vectors

@pytest.mark.parametrize("name", DAG_CASES)
def test_golden_loss_and_grads(golden_dir, name):
    g = load(golden_dir, name)
    m, k, ol, tl = to_dev(g["match"], g["links"], g["out_len"], g["tgt_len"])
    m.requires_grad_(); k.requires_grad_()
    loss = ops().dag_loss(m, k, ol, tl)
    fin = torch.from_numpy(g["finite"]).to(dev())
    ref = torch.from_numpy(g["loss"]).to(dev())
    # reference's own tolerance is rtol 1e-3 / atol 1e-4 (dag_loss.py:478); we hold 1e-5
    torch.testing.assert_close(loss[fin].double(), ref[fin], rtol=1e-5, atol=1e-5)
    assert torch.isneginf(loss[~fin]).all()
    gm, gl = torch.autograd.grad((loss * fin).nan_to_num(neginf=0.0).sum() if False else loss[fin].sum(), [m, k])
    np.testing.assert_allclose(gm.cpu().numpy(), g["grad_match"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(gl.cpu().numpy(), g["grad_links"], rtol=2e-4, atol=1e-6)
    # no-grad path returns alpha[T_b-1, L_b-1]  (dag_loss.py:107-110)
    with torch.no_grad():
        loss2 = ops().dag_loss(m.detach(), k.detach(), ol, tl)
    torch.testing.assert_close(loss2[fin].double(), ref[fin], rtol=1e-5, atol=1e-5)
    assert torch.isneginf(loss2[~fin]).all()


@pytest.mark.parametrize("name", DAG_CASES)
def test_golden_viterbi_bit_exact(golden_dir, name):
    g = load(golden_dir, name)
    m, k, ol, tl = to_dev(g["match"], g["links"], g["out_len"], g["tgt_len"])
    path = ops().dag_best_alignment(m, k, ol, tl)
    assert path.dtype == torch.long and tuple(path.shape) == g["path"].shape
    ok = g["path_valid"]
    np.testing.assert_array_equal(path.cpu().numpy()[ok], g["path"][ok])


@pytest.mark.parametrize("name,dtype", [("lsg_f32", torch.float32), ("lsg_f16", torch.float16)])
def test_golden_logsoftmax_gather(golden_dir, name, dtype):
    g = load(golden_dir, name)
    logits = torch.from_numpy(g["logits"]).to(dev()).to(dtype)
    tgt = torch.from_numpy(g["targets"]).to(dev())
    B, L, V = logits.shape
    x = logits.clone().requires_grad_()
    work = x.clone()                                   # non-leaf, as the model's output is
    out_x, match = ops().dag_logsoftmax_gather_inplace(work, tgt.unsqueeze(1).expand(-1, L, -1))
    assert tuple(match.shape) == (B, L, tgt.shape[1]) and match.dtype == torch.float32
    mt = match.transpose(1, 2)                         # caller's transpose is free: [B,S,L] rows, dense when L is a multiple of 4, else pitched to the next
    assert mt.stride(2) == 1 and mt.stride(1) == (L + 3) // 4 * 4 and (L % 4 != 0 or mt.is_contiguous())
    np.testing.assert_allclose(match.detach().cpu().numpy(), g["match"], rtol=1e-5, atol=1e-5)
    # in-place side effect: softmax in the input dtype (logsoftmax_gather.cu:296-307)
    tol = 1e-6 if dtype == torch.float32 else 1e-3
    np.testing.assert_allclose(out_x.detach().float().cpu().numpy(), g["softmax"], rtol=tol, atol=tol)
    w = torch.from_numpy(g["grad_out"]).to(dev())
    (gx,) = torch.autograd.grad((match * w).sum(), [x])
    gt = 1e-5 if dtype == torch.float32 else 4e-3
    np.testing.assert_allclose(gx.float().cpu().numpy(), g["grad_logits"], rtol=gt, atol=gt)


# ---------------------------------------------------------------------------------------------- oracle, seeded

SHAPES = [
    # B, T, L, TR
    (4, 12, 96, 8),
    (3, 20, 160, 32),
    (2, 9, 70, 69),        # TR = L-1 (README --max-transition-length 99999)
    (5, 33, 257, 64),      # odd sizes, TR > wave
    (2, 2, 2, 1),          # minimum legal sizes
    (1, 40, 1030, 16),     # L > one pass of 1024 threads, 3 column strips
    (2, 30, 1500, 32),     # banded fast path, 3 strips, full window
    (3, 25, 1031, 7),      # odd L (scalar store path), odd TR
    (2, 70, 2048, 32),     # strip boundary exactly at L
    (3, 30, 300, 299),     # dense window TR = L-1 at a realistic graph size (wave-per-column kernels)
    (2, 17, 200, 130),     # dense window, TR < L-1
    (2, 24, 1024, 1023),   # dense window shared by 16 workgroups per (sample, direction): tagged-granule row hand-off
]


@pytest.mark.parametrize("shape", SHAPES)
def test_oracle_alpha_beta_loss(shape):
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(7 + L, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_()
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    a = alpha.cpu().numpy(); b = beta.cpu().numpy()
    assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64))
    fa = np.isfinite(a64); fb = np.isfinite(b64)
    # fp32 DP: error grows with |alpha| (ulp) and with the number of accumulated rows
    np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T)
    np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), b64[:, 0, 0], rtol=3e-6, atol=2e-5 * T)


@pytest.mark.parametrize("shape", SHAPES)
def test_oracle_gradients(shape):
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(11 + L, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_(); k.requires_grad_()
    loss = ops().dag_loss(m, k, o, t)
    fin = torch.isfinite(loss)
    w = torch.linspace(0.5, 1.5, B, device=dev())
    gm, gl = torch.autograd.grad((loss[fin] * w[fin]).sum(), [m, k])
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    go = (w.cpu().numpy() * fin.cpu().numpy()).astype(np.float64)
    gm64, gl64 = orc.dag_grad(go, a64, b64, match, links, ol, tl, np.float64)
    scale = 2e-5 * T                     # exp(x) with |dx| ~ ulp(alpha) * rows
    np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=50 * scale, atol=1e-7)
    np.testing.assert_allclose(gl.cpu().numpy(), gl64, rtol=50 * scale, atol=1e-7)


@pytest.mark.parametrize("shape", SHAPES)
def test_oracle_viterbi_bit_exact(shape):
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(13 + L, B, T, L, TR)
    # quantise so that exact ties occur, exercising the tie rule
    match = np.round(match * 2) / 2
    links = np.where(np.isfinite(links), np.round(links * 2) / 2, links).astype(np.float32)
    m, k, o, t = to_dev(match.astype(np.float32), links, ol, tl)
    path = ops().dag_best_alignment(m, k, o, t).cpu().numpy()
    ref = orc.dag_best_alignment(match.astype(np.float32), links, ol, tl, np.float32)
    np.testing.assert_array_equal(path, ref)


def test_invalid_samples_do_not_trap():
    B, T, L, TR = 3, 6, 20, 2
    match, links, ol, tl = make_dag_inputs(3, B, T, L, TR, ragged=False)
    m, k, o, t = to_dev(match, links, ol, tl)      # (T-1)*TR+1 = 11 < 20: end unreachable for all
    m.requires_grad_()
    loss = ops().dag_loss(m, k, o, t)
    assert torch.isneginf(loss).all()
    (gm,) = torch.autograd.grad(loss.nan_to_num(neginf=0.0).sum() + (m * 0).sum(), [m], allow_unused=True)
    path = ops().dag_best_alignment(m.detach(), k, o, t)
    assert path.shape == (B, L)
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype,V", [(torch.float32, 512), (torch.float16, 1000), (torch.bfloat16, 264), (torch.float32, 37),
                                     (torch.float32, 9000), (torch.float32, 16384), (torch.float16, 20000), (torch.float32, 8196),      # r05: wide rows (16 vectors per lane)
                                     (torch.float16, 6004), (torch.float32, 10001), (torch.bfloat16, 1003)])                            # r05: rows off the 16-byte grid (peeled head / tail)
def test_oracle_logsoftmax_gather(dtype, V):
    B, L, T = 3, 50, 17
    rng = np.random.default_rng(V)
    logits = torch.from_numpy((rng.standard_normal((B, L, V)) * 3).astype(np.float32)).to(dtype)
    tgt = rng.integers(0, V, (B, T))
    lf = logits.float().numpy()
    idx = np.broadcast_to(tgt[:, None, :], (B, L, T))
    ref, sm = orc.logsoftmax_gather(lf, idx, np.float64, want_softmax=True)
    x = logits.to(dev()).requires_grad_()
    work = x.clone()
    tg = torch.from_numpy(tgt).to(dev())
    out_x, match = ops().dag_logsoftmax_gather_inplace(work, tg.unsqueeze(1).expand(-1, L, -1))
    np.testing.assert_allclose(match.detach().cpu().numpy(), ref, rtol=2e-6, atol=2e-6)
    eps = {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    sm_dev = out_x.detach().float().cpu().numpy()          # snapshot: backward overwrites this buffer with the gradient
    np.testing.assert_allclose(sm_dev, sm, rtol=eps, atol=eps * 0.1)
    w = rng.standard_normal((B, L, T)).astype(np.float32)
    (gx,) = torch.autograd.grad((match * torch.from_numpy(w).to(dev())).sum(), [x])
    gref = orc.logsoftmax_gather_bwd(sm_dev, idx, w, np.float64)
    np.testing.assert_allclose(gx.float().cpu().numpy(), gref, rtol=4 * eps, atol=4 * eps)
    # materialised (non-expanded) index tensor gives the same result
    work2 = logits.to(dev()).clone()
    _, match2 = ops().dag_logsoftmax_gather_inplace(work2, tg.unsqueeze(1).expand(-1, L, -1).contiguous())
    assert torch.equal(match2, match.detach())


# ---------------------------------------------------------------------------------------------- full-size properties

def _check_grads_against_oracle(match, links, ol, tl, gm, gl, bs, rtol=2e-3):
    """grad_match / grad_links of utterance `bs` (d loss.sum()) against orc.dag_grad on fp64 alpha / beta of the same inputs."""
    mm, kk = np.asarray(match[bs:bs + 1], np.float64), np.asarray(links[bs:bs + 1], np.float64)
    o1, t1 = ol[bs:bs + 1].cpu().numpy(), tl[bs:bs + 1].cpu().numpy()
    a64, b64 = orc.dag_alpha(mm, kk, o1, t1, np.float64), orc.dag_beta(mm, kk, o1, t1, np.float64)
    gm64, gl64 = orc.dag_grad(np.ones(1), a64, b64, mm, kk, o1, t1, np.float64)
    got_m, got_l = gm[bs].detach().cpu().numpy(), gl[bs].detach().cpu().numpy()
    # a posterior is exp(alpha + beta - match - Z) with |alpha|, |Z| in the thousands: an fp32 ulp of those is the error floor
    floor = 8 * np.spacing(np.float32(np.abs(a64[np.isfinite(a64)]).max()))
    np.testing.assert_allclose(got_m, gm64[0], rtol=rtol + floor, atol=1e-7)
    np.testing.assert_allclose(got_l, gl64[0], rtol=rtol + floor, atol=1e-7)
    assert got_m.sum() == pytest.approx(float(t1[0]), rel=1e-3) and got_l.sum() == pytest.approx(float(t1[0] - 1), rel=1e-3)


def _c2_inputs(TR, B=32, T=512, L=4096, seed=0):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    out_len = L - torch.randint(0, 5, (B,), generator=gen)
    tgt_len = T - torch.randint(0, 5, (B,), generator=gen)
    d = dev()
    g2 = torch.Generator(device=d).manual_seed(seed)
    raw = torch.randn(B, L, TR, device=d, generator=g2)
    i = torch.arange(L, device=d).view(1, L, 1)
    dd = torch.arange(TR, device=d).view(1, 1, TR)
    valid = (i + dd + 1) < out_len.to(d).view(B, 1, 1)
    raw = raw.masked_fill(~valid, float("-inf"))
    dead = ~valid.any(-1, keepdim=True)
    links = torch.log_softmax(raw.masked_fill(dead, 0.0), -1).masked_fill(~valid, float("-inf"))
    match = torch.randn(B, T, L, device=d, generator=g2) - 9.0
    return match, links, out_len.to(d), tgt_len.to(d)


@pytest.mark.parametrize("TR", [32, 64])
def test_full_size_properties(TR):
    """BASELINE.json config 2 (B=32, L=4096, T=512): forward/backward consistency, posterior normalisation,
    transition-count identity, Viterbi path validity by re-scoring (the reference's own check, dag_loss.py:497-512).
    TR = 64 (r06): the same at the bench's `dag_tr64` leg — strip2g forward, the gradient kernel in two planes of 32 transitions, maxstripw."""
    import ctypes
    from daspeech_amd import _lib
    match, links, ol, tl = _c2_inputs(TR)
    B, T, L = match.shape
    m = match.clone().requires_grad_(); k = links.clone().requires_grad_()
    diag = (ctypes.c_uint * 4)(); _lib.load().dsp_dag_debug_k5(diag)
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, ol, tl)
    ar = torch.arange(B, device=dev())
    a_end = alpha[ar, tl - 1, ol - 1]
    assert torch.isfinite(loss).all()
    torch.testing.assert_close(a_end, loss.detach(), rtol=2e-5, atol=0.0)     # beta[0,0] == alpha[end]
    gm, gl = torch.autograd.grad(loss.sum(), [m, k])
    torch.cuda.synchronize(); _lib.load().dsp_dag_debug_k5(diag)
    assert diag[3] == (5 if TR == 32 else 6), diag[3]                          # one fused launch / one launch of two planes: the kernels the bench line times
    rows = gm.sum(-1)                                                          # sum_j posterior(t, j) = 1 for t < T_b
    tmask = torch.arange(T, device=dev()).view(1, T) < tl.view(B, 1)
    assert (rows[tmask] - 1).abs().max() < 0.05, (rows[tmask] - 1).abs().max()
    assert rows[~tmask].abs().max() == 0
    trans = gl.sum((1, 2))                                                     # each path makes T_b - 1 transitions
    assert ((trans - (tl - 1).float()).abs() / (tl - 1).float()).max() < 0.05
    # Viterbi
    path = ops().dag_best_alignment(match, links, ol, tl)
    assert (path[:, 0] == 0).all() and (path[ar, ol - 1] == tl - 1).all()
    on = path >= 0
    assert (on.sum(1) == tl).all()
    # re-score the path
    score = torch.zeros(B, device=dev(), dtype=torch.float64)
    pc = path.cpu().numpy(); mc = match.cpu().numpy(); kc = links.cpu().numpy()
    best = []
    for b in range(B):
        js = np.nonzero(pc[b] >= 0)[0]
        s = float(mc[b, 0, js[0]])
        for t in range(1, len(js)):
            s += float(kc[b, js[t - 1], js[t] - js[t - 1] - 1]) + float(mc[b, t, js[t]])
        best.append(s)
    best = np.array(best)
    assert np.all(best <= loss.detach().cpu().numpy() + 1e-3)                  # best path <= marginal
    # r06: EVERY utterance of the batch against the oracle, not one — the Viterbi paths bit for bit (the oracle's sequential fp32 max-DP +
    # back-trace), the loss and the alpha / beta tables element-wise against the fp64 recurrences (4.3e9 terms per direction on the host cores)
    oln, tln = ol.cpu().numpy(), tl.cpu().numpy()
    np.testing.assert_array_equal(pc, orc.dag_best_alignment(mc, kc, oln, tln, np.float32))
    for name, got, fn in (("alpha", alpha, orc.dag_alpha), ("beta", beta, orc.dag_beta)):
        want = fn(mc, kc, oln, tln, np.float64)
        g = got.detach().cpu().numpy()
        assert np.array_equal(np.isneginf(g), np.isneginf(want)), name
        f = np.isfinite(want)
        np.testing.assert_allclose(g[f], want[f], rtol=3e-6, atol=2e-5 * T, err_msg=name)
        if name == "beta":
            np.testing.assert_allclose(loss.detach().cpu().numpy(), want[:, 0, 0], rtol=3e-6, atol=2e-5 * T)
        del want, g, f
    # ... and four utterances' GRADIENTS element by element against the fp64 oracle (K4 / K5 at full size, not only their row sums)
    for bs in (0, 7, 19, 31):
        _check_grads_against_oracle(mc, kc, ol, tl, gm, gl, bs=bs)


# ---------------------------------------------------------------------------------------------- fast path vs generic

@pytest.mark.parametrize("shape", [(2, 30, 1500, 32), (3, 25, 1031, 7), (2, 70, 2048, 32), (4, 16, 513, 32),
                                   (3, 40, 1028, 32), (2, 33, 2304, 20), (40, 9, 1024, 32), (2, 12, 64, 5)])
def test_fast_paths_match_generic_kernels(shape):
    """The three DP kernel families (generic row-sequential, banded 2-column log-space strips, strip4 exp-space) agree:
    alpha/beta within fp32 DP tolerance, Viterbi traces bit-exact."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(21 + L, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_()
    res = {}
    try:
        for path in (1, 2, 4, 5, 7):
            _lib.set_option("dp_path", path)
            loss, (a, b) = ops().dag_loss_with_alpha_beta(m, k, o, t)
            assert _lib.last_launch_status() == 0
            p = ops().dag_best_alignment(m.detach(), k, o, t)
            assert _lib.last_launch_status() == 0
            res[path] = (loss.detach(), a, b, p)
    finally:
        _lib.set_option("dp_path", 0)
    _, a_g, b_g, p_g = res[1]
    fa = torch.isfinite(a_g); fb = torch.isfinite(b_g)
    for path in (2, 4, 5, 7):
        _, a_f, b_f, p_f = res[path]
        assert torch.equal(torch.isneginf(a_f), torch.isneginf(a_g)), path
        assert torch.equal(torch.isneginf(b_f), torch.isneginf(b_g)), path
        torch.testing.assert_close(a_f[fa], a_g[fa], rtol=3e-6, atol=3e-5 * T)
        torch.testing.assert_close(b_f[fb], b_g[fb], rtol=3e-6, atol=3e-5 * T)
        assert torch.equal(p_f, p_g), path            # Viterbi: bit-exact between kernel families


def test_strip4_exactness_guard():
    """Inputs that force the exp-space kernel's log-space fallback: adjacent vertices whose scores differ by hundreds of
    nats (far beyond the 2^-90 window), plus -inf emissions (force-emit mask) — results must still match the fp64 oracle."""
    from daspeech_amd import _lib
    B, T, L, TR = 2, 24, 1024, 32
    match, links, ol, tl = make_dag_inputs(77, B, T, L, TR, ragged=False)
    rng = np.random.default_rng(5)
    match = match + (rng.integers(0, 2, match.shape) * -300.0).astype(np.float32)     # cliffs of 300 nats between neighbours
    match[0, 5, :] = -np.inf; match[0, 5, 40] = 0.0                                   # force-emit row
    links = np.where(np.isfinite(links), links + (rng.integers(0, 2, links.shape) * -120.0), links).astype(np.float32)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_()
    try:
        res = {}
        for path in (4, 5):
            _lib.set_option("dp_path", path)
            loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
            assert _lib.last_launch_status() == 0
            assert _lib.last_fallback_count() > 0          # the guard really fired
            res[path] = (alpha.cpu().numpy(), beta.cpu().numpy())
    finally:
        _lib.set_option("dp_path", 0)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    for path, (a, b) in res.items():
        assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64)), path
        fa = np.isfinite(a64); fb = np.isfinite(b64)
        np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=1e-3)
        np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=1e-3)


@pytest.mark.parametrize("slope", [2.0, 12.0, 40.0])
def test_exp_space_paths_on_peaked_scores(slope):
    """Sharply peaked emissions (a trained model: the score falls `slope` nats per vertex away from the aligned position) give
    DP rows whose neighbouring vertices differ by tens of binades — the regime where the exp-space kernels' shared exponents
    run out of range and the escape / medium / exact paths take over.  Must match the fp64 oracle."""
    from daspeech_amd import _lib
    B, T, L, TR = 2, 40, 1024, 32
    match, links, ol, tl = make_dag_inputs(123, B, T, L, TR, ragged=True)
    jj = np.arange(L, dtype=np.float32)[None, None, :]
    centre = (np.arange(T, dtype=np.float32) * (L - 1) / (T - 1))[None, :, None]
    match = (match * 0.1 - slope * np.abs(jj - centre)).astype(np.float32)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_()
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    try:
        for path in (5,):
            _lib.set_option("dp_path", path)
            loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
            assert _lib.last_launch_status() == 0
            a, b = alpha.cpu().numpy(), beta.cpu().numpy()
            assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64)), path
            fa = np.isfinite(a64); fb = np.isfinite(b64)
            np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=1e-3)
            np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=1e-3)
    finally:
        _lib.set_option("dp_path", 0)


def test_banded_repeated_launches_reuse_workspace():
    """Tag epochs: many launches back to back on one stream must never see a stale granule."""
    from daspeech_amd import _lib
    B, T, L, TR = 2, 12, 1100, 16
    match, links, ol, tl = make_dag_inputs(99, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    ref = None
    for it in range(20):
        with torch.no_grad():
            loss = ops().dag_loss(m + 0.001 * it, k, o, t)
        if it == 0:
            ref = loss.clone()
        assert _lib.last_launch_status() == 0
    with torch.no_grad():
        again = ops().dag_loss(m, k, o, t)
    assert torch.equal(again, ref)


@pytest.mark.parametrize("shape", [(2, 20, 1024, 32), (3, 17, 516, 20), (2, 9, 260, 7)])
def test_lazy_alignment_ties_bit_exact(shape):
    """Values-only max-DP + lazy back-trace (trace == NULL, dp_path 7 / auto): quantised scores give exact ties on most rows;
    the recomputed arg-max must follow the reference's rule (smallest predecessor index) — compared with the oracle and the
    eager trace kernels."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(71 + L, B, T, L, TR)
    match = (np.round(match) * 1.0).astype(np.float32)
    links = np.where(np.isfinite(links), np.round(links), links).astype(np.float32)
    match[0, 3, :] = -np.inf; match[0, 3, 5] = 0.0                       # a forced vertex and -inf candidates on the way
    m, k, o, t = to_dev(match, links, ol, tl)
    ref = orc.dag_best_alignment(match, links, ol, tl, np.float32)
    try:
        for path in (7, 1, 4):
            _lib.set_option("dp_path", path)
            got = ops().dag_best_alignment(m, k, o, t).cpu().numpy()
            assert _lib.last_launch_status() == 0
            np.testing.assert_array_equal(got, ref, err_msg=f"dp_path {path}")
    finally:
        _lib.set_option("dp_path", 0)


@pytest.mark.parametrize("shape,k5", [((2, 40, 1024, 32), 2), ((2, 40, 1024, 32), 1), ((3, 30, 516, 20), 2), ((2, 40, 260, 7), 2)])   # (T-1)*TR >= L-1: the end is reachable
def test_grad_links_exp_space_weak_and_peaked_links(shape, k5):
    """K5 in exp space: transitions more than 100 binades under 0 (the per-lane exact redo), -inf emissions, short and ragged
    windows; k5_path 1 = the log-space kernel it replaces.  Both against the fp64 oracle."""
    from daspeech_amd import _lib
    import ctypes
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(91 + L, B, T, L, TR)
    rng = np.random.default_rng(3)
    links = np.where(np.isfinite(links), links + (rng.integers(0, 6, links.shape) == 0) * -90.0, links).astype(np.float32)   # ~2^-130
    match[1 % B, 4, :] = -np.inf; match[1 % B, 4, 3 * TR] = -1.0          # forced vertex; the end stays reachable
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_(); k.requires_grad_()
    diag = (ctypes.c_uint * 4)()
    try:
        _lib.set_option("k5_path", k5)
        _lib.load().dsp_dag_debug_k5(diag)
        loss = ops().dag_loss(m, k, o, t)
        fin = torch.isfinite(loss)
        assert fin.any()
        gm, gl = torch.autograd.grad(loss[fin].sum(), [m, k])
        torch.cuda.synchronize()
        _lib.load().dsp_dag_debug_k5(diag)
    finally:
        _lib.set_option("k5_path", 0)
    # the pinned kernel family is the one that ran (autograd's worker thread sees the pin); 5 = the exp-space kernel as ONE launch with grad_match (r06)
    assert diag[3] == (5 if (k5 == 2 and L % 4 == 0) else 1)
    if k5 == 2 and L % 4 == 0:
        assert diag[2] > 0                               # the weak-transition redo really ran
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    go = fin.cpu().numpy().astype(np.float64)
    gm64, gl64 = orc.dag_grad(go, a64, b64, match, links, ol, tl, np.float64)
    np.testing.assert_allclose(gl.cpu().numpy(), gl64, rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("dtype,V", [(torch.float32, 512), (torch.float32, 8192), (torch.float16, 1000), (torch.bfloat16, 264), (torch.float32, 37)])
def test_lazy_softmax_mode_same_match_and_gradient(dtype, V):
    """set_lazy_softmax(True): the forward leaves the logits untouched (two floats of row statistics instead of the in-place
    softmax), the backward recomputes the softmax — match and d/d logits must equal the default mode's."""
    from daspeech_amd.custom_ops import set_lazy_softmax
    B, L, T = 3, 70, 17
    rng = np.random.default_rng(V)
    logits = torch.from_numpy((rng.standard_normal((B, L, V)) * 3).astype(np.float32)).to(dtype).to(dev())
    tg = torch.from_numpy(rng.integers(0, V, (B, T))).to(dev())
    w = torch.from_numpy(rng.standard_normal((B, L, T)).astype(np.float32)).to(dev())
    res = {}
    for lazy in (False, True):
        prev = set_lazy_softmax(lazy)
        try:
            x = logits.clone().requires_grad_()
            work = x.clone()
            out_x, match = ops().dag_logsoftmax_gather_inplace(work, tg.unsqueeze(1).expand(-1, L, -1))
            after_fwd = out_x.detach().clone()
            (gx,) = torch.autograd.grad((match * w).sum(), [x])
        finally:
            set_lazy_softmax(prev)
        res[lazy] = (match.detach().clone(), gx, after_fwd)
    assert torch.equal(res[True][0], res[False][0])                       # same kernel, same match
    assert torch.equal(res[True][2], logits)                              # lazy: logits untouched by the forward
    eps = {torch.float32: 2e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]   # default mode rounds the stored softmax
    torch.testing.assert_close(res[True][1].float(), res[False][1].float(), rtol=eps, atol=eps)


@pytest.mark.parametrize("L", [8192, 8196])
def test_alignment_large_graph_properties(L):
    """The largest graph the trace-free alignment takes (L = 8192: path image + transition window + segments fill the LDS) and
    the first size past it (eager trace kernels): the path must be a valid monotone alignment with the Viterbi score."""
    B, T, TR = 2, 300, 32
    match, links, ol, tl = make_dag_inputs(5 + L, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    path = ops().dag_best_alignment(m, k, o, t).cpu().numpy()
    for b in range(B):
        pos = np.nonzero(path[b] >= 0)[0]
        Tb, Lb = int(tl[b]), int(ol[b])
        assert len(pos) == Tb and pos[0] == 0 and pos[-1] == Lb - 1
        assert np.array_equal(path[b][pos], np.arange(Tb))
        assert np.all(np.diff(pos) >= 1) and np.all(np.diff(pos) <= TR)
    ref = orc.dag_best_alignment(match[:1], links[:1], ol[:1], tl[:1], np.float32)
    np.testing.assert_array_equal(path[:1], ref)


# ---------------------------------------------------------------------------------------------- dense window on the f32 matrix cores
DENSE_SHAPES = [(3, 24, 200, 199), (2, 40, 256, 255), (4, 33, 130, 129), (2, 20, 500, 100), (2, 70, 400, 399), (1, 9, 1024, 1023),
                (3, 18, 192, 191), (2, 50, 640, 639),
                (44, 10, 448, 447),       # 308 workgroups per direction: more than CUs, so the two-workgroups-per-CU builds run
                (3, 30, 300, 33), (2, 25, 520, 64), (2, 40, 390, 48)]     # r05: windows 33 .. 64 are served by the dense-window kernels too


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape", DENSE_SHAPES)
def test_dense_mfma_dp_matches_oracle(shape, masked):
    """dag_dp_dense_mfma.hip (TR > 64: exp-space blocked products on v_mfma_f32_16x16x4_f32 + sequential diagonal blocks) against the
    fp64 oracle: ragged lengths, windows between 64 and L-1, graph sizes that are not multiples of the 64-column block, and
    force-emit style emissions (whole rows -inf except one column, scattered -inf cells: nat_dag_loss.py:130-132)."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(11 + L, B, T, L, TR)
    if masked:
        rng = np.random.default_rng(L)
        match[0, min(3, T - 1), :] = -np.inf; match[0, min(3, T - 1), min(L - 1, 10)] = 0.0
        match[rng.random(match.shape) < 0.1] = -np.inf
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_()
    try:
        res = {}
        # 9 = matrix-core kernel (the auto choice above 64) with 32-row (default) and 16-row chunks, 1 = log-space row-sequential kernels,
        # 0 = auto (r05: windows 33 .. 64 take the banded log-space strips)
        for path, mt in ((9, 0), (9, 1), (1, 0), (0, 0)):
            _lib.set_option("dp_path", path); _lib.set_option("dm_mt", mt)
            loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
            assert _lib.last_launch_status() == 0
            res[(path, mt)] = (alpha.cpu().numpy(), beta.cpu().numpy(), loss.detach().cpu().numpy())
    finally:
        _lib.set_option("dp_path", 0); _lib.set_option("dm_mt", 0)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    for path, (a, b, ls) in res.items():
        assert not np.isnan(a).any() and not np.isnan(b).any(), path
        assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64)), path
        fa = np.isfinite(a64); fb = np.isfinite(b64)
        np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T)
        np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T)
        fl = np.isfinite(b64[:, 0, 0])
        np.testing.assert_allclose(ls[fl], b64[fl, 0, 0], rtol=3e-6, atol=2e-5 * T)


@pytest.mark.parametrize("weak", [False, True])
@pytest.mark.parametrize("shape", [s_ for s_ in DENSE_SHAPES if s_[3] > 64])        # (K5's block products serve TR > 64; 33 .. 64 keep the tiled log-space kernel)
def test_dense_grad_links_block_products_match_oracle(shape, weak):
    """dag_grad_dense.hip (K5 for TR > 64: per-block-pair products over the target axis on v_mfma_f32_16x16x4_f32, diagonal and
    window-edge pairs term by term) against the fp64 oracle and the tiled log-space kernel (k5_path 1): ragged lengths, windows between
    64 and L-1, -inf emissions, transitions ~2^-130 (weak), and the compact layout's entries past the graph (i + d + 1 >= L_b) are
    exactly zero as the reference's at::zeros output leaves them (dag_loss.cu:493)."""
    from daspeech_amd import _lib
    import ctypes
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(17 + L, B, T, L, TR)
    rng = np.random.default_rng(L + 5)
    match[rng.random(match.shape) < 0.08] = -np.inf
    if weak:
        links = np.where(np.isfinite(links), links + (rng.integers(0, 6, links.shape) == 0) * -90.0, links).astype(np.float32)
    res = {}
    try:
        for k5 in (0, 1, 2):
            _lib.set_option("k5_path", k5)
            m, k, o, t = to_dev(match, links, ol, tl)
            m.requires_grad_(); k.requires_grad_()
            loss = ops().dag_loss(m, k, o, t)
            fin = torch.isfinite(loss)
            gm, gl = torch.autograd.grad(loss[fin].sum(), [m, k])
            assert _lib.last_launch_status() == 0
            diag = (ctypes.c_uint * 4)(); _lib.load().dsp_dag_debug_k5(diag)
            # 1 = the tiled log-space kernel really ran when pinned
            want = (1,) if k5 == 1 else ((3,) if (k5 == 2 or TR > 128) else (1, 6))      # auto up to 128: tiled or the planes, by estimated cost
            assert diag[3] in want, (k5, diag[3])
            res[k5] = (gm.cpu().numpy(), gl.cpu().numpy())
    finally:
        _lib.set_option("k5_path", 0)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    gm64, gl64 = orc.dag_grad(fin.cpu().numpy().astype(np.float64), a64, b64, match, links, ol, tl, np.float64)
    i = np.arange(L)[None, :, None]; d = np.arange(TR)[None, None, :]
    outside = (i + d + 1) >= ol[:, None, None]
    for k5, (gm_, gl_) in res.items():
        assert np.isfinite(gl_).all(), k5
        assert (gl_[outside] == 0).all(), k5
        np.testing.assert_allclose(gl_, gl64, rtol=2e-3, atol=1e-7, err_msg=f"k5_path {k5}")
        np.testing.assert_allclose(gm_, gm64, rtol=2e-3, atol=1e-7, err_msg=f"k5_path {k5}")


@pytest.mark.parametrize("weak", [False, True])
@pytest.mark.parametrize("shape", [(3, 30, 300, 33), (2, 25, 520, 64), (2, 40, 392, 48), (2, 20, 500, 100), (2, 33, 1028, 128), (3, 17, 260, 97),
                                   (2, 12, 264, 65), (2, 9, 128, 127), (2, 21, 390, 48), (2, 14, 301, 100)])
def test_windows_33_to_128_backward_in_blocks_of_32_transitions(shape, weak):
    """r06: dag_loss backward on windows 33 .. 128 = the TR <= 32 exp-space kernel with one plane of workgroups per block of 32 transitions
    (gridDim.y), beta read 32 k columns to the right, plane 0 fused with grad_match (family 6, pinned with k5_path 3; rows sit on 16-byte
    boundaries: graph lengths off the grid run with a row pitch, as on the narrow windows).  Against the fp64 oracle and the tiled log-space kernel: ragged
    lengths, -inf emissions, transitions ~2^-130 (weak: the exact redo inside every block), entries past the graph exactly zero
    (dag_loss.cu:461-466, 493)."""
    from daspeech_amd import _lib
    import ctypes
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(23 + L + TR, B, T, L, TR)
    rng = np.random.default_rng(L + TR)
    match[rng.random(match.shape) < 0.08] = -np.inf
    if weak:
        links = np.where(np.isfinite(links), links + (rng.integers(0, 6, links.shape) == 0) * -90.0, links).astype(np.float32)
    res = {}
    try:
        for k5 in (3, 1, 0):
            _lib.set_option("k5_path", k5)
            m, k, o, t = to_dev(match, links, ol, tl)
            m.requires_grad_(); k.requires_grad_()
            loss = ops().dag_loss(m, k, o, t)
            fin = torch.isfinite(loss)
            gm, gl = torch.autograd.grad(loss[fin].sum(), [m, k])
            assert _lib.last_launch_status() == 0
            diag = (ctypes.c_uint * 4)(); _lib.load().dsp_dag_debug_k5(diag)
            want = (1,) if k5 == 1 else ((6,) if k5 == 3 else (1, 6))                  # auto: the cheaper estimate of the two
            assert diag[3] in want, (k5, diag[3])
            if k5 == 3 and weak:
                assert diag[2] > 0                           # the weak-transition redo really ran
            res[k5] = (gm.cpu().numpy(), gl.cpu().numpy())
    finally:
        _lib.set_option("k5_path", 0)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    gm64, gl64 = orc.dag_grad(fin.cpu().numpy().astype(np.float64), a64, b64, match, links, ol, tl, np.float64)
    i = np.arange(L)[None, :, None]; d = np.arange(TR)[None, None, :]
    outside = (i + d + 1) >= ol[:, None, None]
    for k5, (gm_, gl_) in res.items():
        assert np.isfinite(gl_).all() and np.isfinite(gm_).all(), k5
        assert (gl_[outside] == 0).all(), k5
        np.testing.assert_allclose(gl_, gl64, rtol=2e-3, atol=1e-7, err_msg=f"k5_path {k5}")
        np.testing.assert_allclose(gm_, gm64, rtol=2e-3, atol=1e-7, err_msg=f"k5_path {k5}")


@pytest.mark.parametrize("which", ["match", "links"])
@pytest.mark.parametrize("shape,k5", [((3, 40, 520, 32), 0), ((2, 30, 517, 20), 0), ((2, 40, 392, 48), 3), ((2, 33, 1030, 128), 3), ((2, 20, 500, 100), 1),
                                      ((2, 24, 300, 299), 0)])
def test_backward_with_one_gradient_wanted(shape, k5, which):
    """Only `match_all` or only `links` requires a gradient (the DAG frozen / a detached emission): the launchers then run K4 alone, or K5 alone —
    on the narrow windows the un-fused exp-space kernel, on 33 .. 128 the planes without their grad_match part — and return None for the other
    input.  Against the fp64 oracle; the gradient that IS computed equals the one of the both-gradients call bit for bit."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(77 + L + TR, B, T, L, TR)
    try:
        _lib.set_option("k5_path", k5)
        m, k, o, t = to_dev(match, links, ol, tl)
        mb, kb = m.clone().requires_grad_(), k.clone().requires_grad_()
        lb = ops().dag_loss(mb, kb, o, t)
        fin = torch.isfinite(lb)
        gmb, gkb = torch.autograd.grad(lb[fin].sum(), [mb, kb])
        x = (m if which == "match" else k).clone().requires_grad_()
        loss = ops().dag_loss(x, k, o, t) if which == "match" else ops().dag_loss(m, x, o, t)
        (g,) = torch.autograd.grad(loss[fin].sum(), [x])
        assert _lib.last_launch_status() == 0
    finally:
        _lib.set_option("k5_path", 0)
    assert torch.equal(g, gmb if which == "match" else gkb)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    gm64, gl64 = orc.dag_grad(fin.cpu().numpy().astype(np.float64), a64, b64, match, links, ol, tl, np.float64)
    np.testing.assert_allclose(g.cpu().numpy(), gm64 if which == "match" else gl64, rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("shape", [(4, 256, 2048, 2047), (32, 100, 400, 399)])
def test_dense_window_full_size_c1_and_readme_shape(shape):
    """BASELINE configs[0] (C1: B=4, T=256, L=2048, dense window) and the README's training shape (B=32, T=100, L=400,
    --max-transition-length 99999) at FULL size on the HIP ops: loss = beta[0,0] = alpha[T_b-1, L_b-1], the matrix-core DP and the
    log-space DP agree, gradients are a distribution over the alignment (sum_j grad_match[t, j] = d loss for every t), the Viterbi
    path is a valid monotone alignment and scores no more than the marginal."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(3 + L, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_(); k.requires_grad_()
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    assert _lib.last_launch_status() == 0 and torch.isfinite(loss).all()
    ar = torch.arange(B, device=m.device)
    torch.testing.assert_close(alpha[ar, t - 1, o - 1], loss, rtol=3e-6, atol=2e-5 * T)
    try:
        _lib.set_option("dp_path", 1)
        loss_l, (alpha_l, beta_l) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    finally:
        _lib.set_option("dp_path", 0)
    assert torch.equal(torch.isneginf(alpha), torch.isneginf(alpha_l)) and torch.equal(torch.isneginf(beta), torch.isneginf(beta_l))
    fa = torch.isfinite(alpha_l); fb = torch.isfinite(beta_l)
    torch.testing.assert_close(alpha[fa], alpha_l[fa], rtol=6e-6, atol=4e-5 * T)
    torch.testing.assert_close(beta[fb], beta_l[fb], rtol=6e-6, atol=4e-5 * T)
    gm, gk = torch.autograd.grad(loss.sum(), [m, k])
    assert torch.isfinite(gm).all() and torch.isfinite(gk).all()
    rows = gm.sum(-1)                                          # [B, T]: 1 on rows < T_b, 0 beyond
    want = (torch.arange(T, device=m.device).unsqueeze(0) < t.unsqueeze(1)).float()
    torch.testing.assert_close(rows, want, rtol=0, atol=2e-3)
    _check_grads_against_oracle(match, links, o, t, gm, gk, bs=1)              # K4 / K5 element by element vs the fp64 oracle
    path = ops().dag_best_alignment(m.detach(), k.detach(), o, t).cpu().numpy()
    np.testing.assert_array_equal(path[:1], orc.dag_best_alignment(match[:1], links[:1], ol[:1], tl[:1], np.float32))
    for b in range(B):
        pos = np.nonzero(path[b] >= 0)[0]
        Tb, Lb = int(tl[b]), int(ol[b])
        assert len(pos) == Tb and pos[0] == 0 and pos[-1] == Lb - 1 and np.array_equal(path[b][pos], np.arange(Tb))
        score = match[b, np.arange(Tb), pos].sum() + sum(links[b, pos[i], pos[i + 1] - pos[i] - 1] for i in range(Tb - 1))
        assert score <= float(loss[b]) + 1e-3 * abs(float(loss[b]))


def test_dense_window_full_size_c2_tr4095():
    """BASELINE configs[1] with the README's `--max-transition-length 99999` (C2: B=32, T=512, L=4096, TR=4095 — a 2.1 GB transition
    tensor) at FULL size on the HIP ops, through size-independent properties: loss = beta[0,0] = alpha[T_b-1, L_b-1]; the matrix-core
    DP agrees with the row-sequential log-space DP cell by cell; grad_match is a distribution over the vertices for every target row,
    grad_links counts T_b - 1 transitions per utterance; the Viterbi path is a valid monotone alignment scoring no more than the
    marginal and exactly what the max-DP's own end cell says."""
    from daspeech_amd import _lib
    match, links, o, t = _c2_inputs(4095)
    B, T, L = match.shape
    m = match.requires_grad_(); k = links.requires_grad_()
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    assert _lib.last_launch_status() == 0 and torch.isfinite(loss).all() and not _lib.last_dense_gave_up()
    ar = torch.arange(B, device=m.device)
    torch.testing.assert_close(alpha[ar, t - 1, o - 1], loss, rtol=3e-6, atol=2e-5 * T)
    try:
        _lib.set_option("dp_path", 1)
        loss_l, (alpha_l, beta_l) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    finally:
        _lib.set_option("dp_path", 0)
    assert torch.equal(torch.isneginf(alpha), torch.isneginf(alpha_l)) and torch.equal(torch.isneginf(beta), torch.isneginf(beta_l))
    fa = torch.isfinite(alpha_l); fb = torch.isfinite(beta_l)
    torch.testing.assert_close(alpha[fa], alpha_l[fa], rtol=6e-6, atol=4e-5 * T)
    torch.testing.assert_close(beta[fb], beta_l[fb], rtol=6e-6, atol=4e-5 * T)
    del alpha_l, beta_l, fa, fb
    # ... and one utterance of the batch against the fp64 C ORACLE itself (4.3e9 terms per direction; its rows run on all host cores)
    bs = 5
    mm, kk = match[bs:bs + 1].detach().cpu().numpy(), links[bs:bs + 1].detach().cpu().numpy()
    ol1, tl1 = o[bs:bs + 1].cpu().numpy(), t[bs:bs + 1].cpu().numpy()
    for name, got, want in (("alpha", alpha[bs], orc.dag_alpha(mm, kk, ol1, tl1, np.float64)[0]), ("beta", beta[bs], orc.dag_beta(mm, kk, ol1, tl1, np.float64)[0])):
        got = got.detach().cpu().numpy()
        assert np.array_equal(np.isneginf(got), np.isneginf(want)), name
        f = np.isfinite(want)
        np.testing.assert_allclose(got[f], want[f], rtol=3e-6, atol=2e-5 * T + 1e-4, err_msg=name)
    gm, gk = torch.autograd.grad(loss.sum(), [m, k])
    assert torch.isfinite(gm).all() and torch.isfinite(gk).all()
    want = (torch.arange(T, device=m.device).unsqueeze(0) < t.unsqueeze(1)).float()
    # (alpha, beta and Z are ~ -4.5e3 here: one fp32 ulp of them is 4.9e-4, a posterior exp(alpha + beta - match - Z) carries a few)
    torch.testing.assert_close(gm.sum(-1), want, rtol=0, atol=2e-2)
    torch.testing.assert_close(gk.sum((1, 2)), (t - 1).float(), rtol=5e-3, atol=0)
    del gm, gk
    with torch.no_grad():
        path = ops().dag_best_alignment(match.detach(), links.detach(), o, t)
        on = path >= 0
        assert (on.sum(1) == t).all() and (path[:, 0] == 0).all() and (path[ar, o - 1] == t - 1).all()
        for b in range(B):
            pos = on[b].nonzero().flatten()
            assert torch.equal(path[b, pos], torch.arange(int(t[b]), device=m.device))
            score = match[b, torch.arange(int(t[b]), device=m.device), pos].double().sum() + \
                links[b, pos[:-1], pos[1:] - pos[:-1] - 1].double().sum()
            assert float(score) <= float(loss[b]) + 1e-3 * abs(float(loss[b]))
        # ... and utterance 5's path against the ORACLE's sequential max-DP + back-trace, bit for bit (4.3e9 add / compare steps, fp32, the
        # oracle's columns on all host cores)
        np.testing.assert_array_equal(path[bs:bs + 1].cpu().numpy(), orc.dag_best_alignment(mm, kk, ol1, tl1, np.float32))


def test_workspace_sizes_are_reported_and_library_scratch_still_works():
    """The ABI is honest about memory: dsp_dag_workspace_bytes / dsp_dag_alignment_workspace_bytes are non-zero, the Python operators
    pass a torch-allocated workspace, and a caller that passes NULL still gets correct results from the library's own scratch."""
    from daspeech_amd import _lib
    lib = _lib.load()
    assert lib.dsp_dag_workspace_bytes(32, 512, 4096, 32) >= 2 * 32 * 4 * 512 * 32 * 8
    assert lib.dsp_dag_workspace_bytes(4, 256, 2048, 2047) > 0 and lib.dsp_dag_alignment_workspace_bytes(32, 512, 4096, 32) > 0
    B, T, L, TR = 3, 20, 1024, 32
    match, links, ol, tl = make_dag_inputs(31, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    st = _lib.current_stream_handle()
    res = []
    for use_ws in (True, False):
        alpha = torch.empty((B, T, L), device="cuda"); beta = torch.empty_like(alpha); loss = torch.empty(B, device="cuda")
        n = lib.dsp_dag_workspace_bytes(B, T, L, TR)
        ws = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(0xAB) if use_ws else None           # garbage in: the call zeroes what it uses
        _lib.check(lib.dsp_dag_loss_fwd(_lib.ptr(m), _lib.ptr(k), _lib.ptr(o), _lib.ptr(t), _lib.ptr(alpha), _lib.ptr(beta), _lib.ptr(loss),
                                        B, T, L, TR, _lib.ptr(ws), n if use_ws else 0, st), "fwd")
        assert _lib.last_launch_status() == 0
        res.append((alpha.clone(), beta.clone(), loss.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.parametrize("shape", [(4, 20, 512, 32), (3, 24, 200, 199), (3, 20, 1100, 64), (3, 16, 700, 100), (3, 20, 515, 32)])
def test_dag_ops_are_graph_capturable(shape):
    """dag_loss forward + backward and dag_best_alignment captured in a HIP graph (torch.cuda.CUDAGraph) and replayed on new inputs:
    the launches keep no host-side state (caller workspace zeroed on the stream, tags start at 1), so the replay must give what eager
    execution gives."""
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(41 + L, B, T, L, TR)
    m, k, o, t = to_dev(match, links, ol, tl)
    sm = m.clone().requires_grad_(); sk = k.clone().requires_grad_()

    def step():
        loss = ops().dag_loss(sm, sk, o, t)
        gm, gk = torch.autograd.grad(loss.sum(), [sm, sk])
        with torch.no_grad():
            path = ops().dag_best_alignment(sm.detach(), sk.detach(), o, t)
        return loss.detach(), gm, gk, path
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()                                 # warm-up on the capture stream: workspaces, function attributes
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    match2, links2, _, _ = make_dag_inputs(77 + L, B, T, L, TR)
    links2 = np.where(np.isfinite(links), links2, -np.inf).astype(np.float32)                 # same validity pattern as lengths o / t
    with torch.no_grad():
        sm.copy_(torch.from_numpy(match2)); sk.copy_(torch.from_numpy(links2))
    g.replay()
    torch.cuda.synchronize()
    got = [x.clone() for x in out]
    want = step()
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("quant", [False, True])
@pytest.mark.parametrize("shape", DENSE_SHAPES)
def test_dense_alignment_bit_exact_with_ties(shape, quant):
    """dag_dp_dense_max.hip (blocked max-plus DP + trace-free back-trace) for dense windows: paths bit-exact against the f32 oracle —
    also with quantised scores, where most rows hold exact ties (rule: smallest predecessor index) — and identical to the
    row-sequential kernels with a trace tensor (dp_path 1)."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(13 + L, B, T, L, TR)
    if quant:
        match = (np.round(match * 2) / 2).astype(np.float32)
        links = np.where(np.isfinite(links), np.round(links * 2) / 2, links).astype(np.float32)
    m, k, o, t = to_dev(match, links, ol, tl)
    ref = orc.dag_best_alignment(match, links, ol, tl, np.float32)
    mid = 32 < TR <= 128                         # r06: the auto choice for these windows is the values-only max-DP strips (dag_dp_maxstripw.hip); 9 pins the dense kernels
    try:
        for path in (0, 1, 9):
            _lib.set_option("dp_path", path)
            got = ops().dag_best_alignment(m, k, o, t).cpu().numpy()
            assert _lib.last_launch_status() == 0
            np.testing.assert_array_equal(got, ref, err_msg=f"dp_path {path}")
        _lib.set_option("dp_path", 9 if mid else 0)
        for mt in (1, 2):                       # both chunk heights of the max-plus kernel (r05: 32 rows is what the largest launches take)
            _lib.set_option("dx_mt", mt)
            got = ops().dag_best_alignment(m, k, o, t).cpu().numpy()
            assert _lib.last_launch_status() == 0
            np.testing.assert_array_equal(got, ref, err_msg=f"dx_mt {mt}")
    finally:
        _lib.set_option("dp_path", 0); _lib.set_option("dx_mt", 0)
    assert _lib.load().dsp_dag_alignment_trace_optional(L, TR) == 1    # no B*T*L trace tensor for any window above 32 (r06: 33 .. 64 included)


def _relink(links, ol, seed):
    """masked log-softmax transitions for edited lengths (make_dag_inputs' rule)"""
    B, L, TR = links.shape
    for bb in range(B):
        i = np.arange(L)[:, None]; d = np.arange(TR)[None, :]
        valid = (i + d + 1) < ol[bb]
        raw = np.where(valid, np.random.default_rng(seed + bb).standard_normal((L, TR)), -np.inf)
        mx = np.max(np.where(valid, raw, -1e30), axis=-1, keepdims=True)
        e = np.where(valid, np.exp(raw - mx), 0.0); ssum = e.sum(-1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            links[bb] = np.where(valid, raw - mx - np.log(np.where(ssum > 0, ssum, 1.0)), -np.inf).astype(np.float32)


@pytest.mark.parametrize("shape", [(2, 1, 130, 129), (1, 2, 128, 127), (3, 40, 129, 65), (2, 5, 193, 192), (2, 33, 257, 200), (3, 17, 320, 66),
                                   (2, 64, 128, 127), (1, 31, 4160, 4159)])
def test_dense_kernels_edge_shapes(shape):
    """The three dense-window kernel families (forward DP, block-product gradient, max-plus alignment) at the edges of their support:
    T = 1 (the end is unreachable: -inf loss, as the reference's), T = 2, L = 128 and windows of 65 / 66, L not a multiple of the
    64-column block, a graph barely longer than its target, a one-vertex one-token sample, L > 4096 — against the fp64 / f32 oracle."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(23 + L + T, B, T, L, TR, ragged=T > 1)
    if B > 1 and T > 2:
        ol[0] = max(int(tl[0]), min(L, int(tl[0]) + 3))
    if B > 2:
        tl[2] = 1; ol[2] = 1
    if (B > 1 and T > 2) or B > 2:
        _relink(links, ol, L)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_(); k.requires_grad_()
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    assert _lib.last_launch_status() == 0
    fin = torch.isfinite(loss)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    assert np.array_equal(fin.cpu().numpy(), np.isfinite(b64[:, 0, 0]))
    a, b = alpha.cpu().numpy(), beta.cpu().numpy()
    assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64))
    assert not np.isnan(a).any() and not np.isnan(b).any()
    fa, fb = np.isfinite(a64), np.isfinite(b64)
    np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T + 1e-4)
    np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T + 1e-4)
    if fin.any():
        gm, gk = torch.autograd.grad(loss[fin].sum(), [m, k])
        gm64, gl64 = orc.dag_grad(fin.cpu().numpy().astype(np.float64), a64, b64, match, links, ol, tl, np.float64)
        np.testing.assert_allclose(gk.cpu().numpy(), gl64, rtol=2e-3, atol=1e-7)
        np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=2e-3, atol=1e-7)
    if fin.all():
        pref = orc.dag_best_alignment(match, links, ol, tl, np.float32)
        try:
            for mt in (0, 1, 2):                # auto, 16- and 32-row chunks of the max-plus kernel
                _lib.set_option("dx_mt", mt)
                path = ops().dag_best_alignment(m.detach(), k.detach(), o, t).cpu().numpy()
                np.testing.assert_array_equal(path, pref, err_msg=f"dx_mt {mt}")
        finally:
            _lib.set_option("dx_mt", 0)


def _weak_links(seed, B, L, TR, ol, scale):
    """log_softmax of logits with a large spread: most transitions are far weaker than 2^-126 (exp space holds them as exact zeros)."""
    rng = np.random.default_rng(seed)
    raw = (rng.standard_normal((B, L, TR)) * scale).astype(np.float32)
    i = np.arange(L)[None, :, None]; d = np.arange(TR)[None, None, :]
    valid = (i + d + 1) < ol[:, None, None]
    mx = np.max(np.where(valid, raw, -1e30), axis=-1, keepdims=True)
    e = np.where(valid, np.exp(raw - mx), 0.0); ssum = e.sum(-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(valid, raw - mx - np.log(np.where(ssum > 0, ssum, 1.0)), -np.inf).astype(np.float32)


@pytest.mark.parametrize("shape", [(6, 40, 330, 329), (3, 70, 600, 599)])
def test_dense_dp_hands_unrepresentable_batches_to_the_log_space_kernels(shape):
    """Batches the exp-space products are the wrong tool for, with forced emissions as GLAT's (nat_dag_loss.py:130-132).  "weak":
    transitions of -100 ... -400 nats, exact zeros in exp space — the range check ahead of the DP kernel raises the give-up flag and the
    stand-by log-space kernels queued behind it produce the result.  "mild": every weight representable (>= -60 nats) but most sums
    far under the rows' shared exponents — the per-lane-reference pass of the diagonal block resolves them in exp space, no redo, no
    hand-over (r02 spent its budget here).  "faint": every weight in [-84, -70] nats, so every sum is under the guard whatever the
    reference — the DP counts the predecessors its exact redo visits and gives up past its budget (16: gives up; -1: no stand-by,
    every flagged row redone in place).  A benign batch with a budget of one visit goes through the same hand-over if it has a redo
    at all.  Same answer as the fp64 oracle every time."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(5 + L, B, T, L, TR)
    weak = _weak_links(7 + L, B, L, TR, ol, 60.0)
    rng = np.random.default_rng(L)
    forced = match.copy()
    for b in range(B):                                    # forced emissions: one live vertex per glanced target position
        for t in rng.choice(int(tl[b]), size=max(1, int(tl[b]) // 4), replace=False):
            j = int(rng.integers(t, int(ol[b]) - (int(tl[b]) - 1 - t)))
            forced[b, t, :] = -np.inf; forced[b, t, j] = 0.0
    mild = np.where(np.isneginf(weak), weak, np.maximum(weak, -60.0)).astype(np.float32)       # representable; sums far under the shared exponents
    faint = np.where(np.isneginf(weak), weak, np.clip(weak, -84.0, -70.0)).astype(np.float32)  # representable, but EVERY sum is under the guard
    cases = {"weak": (forced, weak, 0, True), "mild": (forced, mild, 0, False), "faint, budget 16": (forced, faint, 16, True),
             "faint, no stand-by": (forced, faint, -1, False), "benign, budget 1": (match, links, 1, None)}
    try:
        for name, (mm, kk, budget, expect_gave_up) in cases.items():
            _lib.set_option("dm_budget", budget)
            m, k, o, t = to_dev(mm, kk, ol, tl)
            m.requires_grad_()
            loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
            assert _lib.last_launch_status() == 0, name
            gave_up, cells = _lib.last_dense_gave_up(), _lib.last_fallback_count()
            if expect_gave_up is None:
                assert gave_up == (cells > 0), (name, cells)
            else:
                assert gave_up == expect_gave_up, (name, cells)
            a64 = orc.dag_alpha(mm, kk, ol, tl, np.float64); b64 = orc.dag_beta(mm, kk, ol, tl, np.float64)
            a, b = alpha.cpu().numpy(), beta.cpu().numpy()
            assert not np.isnan(a).any() and not np.isnan(b).any(), name
            assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64)), name
            fa, fb = np.isfinite(a64), np.isfinite(b64)
            np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T + 1e-4, err_msg=name)
            np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T + 1e-4, err_msg=name)
            l64 = -b64[:, 0, 0]          # (criterion convention checked elsewhere; here: loss finite where beta[0, 0] is)
            fin = torch.isfinite(loss)
            assert np.array_equal(fin.cpu().numpy(), np.isfinite(l64)), name
            # the gradient kernels (block products in exp space, dag_grad_dense.hip) see the same alpha / beta whichever kernel made them
            if fin.any():
                k.requires_grad_()
                loss2 = ops().dag_loss(m, k, o, t)
                gm, gk = torch.autograd.grad(loss2[fin].sum(), [m, k])
                gm64, gl64 = orc.dag_grad(fin.cpu().numpy().astype(np.float64), a64, b64, mm, kk, ol, tl, np.float64)
                np.testing.assert_allclose(gk.cpu().numpy(), gl64, rtol=2e-3, atol=1e-7, err_msg=name)
                np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=2e-3, atol=1e-7, err_msg=name)
    finally:
        _lib.set_option("dm_budget", 0)


@pytest.mark.parametrize("shape", [(3, 53, 838, 837), (2, 78, 551, 83), (1, 51, 144, 143)])
def test_dense_grad_links_with_unrepresentable_transitions(shape):
    """grad_links on the dense-window block products (dag_grad_dense.hip) when most transitions are weaker than fp32 can hold as a
    factor (log-softmax outputs scaled x40 and renormalised: -100 ... -900 nats).  The entry is the posterior of its transition —
    anything up to 1 — so the link has to meet the sum in the log domain (e^link * sum was 0 * inf = NaN, r02 fuzzing), and a block
    pair whose scaled alpha overflows the 2^100 clamp is redone term by term.  Against the fp64 oracle; no NaN."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(77 + L, B, T, L, TR)
    fin_l = np.isfinite(links)
    links = np.where(fin_l, links * 40.0, links)
    mx = np.max(np.where(fin_l, links, -1e30), -1, keepdims=True)
    ssum = np.where(fin_l, np.exp(links - mx), 0).sum(-1, keepdims=True)
    links = np.where(fin_l, links - mx - np.log(np.where(ssum > 0, ssum, 1)), links).astype(np.float32)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_(); k.requires_grad_()
    loss = ops().dag_loss(m, k, o, t)
    assert _lib.last_launch_status() == 0
    fin = torch.isfinite(loss)
    assert fin.any()
    gm, gk = torch.autograd.grad(loss[fin].sum(), [m, k])
    assert not torch.isnan(gk).any() and not torch.isnan(gm).any()
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    assert np.array_equal(fin.cpu().numpy(), np.isfinite(b64[:, 0, 0]))
    gm64, gl64 = orc.dag_grad(fin.cpu().numpy().astype(np.float64), a64, b64, match, links, ol, tl, np.float64)
    np.testing.assert_allclose(gk.cpu().numpy(), gl64, rtol=3e-3, atol=2e-7)
    np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=3e-3, atol=2e-7)


@pytest.mark.parametrize("shape", [(4, 40, 330, 329), (2, 90, 700, 699)])
def test_dense_dp_on_trained_model_like_scores(shape):
    """Emissions near 0 on a band around the alignment, a -20-nat floor elsewhere, 4-sigma transition logits over the whole window
    (what a trained model produces; tools/peaked_dense.py): sums hundreds of binades apart within a row, many cells through the exact
    redo — and, past its budget, the whole batch through the stand-by kernels.  Whichever path runs: alpha, beta, loss and both
    gradients against the fp64 oracle, the alignment bit-exact."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    rng = np.random.default_rng(L)
    ol = np.full(B, L, np.int64); tl = np.full(B, T, np.int64); ol[-1] -= 3; tl[-1] -= 2
    j = np.arange(L)[None, None, :]; c = (np.arange(T) * (L - 1) / (T - 1))[None, :, None]
    match = np.where(np.abs(j - c) < 6, -0.5 + 0.3 * rng.standard_normal((B, T, L)), -20.0 + 3.0 * rng.standard_normal((B, T, L))).astype(np.float32)
    links = _weak_links(3 + L, B, L, TR, ol, 4.0)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_(); k.requires_grad_()
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    assert _lib.last_launch_status() == 0 and torch.isfinite(loss).all()
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    a, b = alpha.detach().cpu().numpy(), beta.detach().cpu().numpy()
    assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64))
    fa, fb = np.isfinite(a64), np.isfinite(b64)
    np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T + 1e-4)
    np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T + 1e-4)
    gm, gk = torch.autograd.grad(loss.sum(), [m, k])
    gm64, gl64 = orc.dag_grad(np.ones(B), a64, b64, match, links, ol, tl, np.float64)
    np.testing.assert_allclose(gk.cpu().numpy(), gl64, rtol=3e-3, atol=2e-7)
    np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=3e-3, atol=2e-7)
    path = ops().dag_best_alignment(m.detach(), k.detach(), o, t).cpu().numpy()
    np.testing.assert_array_equal(path, orc.dag_best_alignment(match, links, ol, tl, np.float32))


def test_dense_window_c1_trained_model_like_scores_stay_on_the_matrix_cores():
    """BASELINE configs[0] at FULL size (B=4, T=256, L=2048, TR=2047) on trained-model-like scores (emissions near 0 on a band around
    the alignment, a -20-nat floor elsewhere, 4-sigma transition logits: rows that climb 5 - 60 binades per column left of the band).
    r02 spent the exact-redo budget on them and handed the batch to the log-space stand-by kernels (7 - 9 ms instead of 1.4); the
    per-lane-reference pass of the diagonal block keeps them in exp space.  Asserted: no hand-over, no exact-redo cell, and one
    utterance of the batch — alpha and beta, every cell — against the fp64 C oracle (not against another HIP kernel)."""
    from daspeech_amd import _lib
    B, T, L, TR = 4, 256, 2048, 2047
    rng = np.random.default_rng(L)
    ol = np.full(B, L, np.int64); tl = np.full(B, T, np.int64); ol[-1] -= 3; tl[-1] -= 2
    j = np.arange(L)[None, None, :]; c = (np.arange(T) * (L - 1) / (T - 1))[None, :, None]
    match = np.where(np.abs(j - c) < 6, -0.5 + 0.3 * rng.standard_normal((B, T, L)), -20.0 + 3.0 * rng.standard_normal((B, T, L))).astype(np.float32)
    links = _weak_links(3 + L, B, L, TR, ol, 4.0)
    m, k, o, t = to_dev(match, links, ol, tl)
    m.requires_grad_()
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    assert _lib.last_launch_status() == 0 and torch.isfinite(loss).all()
    assert not _lib.last_dense_gave_up() and _lib.last_fallback_count() == 0
    alpha, beta = alpha.detach(), beta.detach()
    for bs in (0, B - 1):                                  # a full-length utterance and the ragged one
        sl = slice(bs, bs + 1)
        a64 = orc.dag_alpha(match[sl], links[sl], ol[sl], tl[sl], np.float64)[0]; b64 = orc.dag_beta(match[sl], links[sl], ol[sl], tl[sl], np.float64)[0]
        a, b = alpha[bs].cpu().numpy(), beta[bs].cpu().numpy()
        assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64))
        fa, fb = np.isfinite(a64), np.isfinite(b64)
        np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T + 1e-4)
        np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T + 1e-4)


@pytest.mark.parametrize("shape", [(4, 40, 1024, 32), (3, 33, 516, 20), (2, 50, 260, 7), (5, 70, 2048, 32)])
def test_fused_backward_equals_the_two_launches(shape):
    """r06: grad_match + grad_links of a banded graph in ONE launch (k5_fuse 1 / 2; auto = 2) are bit-identical to K4 followed by K5
    (k5_fuse 3) and match the fp64 oracle — ragged target lengths (the streamed tail rows T_b-1 .. T-1), a sample with an unreachable end
    (every row through the tail, zero gradients) and a -inf emission row included."""
    from daspeech_amd import _lib
    import ctypes
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(170 + L, B, T, L, TR)
    tl[0] = max(2, T // 3)                                   # long tail of padding rows
    if (tl[0] - 1) * TR < ol[0] - 1:
        ol[0] = (tl[0] - 1) * TR + 1                         # keep the end reachable; transitions past the shorter graph do not exist
    ii = np.arange(L)[:, None]; dd = np.arange(TR)[None, :]
    links[0] = np.where(ii + dd + 1 < ol[0], links[0], -np.inf)
    match[1 % B, 3, :] = -np.inf; match[1 % B, 3, 2 * TR] = -0.5
    if B > 2:
        tl[2] = 2; ol[2] = L                                 # end unreachable: loss = -inf, all gradients zero
    m, k, o, t = to_dev(match, links, ol, tl)
    got = {}
    diag = (ctypes.c_uint * 4)()
    try:
        for fuse in (3, 1, 2, 0):
            _lib.set_option("k5_fuse", fuse)
            _lib.load().dsp_dag_debug_k5(diag)
            mm = m.clone().requires_grad_(); kk = k.clone().requires_grad_()
            loss = ops().dag_loss(mm, kk, o, t)
            fin = torch.isfinite(loss)
            w = torch.linspace(0.5, 1.5, B, device=dev())
            gm, gl = torch.autograd.grad((loss.nan_to_num(neginf=0.0) * w).sum(), [mm, kk])
            torch.cuda.synchronize()
            _lib.load().dsp_dag_debug_k5(diag)
            assert diag[3] == {3: 2, 1: 4, 2: 5, 0: 5}[fuse]
            got[fuse] = (gm, gl)
    finally:
        _lib.set_option("k5_fuse", 0)
    for fuse in (1, 2, 0):
        assert torch.equal(got[fuse][0], got[3][0]) and torch.equal(got[fuse][1], got[3][1]), f"k5_fuse {fuse}"
    assert not torch.isnan(got[0][0]).any() and not torch.isnan(got[0][1]).any()
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    go = (w.cpu().numpy() * fin.cpu().numpy()).astype(np.float64)
    gm64, gl64 = orc.dag_grad(go, a64, b64, match, links, ol, tl, np.float64)
    np.testing.assert_allclose(got[0][0].cpu().numpy(), gm64, rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(got[0][1].cpu().numpy(), gl64, rtol=2e-3, atol=1e-7)
    if B > 2:
        assert not fin[2] and got[0][0][2].abs().max() == 0 and got[0][1][2].abs().max() == 0


@pytest.mark.parametrize("shape", [(3, 24, 1023, 32), (2, 30, 518, 32), (2, 17, 261, 9), (3, 12, 97, 32), (1, 2, 5, 3), (2, 40, 1030, 20),
                                   (2, 30, 517, 48), (2, 20, 1031, 100), (3, 12, 261, 64), (2, 9, 133, 128), (2, 16, 774, 33)])     # (windows 33 .. 128: r06)
@pytest.mark.parametrize("source", ["gather", "dense"])
def test_graph_lengths_off_the_16_byte_grid_run_pitched(shape, source):
    """r06: graph lengths that are not multiples of 4 (three real graphs in four) reach the strip kernels (windows <= 128) through ROW PITCHES
    (dsp_dag_loss_fwd_ld / _bwd_ld / dsp_dag_best_alignment_ld), not through F.pad copies: `match` straight from
    dag_logsoftmax_gather_inplace (written with the pitch) or a dense caller tensor (one copy into a pitched buffer).  Loss, alpha, beta, both
    gradients against the fp64 oracle; the Viterbi path bit-exact; the fast kernel families are the ones that ran."""
    from daspeech_amd import _lib
    import ctypes
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(230 + L, B, T, L, TR)
    V = 40
    rng = np.random.default_rng(L)
    m, k, o, t = to_dev(match, links, ol, tl)
    if source == "gather":
        # logits whose log-softmax at the target tokens is the oracle's match: build them from match directly
        tgt = rng.integers(0, V, (B, T))
        logits = rng.standard_normal((B, L, V)).astype(np.float32)
        x = torch.from_numpy(logits).to(dev())
        _, sel = ops().dag_logsoftmax_gather_inplace(x.clone(), torch.from_numpy(tgt).to(dev()).unsqueeze(1).expand(-1, L, -1))
        mt = sel.transpose(1, 2)                                   # [B,T,L] view of the pitched buffer
        assert mt.stride(2) == 1 and mt.stride(1) % 4 == 0 and (L % 4 == 0 or not mt.is_contiguous())
        match = mt.cpu().numpy().astype(np.float32)
        m = mt.detach()
    m = m.clone().requires_grad_() if source == "dense" else m.requires_grad_()
    k.requires_grad_()
    diag = (ctypes.c_uint * 4)()
    _lib.load().dsp_dag_debug_k5(diag)
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    assert alpha.is_contiguous() and alpha.shape == (B, T, L)
    fin = torch.isfinite(loss)
    w = torch.linspace(0.5, 1.5, B, device=dev())
    gm, gl = torch.autograd.grad((loss.nan_to_num(neginf=0.0) * w).sum(), [m, k])
    torch.cuda.synchronize()
    _lib.load().dsp_dag_debug_k5(diag)
    assert diag[3] == 5 or (TR > 32 and diag[3] in (1, 6)), "the fused exp-space gradient kernel ran (pitched rows), not the log-space fallback"
    assert gm.shape == (B, T, L) and gl.shape == (B, L, TR)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
    b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    a = alpha.cpu().numpy(); b = beta.cpu().numpy()
    assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64))
    fa = np.isfinite(a64); fb = np.isfinite(b64)
    np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T)
    np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), b64[:, 0, 0], rtol=3e-6, atol=2e-5 * T)
    go = (w.cpu().numpy() * fin.cpu().numpy()).astype(np.float64)
    gm64, gl64 = orc.dag_grad(go, a64, b64, match, links, ol, tl, np.float64)
    np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(gl.cpu().numpy(), gl64, rtol=2e-3, atol=1e-7)
    path = ops().dag_best_alignment(m.detach(), k.detach(), o, t).cpu().numpy()
    np.testing.assert_array_equal(path, orc.dag_best_alignment(match, links, ol, tl, np.float32))
    assert _lib.last_launch_status() == 0


@pytest.mark.parametrize("shape", [(3, 30, 1031, 64), (2, 40, 2048, 33), (4, 20, 513, 48), (2, 25, 700, 63), (40, 9, 512, 64), (2, 12, 70, 40),
                                   (3, 70, 1537, 64),
                                   (3, 20, 1031, 128), (2, 30, 2048, 65), (4, 16, 513, 96), (2, 25, 700, 127), (70, 6, 256, 128), (2, 12, 140, 100),
                                   (3, 40, 1537, 128)])
def test_windows_33_to_64_exp_space_strips(shape):
    """r06: windows 33 .. 64 run dag_dp_strip2g.hip (exp space, two vertices per lane, one exponent per lane pair) and windows 65 .. 128
    dag_dp_strip1g.hip (one vertex per lane, the window streamed through two register buffers) — dp_path 8 = the auto choice — instead of the
    log-space strips (dp_path 2, windows up to 64) / the dense-window kernels (dp_path 9) and the generic kernel (dp_path 1): alpha / beta /
    loss of all of them against the fp64 oracle, -inf patterns equal, ragged lengths, graph lengths off every grid, several strips per sample
    (hand-off of 64 / 128 boundary columns), more samples than CUs' worth of strips, a forced-emission row."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(300 + L, B, T, L, TR)
    match[0, min(3, T - 1), :] = -np.inf; match[0, min(3, T - 1), min(min(3, T - 1) * (TR // 2), L // 2) if T > 3 else 0] = -0.25        # forced vertex on one row
    m, k, o, t = to_dev(match, links, ol, tl)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    fa, fb = np.isfinite(a64), np.isfinite(b64)
    try:
        for path in ((0, 8, 2, 1) if TR <= 64 else ((0, 8, 9, 1) if L >= 128 else (0, 8, 1))):
            _lib.set_option("dp_path", path)
            mm = m.clone().requires_grad_()
            loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(mm, k, o, t)
            assert _lib.last_launch_status() == 0, path
            a, b = alpha.cpu().numpy(), beta.cpu().numpy()
            assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64)), path
            np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=2e-5 * T, err_msg=str(path))
            np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=2e-5 * T, err_msg=str(path))
            ln = loss.detach().cpu().numpy(); want = b64[:, 0, 0]
            assert np.array_equal(np.isneginf(ln), np.isneginf(want))
            f = np.isfinite(want)
            np.testing.assert_allclose(ln[f], want[f], rtol=3e-6, atol=2e-5 * T)
            # alpha only (no gradient): the one-direction launch
            with torch.no_grad():
                l0 = ops().dag_loss(m, k, o, t).cpu().numpy()
            np.testing.assert_allclose(l0[f], want[f], rtol=3e-6, atol=2e-5 * T)
    finally:
        _lib.set_option("dp_path", 0)


@pytest.mark.parametrize("window", [64, 128, 80])
@pytest.mark.parametrize("slope,scale", [(2.0, 1.0), (12.0, 4.0), (40.0, 12.0)])
def test_windows_33_to_64_on_peaked_scores(slope, scale, window):
    """The 33 .. 64 exp-space strips where their shared exponents run out: emissions that fall `slope` nats per vertex away from the aligned
    position and transition logits stretched by `scale` (weights over 100 binades under their column's strongest) — the single-transition
    shortcut on the diagonal and the exact log-space path must give the fp64 oracle's tables."""
    from daspeech_amd import _lib
    B, T, L, TR = 2, 30, 1100, window
    match, links, ol, tl = make_dag_inputs(777, B, T, L, TR, ragged=True)
    jj = np.arange(L, dtype=np.float32)[None, None, :]
    centre = (np.arange(T, dtype=np.float32) * (L - 1) / (T - 1))[None, :, None]
    match = (match * 0.1 - slope * np.abs(jj - centre)).astype(np.float32)
    fin = np.isfinite(links)
    links = np.where(fin, links * scale, links)
    mx = np.max(np.where(fin, links, -1e30), -1, keepdims=True)
    ssum = np.where(fin, np.exp(links - mx), 0).sum(-1, keepdims=True)
    links = np.where(fin, links - mx - np.log(np.where(ssum > 0, ssum, 1)), links).astype(np.float32)
    m, k, o, t = to_dev(match, links, ol, tl)
    a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
    try:
        _lib.set_option("dp_path", 8)
        loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m.requires_grad_(), k, o, t)
        assert _lib.last_launch_status() == 0
    finally:
        _lib.set_option("dp_path", 0)
    a, b = alpha.cpu().numpy(), beta.cpu().numpy()
    assert np.array_equal(np.isneginf(a), np.isneginf(a64)) and np.array_equal(np.isneginf(b), np.isneginf(b64))
    fa, fb = np.isfinite(a64), np.isfinite(b64)
    np.testing.assert_allclose(a[fa], a64[fa], rtol=3e-6, atol=1e-3)
    np.testing.assert_allclose(b[fb], b64[fb], rtol=3e-6, atol=1e-3)


@pytest.mark.parametrize("shape", [(3, 30, 1031, 64), (2, 40, 2048, 33), (4, 20, 513, 48), (40, 9, 512, 64), (2, 12, 70, 40), (3, 70, 1537, 64),
                                   (3, 20, 1031, 128), (2, 30, 2048, 65), (4, 16, 513, 96), (70, 6, 256, 128), (2, 12, 140, 100), (3, 40, 1537, 128)])
def test_windows_33_to_128_alignment_without_a_trace_tensor(shape):
    """r06: dag_best_alignment for windows 33 .. 128 = values-only max-DP strips (2 x 64 / 1 x 128 transitions per lane) + a back-trace that
    recomputes the arg-max of the cells it visits (dag_dp_maxstripw.hip; the auto choice, dp_path 7) — bit-exact against the oracle's sequential
    max-DP + back-trace and against the kernels that served these windows before (dp_path 2: log-space strips + trace walk, 9: blocked max-plus),
    on scores quantised to half-integers so that exact ties occur (smallest predecessor index wins), ragged lengths, several strips per sample,
    an unreachable sample (path = -1 beyond what the chain visits)."""
    from daspeech_amd import _lib
    B, T, L, TR = shape
    match, links, ol, tl = make_dag_inputs(400 + L, B, T, L, TR)
    match = (np.round(match * 2) / 2).astype(np.float32)
    links = np.where(np.isfinite(links), np.round(links * 2) / 2, links).astype(np.float32)
    if B > 2 and L > 2 * TR + 4:
        tl = tl.copy(); tl[2] = 2                              # two rows cannot span the graph: the end is unreachable
    m, k, o, t = to_dev(match, links, ol, tl)
    ref = orc.dag_best_alignment(match, links, ol, tl, np.float32)
    assert lib_trace_optional(L, TR), "windows 33 .. 128 take no trace tensor"
    try:
        for path in (0, 7) + ((2,) if TR <= 64 else ((9,) if L >= 128 else ())) + (1,):
            _lib.set_option("dp_path", path)
            got = ops().dag_best_alignment(m, k, o, t).cpu().numpy()
            assert _lib.last_launch_status() == 0, path
            np.testing.assert_array_equal(got, ref, err_msg=f"dp_path {path}")
    finally:
        _lib.set_option("dp_path", 0)


def lib_trace_optional(L, TR):
    from daspeech_amd import _lib
    return bool(_lib.load().dsp_dag_alignment_trace_optional(L, TR))
