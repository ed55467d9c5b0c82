"""GPU parity of the inference-side ops (graph decode, posterior, variance-adaptor glue, length regulator) against the CPU
oracle (oracle/dag_oracle.c, functions citing the reference lines) — integer / copy outputs bit-exact."""
import numpy as np
import pytest
import torch

from oracle import dag_oracle as orc
from tests.util_inputs import make_dag_inputs

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def D():
    from daspeech_amd import decode_ops
    return decode_ops


@pytest.mark.parametrize("dtype,V", [(torch.float32, 512), (torch.float16, 97), (torch.bfloat16, 1000)])
def test_argmax_logp(dtype, V):
    rng = np.random.default_rng(V)
    x = torch.from_numpy((rng.standard_normal((3, 41, V)) * 3).astype(np.float32)).to(dtype)
    x[0, 0, 5] = x[0, 0, 9] = x[0, 0].max() + 1          # exact tie -> first index wins
    tok_ref, sc_ref = orc.argmax_logp(x.float().numpy())
    tok, sc = D().argmax_logp(x.to(dev()))
    np.testing.assert_array_equal(tok.cpu().numpy(), tok_ref)
    np.testing.assert_allclose(sc.cpu().numpy(), sc_ref, rtol=2e-6, atol=2e-6)
    assert tok[0, 0].item() == 5


@pytest.mark.parametrize("shape", [(3, 60, 8), (2, 33, 32), (2, 20, 19), (3, 200, 199), (2, 130, 100)])   # TR > 64: one wave per vertex
@pytest.mark.parametrize("greedy", [False, True])
def test_lookahead_next_bit_exact(shape, greedy):
    B, L, TR = shape
    _, links, ol, _ = make_dag_inputs(5 + L, B, 4, L, TR)
    rng = np.random.default_rng(L)
    sc = (-rng.random((B, L)) * 3).astype(np.float32)
    # quantise to provoke ties
    links = np.where(np.isfinite(links), np.round(links * 4) / 4, links).astype(np.float32)
    sc = np.round(sc * 4) / 4
    ref = orc.lookahead_next(links, sc, beta=0.75, greedy=greedy)
    out = D().lookahead_next(torch.from_numpy(links).to(dev()), torch.from_numpy(sc).to(dev()), 0.75, greedy)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # the dense formulation of the reference gives the same indices (s2s_conformer_dag_fastspeech2.py:214/217)
    dense = torch.from_numpy(orc.restore_valid_links(links))
    if greedy:
        idx = dense.max(dim=-1)[1]
    else:
        idx = (dense + torch.from_numpy(sc).unsqueeze(1) * 0.75).max(dim=-1)[1]
    np.testing.assert_array_equal(out.cpu().numpy(), idx.numpy())


def test_graph_decode_matches_reference_loop():
    B, L, TR, V, Dm, pad = 4, 48, 6, 30, 16, 1
    rng = np.random.default_rng(3)
    _, links, ol, _ = make_dag_inputs(3, B, 4, L, TR)
    logits = (rng.standard_normal((B, L, V)) * 2).astype(np.float32)
    logits[:, ::5, pad] += 20.0                                   # some vertices emit <pad>: must be dropped
    logits[:, 1::7, 7] += 20.0; logits[:, 2::7, 7] += 20.0        # repeated tokens: collapsed
    feats = rng.standard_normal((B, L, Dm)).astype(np.float32)
    out_tok, out_feat, mask, lens = D().graph_decode(torch.from_numpy(logits).to(dev()), torch.from_numpy(links).to(dev()),
                                                     torch.from_numpy(feats).to(dev()), torch.from_numpy(ol).to(dev()), pad, 1.0)
    # restatement of the reference's host loop (s2s_conformer_dag_fastspeech2.py:219-243) on oracle tok/next
    tok, sc = orc.argmax_logp(logits)
    nxt = orc.lookahead_next(links, sc, 1.0)
    toks_ref, keep_ref, nf_ref = orc.follow_path(nxt, tok, ol, pad)
    for b in range(B):
        last = tok[b, 0]; j = 0; res = [last]; kept = []
        while j != ol[b] - 1:
            j = nxt[b, j]; now = tok[b, j]
            if now != pad and now != last:
                res.append(now); kept.append(j)
            last = now
        assert nf_ref[b] == len(kept) and lens[b].item() == len(kept)
        got = out_tok[b].cpu().numpy()
        np.testing.assert_array_equal(got[: len(res)], np.array(res))
        assert np.all(got[len(res):] == pad)
        np.testing.assert_array_equal(out_feat[b, : len(kept)].cpu().numpy(), feats[b, kept])       # bit-exact gather
        assert np.all(out_feat[b, len(kept):].cpu().numpy() == 0)
        assert mask[b].cpu().numpy().tolist() == [False] * len(kept) + [True] * (out_feat.shape[1] - len(kept))


def test_posterior_and_expect():
    B, T, L, TR, Dm = 3, 9, 70, 16, 32
    match, links, ol, tl = make_dag_inputs(17, B, T, L, TR)
    a = orc.dag_alpha(match, links, ol, tl, np.float32)
    b = orc.dag_beta(match, links, ol, tl, np.float32)
    feats = np.random.default_rng(0).standard_normal((B, L, Dm)).astype(np.float32)
    score_ref, ex_ref = orc.posterior_expect(a, b, feats)
    score = D().posterior(torch.from_numpy(a).to(dev()), torch.from_numpy(b).to(dev()))
    np.testing.assert_allclose(score.cpu().numpy(), score_ref, rtol=1e-4, atol=1e-6)
    rows = score.sum(-1).cpu().numpy()
    for bb in range(B):
        np.testing.assert_allclose(rows[bb, : tl[bb]], 1.0, rtol=1e-4)
        assert np.all(rows[bb, tl[bb]:] == 0)                      # all -inf rows: NaN -> 0
    ex = D().expect_features(torch.from_numpy(a).to(dev()), torch.from_numpy(b).to(dev()), torch.from_numpy(feats).to(dev()))
    np.testing.assert_allclose(ex.cpu().numpy(), ex_ref[:, 1:], rtol=1e-3, atol=1e-4)


def test_posterior_features_fused_forward_and_backward():
    """dsp_posterior_features / _bwd (no [B,T,L] score tensor) vs the oracle's posterior + matmul and torch autograd of the two-step form:
    ragged lengths (rows past T_b are all -inf -> zero output rows), a feature width that needs two passes, gradient to the features."""
    B, T, L, TR, Dm = 3, 21, 70, 16, 640
    match, links, ol, tl = make_dag_inputs(23, B, T, L, TR)
    a = orc.dag_alpha(match, links, ol, tl, np.float32)
    b = orc.dag_beta(match, links, ol, tl, np.float32)
    feats = np.random.default_rng(1).standard_normal((B, L, Dm)).astype(np.float32)
    _, ex_ref = orc.posterior_expect(a, b, feats)
    ta, tb = torch.from_numpy(a).to(dev()), torch.from_numpy(b).to(dev())
    f1 = torch.from_numpy(feats).to(dev()).requires_grad_()
    out = D().posterior_features(ta, tb, f1)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ex_ref, rtol=1e-4, atol=1e-5)
    for bb in range(B):
        assert np.all(out[bb, tl[bb]:].detach().cpu().numpy() == 0)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    f2 = torch.from_numpy(feats).to(dev()).requires_grad_()
    (torch.matmul(D().posterior(ta, tb), f2) * w).sum().backward()
    torch.testing.assert_close(f1.grad, f2.grad, rtol=1e-4, atol=1e-5)


def test_durations_and_bucketize():
    rng = np.random.default_rng(1)
    ld = (rng.standard_normal((4, 37)) * 1.2 + 1.0).astype(np.float32)
    ld[0, :4] = np.log(np.array([1.5, 2.5, 3.5, 0.2], np.float32))       # (exp-1) = .5, 1.5, 2.5: half-to-even cases
    pm = rng.random((4, 37)) < 0.2
    ref = orc.durations(ld, pm, 1.0)
    out = D().predicted_durations(torch.from_numpy(ld).to(dev()), torch.from_numpy(pm).to(dev()), 1.0)
    # expf on device vs libm can differ in the last ulp exactly on a rounding boundary: allow |diff| <= 1 on < 0.5 % of entries
    diff = np.abs(out.cpu().numpy() - ref)
    assert diff.max() <= 1 and (diff > 0).mean() < 0.005
    assert np.all(out.cpu().numpy()[pm] == 0)
    t = torch.clamp(torch.round((torch.exp(torch.from_numpy(ld)) - 1) * 1.0).long(), min=0).masked_fill(torch.from_numpy(pm), 0)
    assert (np.abs(t.numpy() - out.cpu().numpy()) > 0).mean() < 0.005
    # bucketize + embedding add (pitch / energy path)
    C, nb = 24, 255
    bins = np.linspace(-3.0, 3.0, nb).astype(np.float32)
    v = (rng.standard_normal(4 * 37) * 2).astype(np.float32)
    v[:3] = bins[[0, 100, 254]]                                     # exactly on an edge: right=False -> that index
    emb = rng.standard_normal((nb + 1, C)).astype(np.float32)
    x = rng.standard_normal((4 * 37, C)).astype(np.float32)
    idx_ref = orc.bucketize(v, bins)
    np.testing.assert_array_equal(idx_ref, torch.bucketize(torch.from_numpy(v), torch.from_numpy(bins)).numpy())
    out = D().bucketize_embed_add(torch.from_numpy(x).to(dev()), torch.from_numpy(v).to(dev()), torch.from_numpy(bins).to(dev()),
                                  torch.from_numpy(emb).to(dev()))
    np.testing.assert_array_equal(out.cpu().numpy(), x + emb[idx_ref])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 3, 2), (5, 61, 256), (3, 300, 80), (1, 1, 7)])
def test_length_regulator_bit_exact(shape, dtype):
    B, N, C = shape
    rng = np.random.default_rng(N)
    x = rng.standard_normal((B, N, C)).astype(np.float32)
    dur = rng.poisson(3.0, (B, N)).astype(np.int64)
    dur[:, ::4] = 0
    if shape == (2, 3, 2):
        x = np.arange(12, dtype=np.float32).reshape(2, 3, 2) + 1
        dur = np.array([[2, 0, 1], [1, 1, 0]])                       # SURVEY.md §9.3 example
    xt = torch.from_numpy(x).to(dtype)
    ref, lens_ref = orc.length_regulate(xt.float().numpy(), dur)
    out, lens = D().length_regulate(xt.to(dev()), torch.from_numpy(dur).to(dev()))
    np.testing.assert_array_equal(lens.cpu().numpy(), lens_ref)
    np.testing.assert_array_equal(out.float().cpu().numpy(), ref)
    if shape == (2, 3, 2):
        assert lens.tolist() == [3, 2]


def test_length_regulator_all_zero_durations():
    x = torch.randn(2, 5, 8, device=dev())
    out, lens = D().length_regulate(x, torch.zeros(2, 5, dtype=torch.long, device=dev()))
    assert out.shape == (2, 0, 8) and lens.tolist() == [0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("TRmax", [32, 99999, 7])
def test_fused_extract_links_matches_torch_formulation(TRmax):
    """csrc/extract_links.hip (band only, fused) vs the torch restatement of s2t_conformer_dag.py:171-212 in
    DAGDecoder.extract_links (full [B,L,L,H] content + gather), ragged graph sizes, banded and dense windows."""
    from daspeech_amd.models.daspeech import DAGDecoder, DEFAULT_ARGS, PAD, BOS, EOS, UNK
    from types import SimpleNamespace
    torch.manual_seed(3)
    dev = torch.device("cuda")
    a = SimpleNamespace(**{**DEFAULT_ARGS, "max_transition_length": TRmax})
    dec = DAGDecoder(a).to(dev).eval()
    B, L = 3, 70
    lens = [70, 51, 2]
    prev = torch.full((B, L), PAD, dtype=torch.long, device=dev)
    for b, n in enumerate(lens):
        prev[b, :n] = UNK; prev[b, 0] = BOS; prev[b, n - 1] = EOS
    feats = torch.randn(B, L, a.decoder_embed_dim, device=dev)
    with torch.no_grad():
        dec.fused_links = True
        got = dec.extract_links(feats, prev)
        dec.fused_links = False
        want = dec.extract_links(feats, prev)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert torch.equal(torch.isneginf(got), torch.isneginf(want))
    fin = torch.isfinite(want)
    torch.testing.assert_close(got[fin], want[fin], rtol=1e-5, atol=2e-5)
    # every row with a successor is a distribution over its valid transitions
    rows = fin.any(-1)
    torch.testing.assert_close(torch.logsumexp(got[rows], -1), torch.zeros_like(got[rows][:, 0]), rtol=0, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("TRmax,heads_dim", [(32, 512), (99999, 512), (7, 512), (99999, 256), (20, 1024)])
def test_fused_extract_links_backward_matches_torch_formulation(TRmax, heads_dim):
    """dsp_extract_links_train / dsp_extract_links_bwd (compact band, scores recomputed per tile, no [B,L,L,H] tensor) under autograd vs
    the torch restatement of s2t_conformer_dag.py:171-212: gradients w.r.t. q, k and the gate logits — and through them every parameter
    of the links head — on ragged graphs, banded and dense (TR = L-1) windows, head widths 32 / 64 / 128, with a loss that weights the
    links unevenly and ignores the -inf entries (as dag_loss's gradient does)."""
    from daspeech_amd.models.daspeech import DAGDecoder, DEFAULT_ARGS, PAD, BOS, EOS, UNK
    from types import SimpleNamespace
    torch.manual_seed(11)
    dev = torch.device("cuda")
    a = SimpleNamespace(**{**DEFAULT_ARGS, "max_transition_length": TRmax, "decoder_embed_dim": heads_dim, "decoder_layers": 0})
    dec = DAGDecoder(a).to(dev).train()
    B, L = 3, 70
    lens = [70, 51, 2]
    prev = torch.full((B, L), PAD, dtype=torch.long, device=dev)
    for b, n in enumerate(lens):
        prev[b, :n] = UNK; prev[b, 0] = BOS; prev[b, n - 1] = EOS
    feats0 = torch.randn(B, L, a.decoder_embed_dim, device=dev)
    wgt = None
    res = {}
    for fused in (True, False):
        dec.fused_links = fused
        dec.zero_grad(set_to_none=True)
        feats = feats0.clone().requires_grad_()
        links = dec.extract_links(feats, prev)
        if wgt is None:
            wgt = torch.randn_like(links)
        fin = torch.isfinite(links)
        loss = (links.masked_fill(~fin, 0.0) * wgt).sum() + 0.3 * torch.logsumexp(links.masked_fill(~fin, -1e4), -1).sum()
        loss.backward()
        res[fused] = (links.detach(), feats.grad.detach(), {n: p.grad.detach().clone() for n, p in dec.named_parameters() if p.grad is not None})
    (l1, g1, p1), (l0, g0, p0) = res[True], res[False]
    assert torch.equal(torch.isneginf(l1), torch.isneginf(l0))
    f = torch.isfinite(l0)
    torch.testing.assert_close(l1[f], l0[f], rtol=1e-5, atol=2e-5)
    scale = max(1.0, float(g0.abs().max()))
    assert float((g1 - g0).abs().max()) <= 1e-5 * scale + 2e-5, float((g1 - g0).abs().max())
    assert set(p1) == set(p0) and {"query_linear.weight", "key_linear.weight", "gate_linear.weight"} <= set(p1)
    for n in p0:
        sc = max(1.0, float(p0[n].abs().max()))
        assert float((p1[n] - p0[n]).abs().max()) <= 2e-5 * sc + 2e-5, (n, float((p1[n] - p0[n]).abs().max()), sc)


@pytest.mark.gpu
def test_fused_extract_links_backward_direct_q_k_gates():
    """The autograd function itself: gradients w.r.t. q, k, log_gates against autograd through the torch band formulation (<= 1e-5)."""
    from daspeech_amd import decode_ops
    torch.manual_seed(2)
    B, L, H, CK, TR = 2, 45, 8, 64, 44
    olen = torch.tensor([45, 30], device="cuda")
    q0 = torch.randn(B, L, H, CK, device="cuda") * 0.5; k0 = torch.randn(B, L, H, CK, device="cuda") * 0.5
    g0 = torch.log_softmax(torch.randn(B, L, H, device="cuda"), -1)

    def torch_links(q, k, lg):
        content = torch.einsum("bicf,bjcf->bijc", q, k) / (CK ** 0.5)
        idx = torch.arange(L, device="cuda").unsqueeze(1) + torch.arange(TR, device="cuda").unsqueeze(0) + 1
        invalid = idx.unsqueeze(0) >= olen.view(B, 1, 1)
        band = content.gather(2, idx.unsqueeze(0).masked_fill(invalid, 0).unsqueeze(-1).expand(-1, -1, -1, H))
        nouse = invalid.all(-1)
        band = band.masked_fill(invalid.unsqueeze(-1), float("-inf")).masked_fill(nouse.view(B, L, 1, 1), 0.0)
        band = torch.log_softmax(band, 2).masked_fill(invalid.unsqueeze(-1), -1e30)
        out = torch.logsumexp(band + lg.unsqueeze(2), -1)
        return out.masked_fill(invalid, float("-inf"))

    w = torch.randn(B, L, TR, device="cuda")
    grads = []
    for fn in (lambda q, k, lg: decode_ops.extract_links_autograd(q, k, lg, olen, TR), torch_links):
        q, k, lg = q0.clone().requires_grad_(), k0.clone().requires_grad_(), g0.clone().requires_grad_()
        links = fn(q, k, lg)
        fin = torch.isfinite(links)
        (links.masked_fill(~fin, 0.0) * w).sum().backward()
        grads.append((links.detach(), q.grad, k.grad, lg.grad))
    (la, qa, ka, ga), (lb, qb, kb, gb) = grads
    f = torch.isfinite(lb)
    torch.testing.assert_close(la[f], lb[f], rtol=1e-5, atol=2e-5)
    for x, y, n in ((qa, qb, "q"), (ka, kb, "k"), (ga, gb, "log_gates")):
        assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max())), (n, float((x - y).abs().max()))


@pytest.mark.parametrize("joint", [True, False])
@pytest.mark.parametrize("shape", [(3, 40, 6), (4, 96, 95), (2, 260, 259), (5, 128, 32), (2, 1000, 999)])
def test_viterbi_decode_hip_matches_reference_loop_restatement(shape, joint):
    """viterbi / jointviterbi decode on the HIP max-DP (dsp_dag_max_alpha + dsp_dag_backtrace) vs the torch restatement of the
    reference loop (s2s_conformer_dag_fastspeech2.py:244-304; itself pinned to the per-sample Python loop in
    tests/test_torch_variants.py): identical tokens, lengths, masks and gathered features, with <pad> emissions, repeated
    tokens, ragged graph sizes, banded and full transition windows, and quantised scores that force ties."""
    from daspeech_amd import decode_ops
    B, L, TR = shape
    V, D, pad = 13, 8, 1
    g = torch.Generator().manual_seed(17 + L + TR)
    logits = torch.randn(B, L, V, generator=g) * 2
    logits[:, ::4, pad] += 6
    logits[:, 1::5] = logits[:, 2::5][:, : logits[:, 1::5].shape[1]] if L >= 10 else logits[:, 1::5]     # repeated argmax tokens
    raw = torch.round(torch.randn(B, L, TR, generator=g) * 4) / 4                                        # ties
    out_len = torch.randint(max(3, L - 9), L + 1, (B,), generator=g); out_len[0] = L
    i = torch.arange(L).view(1, L, 1); d = torch.arange(TR).view(1, 1, TR)
    valid = (i + d + 1) < out_len.view(B, 1, 1)
    links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1)
    links = links.masked_fill(~valid, float("-inf"))
    feats = torch.randn(B, L, D, generator=g)
    dev = "cuda"
    for beta, vb in ((1.0, 1.0), (0.5, 1.3)):
        got = decode_ops.viterbi_decode(logits.to(dev), links.to(dev), feats.to(dev), out_len.to(dev), pad, beta, vb, joint, 0.5)
        ref = decode_ops.viterbi_decode_torch(logits.to(dev), links.to(dev), feats.to(dev), out_len.to(dev), pad, beta, vb, joint, 0.5)
        assert torch.equal(got[3], ref[3]), (got[3], ref[3])
        assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]) and torch.equal(got[1], ref[1])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 30, 200, 199), (2, 52, 390, 64), (4, 12, 128, 127), (2, 130, 1030, 1029)])
def test_two_half_alignment_on_the_dense_kernels_equals_the_trace_form(shape):
    """dsp_dag_max_alpha_blocks / dsp_dag_backtrace_blocks (blocked max-plus DP + 2-byte block trace; what Viterbi decode runs for windows wider
    than 32) against dsp_dag_max_alpha / dsp_dag_backtrace (row-sequential DP + 4-byte arg-max trace): alpha_max bit-identical on every cell,
    and the same path from EVERY start row (the decode picks the row after the DP), ragged graphs, quantised weights that force ties."""
    import ctypes
    from daspeech_amd import _lib
    B, T, L, TR = shape
    g = torch.Generator().manual_seed(5 + L)
    match = (torch.round(torch.randn(B, T, L, generator=g) * 4) / 4).cuda()
    out_len = torch.randint(max(T + 2, L - 40), L + 1, (B,), generator=g); out_len[0] = L
    raw = torch.round(torch.randn(B, L, TR, generator=g) * 4) / 4
    i = torch.arange(L).view(1, L, 1); d = torch.arange(TR).view(1, 1, TR)
    valid = (i + d + 1) < out_len.view(B, 1, 1)
    links = raw.masked_fill(~valid, float("-inf")).cuda().contiguous()
    olen = out_len.cuda()
    rows = torch.full((B,), T, dtype=torch.int64, device="cuda")
    lib = _lib.load()
    assert lib.dsp_dag_max_alpha_blocks_supported(L, TR) == 1 and lib.dsp_dag_max_alpha_blocks_supported(L, 32) == 0
    st = _lib.current_stream_handle()
    a0 = torch.empty(B, T, L, device="cuda"); tr0 = torch.empty(B, T, L, dtype=torch.int32, device="cuda")
    a1 = torch.empty(B, T, L, device="cuda"); tr1 = torch.empty(B, T, L, dtype=torch.int16, device="cuda")
    _lib.check(lib.dsp_dag_max_alpha(_lib.ptr(match), _lib.ptr(links), _lib.ptr(olen), _lib.ptr(rows), _lib.ptr(a0), _lib.ptr(tr0), B, T, L, TR, st), "max_alpha")
    _lib.check(lib.dsp_dag_max_alpha_blocks(_lib.ptr(match), _lib.ptr(links), _lib.ptr(olen), _lib.ptr(rows), _lib.ptr(a1), _lib.ptr(tr1), B, T, L, TR, st), "max_alpha_blocks")
    assert torch.equal(a0, a1)
    for start in (T, T - 1, max(2, T // 2), 2):
        sr = torch.full((B,), start, dtype=torch.int64, device="cuda"); sr[-1] = max(2, start - 1)
        p0 = torch.empty(B, L, dtype=torch.int64, device="cuda"); p1 = torch.empty_like(p0)
        _lib.check(lib.dsp_dag_backtrace(_lib.ptr(tr0), _lib.ptr(olen), _lib.ptr(sr), _lib.ptr(p0), B, T, L, st), "backtrace")
        _lib.check(lib.dsp_dag_backtrace_blocks(_lib.ptr(a1), _lib.ptr(tr1), _lib.ptr(links), _lib.ptr(olen), _lib.ptr(sr), _lib.ptr(p1), B, T, L, TR, st), "backtrace_blocks")
        reach = torch.isfinite(a0[torch.arange(B), sr - 1, olen - 1])
        assert reach.any() or start < T            # (a short walk cannot cross a banded graph: nothing to compare for that start row)
        assert torch.equal(p0[reach], p1[reach]), start


@pytest.mark.parametrize("shape", [(3, 77, 64, 31), (2, 5, 256, 31), (1, 200, 8, 3), (4, 33, 128, 15), (2, 64, 32, 7)])
def test_dwconv_bn_silu_matches_torch_module_chain(shape):
    """dsp_dwconv_bn_silu (channels-last, one pass) vs the torch chain it replaces in the Conformer convolution module in eval mode:
    transpose -> Conv1d(C, C, K, groups=C, bias=False) -> BatchNorm1d.eval() -> SiLU -> transpose; fp32, tolerance 1e-5 (different
    summation order; MIOpen's naive kernel accumulates in double)."""
    from daspeech_amd import decode_ops
    B, T, C, K = shape
    torch.manual_seed(5 + T)
    dw = torch.nn.Conv1d(C, C, K, padding=(K - 1) // 2, groups=C, bias=False).cuda()
    bn = torch.nn.BatchNorm1d(C).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.3, 2.0)
    x = torch.randn(B, T, C, device="cuda")
    with torch.no_grad():
        want = torch.nn.functional.silu(bn(dw(x.transpose(1, 2)))).transpose(1, 2)
        got = decode_ops.dwconv_bn_silu(x, dw.weight, bn)
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 77, 256, 1024, 9), (3, 130, 1024, 256, 9), (1, 5, 128, 64, 3), (2, 64, 512, 256, 1), (2, 200, 256, 256, 3)])
def test_split_precision_conv1d_is_fp32_accurate(shape):
    """dsp_conv1d_split (3 x fp16 MFMA, operand splitting) against an fp64 reference of the same Conv1d: its error must be of the order
    of an fp32 convolution's own rounding error (measured beside it: torch's fp32 conv1d vs fp64), far inside the 1e-4 mel tolerance —
    including values across 12 binades (the lo parts must not fall into the fp16 denormals), ReLU, bias, 512-channel input slices."""
    from daspeech_amd.decode_ops import SplitConv1d
    B, T, Cin, Cout, K = shape
    torch.manual_seed(11 + T)
    conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) // 2).cuda()
    x = torch.randn(B, T, Cin, device="cuda") * torch.exp2(torch.randint(-8, 5, (B, T, 1), device="cuda").float())
    sc = SplitConv1d(conv.weight, conv.bias)
    for relu in (False, True):
        with torch.no_grad():
            got = sc(x, relu=relu)
            ref64 = torch.nn.functional.conv1d(x.double().transpose(1, 2), conv.weight.double(), conv.bias.double(), padding=(K - 1) // 2).transpose(1, 2)
            ref32 = conv(x.transpose(1, 2)).transpose(1, 2)
            if relu:
                ref64 = ref64.clamp_min(0); ref32 = ref32.clamp_min(0)
        scale = ref64.abs().max().item()
        err = (got.double() - ref64).abs().max().item() / scale
        err32 = (ref32.double() - ref64).abs().max().item() / scale
        assert got.shape == (B, T, Cout) and torch.isfinite(got).all()
        assert err < 4e-6 and err < 8 * err32 + 1e-6, (err, err32)
    # a channel slice of a wider tensor as input (row stride > Cin)
    if Cin <= 512:
        wide = torch.randn(B, T, Cin + 64, device="cuda")
        xs = wide[:, :, :Cin]
        with torch.no_grad():
            got = sc(xs)
            ref = conv(xs.transpose(1, 2).contiguous()).transpose(1, 2)
        torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(5, 77, 256), (3, 1, 512), (2, 33, 1024), (7, 3, 80), (1, 130, 2048), (4, 9, 36)])
def test_layer_norm_kernel_matches_torch(shape):
    """dsp_layer_norm (one wave per row, values in registers) vs torch.nn.LayerNorm in eval-mode fp32."""
    from daspeech_amd import decode_ops
    torch.manual_seed(3)
    ln = torch.nn.LayerNorm(shape[-1]).cuda().eval()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
        x = torch.randn(*shape, device="cuda") * 3 + 1.5
        got = decode_ops.layer_norm(x, ln); want = ln(x)
    torch.testing.assert_close(got, want, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("shape", [(3, 77, 4), (2, 200, 4), (1, 5, 2), (2, 256, 1), (4, 33, 8), (2, 300, 4)])
def test_fused_relpos_attention_matches_torch_formulation(shape):
    """dsp_relpos_attention vs the torch formulation of the Conformer's relative-position attention (two batched GEMMs, rel_shift,
    masked soft-max, value GEMM) on ragged batches: fp32, tolerance 2e-5 (different summation order)."""
    from daspeech_amd import decode_ops
    from daspeech_amd.models.daspeech import RelPosSelfAttention, rel_positional_encoding
    B, T, H = shape
    torch.manual_seed(9 + T)
    att = RelPosSelfAttention(H * 64, H).cuda().eval()
    with torch.no_grad():
        att.pos_bias_u.normal_(0, 0.5); att.pos_bias_v.normal_(0, 0.5)
    x = torch.randn(B, T, H * 64, device="cuda")
    lens = torch.randint(max(1, T // 2), T + 1, (B,), device="cuda"); lens[0] = T
    pad = torch.arange(T, device="cuda").unsqueeze(0) >= lens.unsqueeze(1)
    pos = rel_positional_encoding(T, H * 64, x.device, x.dtype)
    with torch.no_grad():
        old = decode_ops.set_split_gemm(True)
        try:
            got = att(x, pos, pad)
            decode_ops.set_split_gemm(False)
            want = att(x, pos, pad)
        finally:
            decode_ops.set_split_gemm(old)
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


def _links_case(B, L, CK, TR, lens, seed):
    torch.manual_seed(seed)
    H = 8
    olen = torch.tensor(lens, device="cuda")
    q0 = torch.randn(B, L, H, CK, device="cuda") * 0.5; k0 = torch.randn(B, L, H, CK, device="cuda") * 0.5
    g0 = torch.log_softmax(torch.randn(B, L, H, device="cuda"), -1)
    w = torch.randn(B, L, TR, device="cuda")
    bias = -0.02 * torch.arange(TR, device="cuda").float()
    return olen, q0, k0, g0, w, bias


def _links_fwd_bwd(olen, q0, k0, g0, w, TR, bias):
    from daspeech_amd import decode_ops
    q, k, lg = q0.clone().requires_grad_(), k0.clone().requires_grad_(), g0.clone().requires_grad_()
    links = decode_ops.extract_links_autograd(q, k, lg, olen, TR, bias)
    fin = torch.isfinite(links)
    (links.masked_fill(~fin, 0.0) * w).sum().backward()
    with torch.no_grad():
        inf_links = decode_ops.extract_links(q0, k0, g0, olen, TR, bias)
    return links.detach(), q.grad, k.grad, lg.grad, inf_links


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,CK,TR,lens,tile", [(2, 150, 64, 149, [150, 97], 32), (2, 150, 64, 149, [150, 97], 64), (3, 200, 32, 70, [200, 131, 3], 96),
                                                (2, 90, 128, 89, [90, 41], 32), (2, 300, 64, 299, [300, 1], 128)])
def test_tiled_extract_links_equal_the_one_image_kernels(B, L, CK, TR, lens, tile):
    """r05: dsp_extract_links / _train / _bwd with the window walked in tiles (forced through the xl_tile option on graphs the one-image kernels
    serve too): links, soft-max state and all three gradients against the one-image kernels — same arithmetic, different summation order of
    the soft-max denominators (online vs one pass): <= 2e-6 relative."""
    from daspeech_amd import _lib
    olen, q0, k0, g0, w, bias = _links_case(B, L, CK, TR, lens, 5)
    try:
        _lib.set_option("xl_mfma", 0)
        _lib.set_option("xl_tile", 0)
        _lib.load().dsp_extract_links_debug_ran()
        ref = _links_fwd_bwd(olen, q0, k0, g0, w, TR, bias)
        torch.cuda.synchronize()
        assert _lib.load().dsp_extract_links_debug_ran() == 0b0001001, "reference = the one-image kernels, forward and backward"
        _lib.set_option("xl_tile", tile)
        got = _links_fwd_bwd(olen, q0, k0, g0, w, TR, bias)
        torch.cuda.synchronize()
        assert _lib.load().dsp_extract_links_debug_ran() == 0b0010010, "the pinned tiled kernels ran, in the autograd backward too"
    finally:
        _lib.set_option("xl_tile", 0)
        _lib.set_option("xl_mfma", -1)
    for name, a, b in zip(("links", "dq", "dk", "dgate", "links (inference)"), got, ref):
        assert torch.equal(torch.isneginf(a), torch.isneginf(b)), name
        f = torch.isfinite(b)
        sc = max(1.0, float(b[f].abs().max()))
        assert float((a[f] - b[f]).abs().max()) <= 4e-6 * sc, (name, float((a[f] - b[f]).abs().max()), sc)


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,TR,lens,use_bias", [(2, 150, 149, [150, 97], True), (3, 200, 70, [200, 131, 3], False), (2, 64, 63, [64, 33], True),
                                                 (2, 65, 7, [65, 64], False), (3, 333, 32, [333, 2, 1], True), (2, 31, 30, [31, 17], False),
                                                 (2, 513, 512, [513, 400], False), (1, 700, 300, [650], True), (2, 96, 1, [96, 50], False)])
def test_matrix_core_extract_links_equal_the_fp32_kernels(B, L, TR, lens, use_bias):
    """r05: csrc/extract_links_mfma.hip (scores on the fp16 matrix cores with split operands, contractions on the fp32 matrix-core path; forced
    through the xl_mfma option on graphs of every size) against the fp32-FMA kernels of csrc/extract_links.hip: links, soft-max state and all
    three gradients, inference entry point included; ragged graphs, graph lengths that are not multiples of the 32 / 64-row tiles, narrow and
    dense windows, graphs of one and two vertices.  The split product is good to 2^-22 of sum |q_c k_c|: <= 1e-5 of the largest value."""
    from daspeech_amd import _lib
    olen, q0, k0, g0, w, bias = _links_case(B, L, 64, TR, lens, 9)
    if not use_bias:
        bias = None
    try:
        _lib.set_option("xl_mfma", 0)
        _lib.load().dsp_extract_links_debug_ran()
        ref = _links_fwd_bwd(olen, q0, k0, g0, w, TR, bias)
        torch.cuda.synchronize()
        assert _lib.load().dsp_extract_links_debug_ran() & 0b1100100 == 0, "reference = the fp32-FMA kernels"
        _lib.set_option("xl_mfma", 1)
        gots = []
        for contract in (0, 1):            # the backward's contractions: exact-fp32 MFMAs | bf16-triple products (the default above ~1 500 vertices)
            _lib.set_option("xl_contract", contract)
            gots.append(_links_fwd_bwd(olen, q0, k0, g0, w, TR, bias))
            torch.cuda.synchronize()
            assert _lib.load().dsp_extract_links_debug_ran() == (0b1000100 if contract else 0b0100100), "matrix-core forward + the pinned contraction, in the autograd backward too"
    finally:
        _lib.set_option("xl_mfma", -1)
        _lib.set_option("xl_contract", -1)
    for got in gots:
        for name, a, b in zip(("links", "dq", "dk", "dgate", "links (inference)"), got, ref):
            assert torch.equal(torch.isneginf(a), torch.isneginf(b)), name
            assert torch.isfinite(a[torch.isfinite(b)]).all(), name
            f = torch.isfinite(b)
            sc = max(1.0, float(b[f].abs().max()))
            assert float((a[f] - b[f]).abs().max()) <= 1e-5 * sc, (name, float((a[f] - b[f]).abs().max()), sc)


@pytest.mark.gpu
def test_matrix_core_extract_links_flags_operands_beyond_the_fp16_split():
    """r05 ADVICE: the matrix-core kernels split their operands into fp16 pieces; a k (or q * scale * log2 e) beyond +-65504 has no split.  Such
    an operand is clamped (finite output, no inf / NaN poisoning of the soft-max row) and dsp_extract_links_debug_range() reports it; in-range
    inputs leave the flag clear, a NaN operand stays a NaN and is not reported as a range overflow."""
    from daspeech_amd import _lib, decode_ops
    lib = _lib.load()
    B, L, TR = 2, 96, 95
    olen, q0, k0, g0, w, bias = _links_case(B, L, 64, TR, [96, 70], 3)
    try:
        _lib.set_option("xl_mfma", 1)
        lib.dsp_extract_links_debug_range()
        with torch.no_grad():
            ok = decode_ops.extract_links(q0, k0, g0, olen, TR, bias)
        torch.cuda.synchronize()
        assert lib.dsp_extract_links_debug_ran() & 0b100, "the matrix-core forward ran"
        assert lib.dsp_extract_links_debug_range() == 0
        k1 = k0.clone(); k1[0, 40, 3, 7] = 1.0e5
        with torch.no_grad():
            big = decode_ops.extract_links(q0, k1, g0, olen, TR, bias)
        torch.cuda.synchronize()
        assert lib.dsp_extract_links_debug_range() == 1 and lib.dsp_extract_links_debug_range() == 0       # reported once, then cleared
        assert not torch.isnan(big).any() and torch.equal(torch.isneginf(big), torch.isneginf(ok))
        k2 = k0.clone(); k2[1, 5, 0, 0] = float("nan")
        with torch.no_grad():
            decode_ops.extract_links(q0, k2, g0, olen, TR, bias)
        torch.cuda.synchronize()
        assert lib.dsp_extract_links_debug_range() == 0
    finally:
        _lib.set_option("xl_mfma", -1)


@pytest.mark.gpu
def test_matrix_core_extract_links_at_baseline_graph_size():
    """BASELINE's graph (L = 4096) with the README's dense window (TR = L-1), B = 8 ragged samples: the dispatch must pick the matrix-core kernels
    by itself; size-independent properties (every row with a successor is a distribution over its valid transitions; the -inf pattern is the
    band / graph mask) and the full tensors — links and all three gradients — against the fp32-FMA kernels on the same inputs."""
    import ctypes
    from daspeech_amd import _lib
    B, L = 8, 4096
    TR = L - 1
    lens = [4096, 4000, 3333, 2048, 1025, 64, 2, 1]
    olen, q0, k0, g0, w, _ = _links_case(B, L, 64, TR, lens, 21)
    w = w / L
    n = ctypes.c_size_t(0)
    _lib.check(_lib.load().dsp_extract_links_workspace(B, L, 8, 64, TR, 0, ctypes.byref(n)), "workspace")
    assert n.value > 0, "the dispatch did not choose the matrix-core kernels for BASELINE's graph"
    _lib.load().dsp_extract_links_debug_ran()
    got = _links_fwd_bwd(olen, q0, k0, g0, w, TR, None)
    torch.cuda.synchronize()
    assert _lib.load().dsp_extract_links_debug_ran() == 0b1000100, "default at L = 4096: matrix-core forward, bf16-triple contractions in the backward"
    links = got[0]
    i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
    valid = (i + d + 1) < olen.view(B, 1, 1)
    assert torch.equal(torch.isfinite(links), valid)
    rows = valid.any(-1)
    lse = torch.logsumexp(links.masked_fill(~valid, float("-inf"))[rows], -1)
    assert float(lse.abs().max()) <= 1e-4, float(lse.abs().max())
    del valid, i, d, lse
    try:
        _lib.set_option("xl_mfma", 0)
        ref = _links_fwd_bwd(olen, q0, k0, g0, w, TR, None)
        torch.cuda.synchronize()
        assert _lib.load().dsp_extract_links_debug_ran() == 0b0010010, "reference at this size = the tiled fp32-FMA kernels (the bf16 contraction is checked against THEM)"
    finally:
        _lib.set_option("xl_mfma", -1)
    for name, a, b in zip(("links", "dq", "dk", "dgate", "links (inference)"), got, ref):
        assert torch.equal(torch.isneginf(a), torch.isneginf(b)), name
        f = torch.isfinite(b)
        sc = float(b[f].abs().max())                 # relative to the tensor's own largest value (the gradients are ~1e-3 here: no floor of 1)
        assert sc > 0 and float((a[f] - b[f]).abs().max()) <= 2e-5 * sc, (name, float((a[f] - b[f]).abs().max()), sc)


@pytest.mark.gpu
def test_extract_links_wide_window_stays_on_the_hip_path():
    """A window the one-image kernels cannot hold (L = 1500, TR = 1499: a 768 KB score image) — r04 sent it to the torch band formulation.
    Forward and backward through the model's own dispatch against that formulation; the tiled kernels must be the ones that ran."""
    from daspeech_amd.models.daspeech import DAGDecoder, DEFAULT_ARGS, PAD, BOS, EOS, UNK
    from types import SimpleNamespace
    torch.manual_seed(4)
    dev = torch.device("cuda")
    a = SimpleNamespace(**{**DEFAULT_ARGS, "max_transition_length": 99999, "decoder_layers": 0, "max_target_positions": 2048})
    dec = DAGDecoder(a).to(dev).train()
    B, L = 2, 1500
    lens = [1500, 1203]
    prev = torch.full((B, L), PAD, dtype=torch.long, device=dev)
    for b, n in enumerate(lens):
        prev[b, :n] = UNK; prev[b, 0] = BOS; prev[b, n - 1] = EOS
    feats0 = torch.randn(B, L, a.decoder_embed_dim, device=dev)
    res = {}
    wgt = None
    import daspeech_amd.decode_ops as dops
    calls = []
    orig = dops.extract_links_autograd
    dops.extract_links_autograd = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
    try:
        for fused in (True, False):
            dec.fused_links = fused
            dec.zero_grad(set_to_none=True)
            feats = feats0.clone().requires_grad_()
            links = dec.extract_links(feats, prev)
            if wgt is None:
                wgt = torch.randn_like(links) / L
            fin = torch.isfinite(links)
            ((links.masked_fill(~fin, 0.0) * wgt).sum()).backward()
            res[fused] = (links.detach(), feats.grad.detach())
            del links, feats
            torch.cuda.empty_cache()
    finally:
        dops.extract_links_autograd = orig
    assert calls == [1], "the fused path did not take this window"
    (l1, g1), (l0, g0) = res[True], res[False]
    assert l1.shape == (B, L, L - 1) and torch.equal(torch.isneginf(l1), torch.isneginf(l0))
    f = torch.isfinite(l0)
    torch.testing.assert_close(l1[f], l0[f], rtol=1e-5, atol=3e-5)
    sc = max(1.0, float(g0.abs().max()))
    assert float((g1 - g0).abs().max()) <= 2e-5 * sc + 2e-5, float((g1 - g0).abs().max())
    rows = f.any(-1)
    torch.testing.assert_close(torch.logsumexp(l1[rows], -1), torch.zeros_like(l1[rows][:, 0]), rtol=0, atol=5e-5)
