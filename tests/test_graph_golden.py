"""Python-side rows of the hot path (links producer, graph decode, GLAT, glat-p schedule) against vectors produced by the
reference's own functions (tests/golden/make_golden_graph.py) and against hand-evaluated cases.

CPU part: the oracle (oracle/graph_oracle.py) and the product's torch formulations vs the goldens.
GPU part (-m gpu): the HIP kernels vs the same goldens and vs the oracle.
"""
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import dag_oracle as orc
from oracle import graph_oracle as gorc

PAD = 1
NINF = float("-inf")


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def assert_links_close(got, want, atol=2e-5):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    np.testing.assert_array_equal(np.isneginf(got), np.isneginf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5, atol=atol)


# =================================================================================================== links producer (a9 / f1)
def test_oracle_extract_links_hand_case():
    """One head, one feature: q_i = feat_i, k_j = pos_emb(j): row 0 scores (0, ln 3) -> softmax (1/4, 3/4); row 1 has one valid
    successor -> log 1 = 0; row 2 none -> -inf (s2t_conformer_dag.py:148-155,199-202)."""
    feats = np.array([[[1.0], [1.0], [1.0]]], np.float32)
    prev = np.array([[0, 3, 2]])
    pos_w = np.zeros((6, 1), np.float32)                  # positions of the three vertices: 2, 3, 4
    pos_w[3, 0], pos_w[4, 0] = 0.0, math.log(3.0)
    links = gorc.extract_links(feats, prev, pos_w, q_w=[[1.0, 0.0]], q_b=[0.0], k_w=[[0.0, 1.0]], k_b=[0.0], g_w=[[0.0, 0.0]], g_b=[0.0],
                               max_transition_length=99999, heads=1, pad=PAD)
    want = np.array([[[math.log(0.25), math.log(0.75)], [0.0, NINF], [NINF, NINF]]], np.float32)
    assert_links_close(links, want, atol=1e-6)


@pytest.mark.parametrize("tag", ["band", "full", "wide"])
def test_oracle_extract_links_vs_reference(golden_dir, tag):
    g = load(golden_dir, "graph_links")
    links = gorc.extract_links(g[f"{tag}_feats"], g[f"{tag}_prev"], g[f"{tag}_pos_w"], g[f"{tag}_q_w"], g[f"{tag}_q_b"], g[f"{tag}_k_w"],
                               g[f"{tag}_k_b"], g[f"{tag}_g_w"], g[f"{tag}_g_b"], int(g[f"{tag}_max_transition_length"]), int(g[f"{tag}_heads"]), PAD)
    assert_links_close(links, g[f"{tag}_links"])
    np.testing.assert_array_equal(orc.restore_valid_links(g[f"{tag}_links"]), g[f"{tag}_dense"])       # restore_valid_links (:157-169)


def decoder_from_golden(g, tag, device="cpu"):
    """The product's links head (DAGDecoder) carrying the golden's weights."""
    from daspeech_amd.models.daspeech import DAGDecoder, DEFAULT_ARGS
    L, dim = g[f"{tag}_feats"].shape[1:]
    a = SimpleNamespace(**{**DEFAULT_ARGS, "decoder_embed_dim": int(dim), "decoder_attention_heads": int(g[f"{tag}_heads"]), "decoder_layers": 0,
                           "vocab_size": 8, "max_target_positions": int(L), "max_transition_length": int(g[f"{tag}_max_transition_length"])})
    dec = DAGDecoder(a)
    with torch.no_grad():
        dec.link_positional.weight.copy_(torch.from_numpy(g[f"{tag}_pos_w"]))
        for m, n in ((dec.query_linear, "q"), (dec.key_linear, "k"), (dec.gate_linear, "g")):
            m.weight.copy_(torch.from_numpy(g[f"{tag}_{n}_w"])); m.bias.copy_(torch.from_numpy(g[f"{tag}_{n}_b"]))
    return dec.to(device).eval()


@pytest.mark.parametrize("tag", ["band", "full", "wide"])
def test_product_torch_extract_links_vs_reference(golden_dir, tag):
    """DAGDecoder.extract_links, torch formulation (the training path) on CPU."""
    g = load(golden_dir, "graph_links")
    dec = decoder_from_golden(g, tag)
    with torch.no_grad():
        links = dec.extract_links(torch.from_numpy(g[f"{tag}_feats"]), torch.from_numpy(g[f"{tag}_prev"]))
    assert_links_close(links.numpy(), g[f"{tag}_links"])


def _released_heads_case(g, tag):
    """graph_links_released_heads.npz: 8 heads x 32 / 64 channels, weights rebuilt from the seed (tests/util_inputs.seeded_weights)."""
    from tests.util_inputs import seeded_weights
    feats, prev = g[f"{tag}_feats"], g[f"{tag}_prev"]
    L, dim = feats.shape[1:]
    heads = int(g[f"{tag}_heads"])
    shapes = {"pos.weight": (L + 2, dim), "q.weight": (dim, 2 * dim), "q.bias": (dim,), "k.weight": (dim, 2 * dim), "k.bias": (dim,),
              "g.weight": (heads, 2 * dim), "g.bias": (heads,)}
    w = seeded_weights(shapes, int(g[f"{tag}_seed"]), gain=2.0)
    w["pos.weight"] = w["pos.weight"] * np.float32(4.0)
    w["pos.weight"][PAD] = 0
    return feats, prev, w, heads, int(g[f"{tag}_max_transition_length"])


@pytest.mark.parametrize("tag", ["h32", "h64"])
def test_oracle_extract_links_vs_reference_released_heads(golden_dir, tag):
    g = load(golden_dir, "graph_links_released_heads")
    feats, prev, w, heads, mtl = _released_heads_case(g, tag)
    links = gorc.extract_links(feats, prev, w["pos.weight"], w["q.weight"], w["q.bias"], w["k.weight"], w["k.bias"], w["g.weight"], w["g.bias"],
                               mtl, heads, PAD)
    assert_links_close(links, g[f"{tag}_links"], atol=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["h32", "h64"])
def test_hip_extract_links_vs_reference_and_oracle(golden_dir, tag):
    """dsp_extract_links (fused, band only; head widths 32 / 64 / 128) against links the REFERENCE's extract_links produced for the same
    seeded weights, directly and through the model's dispatch, and against the oracle restatement."""
    from daspeech_amd import decode_ops
    from daspeech_amd.models.daspeech import DAGDecoder, DEFAULT_ARGS
    g = load(golden_dir, "graph_links_released_heads")
    feats_np, prev_np, w, h, mtl = _released_heads_case(g, tag)
    B, L, d = feats_np.shape
    a = SimpleNamespace(**{**DEFAULT_ARGS, "decoder_embed_dim": d, "decoder_attention_heads": h, "decoder_layers": 0, "vocab_size": 8,
                           "max_target_positions": L, "max_transition_length": mtl})
    dec = DAGDecoder(a)
    with torch.no_grad():
        dec.link_positional.weight.copy_(torch.from_numpy(w["pos.weight"]))
        for m, n in ((dec.query_linear, "q"), (dec.key_linear, "k"), (dec.gate_linear, "g")):
            m.weight.copy_(torch.from_numpy(w[f"{n}.weight"])); m.bias.copy_(torch.from_numpy(w[f"{n}.bias"]))
    dec = dec.cuda().eval()
    feats, prev = torch.from_numpy(feats_np).cuda(), torch.from_numpy(prev_np).cuda()
    with torch.no_grad():
        fp = torch.cat([feats, dec.link_positional(dec.positions(prev))], -1)
        q = dec.query_linear(fp).view(B, L, h, d // h); k = dec.key_linear(fp).view(B, L, h, d // h)
        lg = torch.log_softmax(dec.gate_linear(fp), -1, dtype=torch.float)
        TR = min(mtl, L - 1)
        got = decode_ops.extract_links(q, k, lg, prev.ne(PAD).sum(-1), TR)              # the HIP kernel itself
        assert dec.fused_links
        via_model = dec.extract_links(feats, prev)                                       # and through the model's dispatch
    assert_links_close(got.cpu().numpy(), g[f"{tag}_links"], atol=5e-5)
    assert_links_close(via_model.cpu().numpy(), g[f"{tag}_links"], atol=5e-5)
    want = gorc.extract_links(feats_np, prev_np, w["pos.weight"], w["q.weight"], w["q.bias"], w["k.weight"], w["k.bias"], w["g.weight"], w["g.bias"],
                              mtl, h, PAD)
    assert_links_close(got.cpu().numpy(), want, atol=5e-5)


@pytest.mark.gpu
def test_hip_extract_links_vs_oracle_ragged_released_width():
    """Released head geometry (8 heads x 64) on ragged graphs, banded and full windows: HIP vs the oracle."""
    from daspeech_amd import decode_ops
    rng = np.random.default_rng(5)
    B, L, h, ck = 3, 45, 8, 64
    d = h * ck
    for TRmax in (7, 32, 99999):
        feats = rng.standard_normal((B, L, d)).astype(np.float32) * 0.3
        lens = np.array([L, L - 11, 2])
        prev = np.full((B, L), 3, np.int64); prev[np.arange(L)[None] >= lens[:, None]] = PAD
        pos_w = (rng.standard_normal((L + 2, d)) * 0.3).astype(np.float32)
        ws = {n: (rng.standard_normal((o, 2 * d)) * (0.5 / math.sqrt(d))).astype(np.float32) for n, o in (("q", d), ("k", d), ("g", h))}
        bs = {n: (rng.standard_normal(o) * 0.1).astype(np.float32) for n, o in (("q", d), ("k", d), ("g", h))}
        want = gorc.extract_links(feats, prev, pos_w, ws["q"], bs["q"], ws["k"], bs["k"], ws["g"], bs["g"], TRmax, h, PAD)
        t = lambda a: torch.from_numpy(a).cuda()
        fp = torch.cat([t(feats), t(pos_w)[torch.from_numpy(gorc.make_positions(prev, PAD)).cuda()]], -1)
        q = (fp @ t(ws["q"]).T + t(bs["q"])).view(B, L, h, ck); k = (fp @ t(ws["k"]).T + t(bs["k"])).view(B, L, h, ck)
        lg = torch.log_softmax(fp @ t(ws["g"]).T + t(bs["g"]), -1)
        got = decode_ops.extract_links(q, k, lg, t(lens), min(TRmax, L - 1))
        assert_links_close(got.cpu().numpy(), want, atol=5e-5)


# =================================================================================================== graph decode (a12 / f3)
def _hand_graph():
    """4 vertices, argmax tokens (5, 6, 6, 7) with the same max log-prob everywhere (so lookahead = greedy on the links):
    0 -> {1: .2, 2: .8}, 1 -> {2: .5, 3: .5}, 2 -> {3: 1}.  Lookahead / greedy: 0 -> 2 -> 3, tokens 5 6 7, features of vertices 2, 3.
    Viterbi: max_length = int(4/8/0.5) = 1, one step: best j of links[0][j] + links[j][3] = max(ln .2 + ln .5, ln .8 + 0) -> j = 2 -> [6]."""
    V = 8
    logits = np.zeros((1, 4, V), np.float32)
    for j, t in enumerate((5, 6, 6, 7)):
        logits[0, j, t] = 4.0
    ln = math.log
    links = np.array([[[ln(.2), ln(.8), NINF], [ln(.5), ln(.5), NINF], [0.0, NINF, NINF], [NINF, NINF, NINF]]], np.float32)
    feats = np.arange(8, dtype=np.float32).reshape(1, 4, 2)
    prev = np.array([[0, 3, 3, 2]])
    return logits, links, feats, prev


def test_oracle_decode_hand_case():
    logits, links, feats, prev = _hand_graph()
    for strat in ("lookahead", "greedy"):
        tok, f, mask, lens = gorc.forward_decoder(logits, links, feats, prev, strat, PAD)
        assert tok.tolist() == [[5, 6, 7]] and lens.tolist() == [2] and not mask.any()
        np.testing.assert_array_equal(f[0], feats[0, [2, 3]])
    for strat in ("viterbi", "jointviterbi"):
        tok, f, mask, lens = gorc.forward_decoder(logits, links, feats, prev, strat, PAD)
        assert tok.tolist() == [[6]] and lens.tolist() == [1]
        np.testing.assert_array_equal(f[0], feats[0, [2]])


def _check_decode(got, g, tag, strat):
    tok, f, mask, lens = [np.asarray(x) for x in got]
    want_tok, want_f, want_mask = g[f"{tag}_{strat}_tokens"], g[f"{tag}_{strat}_features"], g[f"{tag}_{strat}_mask"]
    np.testing.assert_array_equal(tok, want_tok)
    np.testing.assert_array_equal(mask, want_mask)
    np.testing.assert_array_equal(lens, (~want_mask).sum(1))
    np.testing.assert_array_equal(f, want_f)                          # gathered hidden states: pure copies, bit-exact


@pytest.mark.parametrize("strat", ["lookahead", "greedy", "viterbi", "jointviterbi"])
@pytest.mark.parametrize("tag", ["a", "b", "q"])
def test_oracle_decode_vs_reference(golden_dir, tag, strat):
    g = load(golden_dir, "graph_decode")
    got = gorc.forward_decoder(g[f"{tag}_logits"], g[f"{tag}_links"], g[f"{tag}_feats"], g[f"{tag}_prev"], strat, PAD,
                               float(g[f"{tag}_decode_beta"]), float(g[f"{tag}_viterbibeta"]))
    _check_decode(got, g, tag, strat)


@pytest.mark.parametrize("joint", [False, True])
@pytest.mark.parametrize("tag", ["a", "b", "q"])
def test_product_torch_viterbi_vs_reference(golden_dir, tag, joint):
    """decode_ops.viterbi_decode_torch (batched torch form of the reference loop) on CPU."""
    from daspeech_amd import decode_ops
    g = load(golden_dir, "graph_decode")
    t = torch.from_numpy
    prev = t(g[f"{tag}_prev"])
    got = decode_ops.viterbi_decode_torch(t(g[f"{tag}_logits"]), t(g[f"{tag}_links"]), t(g[f"{tag}_feats"]), prev.ne(PAD).sum(-1), PAD,
                                          float(g[f"{tag}_decode_beta"]), float(g[f"{tag}_viterbibeta"]), joint, 0.5)
    _check_decode([x.numpy() for x in got], g, tag, "jointviterbi" if joint else "viterbi")


@pytest.mark.gpu
@pytest.mark.parametrize("strat", ["lookahead", "greedy", "viterbi", "jointviterbi"])
@pytest.mark.parametrize("tag", ["a", "b", "q", "hand"])
def test_hip_decode_vs_reference(golden_dir, tag, strat):
    """The HIP graph decode (dsp_argmax_logp / dsp_lookahead_next / dsp_follow_path / dsp_gather_rows; dsp_dag_max_alpha +
    dsp_dag_backtrace for the Viterbi strategies) against the reference-produced tokens / features / masks."""
    from daspeech_amd import decode_ops
    if tag == "hand":
        logits, links, feats, prev = _hand_graph()
        dbeta = vbeta = 1.0
        want = gorc.forward_decoder(logits, links, feats, prev, strat, PAD)
    else:
        g = load(golden_dir, "graph_decode")
        logits, links, feats, prev = g[f"{tag}_logits"], g[f"{tag}_links"], g[f"{tag}_feats"], g[f"{tag}_prev"]
        dbeta, vbeta = float(g[f"{tag}_decode_beta"]), float(g[f"{tag}_viterbibeta"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out_len = t(prev).ne(PAD).sum(-1)
    if strat in ("lookahead", "greedy"):
        got = decode_ops.graph_decode(t(logits), t(links), t(feats), out_len, PAD, dbeta, strat)
    else:
        got = decode_ops.viterbi_decode(t(logits), t(links), t(feats), out_len, PAD, dbeta, vbeta, strat == "jointviterbi", 0.5)
    got = [x.cpu().numpy() for x in got]
    if tag == "hand":
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)
    else:
        _check_decode(got, g, tag, strat)


@pytest.mark.gpu
@pytest.mark.parametrize("strat", ["lookahead", "greedy", "viterbi", "jointviterbi"])
@pytest.mark.parametrize("shape", [(3, 48, 7), (2, 96, 95), (4, 130, 32), (3, 120, 1), (2, 400, 3)])
def test_hip_decode_vs_oracle_random(shape, strat):
    """Larger ragged graphs, <pad> emissions and repeated tokens: HIP vs the oracle loop.  The last two shapes have windows so narrow
    that the final vertex is out of reach within the L / 4 steps of the viterbi strategies: every candidate is -inf and the reference
    emits the token of vertex 0 (its arg-maxes over all -inf return index 0)."""
    from daspeech_amd import decode_ops
    B, L, TR = shape
    rng = np.random.default_rng(L + TR)
    V, D = 11, 6
    logits = (rng.standard_normal((B, L, V)) * 2).astype(np.float32)
    logits[:, ::4, PAD] += 5
    logits[:, 1::5] = logits[:, 2::5][:, : logits[:, 1::5].shape[1]]
    lens = rng.integers(max(3, L - 9), L + 1, B); lens[0] = L
    from tests.util_inputs import make_dag_inputs
    _, links, _, _ = make_dag_inputs(L, B, 4, L, TR, ragged=False)
    i = np.arange(L)[None, :, None]; d = np.arange(TR)[None, None, :]
    valid = (i + d + 1) < lens[:, None, None]
    links = np.where(valid, links, -np.inf).astype(np.float32)          # (unnormalised rows are fine for a decode)
    feats = rng.standard_normal((B, L, D)).astype(np.float32)
    prev = np.full((B, L), 3, np.int64); prev[np.arange(L)[None] >= lens[:, None]] = PAD
    want = gorc.forward_decoder(logits, links, feats, prev, strat, PAD, 0.8, 1.2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if strat in ("lookahead", "greedy"):
        got = decode_ops.graph_decode(t(logits), t(links), t(feats), t(lens), PAD, 0.8, strat)
    else:
        got = decode_ops.viterbi_decode(t(logits), t(links), t(feats), t(lens), PAD, 0.8, 1.2, strat == "jointviterbi", 0.5)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a.cpu().numpy(), b)


# =================================================================================================== GLAT (a10)
def test_oracle_glat_hand_case():
    """L = 4, targets (5, 6, 7): emissions force the alignment 0 -> 2 -> 3, so path = (0, -1, 1, 2); matchmask marks (t=0,j=0),
    (1,2), (2,3); oracle = tgt[path.clip(0)] = (5, 5, 6, 7); argmax tokens (5, 9, 9, 7) agree on vertices 0 and 3 -> same_num = 2;
    number-random with p = 1: glance_nums = int((3 - 2) * 1 + 0.5) = 1 (nat_dag_loss.py:224-238)."""
    V = 10
    logits = np.full((1, 4, V), -2.0, np.float32)
    logits[0, 0, 5] = 5; logits[0, 1, 9] = 5; logits[0, 2, 9] = 5; logits[0, 2, 6] = 4; logits[0, 3, 7] = 5
    ln = math.log
    links = np.array([[[ln(.5), ln(.5), NINF], [ln(.5), ln(.5), NINF], [0.0, NINF, NINF], [NINF, NINF, NINF]]], np.float32)
    prev = np.array([[0, 3, 3, 2]]); tgt = np.array([[5, 6, 7]])
    noise = np.array([[0.3, 9.0, -0.2, 0.1]], np.float32)            # vertex 1 is off the path: its 9.0 must not count
    o = gorc.glat(logits, links, prev, tgt, 1.0, "number-random", noise=noise, unif=np.full((1, 4), 0.5, np.float32), pad=PAD)
    assert o["path"].tolist() == [[0, -1, 1, 2]]
    assert o["matchmask"].astype(int).tolist() == [[[1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]]
    assert o["oracle"].tolist() == [[5, 5, 6, 7]] and o["same_num"].tolist() == [2] and o["glance_nums"].tolist() == [1]
    assert o["keep_word_mask"].tolist() == [[True, False, False, False]]          # the highest on-path score is vertex 0's 0.3
    assert o["glat_prev_output_tokens"].tolist() == [[5, 3, 3, 2]]
    o = gorc.glat(logits, links, prev, tgt, 0.6, None, unif=np.array([[0.1, 0.0, 0.3, 0.19]], np.float32), pad=PAD)
    np.testing.assert_allclose(o["keep_prob"], [[0.2, 0, 0.2, 0.2]], rtol=1e-6)      # (3 - 2) / 3 * 0.6 on the path
    assert o["keep_word_mask"].tolist() == [[True, False, False, True]]


def _glat_case(g, tag):
    strategy = str(g[f"{tag}_strategy"])
    return (g[f"{tag}_logits"], g[f"{tag}_links"], g[f"{tag}_prev"], g[f"{tag}_tgt"], float(g[f"{tag}_p"]),
            None if strategy == "None" else strategy, g[f"{tag}_noise"], g[f"{tag}_unif"], g[f"{tag}_unif_n"])


@pytest.mark.parametrize("tag", ["none", "nr", "nr0", "cmlm"])
def test_oracle_glat_vs_reference(golden_dir, tag):
    g = load(golden_dir, "glat")
    logits, links, prev, tgt, p, strategy, noise, unif, unif_n = _glat_case(g, tag)
    o = gorc.glat(logits, links, prev, tgt, p, strategy, noise, unif, PAD, unif_n=unif_n)
    np.testing.assert_array_equal(o["matchmask"], g[f"{tag}_matchmask"])
    np.testing.assert_array_equal(o["keep_word_mask"], g[f"{tag}_keep_word_mask"])
    np.testing.assert_array_equal(o["glat_prev_output_tokens"], g[f"{tag}_glat_prev"])
    np.testing.assert_allclose(o["glat_accu"], g[f"{tag}_glat_accu"], rtol=1e-6)
    np.testing.assert_allclose(o["glat_keep"], g[f"{tag}_glat_keep"], rtol=1e-5)
    if strategy in ("number-random", "cmlm"):                         # count invariant: exactly glance_nums glanced vertices per sample
        np.testing.assert_array_equal(o["keep_word_mask"].sum(1), o["glance_nums"])


def _product_glat(g, tag, device, torch_ops):
    from daspeech_amd.criterions import glat_function
    logits, links, prev, tgt, p, strategy, noise, unif, unif_n = _glat_case(g, tag)
    t = lambda a: torch.from_numpy(a).to(device)
    model = SimpleNamespace(pad=PAD)
    return glat_function(model, t(logits.copy()), t(tgt), t(prev), {"context_p": p}, links=t(links), glance_strategy=strategy,
                         torch_ops=torch_ops, noise=t(noise), unif=t(unif), unif_n=t(unif_n))


def _check_glat(out, g, tag):
    gp, gt, info = out
    np.testing.assert_array_equal(gp.cpu().numpy(), g[f"{tag}_glat_prev"])
    np.testing.assert_array_equal(gt.cpu().numpy(), g[f"{tag}_tgt"])
    np.testing.assert_array_equal(info["matchmask"].cpu().numpy(), g[f"{tag}_matchmask"])
    np.testing.assert_array_equal(info["keep_word_mask"].cpu().numpy(), g[f"{tag}_keep_word_mask"])
    np.testing.assert_allclose(float(info["glat_accu"]), g[f"{tag}_glat_accu"], rtol=1e-6)
    np.testing.assert_allclose(float(info["glat_keep"]), g[f"{tag}_glat_keep"], rtol=1e-5)


@pytest.mark.parametrize("tag", ["none", "nr", "nr0", "cmlm"])
def test_product_glat_torch_ops_vs_reference(golden_dir, tag):
    g = load(golden_dir, "glat")
    _check_glat(_product_glat(g, tag, "cpu", True), g, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["none", "nr", "nr0", "cmlm"])
def test_product_glat_hip_vs_reference_and_oracle(golden_dir, tag):
    """criterions.glat_function on the HIP ops (dag_logsoftmax_gather_inplace + dag_best_alignment) with the stored draws."""
    g = load(golden_dir, "glat")
    out = _product_glat(g, tag, "cuda", False)
    _check_glat(out, g, tag)
    logits, links, prev, tgt, p, strategy, noise, unif, unif_n = _glat_case(g, tag)
    o = gorc.glat(logits, links, prev, tgt, p, strategy, noise, unif, PAD, unif_n=unif_n)
    for key in ("path", "oracle", "same_num"):
        np.testing.assert_array_equal(out[2][key].cpu().numpy(), o[key])


@pytest.mark.gpu
def test_number_random_count_invariant_with_device_rng():
    """With the device's own draws only the COUNT is reproducible: exactly glance_nums aligned vertices are revealed per sample."""
    from daspeech_amd.criterions import glat_function
    from tests.util_inputs import make_dag_inputs
    B, L, T, TR, V = 6, 60, 14, 59, 40
    rng = np.random.default_rng(2)
    _, links, ol, tl = make_dag_inputs(4, B, T, L, TR)
    logits = (rng.standard_normal((B, L, V)) * 2).astype(np.float32)
    prev = np.full((B, L), 3, np.int64); prev[np.arange(L)[None] >= ol[:, None]] = PAD
    tgt = rng.integers(4, V, (B, T)); tgt[np.arange(T)[None] >= tl[:, None]] = PAD
    t = lambda a: torch.from_numpy(a).cuda()
    for p in (0.5, 0.1):
        gp, _, info = glat_function(SimpleNamespace(pad=PAD), t(logits), t(tgt), t(prev), {"context_p": p}, links=t(links),
                                    glance_strategy="number-random")
        want = ((t(tl) - info["same_num"]) * p + 0.5).long()
        assert torch.equal(info["keep_word_mask"].sum(1), want)
        assert (info["keep_word_mask"] & (info["path"] < 0)).sum() == 0


# =================================================================================================== glat-p schedule
def test_anneal_schedule_vs_reference(golden_dir):
    from daspeech_amd import criterions
    g = load(golden_dir, "glat")
    for a, row in zip(g["anneal_args"], g["anneal_values"]):
        for u, want in zip(g["anneal_updates"], row):
            assert gorc.get_anneal_value(gorc.parse_anneal_argument(str(a)), int(u)) == pytest.approx(want, rel=1e-12)
            assert criterions.get_anneal_value(criterions.parse_anneal_argument(str(a)), int(u)) == pytest.approx(want, rel=1e-12)
    c = criterions.NATDAGLoss(glat_p="0.5:0.1@200k", glance_strategy="number-random")
    assert c.glat_p == 0.5
    c.set_update_num(100000)
    assert c.glat_p == pytest.approx(0.5 + (0.1 - 0.5) * 100000 / 200001)
    c.set_update_num(10 ** 7)
    assert c.glat_p == 0.1
