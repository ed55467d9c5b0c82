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


def _viterbi_reference_loop(logits, dense, feats, out_len, pad, beta, vbeta, joint, scale):
    """Per-sample restatement of s2s_conformer_dag_fastspeech2.py:244-304 with Python loops (host back-trace like the reference)."""
    B, L, V = logits.shape
    logp = torch.log_softmax(logits, -1)
    sc, tok = logp.max(-1)
    alpha = dense[:, 0].clone()
    if joint:
        alpha = alpha + sc[:, 0].unsqueeze(1) * beta
    alpha = alpha + sc * beta
    M = max(1, int(L / 8 / scale))
    scores, indexs = [alpha], []
    for _ in range(M - 1):
        alpha, index = torch.max(alpha.unsqueeze(-1) + dense, dim=1)
        if joint:
            alpha = alpha + sc * beta
        scores.append(alpha); indexs.append(index)
    scores = torch.stack(scores, 0)
    link_last = torch.stack([dense[b, :, out_len[b] - 1] for b in range(B)], 0).unsqueeze(0)
    best, max_idx = torch.max(scores + link_last, dim=-1)
    lengths = (torch.arange(M) + 1).unsqueeze(-1).float()
    _, pred = torch.max(best / lengths ** vbeta, dim=0)
    pred = pred + 1
    outs = []
    for b in range(B):
        length = int(pred[b]); j = int(max_idx[length - 1, b])
        last = int(tok[b, j]); res = [last]; fl = [feats[b, j]]
        for k in range(length - 1):
            j = int(indexs[length - k - 2][b, j]); now = int(tok[b, j])
            if now != pad and now != last:
                res.insert(0, now); fl.insert(0, feats[b, j])
            last = now
        outs.append((res, torch.stack(fl)))
    return outs


@pytest.mark.parametrize("joint", [True, False])
def test_viterbi_decode_matches_loop_restatement(joint):
    from daspeech_amd import decode_ops
    torch.manual_seed(5)
    B, L, TR, V, D, pad = 3, 40, 6, 11, 4, 1
    logits = torch.randn(B, L, V) * 2
    logits[:, ::4, pad] += 6                                   # some <pad> emissions
    raw = torch.randn(B, L, TR)
    out_len = torch.tensor([40, 37, 33])
    i = torch.arange(L).view(1, L, 1); d = torch.arange(TR).view(1, 1, TR)
    valid = (i + d + 1) < out_len.view(B, 1, 1)
    links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1)
    links = links.masked_fill(~valid, float("-inf"))
    feats = torch.randn(B, L, D)
    toks, of, mask, n = decode_ops.viterbi_decode_torch(logits, links, feats, out_len, pad, 1.0, 1.0, joint, 0.5)
    ref = _viterbi_reference_loop(logits, decode_ops.restore_valid_links(links), feats, out_len.tolist(), pad, 1.0, 1.0, joint, 0.5)
    for b, (res, fl) in enumerate(ref):
        assert n[b].item() == len(res)
        assert toks[b, : len(res)].tolist() == res and (toks[b, len(res):] == pad).all()
        torch.testing.assert_close(of[b, : len(res)], fl)
        assert (of[b, len(res):] == 0).all() and mask[b].tolist() == [False] * len(res) + [True] * (toks.shape[1] - len(res))


def test_generator_file_sinks_use_the_reference_formats(tmp_path):
    """feat/<id>.npy is [80, T] float32 (generate_features.py:87-91), wav/<id>_generated_e2e.wav is int16 = audio * 32768
    (inference_e2e.py:50-56)."""
    import numpy as np
    from scipy.io.wavfile import read as wav_read
    from daspeech_amd.generator import dump_results
    feat = torch.randn(7, 80)
    wav = torch.tensor([0.0, 0.5, -0.5, 0.999, -1.0])
    out = dump_results(str(tmp_path), ["utt_3", "utt_4"], [{"feature": feat, "waveform": wav}, {"feature": feat[:2]}])
    assert len(out) == 3
    f = np.load(tmp_path / "feat" / "utt_3.npy")
    assert f.shape == (80, 7) and f.dtype == np.float32 and np.array_equal(f, feat.numpy().T)
    sr, a = wav_read(tmp_path / "wav" / "utt_3_generated_e2e.wav")
    assert sr == 22050 and a.dtype == np.int16 and a.tolist() == [0, 16384, -16384, 32735, -32768]
    assert not (tmp_path / "wav" / "utt_4_generated_e2e.wav").exists()


def test_model_loads_a_fairseq_style_checkpoint_dict():
    """checkpoint["model"] with the extra fairseq entries (tied output projection, version buffers, sinusoidal placeholders) loads
    key for key; a renamed or missing parameter is reported, not guessed."""
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
    kw = dict(encoder_layers=1, decoder_layers=1, vocab_size=32, tts=dict(enc_layers=1, dec_layers=1))
    torch.manual_seed(0)
    src = S2SConformerDAGFastSpeech2Model(**kw)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    sd["decoder.output_projection.weight"] = sd["decoder.embed_tokens.weight"].clone()
    sd["decoder.version"] = torch.tensor([3.0]); sd["encoder.version"] = torch.tensor([1.0])
    sd["encoder.embed_positions._float_tensor"] = torch.zeros(1)
    torch.manual_seed(1)
    dst = S2SConformerDAGFastSpeech2Model(**kw)
    missing, unexpected = dst.load_reference_state_dict({"model": sd, "cfg": None})
    assert not missing and not unexpected
    for k, v in src.state_dict().items():
        assert torch.equal(v, dst.state_dict()[k]), k
    bad = dict(sd); bad["decoder.layers.0.fc3.weight"] = torch.zeros(2); del bad["adaptor.fc1.bias"]
    with pytest.raises(KeyError, match="fc3"):
        dst.load_reference_state_dict({"model": bad})
    untied = dict(sd); untied["decoder.output_projection.weight"] = untied["decoder.output_projection.weight"] + 1
    with pytest.raises(ValueError):
        dst.load_reference_state_dict({"model": untied})
