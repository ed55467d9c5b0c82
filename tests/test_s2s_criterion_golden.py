"""S2SDAGFastSpeech2Loss (BASELINE C5's objective) against the REFERENCE's criterion.

tests/golden/make_golden_model.py ran DASpeech/criterions/s2s_dag_fastspeech2_loss.py:93-306 (forward) + loss.backward() through the
reference's whole model in the authoring container: both --training-strategy values, --dag-freezing-steps on / off, criterion.eval(),
--tts-loss-weight 5, GLAT number-random glancing with recorded draws.  The one function of that path the reference has no CPU
implementation of (`dag_loss_with_alpha_beta`, CUDA only) was served by the fp64 C oracle behind the reference's autograd contract — and
checked there, case by case, against the reference's own torch_dag_loss on the same tensors (see `_oracle_alpha_beta_function`)."""
import os

import numpy as np
import pytest
import torch

from tests.test_model_golden import GOLDEN, _manifest, _seeded_product_model
from tests.util_inputs import seeded_fbank

LOG_F = ("loss", "dag-loss", "tts-loss", "l1-loss", "dur-loss", "pitch-loss", "energy-loss", "glat_acc", "glat_keep")
LOG_I = ("ntokens", "nvalidtokens", "nsentences", "invalid_nsentences", "sample_size")
CASES = {  # golden prefix -> criterion options of the product
    "expect": ("expect", dict(training_strategy="expect")),
    "argmax_torch_gather": ("argmax", dict(training_strategy="argmax", torch_dag_logsoftmax_gather=True)),
    "argmax_on_logits": ("argmax", dict(training_strategy="argmax", argmax_on_logits=True)),
    "argmax_reference_gpu_path": ("argmaxq", dict(training_strategy="argmax")),
    "argmax_torch_alignment": ("argmaxq", dict(training_strategy="argmax", torch_dag_best_alignment=True)),
    "frozen": ("frozen", dict(training_strategy="expect", dag_freezing_steps=100)),
    "eval": ("eval", dict(training_strategy="expect")),
}


def _golden():
    return dict(np.load(os.path.join(GOLDEN, "s2s_dag_fastspeech2_loss_reference.npz")))


def test_s2s_criterion_golden_is_self_consistent():
    """The fixture itself (CPU): loss = dag + 5 * (l1 + dur + pitch + energy) in every case, the frozen / eval cases score the DAG from
    alpha alone (same dag-loss, different TTS input than `expect`), and the two argmax cases really differ (the in-place soft-max changes
    the alignment), so the GPU test below discriminates between them."""
    g = _golden()
    for c in ("expect", "argmax", "argmaxq", "frozen", "eval"):
        tts = sum(float(g[f"{c}/log:{k}"]) for k in ("l1-loss", "dur-loss", "pitch-loss", "energy-loss"))
        assert float(g[f"{c}/log:tts-loss"]) == pytest.approx(tts, rel=1e-6)
        assert float(g[f"{c}/loss"]) == pytest.approx(float(g[f"{c}/log:dag-loss"]) + 5.0 * tts, rel=1e-6)
        assert int(g[f"{c}/log:invalid_nsentences"]) == 0 and int(g[f"{c}/log:nsentences"]) == 3
    assert int(g["expect/n_grads"]) == int(g["argmax/n_grads"]) == 695 and int(g["frozen/n_grads"]) == int(g["eval/n_grads"]) == 168
    assert np.abs(g["argmax/adaptor_in"] - g["argmaxq/adaptor_in"]).max() > 1e-3
    assert np.abs(g["expect/adaptor_in"] - g["frozen/adaptor_in"]).max() > 1e-3
    assert np.array_equal(g["frozen/adaptor_in"], g["eval/adaptor_in"])


def _postnet_model(seed, device):
    """The product model with --add-postnet and the reference's seeded weights (manifest keys + the golden's tts.postnet.* keys)."""
    import json
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
    from tests.util_inputs import seeded_model_state
    man = _manifest()
    gp = dict(np.load(os.path.join(GOLDEN, "tts_postnet_reference.npz")))
    pshapes = {str(k): tuple(json.loads(str(v))) for k, v in zip(gp["postnet_keys"], gp["postnet_shapes"])}
    m = S2SConformerDAGFastSpeech2Model(vocab_size=int(man["vocab_size"]), tts=dict(add_postnet=True))
    shapes = {k: tuple(v["shape"]) for k, v in man["keys"].items() if v["dtype"].startswith("float")}
    shapes.update({k: v for k, v in pshapes.items() if not k.endswith("num_batches_tracked")})
    sd = {k: torch.from_numpy(v) for k, v in seeded_model_state(shapes, seed).items()}
    for k, meta in man["keys"].items():
        if k not in sd:
            sd[k] = torch.zeros(tuple(meta["shape"]), dtype=torch.float32 if meta["dtype"].startswith("float") else torch.long)
    for k, shp in pshapes.items():
        if k not in sd:
            sd[k] = torch.zeros(shp, dtype=torch.long)
    missing, unexpected = m.load_reference_state_dict({"model": sd}, strict=True)
    assert missing == [] and unexpected == []
    # the reference's postnet keys, one for one (fastspeech2_noemb.py:128-136 -> tacotron2.py:111-134)
    own = {k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("tts.postnet.")}
    assert own == pshapes and len(m.state_dict()) + 5 == int(gp["n_keys"])            # (+5: the documented ignore list)
    return m.to(device).eval(), gp


def test_postnet_branch_vs_reference_tts():
    """--add-postnet (fastspeech2_noemb.py:128-136,171-173): same state-dict keys as the reference model, strict load, and the TTS half
    teacher-forced reproduces the reference's mel and mel_post = mel + postnet(mel) (CPU, torch path)."""
    e2e = dict(np.load(os.path.join(GOLDEN, "s2st_reference_e2e.npz")))
    m, gp = _postnet_model(int(e2e["seed"]), "cpu")
    with torch.no_grad():
        mel, mel_post, out_lens, _, _, _ = m.tts(torch.from_numpy(gp["x"].copy()), torch.from_numpy(gp["pad"]), durations=torch.from_numpy(gp["dur"]),
                                                 pitches=torch.from_numpy(gp["pitch_in"]), energies=torch.from_numpy(gp["energy_in"]))
    assert out_lens.tolist() == gp["out_lens"].tolist()
    for got, ref in ((mel, gp["mel"]), (mel_post, gp["mel_post"])):
        assert tuple(got.shape) == ref.shape and np.abs(got.numpy() - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-5
    # without the flag the slot is None and the generator keeps `mel` (s2s_nat_generator.py:254-255)
    from daspeech_amd.models.fastspeech2 import FastSpeech2NoEmb
    with torch.no_grad():
        out = FastSpeech2NoEmb(enc_layers=1, dec_layers=1).eval()(torch.randn(1, 4, 256), torch.zeros(1, 4, dtype=torch.bool), durations=torch.full((1, 4), 2))
    assert out[1] is None and tuple(out[0].shape) == (1, 8, 80)


@pytest.mark.gpu
def test_s2s_criterion_with_postnet_vs_reference():
    """The second L1 term (s2s_dag_fastspeech2_loss.py:281-282) and the gradients into the postnet, `expect` case of the reference run."""
    e2e = dict(np.load(os.path.join(GOLDEN, "s2st_reference_e2e.npz")))
    m, _ = _postnet_model(int(e2e["seed"]), "cuda")
    g = dict(np.load(os.path.join(GOLDEN, "s2s_postnet_loss_reference.npz")))
    # (the postnet's five k = 5 convolutions and their gradients go through MIOpen, whose solver choice per shape is made at run time and is not
    #  the same on every box: one full-suite run of r04 missed the common tolerance, three others and every isolated run met it — 4x here)
    grads = _run_case(m, g, "expect", dict(training_strategy="expect"), "expect", int(e2e["seed"]), tol=4.0)
    assert sum(k.startswith("tts.postnet.") for k in grads) == 20 and int(g["expect/n_grads"]) == 715
    assert float(g["expect/log:l1-loss"]) > 2.0                  # two L1 terms


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_s2s_dag_fastspeech2_loss_vs_reference(case):
    pre, opts = CASES[case]
    e2e = dict(np.load(os.path.join(GOLDEN, "s2st_reference_e2e.npz")))
    m = _seeded_product_model(_manifest(), int(e2e["seed"])).cuda().eval()
    _run_case(m, _golden(), pre, opts, case, int(e2e["seed"]))


def _run_case(m, g, pre, opts, case, seed, tol=1.0):
    from daspeech_amd.criterions import S2SDAGFastSpeech2Loss
    frames = [int(x) for x in g["frames"]]
    dev = "cuda"
    sample = {"net_input": {"src_tokens": torch.from_numpy(seeded_fbank(seed + 17, frames)).to(dev),
                            "src_lengths": torch.tensor(frames, device=dev)},
              "update_num": 10}
    for k in ("target_text", "target_text_lengths", "durations", "pitches", "energies", "target_audio", "target_audio_lengths"):
        sample[k] = torch.from_numpy(g[k]).to(dev)
    crit = S2SDAGFastSpeech2Loss(glat_p="0.5", glance_strategy="number-random", tts_loss_weight=5.0, **opts)
    crit.train(case != "eval")
    crit.glat_draws = {"noise": torch.from_numpy(g[pre + "/noise"]).to(dev), "unif": torch.from_numpy(g[pre + "/unif"]).to(dev)}
    captured = {}
    fwd, tts_fwd, ad_fwd = m.forward, m.tts.forward, m.adaptor.forward

    def spy(*a, **k):
        out = fwd(*a, **k)
        captured.update({k2: v for k2, v in out.items() if k2 in ("keep_word_mask", "glat_accu", "glat_keep")})
        return out

    def ad_spy(x):
        captured["adaptor_in"] = x.detach().clone()
        return ad_fwd(x)

    def tts_spy(x, mask, **k):
        captured["tts_padding_mask"] = mask.clone()
        return tts_fwd(x, mask, **k)
    m.forward, m.adaptor.forward, m.tts.forward = spy, ad_spy, tts_spy
    loss, sample_size, log = crit(m, sample)
    assert bool(loss.requires_grad) == bool(g[pre + "/requires_grad"])
    loss.backward()
    # ---- the discrete decisions and what the TTS half was fed
    assert np.array_equal(captured["keep_word_mask"].cpu().numpy(), g[pre + "/keep_word_mask"])
    assert np.array_equal(captured["tts_padding_mask"].cpu().numpy(), g[pre + "/tts_padding_mask"])
    ain, ain_ref = captured["adaptor_in"].float().cpu().numpy(), g[pre + "/adaptor_in"]
    assert ain.shape == ain_ref.shape
    if opts["training_strategy"] == "argmax":
        # features_on_path: the same vertices' hidden states, gathered (s2s_dag_fastspeech2_loss.py:241-246) — a different alignment moves
        # whole rows by O(1)
        assert np.abs(ain - ain_ref).max() <= 2e-4 * np.abs(ain_ref).max()
    else:
        assert np.abs(ain - ain_ref).max() <= 2e-4 * np.abs(ain_ref).max()
    # ---- loss and logging outputs (every key the reference logs)
    assert sample_size == int(g[pre + "/sample_size"]) == 1
    assert float(loss) == pytest.approx(float(g[pre + "/loss"]), rel=3e-5 * tol)
    assert set(log) == set(LOG_F) | set(LOG_I)
    for k in LOG_F:
        assert float(log[k]) == pytest.approx(float(g[f"{pre}/log:{k}"]), rel=3e-5 * tol, abs=1e-7), k
    for k in LOG_I:
        assert int(log[k]) == int(g[f"{pre}/log:{k}"]), k
    # ---- gradients: the same parameters receive one, same total norm, sampled tensors element-wise
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(grads) == sorted(str(x) for x in g[pre + "/grad_names"])
    total = float(torch.sqrt(sum(x.double().pow(2).sum() for x in grads.values())))
    assert total == pytest.approx(float(g[pre + "/grad_total_norm"]), rel=3e-4 * tol)
    keys = [k.split("grad:", 1)[1] for k in g if k.startswith(pre + "/grad:")]
    assert len(keys) >= 10
    for key in keys:
        ref = g[f"{pre}/grad:{key}"]
        got = grads[key].detach().float().cpu().numpy().reshape(-1)[: ref.size].reshape(ref.shape)
        # a global scalar (tts.pos_emb_alpha, dec_pos_emb_alpha) is one cancelling sum over every position and channel of a gradient that
        # reached it through atomic scatter-adds (the length regulator's gather backward): run to run it moves by ~1e-3 of itself on the
        # GPU (seen: 1.3231 vs 1.3244 in one of three identical runs), so scalars get ten times the tensors' tolerance
        w = 10.0 if ref.size <= 4 else 1.0
        assert np.abs(got - ref).max() <= 3e-4 * tol * w * np.abs(ref).max() + 1e-7 * tol * total, (key, np.abs(got - ref).max(), np.abs(ref).max())
    return grads


@pytest.mark.gpu
def test_expect_strategy_runs_without_gradient_and_at_update_zero():
    """criterion.eval() under torch.no_grad() (fairseq's validation) and the first training update (update_num = 0 is not
    > --dag-freezing-steps 0) take the expect branch with no beta kernel — the reference computes with its all-zero beta there."""
    from daspeech_amd.criterions import S2SDAGFastSpeech2Loss
    from daspeech_amd.synthetic import make_s2st_batch
    from tests.test_gpu_model import small_model
    m = small_model().eval()
    s = make_s2st_batch(3, "cuda", seed=1, min_frames=120, max_frames=200)
    crit = S2SDAGFastSpeech2Loss(glat_p="0.5", glance_strategy="number-random", tts_loss_weight=5.0, dag_freezing_steps=0)
    s["update_num"] = 0
    loss, _, log = crit(m.train(), s)
    assert torch.isfinite(loss) and loss.requires_grad
    loss.backward()
    assert m.decoder.query_linear.weight.grad is None and m.tts.out_proj.weight.grad is not None      # DAG frozen, TTS trained
    crit.eval()
    with torch.no_grad():
        loss, _, log = crit(m.eval(), s)
    assert torch.isfinite(loss) and not loss.requires_grad and torch.isfinite(log["tts-loss"])
