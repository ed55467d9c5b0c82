"""GPU smoke + consistency tests of the model glue (reduced depth, released widths)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def small_model():
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
    torch.manual_seed(0)
    return S2SConformerDAGFastSpeech2Model(encoder_layers=2, decoder_layers=1, tts=dict(enc_layers=1, dec_layers=1)).cuda()


def test_registry_names():
    from daspeech_amd import registry
    assert set(registry.MODEL_REGISTRY) == {"s2t_conformer_dag", "s2s_conformer_dag_fastspeech2"}
    assert set(registry.CRITERION_REGISTRY) == {"nat_dag_loss", "s2s_dag_fastspeech2_loss"}
    assert set(registry.TASK_REGISTRY) == {"nat_speech_to_text", "nat_speech_to_speech"}


def test_training_objective_backward():
    """S2SDAGFastSpeech2Loss.forward(model, sample) — the reference's criterion contract (s2s_dag_fastspeech2_loss.py:93-306) with the
    released recipe (number-random glancing, expect strategy): finite loss, the logging keys, and gradients everywhere the reference's
    graph has them (the mel loss must reach the adaptor, the encoder FFT layers and both variance embeddings)."""
    from daspeech_amd.criterions import S2SDAGFastSpeech2Loss
    from daspeech_amd.synthetic import make_s2st_batch
    m = small_model().train()
    s = make_s2st_batch(3, "cuda", seed=1, min_frames=120, max_frames=200)
    s["update_num"] = 1000
    crit = S2SDAGFastSpeech2Loss(glat_p="0.5:0.1@200k", glance_strategy="number-random", tts_loss_weight=5.0)
    loss, sample_size, log = crit(m, s)
    assert sample_size == 1 and torch.isfinite(loss) and int(log["invalid_nsentences"]) == 0
    for key in ("loss", "dag-loss", "tts-loss", "l1-loss", "dur-loss", "pitch-loss", "energy-loss", "ntokens", "nvalidtokens", "nsentences",
                "glat_acc", "glat_keep"):
        assert key in log
    assert crit.glat_p == pytest.approx(0.5 + (0.1 - 0.5) * 1000 / 200001)
    loss.backward()
    g = [p.grad for p in m.parameters() if p.grad is not None]
    assert len(g) > 50 and all(torch.isfinite(x).all() for x in g)
    assert m.decoder.query_linear.weight.grad.abs().sum() > 0          # links head receives gradient through the HIP DP ops
    assert m.tts.out_proj.weight.grad.abs().sum() > 0


def test_mel_loss_alone_reaches_adaptor_encoder_and_embeddings():
    """with ONLY the mel L1 term the gradient must flow through the length regulator and `x + embed(...)` into
    model.adaptor, the TTS encoder FFT layers and embed_pitch / embed_energy."""
    from daspeech_amd.criterions import S2SDAGFastSpeech2Loss
    from daspeech_amd.synthetic import make_s2st_batch
    import torch.nn.functional as F
    m = small_model().train()
    s = make_s2st_batch(2, "cuda", seed=4, min_frames=120, max_frames=160)
    feats = torch.randn(2, int(s["target_text_lengths"].max()) - 1, 512, device="cuda")
    tlen = s["target_text_lengths"] - 1
    pmask = torch.arange(feats.shape[1], device="cuda").unsqueeze(0) >= tlen.unsqueeze(1)
    mel, _, out_lens, _, _, _ = m.tts(m.adaptor(feats), pmask, durations=s["durations"], pitches=s["pitches"], energies=s["energies"])
    assert out_lens.tolist() == s["durations"].sum(1).tolist()
    F.l1_loss(mel, torch.zeros_like(mel)).backward()
    for p in (m.adaptor.fc1.weight, m.tts.encoder_fft_layers[0].ffn.ffn[0].weight, m.tts.var_adaptor.embed_pitch.weight,
              m.tts.var_adaptor.embed_energy.weight):
        assert p.grad is not None and p.grad.abs().sum() > 0


def test_training_path_adaptor_matches_hip_inference_path():
    """The differentiable torch formulation (training) and the HIP glue kernels (inference) of the variance adaptor are the same function."""
    m = small_model().eval()
    torch.manual_seed(3)
    x = torch.randn(3, 11, 256, device="cuda")
    pmask = torch.arange(11, device="cuda").unsqueeze(0) >= torch.tensor([11, 7, 2], device="cuda").unsqueeze(1)
    dur = torch.randint(0, 5, (3, 11), device="cuda").masked_fill(pmask, 0)
    pit = torch.rand(3, 11, device="cuda") * 10 - 4.6; ene = torch.rand(3, 11, device="cuda") * 8 - 4.9
    va = m.tts.var_adaptor
    with torch.no_grad():
        a = va(x, pmask, dur, pit, ene)
    with torch.enable_grad():
        b = va(x.clone().requires_grad_(), pmask, dur, pit, ene)
    assert a[1].tolist() == b[1].tolist()
    torch.testing.assert_close(a[0], b[0].detach(), rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        a = va(x, pmask)                                 # predicted durations / pitch / energy
    with torch.enable_grad():
        b = va(x.clone().requires_grad_(), pmask)
    assert a[1].tolist() == b[1].tolist()
    torch.testing.assert_close(a[0], b[0].detach(), rtol=1e-5, atol=1e-5)


def test_nat_dag_loss_criterion_contract():
    """NATDAGLoss.forward(model, sample) -> (loss, sample_size, logging_output) (nat_dag_loss.py:164-300) on the S2TT model, HIP ops vs
    the --torch-dag-* variants of the same criterion: same loss (the GLAT draws are replayed by seeding the device generator)."""
    from daspeech_amd.criterions import NATDAGLoss
    from daspeech_amd.models.daspeech import S2TConformerDAGModel
    from daspeech_amd.synthetic import make_s2st_batch
    torch.manual_seed(0)
    m = S2TConformerDAGModel(encoder_layers=2, decoder_layers=1).cuda().eval()      # eval: no dropout anywhere -> deterministic
    s = make_s2st_batch(3, "cuda", seed=6, min_frames=100, max_frames=150)
    s["target"] = s["target_text"]
    s["update_num"] = 5
    losses, grads = [], []
    for torch_ops in (False, True):
        crit = NATDAGLoss(glat_p="0.5", glance_strategy="number-random", torch_dag_loss=torch_ops, torch_dag_best_alignment=torch_ops,
                          torch_dag_logsoftmax_gather=torch_ops)
        crit.train()                                 # (the criterion's training flag: the DAG branch with gradients; the model stays in eval)
        torch.manual_seed(123)
        m.zero_grad(set_to_none=True)
        loss, sample_size, log = crit(m, s)
        assert sample_size == 1 and torch.isfinite(loss) and "dag-loss" in log and "dag_nll-loss" in log
        loss.backward()
        losses.append(float(loss))
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert losses[0] == pytest.approx(losses[1], rel=2e-5)
    # ... and the same gradient for every parameter: the HIP operators' backward (K1 scatter, K4, K5) against autograd through the
    # reference's torch formulations (dag_loss.py:303-425), through the whole model
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 50
    gscale = max(float(g.abs().max()) for g in grads[1].values())        # (a key projection's bias has a zero gradient: noise on both sides)
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        scale = float(b.abs().max()) + 1e-3 * gscale
        # (fp32 both ways, different summation orders through the DP rows and the layers: observed <= 2.1e-4 of the largest entry)
        assert float((a - b).abs().max()) <= 1e-3 * scale + 1e-8, (n, float((a - b).abs().max()), scale)


def test_graph_decode_matches_dense_torch_formulation():
    """forward_decoder (HIP, compact links) vs the reference's dense formulation restated in torch."""
    from daspeech_amd.synthetic import make_s2st_batch
    from oracle import dag_oracle as orc
    m = small_model().eval()
    s = make_s2st_batch(2, "cuda", seed=2, min_frames=100, max_frames=160)
    with torch.no_grad():
        enc = m.forward_encoder(s["net_input"]["src_tokens"], s["net_input"]["src_lengths"])
        prev = m.initialize_output_tokens_by_src(s["net_input"]["src_lengths"])
        logits, links, feats = m.decode_graph(prev, enc)
        dec = m.forward_decoder(prev, enc)
    logp = torch.log_softmax(logits, -1)
    sc, tok = logp.max(-1)
    dense = torch.from_numpy(orc.restore_valid_links(links.cpu().numpy())).cuda()
    nxt = (dense + sc.unsqueeze(1) * 1.0).max(-1)[1].cpu().tolist()
    tok = tok.cpu().tolist()
    out_len = prev.ne(m.pad).sum(-1).tolist()
    for b in range(2):
        last = tok[b][0]; j = 0; res = [last]; kept = []
        while j != out_len[b] - 1:
            j = nxt[b][j]; now = tok[b][j]
            if now != m.pad and now != last:
                res.append(now); kept.append(j)
            last = now
        got = dec["output_tokens"][b].cpu().tolist()
        assert got[: len(res)] == res and dec["feature_lengths"][b].item() == len(kept)
        assert torch.equal(dec["features"][b, : len(kept)], feats[b, kept])


def test_generator_end_to_end():
    from daspeech_amd.generator import S2SNATGenerator
    from daspeech_amd.models import HiFiGANGenerator
    from daspeech_amd.synthetic import make_s2st_batch
    from daspeech_amd.synthetic import calibrate_synthetic_weights
    m = calibrate_synthetic_weights(small_model().eval())
    voc = HiFiGANGenerator(conv_backend="hip").cuda().eval()               # released V1 widths (the HIP kernels need >= 32 channels)
    gen = S2SNATGenerator(voc, torch.zeros(80), torch.ones(80), vocoder_group=2)
    s = make_s2st_batch(3, "cuda", seed=3, min_frames=100, max_frames=140)
    out = gen.generate(m, s)
    assert len(out) == 3
    # grouped vocoding (HIP kernels, per-utterance lengths) == the reference's one-file-at-a-time loop (inference_e2e.py:47-56)
    for o in out:
        alone = voc(o["feature"].t().unsqueeze(0).contiguous())[0, 0]
        assert torch.equal(alone, o["waveform"])
    assert all(8 <= o["feature"].shape[0] <= 1200 for o in out)          # calibrated shapes: tens of phonemes x ~7.5 frames
    for o in out:
        assert o["feature"].shape[1] == 80 and o["waveform"].shape[0] == o["feature"].shape[0] * 256
        assert torch.isfinite(o["waveform"]).all()


def test_generator_batch_pipeline_equals_one_batch_at_a_time():
    """generator.generate_batches / submit / flush (vocoder of batch k-1 on a second stream under the acoustic model of batch k) returns,
    batch by batch and in order, what generate() returns.  The library GEMMs of the acoustic model are not run-to-run deterministic
    (two generate() calls differ by ~1e-6 in the mel, tools/gen_det_check.py), so: tokens identical, mel within 1e-5 of the
    one-at-a-time run — and each returned waveform is, BIT FOR BIT, the vocoding of the returned mel alone on the main stream, which a
    race between the streams (allocator reuse, the vocoder's cached workspace) would break."""
    from daspeech_amd.generator import S2SNATGenerator
    from daspeech_amd.models import HiFiGANGenerator
    from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
    m = calibrate_synthetic_weights(small_model().eval())
    voc = HiFiGANGenerator(conv_backend="hip").cuda().eval()
    gen = S2SNATGenerator(voc, torch.zeros(80), torch.ones(80), vocoder_group=2)
    batches = [make_s2st_batch(3, "cuda", seed=20 + i, min_frames=90 + 10 * i, max_frames=150) for i in range(4)]
    want = [gen.generate(m, s) for s in batches]
    torch.cuda.synchronize()
    for rep in range(2):                               # twice: the side stream and the vocoder's cached buffers are reused
        got = list(gen.generate_batches(m, batches))
        assert len(got) == len(want)
        for gb, wb in zip(got, want):
            assert len(gb) == len(wb)
            for g, w in zip(gb, wb):
                assert torch.equal(g["tokens"], w["tokens"]) and g["feature"].shape == w["feature"].shape
                torch.testing.assert_close(g["feature"], w["feature"], rtol=0, atol=1e-5)
                torch.testing.assert_close(g["waveform"], w["waveform"], rtol=0, atol=1e-3)
        torch.cuda.synchronize()
        for gb in got:
            for g in gb:
                alone = voc(g["feature"].t().unsqueeze(0).contiguous())[0, 0]
                assert torch.equal(alone, g["waveform"])
    assert gen.flush() is None and gen.submit(m, batches[0]) is None and len(gen.flush()) == 3


def test_split_gemm_inference_matches_torch_fp32_within_mel_tolerance():
    """The eval-mode inference path runs its Linear layers and FFT convolutions as fp32-accurate split GEMMs on the fp16 matrix cores
    (decode_ops.split_linear / SplitConv1d).  Same batch through the released architecture with the path on and off: same decoded
    tokens, mel-spectrogram frames within the north-star tolerance (1e-4 relative to the mel range)."""
    from daspeech_amd import decode_ops
    from daspeech_amd.generator import S2SNATGenerator
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
    from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
    torch.manual_seed(77)
    model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).cuda().eval()
    gen = S2SNATGenerator(None, torch.zeros(80, device="cuda"), torch.ones(80, device="cuda"))
    batch = make_s2st_batch(4, "cuda", seed=9, min_frames=300, max_frames=420)
    old = decode_ops.set_split_gemm(True)
    try:
        a = gen.generate(model, batch, generate_waveform=False)
        decode_ops.set_split_gemm(False)
        b = gen.generate(model, batch, generate_waveform=False)
    finally:
        decode_ops.set_split_gemm(old)
    for x, y in zip(a, b):
        assert torch.equal(x["tokens"], y["tokens"]) and x["feature"].shape == y["feature"].shape
        scale = y["feature"].abs().max().item()
        assert (x["feature"] - y["feature"]).abs().max().item() <= 1e-4 * scale, ((x["feature"] - y["feature"]).abs().max().item(), scale)


def _released_model(kind):
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model, S2TConformerDAGModel
    from daspeech_amd.synthetic import calibrate_synthetic_weights
    torch.manual_seed(1234)
    return calibrate_synthetic_weights(S2TConformerDAGModel() if kind == "s2tt" else S2SConformerDAGFastSpeech2Model()).cuda().eval()


@pytest.mark.parametrize("strategy", ["lookahead", "jointviterbi"])
def test_c3_s2tt_full_size_properties(strategy):
    """BASELINE configs[2] at its workload size: S2TT Conformer-DAG full forward, batch 64, released depth (12 + 4 layers), synthetic
    fr-en fbank80 of 300-800 frames.  Size-independent properties: every output token is a vocabulary id, no <pad> inside a
    sequence, no immediate repeats (the decode collapses them), lengths consistent with the padding, graph sizes = 0.5 x frames."""
    from daspeech_amd.synthetic import make_s2st_batch
    m = _released_model("s2tt")
    m.args.decode_strategy = strategy
    s = make_s2st_batch(64, "cuda", seed=11)
    ni = s["net_input"]
    with torch.no_grad():
        enc = m.forward_encoder(ni["src_tokens"], ni["src_lengths"])
        prev = m.initialize_output_tokens_by_src(ni["src_lengths"], max_src_len=ni["src_tokens"].shape[1])
        dec = m.forward_decoder(prev, enc)
    assert prev.ne(m.pad).sum(-1).tolist() == (ni["src_lengths"].float() * 0.5).long().clamp(2, 1024).tolist()
    toks, lens = dec["output_tokens"], dec["feature_lengths"]
    assert toks.shape[0] == 64 and (toks >= 0).all() and (toks < m.args.vocab_size).all()
    nt = lens + (1 if strategy == "lookahead" else 0)                # lookahead keeps <bos> in the token row, its features exclude it
    for b in range(64):
        row = toks[b, : int(nt[b])]
        assert (row != m.pad).all() and (toks[b, int(nt[b]):] == m.pad).all()
        assert (row[1:] != row[:-1]).all()
    assert dec["features"].shape[:2] == (64, int(lens.max())) and torch.isfinite(dec["features"]).all()
    assert torch.equal(dec["features_padding_mask"], torch.arange(int(lens.max()), device="cuda").unsqueeze(0) >= lens.unsqueeze(1))


def test_c4_s2st_full_size_properties():
    """BASELINE configs[3] at its workload size: the full S2ST pipeline (s2s_conformer_dag_fastspeech2 + HiFi-GAN V1, HIP vocoder),
    lookahead decode, batch 32, released depth.  Properties: mel length = sum of the predicted durations, waveform length =
    256 x mel frames, finite values in (-1, 1), and every utterance's waveform equals vocoding its own mel alone."""
    from daspeech_amd.generator import S2SNATGenerator
    from daspeech_amd.models import HiFiGANGenerator
    from daspeech_amd.synthetic import make_s2st_batch
    m = _released_model("s2st")
    voc = HiFiGANGenerator(conv_backend="hip").cuda().eval()
    gen = S2SNATGenerator(voc, torch.zeros(80, device="cuda"), torch.ones(80, device="cuda"), vocoder_group=8)
    s = make_s2st_batch(32, "cuda", seed=12)
    out = gen.generate(m, s)
    assert len(out) == 32
    ni = s["net_input"]
    with torch.no_grad():                       # the durations the pipeline used, recomputed through the same modules
        enc = m.forward_encoder(ni["src_tokens"], ni["src_lengths"])
        prev = m.initialize_output_tokens_by_src(ni["src_lengths"], max_src_len=ni["src_tokens"].shape[1])
        dec = m.forward_decoder(prev, enc)
        _, _, out_lens, log_dur, _, _ = m.tts(m.adaptor(dec["features"]), dec["features_padding_mask"])
        dur = torch.clamp(torch.round(torch.exp(log_dur) - 1).long(), min=0).masked_fill(dec["features_padding_mask"], 0)
    assert out_lens.tolist() == dur.sum(1).tolist()
    for b, o in enumerate(out):
        n = int(out_lens[b])
        assert o["feature"].shape == (max(n, 1), 80) and torch.isfinite(o["feature"]).all()
        assert o["waveform"].shape[0] == max(n, 1) * 256 and torch.isfinite(o["waveform"]).all() and o["waveform"].abs().max() <= 1.0
    for b in (0, 7, 31):
        alone = voc(out[b]["feature"].t().unsqueeze(0).contiguous())[0, 0]
        assert torch.equal(alone, out[b]["waveform"])


def test_c5_training_step_full_size_properties():
    """BASELINE configs[4] per GPU at its workload size: three optimizer steps of the released architecture on a batch of 32
    (s2s_dag_fastspeech2_loss with GLAT number-random glancing, fp16 autocast + loss scaling as the reference's --fp16, fp32 HIP DAG
    ops, clip-norm 1, Adam).  Properties: finite loss and logging outputs, a finite gradient for EVERY parameter, no DP launch error,
    the loss on the SAME batch goes down."""
    from daspeech_amd import _lib
    from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
    from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
    torch.manual_seed(3)
    model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).cuda().train()
    batch = make_s2st_batch(32, "cuda", seed=4)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=0.01, fused=True)
    scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=2.0 ** 7)
    losses = []
    for i in range(3):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(100)                      # same glancing draw every step: the three losses are comparable
        with torch.autocast("cuda", dtype=torch.float16):
            loss, log = s2s_dag_fastspeech2_loss(model, batch, glat_p="0.5:0.1@200k", update_num=100000)
        assert torch.isfinite(loss) and _lib.last_launch_status() == 0
        assert all(np.isfinite(float(v)) for v in log.values() if isinstance(v, (int, float)) or torch.is_tensor(v) and v.numel() == 1)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        missing = [n for n, p in model.named_parameters() if p.grad is None]
        assert not missing, missing[:8]          # (r02: autocast's weight cache, filled by the no-grad GLAT pass, had cut off 86 of 695)
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        scaler.step(opt); scaler.update()
        losses.append(float(loss))
    assert losses[2] < losses[0], losses


@pytest.mark.parametrize("amp", [None, torch.float16, torch.bfloat16])
def test_glat_two_pass_reaches_every_decoder_weight(amp):
    """The GLAT forward runs the decoder twice: once without gradient to pick the glanced positions, once with
    (s2s_conformer_dag_fastspeech2.py:143-173).  Under torch.autocast the second pass must not reuse the weight copies the first one
    cached without gradient history: every decoder / link-predictor parameter gets a gradient, with and without autocast."""
    from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
    from daspeech_amd.synthetic import make_s2st_batch
    model = small_model().train()
    batch = make_s2st_batch(3, "cuda", seed=2, min_frames=120, max_frames=160)
    with torch.autocast("cuda", dtype=amp or torch.float16, enabled=amp is not None):
        loss, _ = s2s_dag_fastspeech2_loss(model, batch, glat_p="0.5:0.1@200k", update_num=100000)
    loss.backward()
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing[:8]
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_fp16_model_training_step_as_the_reference_trains():
    """The reference's --fp16 scheme (fairseq FP16Optimizer: model.half(), fp16 batch, flat fp32 master, dynamic loss scale — not autocast):
    three steps of the released objective on a small model: finite loss, a gradient for every parameter, the loss on the same batch goes
    down, the fp16 parameters follow the fp32 master, and an overflow halves the scale and skips the update."""
    from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
    from daspeech_amd.fp16_trainer import FP16FlatOptimizer, half_sample
    from daspeech_amd.synthetic import make_s2st_batch
    model = small_model().half().train()
    batch = half_sample(make_s2st_batch(3, "cuda", seed=2, min_frames=120, max_frames=160))
    assert batch["net_input"]["src_tokens"].dtype == torch.float16 and batch["durations"].dtype == torch.long
    opt = FP16FlatOptimizer(model.parameters(), lr=3e-4, init_scale=2.0 ** 7)
    w0 = model.tts.out_proj.weight.detach().clone()
    losses = []
    for i in range(3):
        opt.zero_grad()
        torch.manual_seed(7)                                   # same glancing and dropout draws: comparable losses
        loss, log = s2s_dag_fastspeech2_loss(model, batch, glat_p="0.5:0.1@200k", update_num=100000)
        assert torch.isfinite(loss)
        opt.backward(loss)
        missing = [n for n, p in model.named_parameters() if p.grad is None]
        assert not missing, missing[:8]
        assert all(p.grad.dtype == torch.float16 for p in model.parameters())
        assert opt.step() and np.isfinite(opt.last_grad_norm)
        losses.append(float(loss))
    assert losses[2] < losses[0], losses
    assert not torch.equal(model.tts.out_proj.weight, w0)
    n = model.tts.out_proj.weight.numel()
    views = dict(zip((id(p) for p in opt.params), opt._mviews))
    torch.testing.assert_close(model.tts.out_proj.weight.float(), views[id(model.tts.out_proj.weight)].half().float(), rtol=0, atol=0)
    # overflow: an inf gradient -> the step is skipped, the scale halves, the master is untouched
    scale, before = opt.scaler.loss_scale, opt.master.detach().clone()
    opt.zero_grad()
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    model.tts.out_proj.weight.grad[0, 0] = float("inf")
    assert opt.step() is False and opt.scaler.loss_scale == scale / 2 and torch.equal(opt.master.detach(), before)
