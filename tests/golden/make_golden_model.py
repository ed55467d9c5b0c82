#!/usr/bin/env python3
"""Whole-model goldens from the REFERENCE implementation, generated in the authoring container (not on the GPU box):

  ckpt_manifest.json     every state-dict key -> shape of the reference's S2SConformerDAGFastSpeech2Model built with the README's
                         finetuning flags (README.md:288-323) and a 104-symbol dictionary: what `fairseq-train` would save under
                         ckpt["model"] (fairseq/fairseq/checkpoint_utils.py:288).  Pins `load_reference_state_dict`.
  s2st_reference_e2e.npz the reference's own S2SNATGenerator.generate (DASpeech/generator/s2s_nat_generator.py:49-271) on a seeded
                         filter-bank batch with weights rebuilt from a seed BY PARAMETER NAME (tests/util_inputs.seeded_model_state):
                         decoded tokens, feature lengths, the mel frames of every utterance, and small slices of the intermediates
                         (encoder output, vertex arg-max tokens, links).  Only inputs' seeds and outputs are stored.

The reference imports fairseq, which imports omegaconf / hydra / bitarray / sacrebleu — absent here.  None of them is touched by
model construction or inference, so they are replaced by inert stand-ins for the IMPORT only (`_install_import_stubs`); the model,
the decoder, the generator and every tensor op below are the reference's own code.  `torch.cuda.random.*` is patched to no-ops: the
reference's `torch_seed` context saves the CUDA RNG state unconditionally (s2t_conformer_dag.py:43), which needs a GPU."""
import argparse
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SEED, FRAMES, NSYM = 2024, (420, 333, 260), 100


def _install_import_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Inert:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return Inert()
        def __iter__(self): return iter(())
        def __getattr__(self, n):
            if n.startswith("__") and n.endswith("__"):
                raise AttributeError(n)
            return Inert()

    class DictConfig(dict):
        pass

    class OmegaConf:
        is_config = staticmethod(lambda o: isinstance(o, DictConfig))
        is_dict = staticmethod(lambda o: isinstance(o, DictConfig))
        set_struct = staticmethod(lambda *a, **k: None)
        create = staticmethod(lambda x=None, *a, **k: x)
        to_container = staticmethod(lambda x, *a, **k: x)

    stub("omegaconf", DictConfig=DictConfig, OmegaConf=OmegaConf, open_dict=Inert(), II=lambda x: x, MISSING="???", _utils=Inert())
    stub("omegaconf._utils", is_primitive_type=lambda x: True)
    stub("hydra"); stub("hydra.core"); stub("hydra.core.config_store", ConfigStore=Inert()); stub("hydra.core.global_hydra", GlobalHydra=Inert())
    stub("hydra.experimental", compose=Inert(), initialize=Inert())
    ba = stub("bitarray", bitarray=Inert)
    ba.util = stub("bitarray.util", ba2int=Inert(), int2ba=Inert(), make_endian=Inert())
    sb = stub("sacrebleu", __version__="2.0.0", TOKENIZERS={}, DEFAULT_TOKENIZER="13a", metrics=Inert(),
              BLEU=type("BLEU", (), {"TOKENIZERS": ["none", "13a", "intl", "zh", "ja-mecab", "char"]}))
    sb.tokenizers = stub("sacrebleu.tokenizers", BaseTokenizer=Inert())


def build_reference(**over):
    sys.path.insert(0, os.path.join(REF, "fairseq")); sys.path.insert(0, REF)
    _install_import_stubs()
    import torch
    for fn in ("get_rng_state", "set_rng_state", "manual_seed"):
        setattr(torch.cuda.random, fn, (lambda *a, **k: torch.zeros(1, dtype=torch.uint8)) if fn == "get_rng_state" else (lambda *a, **k: None))
    torch.cuda.manual_seed = lambda *a, **k: None
    import fairseq  # noqa
    import DASpeech  # noqa
    from fairseq.data import Dictionary
    from DASpeech.models.s2s_conformer_dag_fastspeech2 import S2SConformerDAGFastSpeech2Model
    d = Dictionary()
    for i in range(NSYM):
        d.add_symbol(f"p{i}")
    # README.md:288-323 (DASpeech finetuning); pitch / energy ranges come from the data config (any finite range serves)
    kw = dict(arch="s2s_conformer_dag_fastspeech2", share_decoder_input_output_embed=True, pos_enc_type="rel_pos", decoder_learned_pos=True,
              attn_type="espnet", activation_fn="gelu", apply_bert_init=True, encoder_layers=12, encoder_embed_dim=256, encoder_ffn_embed_dim=2048,
              encoder_attention_heads=4, decoder_layers=4, decoder_embed_dim=512, decoder_ffn_embed_dim=2048, decoder_attention_heads=8,
              tts_encoder_layers=4, tts_encoder_embed_dim=256, tts_encoder_attention_heads=4, tts_decoder_layers=4, tts_decoder_embed_dim=256,
              tts_decoder_attention_heads=4, fft_hidden_dim=1024, adaptor_ffn_dim=1024, n_frames_per_step=1, links_feature="feature:position",
              decode_strategy="lookahead", decode_beta=1.0, decode_viterbibeta=1.0, max_source_positions=6000, max_target_positions=1024,
              max_target_audio_positions=1200, src_upsample_scale=0.5, max_transition_length=99999, dropout=0.1, attention_dropout=0.1,
              relu_dropout=0.1, input_feat_per_channel=80, input_channels=1, fp16=False, pitch_min=-4.6600, pitch_max=5.7333, energy_min=-4.9544, energy_max=3.2244)
    kw.update(over)
    args = argparse.Namespace(**kw)
    task = types.SimpleNamespace(target_dictionary=d, tgt_dict=d, source_dictionary=None, data_cfg=types.SimpleNamespace())
    return S2SConformerDAGFastSpeech2Model.build_model(args, task), args, d


def main():
    import torch
    sp = importlib.util.spec_from_file_location("util_inputs", os.path.join(os.path.dirname(HERE), "util_inputs.py"))
    ui = importlib.util.module_from_spec(sp); sp.loader.exec_module(ui)
    model, args, d = build_reference()
    model.eval()
    sd = model.state_dict()
    manifest = {"source": "DASpeech/models/s2s_conformer_dag_fastspeech2.py:46-83 build_model with README.md:288-323 flags, 104-symbol dictionary",
                "vocab_size": len(d), "pad": d.pad(), "bos": d.bos(), "eos": d.eos(), "unk": d.unk(),
                "args": {k: v for k, v in sorted(vars(args).items()) if isinstance(v, (int, float, str, bool))},
                "keys": {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")} for k, v in sd.items()}}
    json.dump(manifest, open(os.path.join(HERE, "ckpt_manifest.json"), "w"), indent=0, sort_keys=False)
    print("manifest:", len(sd), "keys,", sum(v.numel() for v in sd.values()), "elements")

    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}
    w = ui.seeded_model_state(shapes, SEED)
    model.load_state_dict({k: (torch.from_numpy(w[k]) if k in w else v) for k, v in sd.items()})
    from DASpeech.generator.s2s_nat_generator import S2SNATGenerator
    gen = S2SNATGenerator(d, None, types.SimpleNamespace(global_cmvn_stats_npz=None), max_iter=0, adaptive=False)
    src = torch.from_numpy(ui.seeded_fbank(SEED + 1, FRAMES)); lens = torch.tensor(FRAMES)
    with torch.no_grad():
        enc = model.forward_encoder([src, lens])
        prev = model.initialize_output_tokens(enc, src, lens)
        logits, links, feats = model.extract_features(prev.output_tokens, enc, 1, require_links=True)
        dec = model.forward_decoder(prev._replace(step=0, max_step=1), enc)
        out = gen.generate(model, {"net_input": {"src_tokens": src, "src_lengths": lens}}, generate_waveform=False)
        # margins of the discrete decisions (a golden with a near-tie would be a coin flip in fp32 on another device)
        lp = torch.log_softmax(logits.float(), -1)
        top2 = lp.topk(2, -1).values
        tok_margin = float((top2[..., 0] - top2[..., 1])[prev.output_tokens.ne(d.pad())].min())
        dense = model.restore_valid_links(links) if hasattr(model, "restore_valid_links") else None
    store = {"seed": np.int64(SEED), "frames": np.array(FRAMES), "vocab_size": np.int64(len(d)),
             "graph_tokens": prev.output_tokens.numpy(), "vertex_argmax": logits.argmax(-1).numpy(), "tokens": dec.output_tokens.numpy(),
             "n_features": (~dec.features_padding_mask).sum(1).numpy(), "tok_margin": np.float32(tok_margin),
             "encoder_out_slice": enc["encoder_out"][0][:6, :, :8].numpy(), "encoder_len": np.array([int((~m).sum()) for m in enc["encoder_padding_mask"][0]]),
             "links_slice": links[:, :8, :8].float().numpy(), "logits_slice": logits[:, :6, :10].float().numpy()}
    for b, o in enumerate(out):
        store[f"mel{b}"] = o["feature"].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "s2st_reference_e2e.npz"), **store)
    print("tokens per utterance", [int((t != d.pad()).sum()) for t in dec.output_tokens], "mel frames", [o["feature"].shape[0] for o in out], "token margin", tok_margin)
    criterion_golden(model, d, ui)
    s2s_criterion_golden(model, d, ui)
    postnet_golden(ui)


def criterion_golden(model, d, ui):
    """nat_dag_loss_reference.npz: the reference's NATDAGLoss.forward (DASpeech/criterions/nat_dag_loss.py:164-300) — GLAT two-pass forward
    with number-random glancing at p = 0.5, force-emit, torch DAG ops (its own --torch-dag-* CPU path) — and loss.backward() through the
    whole reference model (eval mode: no dropout draws), for the seeded weights above.  The two random draws of the glancing are replayed
    from the seed in the reference's order (randn(B, L) then rand(B, L); nothing else draws in eval mode) and stored."""
    import torch
    from DASpeech.criterions.nat_dag_loss import NATDAGLoss
    frames = (300, 236)
    src = torch.from_numpy(ui.seeded_fbank(SEED + 7, frames)); lens = torch.tensor(frames)
    rng = np.random.default_rng(SEED + 8)
    tl = (17, 12)
    T = max(tl) + 2
    tgt = np.full((len(frames), T), d.pad(), np.int64)
    for b, n in enumerate(tl):
        tgt[b, 0] = d.bos(); tgt[b, 1:n + 1] = rng.integers(4, len(d), n); tgt[b, n + 1] = d.eos()
    cfg = types.SimpleNamespace(label_smoothing=0, glat_p="0.5", glance_strategy="number-random", no_force_emit=False,
                                torch_dag_logsoftmax_gather=True, torch_dag_best_alignment=True, torch_dag_loss=True)
    crit = NATDAGLoss(cfg, types.SimpleNamespace(tgt_dict=d, target_dictionary=d))
    model.eval()
    model.zero_grad(set_to_none=True)
    captured = {}
    fwd = model.forward

    def spy(*a, **k):
        out = fwd(*a, **k)
        captured.update({k2: v for k2, v in out.items() if k2 in ("keep_word_mask", "glat_accu", "glat_keep")})
        return out
    model.forward = spy
    sample = {"net_input": {"src_tokens": src, "src_lengths": lens}, "target": torch.from_numpy(tgt), "update_num": 10}
    draw_seed = 4242
    torch.manual_seed(draw_seed)
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    model.forward = fwd
    L = int(max(frames) * 0.5)
    torch.manual_seed(draw_seed)
    noise = torch.randn(len(frames), L); unif = torch.rand(len(frames), L)
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    pick = ["decoder.gate_linear.weight", "decoder.query_linear.bias", "decoder.key_linear.bias", "encoder.linear.bias", "decoder.layers.3.fc2.bias",
            "encoder.conformer_layers.11.final_layer_norm.weight", "decoder.embed_positions.weight"]
    store = {"frames": np.array(frames), "target": tgt, "draw_seed": np.int64(draw_seed), "noise": noise.numpy(), "unif": unif.numpy(),
             "loss": np.float64(float(loss)), "keep_word_mask": captured["keep_word_mask"].numpy(), "glat_accu": np.float32(float(captured["glat_accu"])),
             "glat_keep": np.float32(float(captured["glat_keep"])), "n_grads": np.int64(len(grads))}
    for k in ("ntokens", "nvalidtokens", "nsentences", "invalid_nsentences"):
        store["log_" + k] = np.int64(int(log[k]))
    store["log_dag_nll_loss"] = np.float64(float(log["dag_nll-loss"]))
    for k in pick:
        g = grads[k]
        store["grad:" + k] = (g if g.numel() <= 4096 else g.reshape(-1)[:4096]).numpy().astype(np.float32)
        store["gradnorm:" + k] = np.float64(float(g.double().norm()))
    store["grad_total_norm"] = np.float64(float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values()))))
    np.savez_compressed(os.path.join(HERE, "nat_dag_loss_reference.npz"), **store)
    print("criterion: loss", float(loss), "glanced", captured["keep_word_mask"].sum(1).tolist(), "grads", len(grads), "total norm", float(store["grad_total_norm"]))


def _oracle_alpha_beta_function():
    """CPU stand-in for the ONE function of this path the reference has no CPU implementation of: `dag_loss_with_alpha_beta`
    (DASpeech/custom_ops/dag_loss.py:123-188 is CUDA-only and `S2SDAGFastSpeech2Loss` asserts `not torch_dag_loss`,
    s2s_dag_fastspeech2_loss.py:75).  Same autograd contract as `DagLossWithAlphaBetaFunc` — forward returns `(res, (alpha, beta))` with
    `res = beta[:,0,0]` when a gradient is required and `alpha[b, T_b-1, L_b-1]` otherwise, `beta` = zeros without gradient
    (dag_loss.cu:339-340: both tables are `at::zeros`, the beta kernel is launched only with `require_gradient`), backward =
    `dag_loss_backward` — evaluated by the fp64 C oracle (oracle/dag_oracle.c, itself pinned to the reference's torch_dag_loss and its
    autograd gradients by tests/test_oracle_golden.py) and rounded to fp32 like the CUDA op's outputs.  `main` asserts, on the very
    tensors of every case, that this stand-in's loss equals the REFERENCE's own `torch_dag_loss` on the restored dense links."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import dag_oracle as orc

    class OracleDagLossWithAlphaBeta(torch.autograd.Function):
        calls = []

        @staticmethod
        def forward(ctx, match_all, links, output_length, target_length):
            need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
            m = match_all.detach().double().numpy(); k = links.detach().double().numpy()
            ol = output_length.numpy(); tl = target_length.numpy()
            a = orc.dag_alpha(m, k, ol, tl, np.float64)
            if need:
                b = orc.dag_beta(m, k, ol, tl, np.float64)
                res = b[:, 0, 0].copy()
            else:
                b = np.zeros_like(a)
                res = a[np.arange(a.shape[0]), tl - 1, ol - 1].copy()
            ctx.np = (a, b, m, k, ol, tl)
            OracleDagLossWithAlphaBeta.calls.append({"match": m, "links": k, "ol": ol, "tl": tl, "res": res, "need": need})
            alpha = torch.from_numpy(a).float(); beta = torch.from_numpy(b).float()
            ctx.mark_non_differentiable(alpha, beta)
            return torch.from_numpy(res).to(match_all.dtype), (alpha, beta)

        @staticmethod
        def backward(ctx, grad_output, unused):
            a, b, m, k, ol, tl = ctx.np
            gm, gl = orc.dag_grad(grad_output.double().numpy(), a, b, m, k, ol, tl, np.float64)
            return torch.from_numpy(gm).float(), torch.from_numpy(gl).float(), None, None

    return OracleDagLossWithAlphaBeta


def _inplace_gather_function():
    """CPU stand-in with the CUDA operator's SIDE EFFECT: `dag_logsoftmax_gather_inplace` overwrites the logits with their soft-max when a
    gradient is required (dag_loss.py:238-299, logsoftmax_gather.cu:303-305) — which the reference's argmax strategy then READS
    (s2s_dag_fastspeech2_loss.py:215 clones `outputs["word_ins"]["out"]` after the loss has run on it, so on the CUDA path the alignment is
    taken on log_softmax(softmax(x))).  Values through torch's own log_softmax / softmax; backward as dag_loss.py:283-297."""
    import torch

    class InplaceGather(torch.autograd.Function):
        @staticmethod
        def forward(ctx, word_ins_out, select_idx):
            need = ctx.needs_input_grad[0]
            lp = torch.log_softmax(word_ins_out.detach(), -1, dtype=torch.float32)
            match = lp.gather(-1, select_idx)
            if need:
                word_ins_out.data.copy_(lp.exp())
                ctx.mark_dirty(word_ins_out)
                ctx.save_for_backward(word_ins_out, select_idx)
            return word_ins_out, match

        @staticmethod
        def backward(ctx, g_word, g_match):
            sm, idx = ctx.saved_tensors
            gx = sm * (-g_match.sum(-1, keepdim=True))
            gx.scatter_add_(-1, idx, g_match)
            return gx, None

    return InplaceGather.apply


def s2s_criterion_golden(model, d, ui, only=None, out="s2s_dag_fastspeech2_loss_reference.npz"):
    """s2s_dag_fastspeech2_loss_reference.npz: the reference's S2SDAGFastSpeech2Loss.forward (DASpeech/criterions/
    s2s_dag_fastspeech2_loss.py:93-306) + loss.backward() through its whole model, seeded weights (as the other goldens), eval-mode modules
    (no dropout draws), criterion in training mode unless stated.  Cases (prefix):
      expect/        --training-strategy expect, --glat-p 0.5 --glance-strategy number-random, --tts-loss-weight 5, update_num 10
      argmax/        --training-strategy argmax on the reference's torch gather (logits untouched)
      argmaxq/       --training-strategy argmax with the CUDA gather's in-place soft-max side effect emulated (the alignment is then taken
                     on log_softmax(softmax(x)), as on the reference's GPU path)
      frozen/        expect with --dag-freezing-steps 100 > update_num 10: the DA-Transformer forward runs without gradient, beta = zeros
      eval/          expect with criterion.eval() (validation): no gradient anywhere
    Torch DAG ops (--torch-dag-logsoftmax-gather / --torch-dag-best-alignment) wherever the reference has them; `dag_loss_with_alpha_beta`
    through `_oracle_alpha_beta_function` (see there).  Random draws (glancing) are replayed from a seed and stored."""
    import torch
    import DASpeech.criterions.s2s_dag_fastspeech2_loss as ref_mod
    ref_ops = sys.modules["DASpeech.custom_ops.dag_loss"]           # (the package re-exports a function of the same name)
    OracleAB = _oracle_alpha_beta_function()
    ref_mod.dag_loss_with_alpha_beta = OracleAB.apply
    orig_gather = ref_mod.dag_logsoftmax_gather_inplace
    frames = (300, 236, 264)
    src = torch.from_numpy(ui.seeded_fbank(SEED + 17, frames)); lens = torch.tensor(frames)
    rng = np.random.default_rng(SEED + 18)
    tl = (17, 12, 15)
    B = len(frames)
    T = max(tl) + 2
    tgt = np.full((B, T), d.pad(), np.int64)
    dur = np.zeros((B, T - 1), np.int64)
    pit = np.zeros((B, T - 1), np.float32); ene = np.zeros((B, T - 1), np.float32)
    for b, n in enumerate(tl):
        tgt[b, 0] = d.bos(); tgt[b, 1:n + 1] = rng.integers(4, len(d), n); tgt[b, n + 1] = d.eos()
        dur[b, :n] = 1 + rng.poisson(3.0, n)                                   # <eos> gets 0 frames
        pit[b, :n + 1] = rng.uniform(-4.66, 5.73, n + 1); ene[b, :n + 1] = rng.uniform(-4.95, 3.22, n + 1)
    mel_len = dur.sum(1)
    mel = rng.standard_normal((B, int(mel_len.max()), 80)).astype(np.float32)
    for b in range(B):
        mel[b, mel_len[b]:] = 0
    store = {"frames": np.array(frames), "target_text": tgt, "target_text_lengths": np.array(tl) + 2, "durations": dur, "pitches": pit,
             "energies": ene, "target_audio": mel, "target_audio_lengths": mel_len}
    L = int(max(frames) * 0.5)
    pick = ["decoder.gate_linear.weight", "decoder.query_linear.bias", "encoder.linear.bias", "decoder.layers.3.fc2.bias",
            "encoder.conformer_layers.11.final_layer_norm.weight", "decoder.embed_positions.weight", "adaptor.fc1.weight", "adaptor.fc2.bias",
            "tts.encoder_fft_layers.0.self_attn.q_proj.weight", "tts.encoder_fft_layers.3.ffn.ffn.2.bias", "tts.var_adaptor.embed_pitch.weight",
            "tts.var_adaptor.embed_energy.weight", "tts.var_adaptor.duration_predictor.proj.weight", "tts.var_adaptor.pitch_predictor.conv1.0.weight",
            "tts.var_adaptor.energy_predictor.ln2.weight", "tts.decoder_fft_layers.3.layer_norm.weight", "tts.out_proj.weight", "tts.pos_emb_alpha",
            "tts.dec_pos_emb_alpha"]
    cases = {"expect": dict(training_strategy="expect"), "argmax": dict(training_strategy="argmax"),
             "argmaxq": dict(training_strategy="argmax", _inplace=True), "frozen": dict(training_strategy="expect", dag_freezing_steps=100),
             "eval": dict(training_strategy="expect", _eval=True)}
    for name, kw in cases.items():
        if only and name not in only:
            continue
        inplace = kw.pop("_inplace", False); is_eval = kw.pop("_eval", False)
        cfg = types.SimpleNamespace(label_smoothing=0, glat_p="0.5", glance_strategy="number-random", no_force_emit=False,
                                    torch_dag_logsoftmax_gather=not inplace, torch_dag_best_alignment=True, torch_dag_loss=False,
                                    tts_loss_weight=5.0, dag_freezing_steps=-1)
        cfg.__dict__.update(kw)
        ref_mod.dag_logsoftmax_gather_inplace = _inplace_gather_function() if inplace else orig_gather
        crit = ref_mod.S2SDAGFastSpeech2Loss(cfg, types.SimpleNamespace(tgt_dict=d, target_dictionary=d))
        crit.train(not is_eval)
        model.eval(); model.zero_grad(set_to_none=True)
        captured = {}
        fwd, tts_fwd, ad_fwd = model.forward, model.tts.forward, model.adaptor.forward

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
        model.forward, model.adaptor.forward, model.tts.forward = spy, ad_spy, tts_spy
        sample = {"net_input": {"src_tokens": src.clone(), "src_lengths": lens}, "target_text": torch.from_numpy(tgt),
                  "target_text_lengths": torch.from_numpy(store["target_text_lengths"]), "durations": torch.from_numpy(dur),
                  "pitches": torch.from_numpy(pit), "energies": torch.from_numpy(ene), "target_audio": torch.from_numpy(mel),
                  "target_audio_lengths": torch.from_numpy(mel_len), "update_num": 10}
        draw_seed = 4343
        OracleAB.calls.clear()
        torch.manual_seed(draw_seed)
        loss, sample_size, log = crit(model, sample)
        if loss.requires_grad:
            loss.backward()
        model.forward, model.adaptor.forward, model.tts.forward = fwd, ad_fwd, tts_fwd
        # the stand-in against the REFERENCE's own torch_dag_loss on the very tensors of this case (dense links restored as
        # dag_loss.py:439-448 does)
        c = OracleAB.calls[-1]
        dense = model.restore_valid_links(torch.from_numpy(c["links"]).float()).double()      # (fp32 -> fp64 round trip is exact)
        ref_loss = ref_ops.torch_dag_loss(torch.from_numpy(c["match"]), dense, torch.from_numpy(c["ol"]), torch.from_numpy(c["tl"]))
        assert np.allclose(ref_loss.numpy(), c["res"], rtol=1e-10, atol=1e-9), (ref_loss, c["res"])
        torch.manual_seed(draw_seed)
        noise = torch.randn(B, L); unif = torch.rand(B, L)
        grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        pre = name + "/"
        store.update({pre + "draw_seed": np.int64(draw_seed), pre + "noise": noise.numpy(), pre + "unif": unif.numpy(),
                      pre + "loss": np.float64(float(loss)), pre + "sample_size": np.int64(sample_size),
                      pre + "keep_word_mask": captured["keep_word_mask"].numpy(), pre + "glat_accu": np.float32(float(captured["glat_accu"])),
                      pre + "glat_keep": np.float32(float(captured["glat_keep"])), pre + "n_grads": np.int64(len(grads)),
                      pre + "requires_grad": np.bool_(loss.requires_grad), pre + "dag_res": c["res"],
                      pre + "adaptor_in": captured["adaptor_in"].numpy().astype(np.float32), pre + "tts_padding_mask": captured["tts_padding_mask"].numpy()})
        for k in ("loss", "dag-loss", "tts-loss", "l1-loss", "dur-loss", "pitch-loss", "energy-loss"):
            store[pre + "log:" + k] = np.float64(float(log[k]))
        for k in ("ntokens", "nvalidtokens", "nsentences", "invalid_nsentences", "sample_size"):
            store[pre + "log:" + k] = np.int64(int(log[k]))
        store[pre + "log:glat_acc"] = np.float64(float(log["glat_acc"])); store[pre + "log:glat_keep"] = np.float64(float(log["glat_keep"]))
        for k in pick:
            if k not in grads:
                continue
            g = grads[k]
            store[pre + "grad:" + k] = (g if g.numel() <= 1024 else g.reshape(-1)[:1024]).numpy().astype(np.float32)
            store[pre + "gradnorm:" + k] = np.float64(float(g.double().norm()))
        store[pre + "grad_names"] = np.array(sorted(grads))
        if grads:
            store[pre + "grad_total_norm"] = np.float64(float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values()))))
        print(f"s2s criterion [{name}]: loss {float(loss):.6f} dag {float(log['dag-loss']):.6f} l1 {float(log['l1-loss']):.6f} dur {float(log['dur-loss']):.6f} "
              f"pitch {float(log['pitch-loss']):.6f} energy {float(log['energy-loss']):.6f} grads {len(grads)} adaptor_in {tuple(captured['adaptor_in'].shape)}")
    ref_mod.dag_logsoftmax_gather_inplace = orig_gather
    np.savez_compressed(os.path.join(HERE, out), **store)
    return store


def postnet_golden(ui):
    """tts_postnet_reference.npz: the reference model built with --add-postnet (fastspeech2_noemb.py:128-136: fairseq's tacotron2 Postnet,
    5 x [Conv1d k=5 -> BatchNorm1d -> tanh] of width 512) on seeded weights: the `tts.postnet.*` state-dict keys and shapes, the TTS
    half alone teacher-forced (mel and mel_post = mel + postnet(mel), :171-173), and the criterion's `expect` case through
    s2s_criterion_golden with the second L1 term (s2s_dag_fastspeech2_loss.py:281-282) -> s2s_postnet_loss_reference.npz."""
    import torch
    model, args, d = build_reference(add_postnet=True)
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}
    w = ui.seeded_model_state(shapes, SEED)
    model.load_state_dict({k: (torch.from_numpy(w[k]) if k in w else v) for k, v in sd.items()})
    model.eval()
    rng = np.random.default_rng(SEED + 31)
    B, N = 3, 11
    x = rng.standard_normal((B, N, 256)).astype(np.float32)
    lens = np.array([11, 8, 3])
    pad = np.arange(N)[None, :] >= lens[:, None]
    dur = rng.poisson(3.0, (B, N)); dur[pad] = 0
    pit = rng.uniform(-4.66, 5.73, (B, N)).astype(np.float32); ene = rng.uniform(-4.95, 3.22, (B, N)).astype(np.float32)
    with torch.no_grad():
        mel, mel_post, out_lens, log_dur, pitch, energy = model.tts(torch.from_numpy(x.copy()), torch.from_numpy(pad), durations=torch.from_numpy(dur),
                                                                    pitches=torch.from_numpy(pit), energies=torch.from_numpy(ene))
    keys = {k: list(v.shape) for k, v in sd.items() if k.startswith("tts.postnet.")}
    np.savez_compressed(os.path.join(HERE, "tts_postnet_reference.npz"), x=x, pad=pad, dur=dur, pitch_in=pit, energy_in=ene, mel=mel.numpy(),
                        mel_post=mel_post.numpy(), out_lens=out_lens.numpy(), postnet_keys=np.array(sorted(keys)),
                        postnet_shapes=np.array([json.dumps(keys[k]) for k in sorted(keys)]), n_keys=np.int64(len(sd)))
    print("postnet: keys", len(keys), "of", len(sd), "mel", tuple(mel.shape), "|mel_post - mel| max", float((mel_post - mel).abs().max()))
    s2s_criterion_golden(model, d, ui, only=["expect"], out="s2s_postnet_loss_reference.npz")


def s2s_only():
    import torch
    sp = importlib.util.spec_from_file_location("util_inputs", os.path.join(os.path.dirname(HERE), "util_inputs.py"))
    ui = importlib.util.module_from_spec(sp); sp.loader.exec_module(ui)
    model, args, d = build_reference()
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}
    w = ui.seeded_model_state(shapes, SEED)
    model.load_state_dict({k: (torch.from_numpy(w[k]) if k in w else v) for k, v in sd.items()})
    s2s_criterion_golden(model, d, ui)


if __name__ == "__main__":
    if "--s2s-only" in sys.argv:
        s2s_only()
    elif "--postnet-only" in sys.argv:
        sp = importlib.util.spec_from_file_location("util_inputs", os.path.join(os.path.dirname(HERE), "util_inputs.py"))
        ui = importlib.util.module_from_spec(sp); sp.loader.exec_module(ui)
        postnet_golden(ui)
    else:
        main()
