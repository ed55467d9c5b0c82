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


if __name__ == "__main__":
    main()
