#!/usr/bin/env python3
"""Golden vectors for the TTS side, produced by the REFERENCE's own modules (authoring container only).

  * fairseq/fairseq/models/text_to_speech/hifigan.py : Generator (:111-179), V1 topology at reduced width (initial channels 32
    instead of 512 so the weights fit in a fixture), weight-norm removed as hifi-gan/inference_e2e.py:44-45 does.
  * fairseq/fairseq/models/text_to_speech/fastspeech2.py : LengthRegulator (:98-114), VariancePredictor (:117-151),
    VarianceAdaptor (:154-216) — imported with the stub recipe of SURVEY.md §9.4 (fake `fairseq.*` modules, no fairseq install).
  * FULL-WIDTH vectors (hifigan_v1_seeded.npz, fastspeech2_noemb_seeded.npz): the released widths (HiFi-GAN V1 config: 13.9 M
    parameters; FastSpeech2EncoderNoEmb of DASpeech/models/fastspeech2_noemb.py:69-174 at README.md:295-300 sizes: 4 + 4 FFT layers of
    256 x 1024, kernel 9) with weights drawn by `tests/util_inputs.seeded_weights` from a seed — the fixture holds {seed, inputs,
    reference outputs} only and the tests rebuild the identical weights on the GPU box.
Outputs are data only (inputs, weights or their seed, expected outputs).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util_inputs import seeded_weights  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def hifigan_golden():
    hg = load_by_path("ref_hifigan", f"{REF}/fairseq/fairseq/models/text_to_speech/hifigan.py")
    cfg = json.load(open(f"{REF}/hifi-gan/config_v1.json"))
    cfg["upsample_initial_channel"] = 32
    torch.manual_seed(7)
    g = hg.Generator(cfg)
    g.remove_weight_norm()
    g.eval()
    rng = np.random.default_rng(7)
    with torch.no_grad():
        for n, p in g.named_parameters():
            fan = p[0].numel() if p.dim() > 1 else 1
            scale = (1.0 / np.sqrt(fan)) if p.dim() > 1 else 0.05
            p.copy_(torch.from_numpy((rng.standard_normal(tuple(p.shape)) * scale).astype(np.float32)))
    mel = torch.from_numpy(rng.standard_normal((2, 80, 11)).astype(np.float32))
    with torch.no_grad():
        wav = g(mel)
    out = {"mel": mel.numpy(), "wav": wav.numpy(),
           "cfg_json": np.frombuffer(json.dumps({k: cfg[k] for k in ("upsample_rates", "upsample_kernel_sizes",
                                     "upsample_initial_channel", "resblock_kernel_sizes", "resblock_dilation_sizes")}).encode(), dtype=np.uint8)}
    for n, p in g.state_dict().items():
        out["w:" + n] = p.numpy()
    np.savez_compressed(os.path.join(HERE, "hifigan_small.npz"), **out)
    print("hifigan_small", wav.shape, float(wav.abs().max()), sum(p.numel() for p in g.parameters()))


def stub_fairseq():
    def mod(name):
        m = types.ModuleType(name); sys.modules[name] = m; return m
    fs = mod("fairseq")
    u = mod("fairseq.utils"); u.item = lambda t: t.item() if hasattr(t, "item") else t
    fs.utils = u
    d = mod("fairseq.data"); du = mod("fairseq.data.data_utils")

    def lengths_to_padding_mask(lens):
        bsz, max_lens = lens.size(0), int(torch.max(lens).item())
        mask = torch.arange(max_lens).to(lens.device).view(1, max_lens).expand(bsz, -1)
        return mask >= lens.view(bsz, 1).expand(-1, max_lens)
    du.lengths_to_padding_mask = lengths_to_padding_mask
    d.data_utils = du
    ms = mod("fairseq.models")
    ms.FairseqEncoder = nn.Module; ms.FairseqEncoderModel = nn.Module
    ms.register_model = lambda *a, **k: (lambda c: c)
    ms.register_model_architecture = lambda *a, **k: (lambda c: c)
    tts = mod("fairseq.models.text_to_speech")
    hub = mod("fairseq.models.text_to_speech.hub_interface"); hub.TTSHubInterface = object
    tac = mod("fairseq.models.text_to_speech.tacotron2"); tac.Postnet = nn.Module
    mods = mod("fairseq.modules")

    class FairseqDropout(nn.Module):
        def __init__(self, p, module_name=None):
            super().__init__(); self.p = p

        def forward(self, x, inplace=False):
            return nn.functional.dropout(x, self.p, self.training)
    mods.FairseqDropout = FairseqDropout
    mods.LayerNorm = nn.LayerNorm
    mods.MultiheadAttention = nn.MultiheadAttention
    mods.PositionalEmbedding = lambda *a, **k: None
    fs.models = ms; fs.modules = mods
    return lengths_to_padding_mask


def fastspeech2_golden():
    stub_fairseq()
    fs2 = load_by_path("ref_fs2", f"{REF}/fairseq/fairseq/models/text_to_speech/fastspeech2.py")
    rng = np.random.default_rng(11)
    # LengthRegulator
    x = torch.from_numpy(rng.standard_normal((3, 9, 6)).astype(np.float32))
    dur = torch.from_numpy(rng.poisson(2.0, (3, 9))).long()
    dur[0, 2] = 0; dur[2, :] = torch.tensor([1, 0, 0, 3, 0, 0, 0, 0, 2])
    out, lens = fs2.LengthRegulator()(x, dur)
    # VarianceAdaptor at inference (predicted durations / pitch / energy)
    args = types.SimpleNamespace(encoder_embed_dim=16, var_pred_hidden_dim=16, var_pred_kernel_size=3, var_pred_dropout=0.5,
                                 var_pred_n_bins=32, pitch_min=-2.0, pitch_max=3.0, energy_min=-1.5, energy_max=2.5)
    torch.manual_seed(5)
    va = fs2.VarianceAdaptor(args).eval()
    with torch.no_grad():
        for p in va.parameters():
            p.copy_(torch.from_numpy((rng.standard_normal(tuple(p.shape)) * (0.4 if p.dim() > 1 else 0.2)).astype(np.float32)))
        va.duration_predictor.proj.bias.fill_(0.9)              # durations around exp(0.9)-1 ~ 1.5 frames
    xin = torch.from_numpy(rng.standard_normal((2, 7, 16)).astype(np.float32))
    pad = torch.tensor([[False] * 7, [False] * 5 + [True] * 2])
    with torch.no_grad():
        y, out_lens, log_dur, pitch, energy = va(xin, pad)
    store = {"lr_x": x.numpy(), "lr_dur": dur.numpy(), "lr_out": out.numpy(), "lr_lens": lens.numpy(),
             "va_x": xin.numpy(), "va_pad": pad.numpy(), "va_out": y.numpy(), "va_out_lens": out_lens.numpy(),
             "va_log_dur": log_dur.numpy(), "va_pitch": pitch.numpy(), "va_energy": energy.numpy(),
             "va_pitch_bins": va.pitch_bins.numpy(), "va_energy_bins": va.energy_bins.numpy()}
    for n, p in va.state_dict().items():
        store["va_w:" + n] = p.numpy()
    np.savez_compressed(os.path.join(HERE, "fastspeech2_pieces.npz"), **store)
    print("fastspeech2_pieces: lr", tuple(out.shape), lens.tolist(), "va", tuple(y.shape), out_lens.tolist())


def hifigan_full_golden(seed=20240):
    """The reference Generator (hifi-gan/models.py:75-119 twin in fairseq) at the released V1 widths on seeded weights."""
    hg = load_by_path("ref_hifigan", f"{REF}/fairseq/fairseq/models/text_to_speech/hifigan.py")
    cfg = json.load(open(f"{REF}/hifi-gan/config_v1.json"))
    g = hg.Generator(cfg)
    g.remove_weight_norm()
    g.eval()
    sd = g.state_dict()
    w = seeded_weights({k: tuple(v.shape) for k, v in sd.items()}, seed)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    rng = np.random.default_rng(seed)
    lens = np.array([24, 17, 9])
    mel = (rng.standard_normal((3, 80, 24)) * 1.2 - 4.0).astype(np.float32)        # de-normalised log-mel magnitudes
    out = {"seed": np.int64(seed), "mel": mel, "lens": lens}
    with torch.no_grad():
        for b, n in enumerate(lens):                                              # one file at a time, inference_e2e.py:47-56
            out[f"wav{b}"] = g(torch.from_numpy(mel[b:b + 1, :, :n])).numpy()[0, 0]
    np.savez_compressed(os.path.join(HERE, "hifigan_v1_seeded.npz"), **out)
    print("hifigan_v1_seeded", [out[f"wav{b}"].shape for b in range(3)], [float(np.abs(out[f"wav{b}"]).max()) for b in range(3)],
          [float(np.abs(out[f"wav{b}"]).mean()) for b in range(3)], sum(p.numel() for p in g.parameters()))


def stub_fairseq_for_noemb():
    """On top of stub_fairseq(): what DASpeech/models/fastspeech2_noemb.py and FFTLayer need.  PositionalEmbedding is the REAL
    fairseq SinusoidalPositionalEmbedding loaded from its file; MultiheadAttention is a parameter-compatible stand-in that calls
    torch's multi_head_attention_forward the way fairseq's module does (multihead_attention.py:539-561)."""
    import torch.nn.functional as F
    lengths_to_padding_mask = stub_fairseq()
    u = sys.modules["fairseq.utils"]

    def make_positions(tensor, padding_idx, onnx_trace=False):            # fairseq/utils.py:256-266
        mask = tensor.ne(padding_idx).int()
        return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx
    u.make_positions = make_positions
    spe = load_by_path("ref_sinpos", f"{REF}/fairseq/fairseq/modules/sinusoidal_positional_embedding.py")

    def PositionalEmbedding(num_embeddings, embedding_dim, padding_idx, learned=False):      # modules/positional_embedding.py (non-learned branch)
        return spe.SinusoidalPositionalEmbedding(embedding_dim, padding_idx, init_size=num_embeddings + padding_idx + 1)

    class MultiheadAttention(nn.Module):
        def __init__(self, embed_dim, num_heads, dropout=0.0, self_attention=False, **kw):
            super().__init__()
            self.embed_dim, self.num_heads = embed_dim, num_heads
            self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(embed_dim, embed_dim) for _ in range(4))

        def forward(self, query, key, value, key_padding_mask=None, need_weights=False, **kw):
            return F.multi_head_attention_forward(
                query, key, value, self.embed_dim, self.num_heads, torch.empty([0]),
                torch.cat((self.q_proj.bias, self.k_proj.bias, self.v_proj.bias)), None, None, False, 0.0,
                self.out_proj.weight, self.out_proj.bias, False, key_padding_mask.bool() if key_padding_mask is not None else None,
                need_weights, None, use_separate_proj_weight=True, q_proj_weight=self.q_proj.weight, k_proj_weight=self.k_proj.weight,
                v_proj_weight=self.v_proj.weight)
    mods = sys.modules["fairseq.modules"]
    mods.PositionalEmbedding = PositionalEmbedding
    mods.MultiheadAttention = MultiheadAttention
    ms = sys.modules["fairseq.models"]

    class FairseqEncoder(nn.Module):
        def __init__(self, dictionary=None):
            super().__init__()
            self.dictionary = dictionary
    ms.FairseqEncoder = FairseqEncoder
    ms.FairseqEncoderModel = nn.Module
    fs2 = load_by_path("fairseq.models.text_to_speech.fastspeech2", f"{REF}/fairseq/fairseq/models/text_to_speech/fastspeech2.py")
    sys.modules["fairseq.models.text_to_speech.fastspeech2"] = fs2
    fs2.Postnet = sys.modules["fairseq.models.text_to_speech.tacotron2"].Postnet
    return lengths_to_padding_mask


NOEMB_SKIP = ("embed_tokens.", "embed_positions.")       # parameters of the reference module the NoEmb forward never reads


def fastspeech2_noemb_golden(seed=20385):
    """FastSpeech2EncoderNoEmb.forward (fastspeech2_noemb.py:140-174) with FFTLayer / PositionwiseFeedForward (fastspeech2.py:42-95) at
    the released sizes, inference (predicted durations / pitch / energy) and teacher-forced (training inputs)."""
    stub_fairseq_for_noemb()
    noemb = load_by_path("ref_noemb", f"{REF}/DASpeech/models/fastspeech2_noemb.py")
    args = types.SimpleNamespace(
        n_frames_per_step=1, output_frame_dim=80, tts_encoder_embed_dim=256, tts_decoder_embed_dim=256, speaker_embed_dim=64, dropout=0.2,
        max_target_positions=1200, tts_encoder_attention_heads=4, tts_decoder_attention_heads=4, fft_hidden_dim=1024, fft_kernel_size=9,
        attention_dropout=0.0, tts_encoder_layers=4, tts_decoder_layers=4, add_postnet=False, var_pred_hidden_dim=256, var_pred_kernel_size=3,
        var_pred_dropout=0.5, var_pred_n_bins=256, pitch_min=-4.6600, pitch_max=5.7333, energy_min=-4.9544, energy_max=3.2244)
    src_dict = types.SimpleNamespace(pad=lambda: 1, __len__=lambda: 8)

    class Dict8:
        def pad(self):
            return 1

        def __len__(self):
            return 8
    enc = noemb.FastSpeech2EncoderNoEmb(args, Dict8(), None).eval()
    sd = enc.state_dict()
    names = {k: tuple(v.shape) for k, v in sd.items() if not k.startswith(NOEMB_SKIP)}
    w = seeded_weights(names, seed)
    # durations around exp(1.4) - 1 ~ 3 frames per phoneme, spread by the predictor's input
    w["var_adaptor.duration_predictor.proj.bias"] = np.full((1,), 1.4, np.float32)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    rng = np.random.default_rng(seed)
    B, N = 3, 13
    x = rng.standard_normal((B, N, 256)).astype(np.float32)
    lens = np.array([13, 9, 4])
    pad = np.arange(N)[None, :] >= lens[:, None]
    store = {"seed": np.int64(seed), "x": x, "pad": pad, "dur_bias": np.float32(1.4)}
    with torch.no_grad():
        mel, _, out_lens, log_dur, pitch, energy = enc(torch.from_numpy(x.copy()), torch.from_numpy(pad))
    # the integer decisions of the adaptor (duration rounding, pitch / energy buckets) must not sit on an edge: an fp32-rounding-sized
    # difference in an implementation under test would flip them.  Walk the seed until every margin is comfortable.
    v = (np.exp(log_dur.numpy()) - 1)[~pad]
    margins = [np.abs((v - np.floor(v)) - 0.5).min()]
    for val, bins in ((pitch, enc.var_adaptor.pitch_bins), (energy, enc.var_adaptor.energy_bins)):
        margins.append(np.abs(val.numpy()[~pad][:, None] - bins.numpy()[None, :]).min())
    if min(margins) < 5e-4:
        print("seed", seed, "margins", margins, "-> next seed")
        return fastspeech2_noemb_golden(seed + 1)
    store.update({"inf_mel": mel.numpy(), "inf_out_lens": out_lens.numpy(), "inf_log_dur": log_dur.numpy(), "inf_pitch": pitch.numpy(),
                  "inf_energy": energy.numpy()})
    # teacher-forced: the training call (s2s_dag_fastspeech2_loss.py:267-273)
    dur = rng.poisson(3.0, (B, N)); dur[pad] = 0; dur[0, 3] = 0
    pit = (rng.random((B, N)) * 10.39 - 4.66).astype(np.float32); ene = (rng.random((B, N)) * 8.18 - 4.95).astype(np.float32)
    with torch.no_grad():
        mel2, _, out_lens2, log_dur2, pitch2, energy2 = enc(torch.from_numpy(x.copy()), torch.from_numpy(pad), durations=torch.from_numpy(dur),
                                                            pitches=torch.from_numpy(pit), energies=torch.from_numpy(ene))
    store.update({"tf_dur": dur, "tf_pitch_in": pit, "tf_energy_in": ene, "tf_mel": mel2.numpy(), "tf_out_lens": out_lens2.numpy(),
                  "tf_log_dur": log_dur2.numpy(), "tf_pitch": pitch2.numpy(), "tf_energy": energy2.numpy()})
    # one FFTLayer on its own (first encoder layer), ragged mask
    with torch.no_grad():
        y = enc.encoder_fft_layers[0](torch.from_numpy(x.copy()), torch.from_numpy(pad))
    store["fft0_out"] = y.numpy()
    store["param_names"] = np.array(sorted(names))
    np.savez_compressed(os.path.join(HERE, "fastspeech2_noemb_seeded.npz"), **store)
    print("fastspeech2_noemb_seeded: mel", tuple(mel.shape), out_lens.tolist(), "tf", tuple(mel2.shape), out_lens2.tolist(),
          "mel abs max", float(mel.abs().max()), "params", sum(int(np.prod(s)) for s in names.values()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "full"]
    if "small" in which:
        hifigan_golden()
        fastspeech2_golden()
    if "full" in which:
        hifigan_full_golden()
        fastspeech2_noemb_golden()
