#!/usr/bin/env python3
"""Golden vectors for the TTS side, produced by the REFERENCE's own modules (authoring container only).

  * fairseq/fairseq/models/text_to_speech/hifigan.py : Generator (:111-179), V1 topology at reduced width (initial channels 32
    instead of 512 so the weights fit in a fixture), weight-norm removed as hifi-gan/inference_e2e.py:44-45 does.
  * fairseq/fairseq/models/text_to_speech/fastspeech2.py : LengthRegulator (:98-114), VariancePredictor (:117-151),
    VarianceAdaptor (:154-216) — imported with the stub recipe of SURVEY.md §9.4 (fake `fairseq.*` modules, no fairseq install).
Outputs are data only (inputs, weights, expected outputs).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

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


if __name__ == "__main__":
    hifigan_golden()
    fastspeech2_golden()
