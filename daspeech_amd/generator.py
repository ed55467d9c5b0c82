"""S2SNATGenerator — fbank -> waveform in one process (the reference splits it over generate_features.py and
hifi-gan/inference_e2e.py with .npy files in between).

Mirrors DASpeech/generator/s2s_nat_generator.py:49-285: forward_encoder -> initialize_output_tokens -> forward_decoder (graph
decode, max_iter = 0) -> adaptor -> tts with predicted durations -> gcmvn de-normalisation (:273-281) -> per-utterance slices
(:260-269) -> vocoder.
"""
from typing import Dict, List, Optional

import torch
from torch import Tensor


class S2SNATGenerator:
    def __init__(self, vocoder=None, gcmvn_mean: Optional[Tensor] = None, gcmvn_std: Optional[Tensor] = None):
        self.vocoder, self.mean, self.std = vocoder, gcmvn_mean, gcmvn_std

    def gcmvn_denormalize(self, x: Tensor) -> Tensor:
        if self.mean is None:
            return x
        return x * self.std.view(1, 1, -1).to(x) + self.mean.view(1, 1, -1).to(x)

    @torch.no_grad()
    def generate(self, model, sample: Dict, generate_waveform: bool = True) -> List[Dict[str, Tensor]]:
        net = sample["net_input"]
        enc = model.forward_encoder(net["src_tokens"], net["src_lengths"])
        prev = model.initialize_output_tokens_by_src(net["src_lengths"])
        dec = model.forward_decoder(prev, enc)
        tts_in = model.adaptor(dec["features"])
        mel, out_lens, _, _, _ = model.tts(tts_in, dec["features_padding_mask"])
        mel = self.gcmvn_denormalize(mel)
        wav = None
        if generate_waveform and self.vocoder is not None and mel.shape[1] > 0:
            fmask = torch.arange(mel.shape[1], device=mel.device).unsqueeze(0) >= out_lens.unsqueeze(1)
            wav = self.vocoder(mel.masked_fill(fmask.unsqueeze(-1), 0).transpose(1, 2)).squeeze(1)   # batched, by length
        hop = getattr(self.vocoder, "hop", 256)
        lens = out_lens.tolist()
        res = []
        for b, n in enumerate(lens):
            feat = mel[b, :n] if n > 0 else mel.new_zeros(1, mel.shape[-1])              # zeros[1,80] when empty (:263)
            item = {"tokens": dec["output_tokens"][b], "feature": feat}
            if wav is not None:
                item["waveform"] = wav[b, : max(n, 1) * hop]
            res.append(item)
        return res
