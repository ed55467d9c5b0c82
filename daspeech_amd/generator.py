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
    def __init__(self, vocoder=None, gcmvn_mean: Optional[Tensor] = None, gcmvn_std: Optional[Tensor] = None,
                 vocoder_group: int = 8):
        self.vocoder, self.mean, self.std, self.vocoder_group = vocoder, gcmvn_mean, gcmvn_std, vocoder_group

    def gcmvn_denormalize(self, x: Tensor) -> Tensor:
        if self.mean is None:
            return x
        return x * self.std.view(1, 1, -1).to(x) + self.mean.view(1, 1, -1).to(x)

    @torch.no_grad()
    def generate(self, model, sample: Dict, generate_waveform: bool = True) -> List[Dict[str, Tensor]]:
        net = sample["net_input"]
        enc = model.forward_encoder(net["src_tokens"], net["src_lengths"])
        # the encoder's 4x subsampling keeps lengths on the device; the padded frame count is a host integer already
        prev = model.initialize_output_tokens_by_src(net["src_lengths"], max_src_len=net["src_tokens"].shape[1])
        dec = model.forward_decoder(prev, enc)
        tts_in = model.adaptor(dec["features"])
        mel, out_lens, _, _, _ = model.tts(tts_in, dec["features_padding_mask"])
        mel = self.gcmvn_denormalize(mel)
        hop = getattr(self.vocoder, "hop", 256)
        lens = out_lens.tolist()
        wavs = [None] * len(lens)
        if generate_waveform and self.vocoder is not None and mel.shape[1] > 0:
            # vocode in length-sorted groups: the batch is padded to each GROUP's maximum, not the batch maximum
            # (the reference vocodes one file at a time, hifi-gan/inference_e2e.py:47-56)
            # (sorted and regrouped on the device: the only host data needed are the lengths fetched above)
            order = sorted(range(len(lens)), key=lambda i: lens[i])
            dev_order = torch.argsort(out_lens, stable=True)
            mel_sorted = mel.index_select(0, dev_order)
            len_sorted = out_lens.index_select(0, dev_order)
            gsz = max(1, self.vocoder_group)
            for g0 in range(0, len(order), gsz):
                idx = order[g0:g0 + gsz]
                gmax = max(1, max(lens[i] for i in idx))
                sub = mel_sorted[g0:g0 + gsz, :gmax]
                fmask = torch.arange(gmax, device=mel.device).unsqueeze(0) >= len_sorted[g0:g0 + gsz].unsqueeze(1)
                # per-utterance lengths go down to the vocoder: each utterance's samples are those of vocoding it alone
                w = self.vocoder(sub.masked_fill(fmask.unsqueeze(-1), 0).transpose(1, 2), lengths=len_sorted[g0:g0 + gsz].clamp(min=1)).squeeze(1)
                for k, i in enumerate(idx):
                    wavs[i] = w[k, : max(lens[i], 1) * hop]
        res = []
        for b, n in enumerate(lens):
            feat = mel[b, :n] if n > 0 else mel.new_zeros(1, mel.shape[-1])              # zeros[1,80] when empty (:263)
            item = {"tokens": dec["output_tokens"][b], "feature": feat}
            if wavs[b] is not None:
                item["waveform"] = wavs[b]
            res.append(item)
        return res


MAX_WAV_VALUE = 32768.0          # hifi-gan/meldataset.py:13, inference_e2e.py:52


def dump_results(results_path, sample_ids, results, sampling_rate: int = 22050, write_features: bool = True, write_waveforms: bool = True):
    """Optional file sinks in the reference's formats, for its downstream ASR-BLEU scripts:
      feat/<id>.npy                 float32 [80, T]   (generate_features.py:87-91 — the feature transposed)
      wav/<id>_generated_e2e.wav    int16 mono        (hifi-gan/inference_e2e.py:50-56 — audio * 32768 cast to int16)
    `results` is the list `S2SNATGenerator.generate` returns.  Returns the written paths."""
    import os
    import numpy as np
    from scipy.io.wavfile import write as wav_write
    written = []
    feat_dir, wav_dir = os.path.join(results_path, "feat"), os.path.join(results_path, "wav")
    for sid, item in zip(sample_ids, results):
        if write_features and item.get("feature") is not None:
            os.makedirs(feat_dir, exist_ok=True)
            path = os.path.join(feat_dir, f"{sid}.npy")
            np.save(path, item["feature"].detach().float().cpu().numpy().transpose(1, 0))
            written.append(path)
        if write_waveforms and item.get("waveform") is not None:
            os.makedirs(wav_dir, exist_ok=True)
            path = os.path.join(wav_dir, f"{sid}_generated_e2e.wav")
            audio = (item["waveform"].detach().float().cpu().reshape(-1) * MAX_WAV_VALUE).numpy().astype("int16")
            wav_write(path, sampling_rate, audio)
            written.append(path)
    return written
