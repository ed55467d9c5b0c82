"""S2SNATGenerator — fbank -> waveform in one process (the reference splits it over generate_features.py and
hifi-gan/inference_e2e.py with .npy files in between).

Mirrors DASpeech/generator/s2s_nat_generator.py:49-285: forward_encoder -> initialize_output_tokens -> forward_decoder (graph
decode, max_iter = 0) -> adaptor -> tts with predicted durations -> gcmvn de-normalisation (:273-281) -> per-utterance slices
(:260-269) -> vocoder.
"""
from typing import Dict, List, Optional

import torch
from torch import Tensor


def _accepts_lengths(vocoder) -> bool:
    """Does the vocoder callable take the per-utterance `lengths` keyword (models/hifigan.HiFiGANGenerator does)?"""
    if vocoder is None:
        return False
    import inspect
    try:
        fn = vocoder.forward if hasattr(vocoder, "forward") else vocoder
        params = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return "lengths" in params or any(p.kind == inspect.Parameter.VAR_KEYWORD for p in params.values())


class S2SNATGenerator:
    """`generate(model, sample)` is the reference's call (one batch, in order).  `submit` / `flush` / `generate_batches` run the same
    two stages as a two-deep pipeline over consecutive batches: the acoustic model of batch k is ISSUED (≈14 ms of host time for ≈1000
    small kernels at B=32 — the stage is launch bound, the GPU idles between them) while the vocoder of batch k-1 (≈50 launches, 11 ms
    of dense MFMA work) runs on a second, lower-priority stream and fills those gaps.  Same kernels, same results, batch by batch."""

    def __init__(self, vocoder=None, gcmvn_mean: Optional[Tensor] = None, gcmvn_std: Optional[Tensor] = None,
                 vocoder_group: int = 8):
        self.vocoder, self.mean, self.std, self.vocoder_group = vocoder, gcmvn_mean, gcmvn_std, vocoder_group
        self._vocoder_takes_lengths = _accepts_lengths(vocoder)
        self._side = None            # vocoder stream of the pipelined mode
        self._pending = None         # acoustic stage issued, vocoder not yet: (stage-1 outputs, completion event)
        self._inflight = None        # vocoder issued on the side stream: (results, completion event)

    def gcmvn_denormalize(self, x: Tensor) -> Tensor:
        if self.mean is None:
            return x
        return x * self.std.view(1, 1, -1).to(x) + self.mean.view(1, 1, -1).to(x)

    # ---- stage 1: fbank -> mel (forward_encoder -> graph decode -> adaptor -> tts; s2s_nat_generator.py:49-258), no host sync
    def _acoustic(self, model, sample: Dict) -> Dict[str, Tensor]:
        net = sample["net_input"]
        enc = model.forward_encoder(net["src_tokens"], net["src_lengths"])
        # the encoder's 4x subsampling keeps lengths on the device; the padded frame count is a host integer already
        prev = model.initialize_output_tokens_by_src(net["src_lengths"], max_src_len=net["src_tokens"].shape[1])
        dec = model.forward_decoder(prev, enc)
        tts_in = model.adaptor(dec["features"])
        mel, mel_post, out_lens, _, _, _ = model.tts(tts_in, dec["features_padding_mask"])
        if mel_post is not None:                                                                             # s2s_nat_generator.py:254-255
            mel = mel_post
        return {"mel": self.gcmvn_denormalize(mel), "out_lens": out_lens, "tokens": dec["output_tokens"]}

    # ---- stage 2: mel -> waveforms on the CURRENT stream; `lens` = out_lens on the host
    def _vocode(self, mel: Tensor, out_lens: Tensor, lens: List[int]) -> List[Optional[Tensor]]:
        hop = getattr(self.vocoder, "hop", 256)
        wavs: List[Optional[Tensor]] = [None] * len(lens)
        if self.vocoder is None or mel.shape[1] == 0:
            return wavs
        # vocode in length-sorted groups: the batch is padded to each GROUP's maximum, not the batch maximum
        # (the reference vocodes one file at a time, hifi-gan/inference_e2e.py:47-56)
        # (sorted and regrouped on the device: the only host data needed are the lengths)
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
            # per-utterance lengths go down to the vocoder: each utterance's samples are those of vocoding it alone.  (A vocoder
            # callable that does not take `lengths` — any module with the reference generator's forward(mel) — gets the padded group.)
            x = sub.masked_fill(fmask.unsqueeze(-1), 0).transpose(1, 2)
            w = (self.vocoder(x, lengths=len_sorted[g0:g0 + gsz].clamp(min=1)) if self._vocoder_takes_lengths else self.vocoder(x)).squeeze(1)
            for k, i in enumerate(idx):
                wavs[i] = w[k, : max(lens[i], 1) * hop]
        return wavs

    @staticmethod
    def _assemble(ac: Dict[str, Tensor], lens: List[int], wavs: List[Optional[Tensor]]) -> List[Dict[str, Tensor]]:
        mel, res = ac["mel"], []
        for b, n in enumerate(lens):
            feat = mel[b, :n] if n > 0 else mel.new_zeros(1, mel.shape[-1])              # zeros[1,80] when empty (:263)
            item = {"tokens": ac["tokens"][b], "feature": feat}
            if wavs[b] is not None:
                item["waveform"] = wavs[b]
            res.append(item)
        return res

    @torch.no_grad()
    def generate(self, model, sample: Dict, generate_waveform: bool = True) -> List[Dict[str, Tensor]]:
        ac = self._acoustic(model, sample)
        lens = ac["out_lens"].tolist()
        wavs = self._vocode(ac["mel"], ac["out_lens"], lens) if generate_waveform else [None] * len(lens)
        return self._assemble(ac, lens, wavs)

    # ------------------------------------------------------------------------------------------------ pipelined over batches
    def _issue_vocoder_of_pending(self):
        ac, ev = self._pending
        self._pending = None
        ev.synchronize()                                   # the host needs the mel lengths of that batch
        lens = ac["out_lens"].tolist()
        main = torch.cuda.current_stream()
        if self._side is None:
            least, _greatest = torch.cuda.Stream.priority_range()
            self._side = torch.cuda.Stream(device=ac["mel"].device, priority=least)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            wavs = self._vocode(ac["mel"], ac["out_lens"], lens)
            vev = torch.cuda.Event()
            vev.record(self._side)
        for t in (ac["mel"], ac["out_lens"]):
            t.record_stream(self._side)                    # allocated on the main stream, read on the side stream
        for w in wavs:
            if w is not None:
                w.record_stream(main)                      # and the other way round for the consumer
        self._inflight = (self._assemble(ac, lens, wavs), vev)

    @torch.no_grad()
    def submit(self, model, sample: Dict) -> Optional[List[Dict[str, Tensor]]]:
        """Feed the next batch; returns the finished results of the PREVIOUS batch (None for the first).  The current stream is made to
        wait for their vocoder, so they can be used on it without further synchronisation."""
        done = None
        if self._pending is not None:
            self._issue_vocoder_of_pending()               # vocoder(k-1) is queued before the long host-side issue of acoustic(k)
            done = self._inflight
        ac = self._acoustic(model, sample)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._pending = (ac, ev)
        if done is None:
            return None
        self._inflight = None
        torch.cuda.current_stream().wait_event(done[1])
        return done[0]

    @torch.no_grad()
    def flush(self) -> Optional[List[Dict[str, Tensor]]]:
        """Vocode and return the last submitted batch (None if nothing is pending)."""
        if self._pending is None:
            return None
        self._issue_vocoder_of_pending()
        res, vev = self._inflight
        self._inflight = None
        torch.cuda.current_stream().wait_event(vev)
        return res

    def generate_sharded(self, model, sample: Dict, rank: int = None, world_size: int = None, generate_waveform: bool = True):
        """Data-parallel inference over one pool of utterances: every rank takes its length-balanced shard (distributed.balanced_shards on
        src_lengths — SURVEY §8e), generates it, and returns (indices into the pool, results) — no collective: utterances are independent."""
        import torch.distributed as dist
        from .distributed import balanced_shards, shard_sample
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        idx = balanced_shards(sample["net_input"]["src_lengths"].cpu(), world_size)[rank]
        if not idx:
            return [], []
        return idx, self.generate(model, shard_sample(sample, idx), generate_waveform)

    def generate_batches(self, model, samples):
        """for results in generator.generate_batches(model, batch_iterable): ...  — same results as calling generate() per batch."""
        for sample in samples:
            out = self.submit(model, sample)
            if out is not None:
                yield out
        out = self.flush()
        if out is not None:
            yield out


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
