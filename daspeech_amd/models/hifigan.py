"""HiFi-GAN generator (V1 topology) — inference only, weight-norm folded at load time.

What it mirrors: `Generator.forward` + `ResBlock.forward` of hifi-gan/models.py:35-43,100-119 (twin:
fairseq/fairseq/models/text_to_speech/hifigan.py:111-170) as driven by hifi-gan/inference_e2e.py:34-57.
Parameter names follow the reference checkpoint (`generator_v1`: conv_pre, ups.N, resblocks.N.convs1/2.M, conv_post) so a
reference state dict — with or without weight-norm (`weight_g`/`weight_v`) — loads directly.

Compute: activations stay channels-first [B, C, T]; every conv goes through `conv_backend`:
  * "torch"  — torch.nn.functional conv1d / conv_transpose1d (MIOpen on ROCm); the functional reference of this repo;
  * "hip"    — hand-written gfx950 kernels: fp32 activations / weights as the reference (operand splitting on the fp16 matrix cores,
                daspeech_amd/csrc/hifigan_conv_f32.hip; waveform within 1e-4 of the reference generator);
  * "hip_fp16" — the same stack with fp16 activation / weight storage (csrc/hifigan_conv.hip): ~3x faster, 1.5e-3 off the reference.
"""
import json
from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor, nn

# hifi-gan/config_v1.json:11-15
HIFIGAN_V1 = {
    "upsample_rates": [8, 8, 2, 2],
    "upsample_kernel_sizes": [16, 16, 4, 4],
    "upsample_initial_channel": 512,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "model_in_dim": 80,
}

LRELU_SLOPE = 0.1


class _ResBlock1(nn.Module):
    """x + conv_d1(lrelu(conv_dk(lrelu(x)))) for the three dilations, chained (models.py:35-43)."""

    def __init__(self, channels: int, kernel: int, dilations):
        super().__init__()
        self.kernel, self.dilations = kernel, list(dilations)
        self.convs1 = nn.ModuleList(nn.Conv1d(channels, channels, kernel, dilation=d, padding=(kernel - 1) * d // 2)
                                    for d in self.dilations)
        self.convs2 = nn.ModuleList(nn.Conv1d(channels, channels, kernel, padding=(kernel - 1) // 2) for _ in self.dilations)


class HiFiGANGenerator(nn.Module):
    def __init__(self, cfg: Dict = None, conv_backend: str = "torch"):
        super().__init__()
        cfg = dict(HIFIGAN_V1 if cfg is None else cfg)
        cfg.setdefault("model_in_dim", 80)
        self.cfg = cfg
        self.conv_backend = conv_backend
        c0 = cfg["upsample_initial_channel"]
        self.rates, self.up_kernels = list(cfg["upsample_rates"]), list(cfg["upsample_kernel_sizes"])
        self.rb_kernels = list(cfg["resblock_kernel_sizes"])
        self.conv_pre = nn.Conv1d(cfg["model_in_dim"], c0, 7, padding=3)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        ch = c0
        for i, (u, k) in enumerate(zip(self.rates, self.up_kernels)):
            self.ups.append(nn.ConvTranspose1d(ch, ch // 2, k, stride=u, padding=(k - u) // 2))
            ch //= 2
            for rk, rd in zip(self.rb_kernels, cfg["resblock_dilation_sizes"]):
                self.resblocks.append(_ResBlock1(ch, rk, rd))
        self.conv_post = nn.Conv1d(ch, 1, 7, padding=3)
        self.hop = 1
        for u in self.rates:
            self.hop *= u

    # ------------------------------------------------------------------ checkpoint loading
    @staticmethod
    def fold_weight_norm(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """w = g * v / ||v||  (norm over all dims but 0) — what torch's remove_weight_norm leaves (inference_e2e.py:44-45)."""
        out = {}
        for k, v in sd.items():
            if k.endswith(".weight_g"):
                base = k[: -len(".weight_g")]
                vv = sd[base + ".weight_v"]
                norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (vv.dim() - 1))
                out[base + ".weight"] = v * vv / norm
            elif k.endswith(".weight_v"):
                continue
            else:
                out[k] = v
        return out

    def load_reference_state_dict(self, sd: Dict[str, Tensor]):
        if "generator" in sd and isinstance(sd["generator"], dict):          # hifi-gan checkpoint file layout
            sd = sd["generator"]
        self.load_state_dict(self.fold_weight_norm(sd), strict=True)
        self._hip_runner = None                       # packed fp16 weights of the HIP backend are stale now
        return self

    @classmethod
    def from_config_json(cls, path: str, **kw):
        return cls(json.load(open(path)), **kw)

    # ------------------------------------------------------------------ forward
    def _conv(self, x: Tensor, m: nn.Conv1d, pre_slope: float = None, residual: Tensor = None) -> Tensor:
        """[residual +] conv(leaky_relu(x)) — torch backend (the HIP backend runs the whole stack in HiFiGANHipRunner)."""
        if pre_slope is not None:
            x = F.leaky_relu(x, pre_slope)
        y = F.conv1d(x, m.weight, m.bias, dilation=m.dilation, padding=m.padding)
        return y if residual is None else y + residual

    def _up(self, x: Tensor, m: nn.ConvTranspose1d, pre_slope: float) -> Tensor:
        return F.conv_transpose1d(F.leaky_relu(x, pre_slope), m.weight, m.bias, stride=m.stride, padding=m.padding)

    def forward(self, mel: Tensor, lengths: Tensor = None) -> Tensor:
        """mel [B, 80, T] (de-normalised log-mel) -> waveform [B, 1, T*256] in (-1, 1).  `lengths` [B]: mel frames per utterance of a
        zero-padded batch; waveform[b, :, :lengths[b]*hop] then equals the utterance vocoded on its own, which is what the reference
        does (hifi-gan/inference_e2e.py:47-56) — the HIP backend masks per layer inside its kernels, the torch backend loops."""
        if self.conv_backend in ("hip", "hip_fp16") and mel.is_cuda:
            key = (self.conv_backend,) + tuple((p.data_ptr(), p._version) for p in self.parameters())
            if getattr(self, "_hip_runner", None) is None or getattr(self, "_hip_runner_key", None) != key:
                from ..hifigan_ops import HiFiGANHipRunner
                prec = "fp32" if self.conv_backend == "hip" else "fp16"
                self._hip_runner, self._hip_runner_key = HiFiGANHipRunner(self, precision=prec), key   # re-packed whenever a weight changed
            return self._hip_runner(mel, lengths)
        if lengths is not None:
            out = mel.new_zeros(mel.shape[0], 1, mel.shape[2] * self.hop)
            for b, n in enumerate(lengths.tolist()):
                if n > 0:
                    out[b, :, : n * self.hop] = self.forward(mel[b:b + 1, :, :n])[0]
            return out
        x = self._conv(mel, self.conv_pre)
        nk = len(self.rb_kernels)
        for i, up in enumerate(self.ups):
            x = self._up(x, up, LRELU_SLOPE)                                   # models.py:103-104
            acc = None
            for j in range(nk):                                                # MRF: mean of the three kernel sizes  :105-111
                rb = self.resblocks[i * nk + j]
                y = x
                for c1, c2 in zip(rb.convs1, rb.convs2):
                    h = self._conv(y, c1, LRELU_SLOPE)
                    y = self._conv(h, c2, LRELU_SLOPE, residual=y)
                acc = y if acc is None else acc + y
            x = acc / nk
        x = self._conv(x, self.conv_post, 0.01)                                # default-slope leaky_relu  :112
        return torch.tanh(x)

    @torch.no_grad()
    def synthesize_int16(self, mel: Tensor) -> Tensor:
        """inference_e2e.py:50-53: audio * 32768 as int16."""
        return (self.forward(mel).squeeze(1) * 32768.0).to(torch.int16)
