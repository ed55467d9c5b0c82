"""HIP (MFMA) execution of the HiFi-GAN generator — host side of include/daspeech_hifigan.h.

`HiFiGANHipRunner(generator)` re-packs the weights of a `daspeech_amd.models.HiFiGANGenerator` (reference-checkpoint layout) once:
  Conv1d           weight [Cout,Cin,K]        -> taps [K][Cout][Cin] fp16, shift_k = (k-(K-1)/2)*dilation
  ConvTranspose1d  weight [Cin,Cout,2u]       -> 2 taps [2][u*Cout][Cin] fp16 (phase-major rows), shifts {0,-1}, pad u/2
and runs Generator.forward (hifi-gan/models.py:100-119) as a chain of `dsp_hifigan_conv` launches on channels-last fp16
activations; the residual add of each ResBlock1 unit (models.py:41-42) and the mean over the three kernel sizes (:105-111) are
fused into the conv epilogues.  fp16 storage / fp32 accumulate: the waveform differs from the fp32 torch path by ~1e-3 (tested).
"""
import ctypes
from typing import List

import torch
from torch import Tensor

from . import _lib

OUT_STORE, OUT_ACCUM, OUT_UPSAMPLE = 0, 1, 2


class _Layer:
    __slots__ = ("w", "bias", "shifts", "ntaps", "CI", "M", "Cout", "mode", "u", "pad")


def _shifts_array(sh: List[int]):
    return (ctypes.c_int * len(sh))(*sh)


class HiFiGANHipRunner:
    def __init__(self, gen):
        dev = next(gen.parameters()).device
        assert dev.type == "cuda", "HiFiGANHipRunner needs the generator on a GPU"
        self.dev = dev
        self.nk = len(gen.rb_kernels)
        self.hop = gen.hop
        self.in_dim = gen.conv_pre.weight.shape[1]
        self.in_pad = (self.in_dim + 31) // 32 * 32
        self.pre = self._conv_layer(gen.conv_pre, ci_pad=self.in_pad)
        self.ups = [self._up_layer(u) for u in gen.ups]
        self.blocks = []
        for rb in gen.resblocks:
            self.blocks.append([(self._conv_layer(c1), self._conv_layer(c2)) for c1, c2 in zip(rb.convs1, rb.convs2)])
        cp = gen.conv_post
        self.post_w = cp.weight.detach()[0].t().contiguous().float()          # [K][C]
        self.post_b = float(cp.bias.detach()[0])
        self.post_k = cp.weight.shape[2]

    def _conv_layer(self, m, ci_pad=None):
        L = _Layer()
        w = m.weight.detach()                                   # [Cout, Cin, K]
        Cout, Cin, K = w.shape
        if ci_pad and ci_pad != Cin:
            w = torch.nn.functional.pad(w, (0, 0, 0, ci_pad - Cin))
            Cin = ci_pad
        L.w = w.permute(2, 0, 1).contiguous().to(torch.float16)
        L.bias = m.bias.detach().float().contiguous() if m.bias is not None else None
        d = m.dilation[0]
        L.shifts = [(k - (K - 1) // 2) * d for k in range(K)]
        L.ntaps, L.CI, L.M, L.Cout, L.mode, L.u, L.pad = K, Cin, Cout, Cout, OUT_STORE, 1, 0
        return L

    def _up_layer(self, m):
        L = _Layer()
        w = m.weight.detach()                                   # [Cin, Cout, K = 2u]
        Cin, Cout, K = w.shape
        u = m.stride[0]
        assert K == 2 * u and m.padding[0] == (K - u) // 2, "expects the HiFi-GAN upsampler geometry (kernel 2u, pad u/2)"
        L.w = w.permute(2, 1, 0).reshape(2, u * Cout, Cin).contiguous().to(torch.float16)      # tap j, row (r, co): k = j*u + r
        L.bias = m.bias.detach().float().contiguous() if m.bias is not None else None
        L.shifts = [0, -1]
        L.ntaps, L.CI, L.M, L.Cout, L.mode, L.u, L.pad = 2, Cin, u * Cout, Cout, OUT_UPSAMPLE, u, (K - u) // 2
        return L

    def _run(self, L, x: Tensor, slope: float, res: Tensor = None, out: Tensor = None, mode=None, scale: float = 1.0) -> Tensor:
        B, T, CI = x.shape
        assert CI == L.CI and x.dtype == torch.float16 and x.is_contiguous()
        mode = L.mode if mode is None else mode
        Tout = T * L.u if L.mode == OUT_UPSAMPLE else T
        if out is None:
            out = torch.empty((B, Tout, L.Cout), dtype=torch.float16, device=x.device)
        lib = _lib.load()
        rc = lib.dsp_hifigan_conv(_lib.ptr(x), _lib.ptr(L.w), _lib.ptr(L.bias), _lib.ptr(res), _lib.ptr(out), B, T, CI, L.M, L.ntaps,
                                  _shifts_array(L.shifts), float(slope), float(scale), int(mode), L.u, L.pad, Tout, L.Cout,
                                  _lib.current_stream_handle())
        _lib.check(rc, "dsp_hifigan_conv")
        return out

    @torch.no_grad()
    def __call__(self, mel: Tensor) -> Tensor:
        """mel [B, 80, T] fp32 -> waveform [B, 1, T*hop] fp32."""
        lib = _lib.load()
        B, C, T = mel.shape
        with torch.cuda.device(mel.device):
            mt = mel.detach().float().transpose(1, 2).contiguous()
            x = torch.empty((B, T, self.in_pad), dtype=torch.float16, device=mel.device)
            _lib.check(lib.dsp_hifigan_pack_input(_lib.ptr(mt), _lib.ptr(x), B, T, C, self.in_pad, _lib.current_stream_handle()),
                       "dsp_hifigan_pack_input")
            x = self._run(self.pre, x, 1.0)
            for i, up in enumerate(self.ups):
                x = self._run(up, x, 0.1)                                           # lrelu(0.1) -> ConvTranspose1d
                acc = None
                for j in range(self.nk):
                    y = x
                    units = self.blocks[i * self.nk + j]
                    for n, (c1, c2) in enumerate(units):
                        h = self._run(c1, y, 0.1)
                        if n + 1 < len(units):
                            y = self._run(c2, h, 0.1, res=y)
                        elif acc is None:                                           # last unit: fold the MRF mean in
                            acc = self._run(c2, h, 0.1, res=y, scale=1.0 / self.nk)
                        else:
                            self._run(c2, h, 0.1, res=y, out=acc, mode=OUT_ACCUM, scale=1.0 / self.nk)
                x = acc
            Tw = x.shape[1]
            wav = torch.empty((B, Tw), dtype=torch.float32, device=mel.device)
            _lib.check(lib.dsp_hifigan_post(_lib.ptr(x), _lib.ptr(self.post_w), self.post_b, _lib.ptr(wav), B, Tw, x.shape[2], self.post_k,
                                            0.01, _lib.current_stream_handle()), "dsp_hifigan_post")
        return wav.unsqueeze(1)


def lrelu_conv1d(*a, **k):          # pragma: no cover - per-layer entry points are not used; the runner owns the whole stack
    raise NotImplementedError("use HiFiGANHipRunner")


lrelu_conv_transpose1d = lrelu_conv1d
