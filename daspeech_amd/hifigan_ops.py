"""HIP (MFMA) execution of the HiFi-GAN generator — host side of include/daspeech_hifigan.h.

`HiFiGANHipRunner(generator)` re-packs the weights of a `daspeech_amd.models.HiFiGANGenerator` (reference-checkpoint layout) once:
  Conv1d           weight [Cout,Cin,K]        -> taps [K][Cout][Cin] fp16, shift_k = (k-(K-1)/2)*dilation
  ConvTranspose1d  weight [Cin,Cout,2u]       -> 2 taps [2][u*Cout][Cin] fp16 (phase-major rows), shifts {0,-1}, pad u/2
and runs Generator.forward (hifi-gan/models.py:100-119) as a chain of `dsp_hifigan_conv` launches on channels-last fp16
activations; the residual add of each ResBlock1 unit (models.py:41-42) and the mean over the three kernel sizes (:105-111) are
fused into the conv epilogues.  Two arithmetic modes (`precision`):
  "fp16"  fp16 activations + weights, fp32 accumulate (csrc/hifigan_conv.hip): the waveform differs from the reference's fp32 generator by
          ~1.5e-3 — NARROWER than the reference, the fast mode;
  "fp32"  fp32 activations + weights (the reference's arithmetic, hifi-gan/inference_e2e.py:47-56) with split operands on the fp16 matrix
          cores (csrc/hifigan_conv_f32.hip): waveform within 1e-4 of the reference (tests/test_tts_golden.py).
"""
import ctypes
from typing import List

import torch
from torch import Tensor

from . import _lib

OUT_STORE, OUT_ACCUM, OUT_UPSAMPLE = 0, 1, 2
MAX_TAPS = 16


class HgLayerDesc(ctypes.Structure):          # include/daspeech_hifigan.h: dsp_hg_layer
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("res", ctypes.c_void_p),
                ("out", ctypes.c_void_p),
                ("T", ctypes.c_int), ("CI", ctypes.c_int), ("M", ctypes.c_int), ("ntaps", ctypes.c_int),
                ("shifts", ctypes.c_int * MAX_TAPS),
                ("pre_slope", ctypes.c_float), ("scale", ctypes.c_float),
                ("out_mode", ctypes.c_int), ("up_u", ctypes.c_int), ("up_pad", ctypes.c_int), ("Tout", ctypes.c_int),
                ("Cout", ctypes.c_int), ("w2", ctypes.c_void_p), ("bias2", ctypes.c_void_p)]


class _Layer:
    __slots__ = ("w", "w_lo", "bias", "shifts", "ntaps", "CI", "M", "Cout", "mode", "u", "pad", "dil")


def _shifts_array(sh: List[int]):
    return (ctypes.c_int * len(sh))(*sh)


def pack_weights(w_tap_major: Tensor) -> Tensor:
    """[ntaps][M][CI] fp16 -> the MFMA fragment order the kernels read (include/daspeech_hifigan.h: dsp_hifigan_pack_weights)."""
    lib = _lib.load()
    w = w_tap_major.contiguous()
    K, M, CI = w.shape
    assert w.dtype == torch.float16 and w.is_cuda
    out = torch.empty((lib.dsp_hifigan_packed_weight_elems(K, M, CI),), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.dsp_hifigan_pack_weights(_lib.ptr(w), _lib.ptr(out), K, M, CI, _lib.current_stream_handle()), "dsp_hifigan_pack_weights")
    return out


def pack_weights_f32(w_tap_major: Tensor) -> Tensor:
    """[ntaps][M][CI] fp32 -> ONE fp16 buffer [hi | lo] in fragment order for the split-precision kernels (dsp_hifigan_pack_weights_f32)."""
    lib = _lib.load()
    w = w_tap_major.contiguous().float()
    K, M, CI = w.shape
    n = lib.dsp_hifigan_packed_weight_elems(K, M, CI)
    out = torch.empty((2 * n,), dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.dsp_hifigan_pack_weights_f32(_lib.ptr(w), _lib.ptr(out), K, M, CI, _lib.current_stream_handle()),
                   "dsp_hifigan_pack_weights_f32")
    return out


class HiFiGANHipRunner:
    def __init__(self, gen, fuse_units: bool = True, precision: str = "fp16"):
        """fuse_units: run a ResBlock unit (conv, conv, residual) as ONE launch where dsp_hifigan_resunit supports its shape
        (C <= 128); False keeps the layer-at-a-time chain (same bits, 2.5x the activation traffic) for comparison.
        precision: "fp16" (fp16 storage, fp32 accumulate) or "fp32" (the reference's arithmetic by operand splitting)."""
        assert precision in ("fp16", "fp32")
        self.precision = precision
        self.f32 = precision == "fp32"
        self.fuse_units = fuse_units
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
        self._plans = {}

    def _pack(self, L, w_tap_major: Tensor):
        if self.f32:
            L.w, L.w_lo = pack_weights_f32(w_tap_major), None
        else:
            L.w, L.w_lo = pack_weights(w_tap_major.to(torch.float16)), None

    def _conv_layer(self, m, ci_pad=None):
        L = _Layer()
        w = m.weight.detach()                                   # [Cout, Cin, K]
        Cout, Cin, K = w.shape
        if ci_pad and ci_pad != Cin:
            w = torch.nn.functional.pad(w, (0, 0, 0, ci_pad - Cin))
            Cin = ci_pad
        self._pack(L, w.permute(2, 0, 1).contiguous())
        L.bias = m.bias.detach().float().contiguous() if m.bias is not None else None
        d = m.dilation[0]
        L.shifts = [(k - (K - 1) // 2) * d for k in range(K)]
        L.ntaps, L.CI, L.M, L.Cout, L.mode, L.u, L.pad, L.dil = K, Cin, Cout, Cout, OUT_STORE, 1, 0, d
        return L

    def _up_layer(self, m):
        L = _Layer()
        w = m.weight.detach()                                   # [Cin, Cout, K = 2u]
        Cin, Cout, K = w.shape
        u = m.stride[0]
        assert K == 2 * u and m.padding[0] == (K - u) // 2, "expects the HiFi-GAN upsampler geometry (kernel 2u, pad u/2)"
        self._pack(L, w.permute(2, 1, 0).reshape(2, u * Cout, Cin).contiguous())      # tap j, row (r, co): k = j*u + r
        L.bias = m.bias.detach().float().contiguous() if m.bias is not None else None
        L.shifts = [0, -1]
        L.ntaps, L.CI, L.M, L.Cout, L.mode, L.u, L.pad = 2, Cin, u * Cout, Cout, OUT_UPSAMPLE, u, (K - u) // 2
        return L

    def _run(self, L, x: Tensor, slope: float, res: Tensor = None, out: Tensor = None, mode=None, scale: float = 1.0) -> Tensor:
        B, T, CI = x.shape
        assert not self.f32, "single-layer calls are offered by the fp16-storage kernels only"
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

    # ---- one layer table per input shape -----------------------------------------------------------------------
    # Generator.forward unrolled into ~100 dsp_hg_layer records over ONE fp16 workspace; the table (and its workspace) is
    # cached per (B, T), so a repeated shape costs three ctypes calls: pack, chain, post.
    def _plan(self, B: int, T: int, dev):
        key = (B, T)
        hit = self._plans.get(key)
        if hit is not None:
            return hit
        sizes, recs, free = [], [], []      # physical buffers (capacity in halves), layer records, released buffer ids
        lib = _lib.load()                   # records: (layer, x_buf, res_buf, out_buf, T, slope, mode, scale[, layer2])

        supported = lib.dsp_hifigan_resunit_f32_supported if self.f32 else lib.dsp_hifigan_resunit_supported

        def fusable(c1, c2):
            return (self.fuse_units and c1.CI == c1.M == c2.CI == c2.M and c1.ntaps == c2.ntaps and c2.dil == 1
                    and bool(supported(c1.CI, c1.ntaps, c1.dil)))

        # The launches run in order on one stream, so a buffer can be handed out again as soon as its last reader has been recorded:
        # 5 live buffers per stage (stage input, running MRF sum, two ping-pong unit outputs, the unfused chain's intermediate) instead of
        # one per layer — 1.4 GB instead of 5 GB of workspace at B=32 x 330 frames.
        def new_buf(t, c):
            n = B * t * c
            best = None
            for k in free:
                if sizes[k] >= n and (best is None or sizes[k] < sizes[best]):
                    best = k
            if best is not None:
                free.remove(best)
                return best
            sizes.append(n)
            return len(sizes) - 1

        def release(k):
            if k is not None and k not in free:
                free.append(k)
        x = new_buf(T, self.in_pad)
        x_first = x
        t = T
        y = new_buf(t, self.pre.Cout); recs.append((self.pre, x, None, y, t, 1.0, None, 1.0)); x = y      # (the packed input stays: x_in)
        for i, up in enumerate(self.ups):
            y = new_buf(t * up.u, up.Cout); recs.append((up, x, None, y, t, 0.1, None, 1.0)); release(x); x = y; t = t * up.u
            acc = new_buf(t, up.Cout)
            first = True
            for j in range(self.nk):
                yb, cur = x, None
                units = self.blocks[i * self.nk + j]
                for n, (c1, c2) in enumerate(units):
                    last = n + 1 == len(units)
                    if fusable(c1, c2):                       # x -> c1 -> c2 -> + x in one launch, no intermediate buffer
                        if not last:
                            o = new_buf(t, c2.Cout); recs.append((c1, yb, None, o, t, 0.1, OUT_STORE, 1.0, c2))
                            release(cur); cur = o; yb = o
                        else:
                            recs.append((c1, yb, None, acc, t, 0.1, OUT_STORE if first else OUT_ACCUM, 1.0 / self.nk, c2))
                        continue
                    h = new_buf(t, c1.Cout); recs.append((c1, yb, None, h, t, 0.1, None, 1.0))
                    if not last:
                        o = new_buf(t, c2.Cout); recs.append((c2, h, yb, o, t, 0.1, None, 1.0))
                        release(cur); cur = o; yb = o
                    else:
                        recs.append((c2, h, yb, acc, t, 0.1, None if first else OUT_ACCUM, 1.0 / self.nk))
                    release(h)
                release(cur)
                first = False
            release(x); x = acc
        offs, tot = [], 0
        esz = 4 if self.f32 else 2                                 # bytes per activation element
        for sz in sizes:
            offs.append(tot); tot += (sz + 63) // 64 * 64          # 128-byte (fp32: 256-byte) aligned buffers
        # ONE grow-only workspace shared by all cached plans (a plan is offsets + a launch table); growing it invalidates the tables.
        # Calls of one runner are therefore ordered on ONE stream at a time (use one runner per stream for concurrent vocoding).
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < tot or ws.device != dev:
            self._plans.clear()
            ws = self._ws = torch.empty((tot,), dtype=torch.float32 if self.f32 else torch.float16, device=dev)
        base = ws.data_ptr()
        table = (HgLayerDesc * len(recs))()
        for d, rec in zip(table, recs):
            L, xb, rb, ob, tt, slope, mode, scale = rec[:8]
            L2 = rec[8] if len(rec) > 8 else None
            d.w2 = L2.w.data_ptr() if L2 is not None else None
            d.bias2 = L2.bias.data_ptr() if (L2 is not None and L2.bias is not None) else None
            d.x = base + esz * offs[xb]; d.res = (base + esz * offs[rb]) if rb is not None else None; d.out = base + esz * offs[ob]
            d.w = L.w.data_ptr(); d.bias = L.bias.data_ptr() if L.bias is not None else None
            d.T, d.CI, d.M, d.ntaps = tt, L.CI, L.M, L.ntaps
            for k, sh in enumerate(L.shifts):
                d.shifts[k] = sh
            d.pre_slope, d.scale = slope, scale
            d.out_mode = L.mode if mode is None else mode
            d.up_u, d.up_pad, d.Tout, d.Cout = L.u, L.pad, (tt * L.u if L.mode == OUT_UPSAMPLE else tt), L.Cout
        plan = (ws, table, base + esz * offs[x_first], base + esz * offs[x], t, self.ups[-1].Cout if self.ups else self.pre.Cout)
        if len(self._plans) >= 32:
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = plan
        return plan

    @torch.no_grad()
    def __call__(self, mel: Tensor, lengths: Tensor = None) -> Tensor:
        """mel [B, 80, T] fp32 -> waveform [B, 1, T*hop] fp32.  `lengths` [B] (mel frames per utterance of a padded batch): every layer
        then treats rows beyond an utterance's own length as zero, so waveform[b, :, :lengths[b]*hop] is what the utterance gives when
        vocoded alone (the reference's one-file-at-a-time loop, hifi-gan/inference_e2e.py:47-56); samples beyond are undefined."""
        lib = _lib.load()
        B, C, T = mel.shape
        with torch.cuda.device(mel.device):
            ws, table, x_in, x_last, Tw, Cl = self._plan(B, T, mel.device)
            st = _lib.current_stream_handle()
            mt = mel.detach().float().transpose(1, 2).contiguous()
            wav = torch.empty((B, Tw), dtype=torch.float32, device=mel.device)
            if self.f32:
                lens = None if lengths is None else lengths.to(device=mel.device, dtype=torch.int32).contiguous()
                assert lens is None or lens.shape == (B,)
                _lib.check(lib.dsp_hifigan_pad_input_f32(_lib.ptr(mt), ctypes.c_void_p(x_in), B, T, C, self.in_pad, st), "dsp_hifigan_pad_input_f32")
                _lib.check(lib.dsp_hifigan_conv_chain_f32(table, len(table), B, _lib.ptr(lens), T, st), "dsp_hifigan_conv_chain_f32")
                _lib.check(lib.dsp_hifigan_post_f32(ctypes.c_void_p(x_last), _lib.ptr(self.post_w), self.post_b, _lib.ptr(wav), B, Tw, Cl,
                                                    self.post_k, 0.01, _lib.ptr(lens), Tw // T, st), "dsp_hifigan_post_f32")
                return wav.unsqueeze(1)
            _lib.check(lib.dsp_hifigan_pack_input(_lib.ptr(mt), ctypes.c_void_p(x_in), B, T, C, self.in_pad, st), "dsp_hifigan_pack_input")
            if lengths is None:
                _lib.check(lib.dsp_hifigan_conv_chain(table, len(table), B, st), "dsp_hifigan_conv_chain")
                _lib.check(lib.dsp_hifigan_post(ctypes.c_void_p(x_last), _lib.ptr(self.post_w), self.post_b, _lib.ptr(wav), B, Tw, Cl, self.post_k,
                                                0.01, st), "dsp_hifigan_post")
            else:
                lens = lengths.to(device=mel.device, dtype=torch.int32).contiguous()
                assert lens.shape == (B,)
                _lib.check(lib.dsp_hifigan_conv_chain_lens(table, len(table), B, _lib.ptr(lens), T, st), "dsp_hifigan_conv_chain_lens")
                _lib.check(lib.dsp_hifigan_post_lens(ctypes.c_void_p(x_last), _lib.ptr(self.post_w), self.post_b, _lib.ptr(wav), B, Tw, Cl,
                                                     self.post_k, 0.01, _lib.ptr(lens), Tw // T, st), "dsp_hifigan_post_lens")
        return wav.unsqueeze(1)
