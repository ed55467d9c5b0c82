"""FastSpeech2 without token embedding (the DASpeech TTS half) on top of the HIP variance-adaptor glue.

Mirrors, with fairseq-compatible parameter names so a reference checkpoint's `tts.*` keys load:
  FFNAdapter                 DASpeech/models/s2s_conformer_dag_fastspeech2.py:24-39
  PositionwiseFeedForward / FFTLayer / VariancePredictor / VarianceAdaptor
                             fairseq/fairseq/models/text_to_speech/fastspeech2.py:42-216
  FastSpeech2EncoderNoEmb    DASpeech/models/fastspeech2_noemb.py:69-174
The integer / copy steps (durations, bucketize + embedding add, length regulator) run as HIP kernels
(daspeech_amd/decode_ops.py); dense layers are PyTorch-ROCm (hipBLASLt / MIOpen), as the north star prescribes.
"""
import math
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import decode_ops

# README.md:295-300 model sizes of the released DASpeech configuration
DEFAULT_TTS_ARGS = dict(
    adaptor_in=512, adaptor_hidden=1024, embed_dim=256, heads=4, fft_hidden_dim=1024, fft_kernel_size=9, enc_layers=4,
    dec_layers=4, var_pred_hidden_dim=256, var_pred_kernel_size=3, var_pred_n_bins=256, pitch_min=-4.6600, pitch_max=5.7333,
    energy_min=-4.9544, energy_max=3.2244, out_dim=80, max_positions=1200,
    # --add-postnet and its sizes (s2s_conformer_dag_fastspeech2.py:114-119, defaults :430-435); off in the released recipe
    # dropout rates (train() mode only): --dropout / --attention-dropout shared with the translation model, var_pred_dropout 0.5
    # (fastspeech2.py base_architecture)
    dropout=0.1, attention_dropout=0.1, var_pred_dropout=0.5,
    add_postnet=False, postnet_dropout=0.5, postnet_layers=5, postnet_conv_dim=512, postnet_conv_kernel_size=5,
)


def _drop(x: Tensor, p: float, training: bool) -> Tensor:
    return F.dropout(x, p, True) if training and p > 0 else x


class FFNAdapter(nn.Module):
    def __init__(self, input_size: int, hidden_size: int, output_size: int, dropout: float = 0.1):
        super().__init__()
        self.fc1 = nn.Linear(input_size, hidden_size)
        self.fc2 = nn.Linear(hidden_size, output_size)
        self.p = dropout

    def forward(self, x: Tensor) -> Tensor:
        from ..decode_ops import linear
        return linear(_drop(linear(x, self.fc1, act="relu"), self.p, self.training), self.fc2)


class _SelfAttention(nn.Module):
    """fairseq MultiheadAttention parameter names (q_proj / k_proj / v_proj / out_proj), self-attention only."""

    def __init__(self, dim: int, heads: int, dropout: float = 0.0):
        super().__init__()
        self.heads = heads
        self.p_attn = dropout
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(dim, dim) for _ in range(4))

    def forward(self, x: Tensor, padding_mask: Optional[Tensor], lens: Optional[Tensor] = None, slack: int = 0,
                residual: Optional[Tensor] = None) -> Tensor:
        """[residual +] out_proj(attention(x)) — the residual rides in the projection's epilogue on the split-GEMM path"""
        B, N, C = x.shape
        h = self.heads
        from ..decode_ops import linear as L_                 # fp32-accurate split GEMM in eval-mode fp32 inference, torch otherwise
        if not self.training:
            from ..decode_ops import attention, linear_fused   # eval: one stacked q|k|v projection + the fp32-accurate matrix-core attention
            qf, kf, vf = linear_fused(x, (self.q_proj, self.k_proj, self.v_proj), lens=lens, slack=slack)
            o = attention(qf, kf, vf, padding_mask, h, q_lens=lens, q_slack=slack)
            if o is not None:
                return L_(o, self.out_proj, residual=residual, lens=lens, slack=slack)
            q, k, v = (t.reshape(B, N, h, C // h).transpose(1, 2) for t in (qf, kf, vf))
        else:
            q = L_(x, self.q_proj).view(B, N, h, C // h).transpose(1, 2)
            k = L_(x, self.k_proj).view(B, N, h, C // h).transpose(1, 2)
            v = L_(x, self.v_proj).view(B, N, h, C // h).transpose(1, 2)
        mask = None
        if padding_mask is not None:
            mask = torch.zeros(B, 1, 1, N, dtype=x.dtype, device=x.device).masked_fill(padding_mask.view(B, 1, 1, N), float("-inf"))
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=self.p_attn if self.training else 0.0)
        return L_(o.transpose(1, 2).reshape(B, N, C), self.out_proj, residual=residual)


class _ConvFFN(nn.Module):
    def __init__(self, dim: int, hidden: int, kernel: int, dropout: float = 0.0):
        super().__init__()
        self.p = dropout                                     # after the second convolution (fastspeech2.py:62-70)
        pad = (kernel - 1) // 2
        self.ffn = nn.Sequential(nn.Conv1d(dim, hidden, kernel, padding=pad), nn.ReLU(), nn.Conv1d(hidden, dim, kernel, padding=pad))
        self.layer_norm = nn.LayerNorm(dim)

    def forward(self, x: Tensor, lens: Optional[Tensor] = None, slack: int = 0) -> Tensor:
        from .. import decode_ops as _dops
        if (_dops.SPLIT_GEMM and not self.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
                and not torch.is_autocast_enabled() and self.ffn[0].weight.dtype == torch.float32 and x.is_contiguous()):
            # eval, fp32: both convolutions on the matrix cores at fp32 accuracy (operand splitting), channels-last, no transposes
            key = (self.ffn[0].weight.data_ptr(), self.ffn[0].weight._version, self.ffn[2].weight.data_ptr(), self.ffn[2].weight._version)
            if getattr(self, "_split_key", None) != key:
                from ..decode_ops import SplitConv1d
                c1, c2 = self.ffn[0], self.ffn[2]
                ok = all(c.stride == (1,) and c.dilation == (1,) and c.groups == 1 and c.padding == ((c.kernel_size[0] - 1) // 2,) for c in (c1, c2))
                ok = ok and all(ci % 128 == 0 and (ci <= 512 and ci in (128, 256, 512) or ci % 512 == 0) for ci in (c1.in_channels, c2.in_channels))
                self._split = (SplitConv1d(c1.weight, c1.bias), SplitConv1d(c2.weight, c2.bias)) if ok else None
                self._split_key = key
            if self._split is not None:
                return _dops.layer_norm(self._split[1](self._split[0](x, relu=True, lens=lens, slack=slack), residual=x, lens=lens, slack=slack),
                                        self.layer_norm)
        return self.layer_norm(_drop(self.ffn(x.transpose(1, 2)).transpose(1, 2), self.p, self.training) + x)


class FFTLayer(nn.Module):
    def __init__(self, dim: int, heads: int, hidden: int, kernel: int, dropout: float = 0.0, attention_dropout: float = 0.0):
        super().__init__()
        self.self_attn = _SelfAttention(dim, heads, attention_dropout)
        self.layer_norm = nn.LayerNorm(dim)
        self.ffn = _ConvFFN(dim, hidden, kernel, dropout)

    def forward(self, x: Tensor, padding_mask: Optional[Tensor] = None, lens: Optional[Tensor] = None, slack: int = 0) -> Tensor:
        """lens [B] int32 + slack (eval-mode inference on the GPU only): rows from lens[b] + slack on are padding that no valid frame of
        the model's output depends on; the matrix-core kernels skip their tiles and return zeros there."""
        from ..decode_ops import layer_norm as _ln
        x = _ln(self.self_attn(x, padding_mask, lens, slack, residual=x), self.layer_norm)
        return self.ffn(x, lens, slack)


class VariancePredictor(nn.Module):
    """Conv1d-ReLU-LN-Conv1d-ReLU-LN-Linear (fastspeech2.py:117-151); dropout is identity at inference."""

    def __init__(self, dim: int, hidden: int, kernel: int, dropout: float = 0.0):
        super().__init__()
        self.p = dropout                                     # after each LayerNorm (fastspeech2.py:128-130,147-150)
        self.conv1 = nn.Sequential(nn.Conv1d(dim, hidden, kernel, padding=(kernel - 1) // 2), nn.ReLU())
        self.ln1 = nn.LayerNorm(hidden)
        self.conv2 = nn.Sequential(nn.Conv1d(hidden, hidden, kernel, padding=1), nn.ReLU())
        self.ln2 = nn.LayerNorm(hidden)
        self.proj = nn.Linear(hidden, 1)

    def forward(self, x: Tensor) -> Tensor:
        from .. import decode_ops as _dops
        c1, c2 = self.conv1[0], self.conv2[0]
        if (_dops.SPLIT_GEMM and not self.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
                and not torch.is_autocast_enabled() and c1.weight.dtype == torch.float32 and x.is_contiguous()):
            # eval, fp32: both convolutions on the matrix cores at fp32 accuracy (operand splitting), channels-last, no transposes
            # (MIOpen runs these small k=3 convolutions as one im2col + GEMM PER SAMPLE: ~190 launches and 2 ms per batch of 32)
            key = (c1.weight.data_ptr(), c1.weight._version, c2.weight.data_ptr(), c2.weight._version)
            if getattr(self, "_split_key", None) != key:
                ok = all(c.stride == (1,) and c.dilation == (1,) and c.groups == 1 and c.padding == ((c.kernel_size[0] - 1) // 2,) for c in (c1, c2))
                ok = ok and all(ci in (128, 256, 512) or ci % 512 == 0 for ci in (c1.in_channels, c2.in_channels))
                ok = ok and all(c.out_channels % 4 == 0 for c in (c1, c2))
                self._split = (_dops.SplitConv1d(c1.weight, c1.bias), _dops.SplitConv1d(c2.weight, c2.bias)) if ok else None
                self._split_key = key
            if self._split is not None:
                h = _dops.layer_norm(self._split[0](x, relu=True), self.ln1)
                h = _dops.layer_norm(self._split[1](h, relu=True), self.ln2)
                return self.proj(h).squeeze(2)
        x = _drop(self.ln1(self.conv1(x.transpose(1, 2)).transpose(1, 2)), self.p, self.training)
        x = _drop(self.ln2(self.conv2(x.transpose(1, 2)).transpose(1, 2)), self.p, self.training)
        return self.proj(x).squeeze(2)


class VarianceAdaptor(nn.Module):
    """fastspeech2.py:154-216.  Without gradient (inference, predicted or supplied durations / pitch / energy) the glue runs on the
    HIP ops; with gradient enabled (training) it runs on differentiable torch ops so that the mel loss reaches the encoder FFT
    layers, the adaptor and both embedding tables as in the reference."""

    def __init__(self, dim: int, hidden: int, kernel: int, n_bins: int, pitch_min: float, pitch_max: float,
                 energy_min: float, energy_max: float, dropout: float = 0.0):
        super().__init__()
        self.duration_predictor = VariancePredictor(dim, hidden, kernel, dropout)
        self.pitch_predictor = VariancePredictor(dim, hidden, kernel, dropout)
        self.energy_predictor = VariancePredictor(dim, hidden, kernel, dropout)
        self.register_buffer("pitch_bins", torch.linspace(pitch_min, pitch_max, n_bins - 1), persistent=False)
        self.register_buffer("energy_bins", torch.linspace(energy_min, energy_max, n_bins - 1), persistent=False)
        self.embed_pitch = nn.Embedding(n_bins, dim)
        self.embed_energy = nn.Embedding(n_bins, dim)

    def forward(self, x: Tensor, padding_mask: Tensor, durations: Optional[Tensor] = None, pitches: Optional[Tensor] = None,
                energies: Optional[Tensor] = None, d_factor: float = 1.0, p_factor: float = 1.0, e_factor: float = 1.0):
        B, N, C = x.shape
        log_dur_out = self.duration_predictor(x)
        pitch_out = self.pitch_predictor(x)
        pv = pitch_out * p_factor if pitches is None else pitches
        if torch.is_grad_enabled() or not x.is_cuda:
            # training (teacher durations / pitch / energy, gradients into x and both embedding tables — the reference's
            # `x + embed(bucketize(v))` and LengthRegulator are differentiable, fastspeech2.py:98-114,169-214) or CPU tensors:
            # plain torch ops.  The HIP glue kernels below are forward-only.
            x = x + F.embedding(torch.bucketize(pv.detach(), self.pitch_bins), self.embed_pitch.weight)
            energy_out = self.energy_predictor(x)
            ev = energy_out * e_factor if energies is None else energies
            x = x + F.embedding(torch.bucketize(ev.detach(), self.energy_bins), self.embed_energy.weight)
            if durations is None:
                durations = torch.clamp(torch.round((torch.exp(log_dur_out.detach()) - 1) * d_factor).long(), min=0).masked_fill(padding_mask, 0)
            out_lens = durations.sum(1)
            maxlen = int(out_lens.max()) if B else 0
            # length regulator as one gather: frame f of sample b comes from phoneme searchsorted(cumsum(dur), f, right)
            cum = durations.cumsum(1)
            frames = torch.arange(maxlen, device=x.device).unsqueeze(0).expand(B, -1)
            src = torch.searchsorted(cum, frames.contiguous(), right=True).clamp(max=max(N - 1, 0))
            x = x.gather(1, src.unsqueeze(-1).expand(-1, -1, C)) * (frames < out_lens.unsqueeze(1)).unsqueeze(-1).to(x.dtype)
        else:
            dur_out = decode_ops.predicted_durations(log_dur_out, padding_mask, d_factor)                     # :202-205
            x = decode_ops.bucketize_embed_add(x.reshape(B * N, C), pv.reshape(-1), self.pitch_bins, self.embed_pitch.weight).view(B, N, C)
            energy_out = self.energy_predictor(x)                                                             # on x + pitch_emb  :209
            ev = energy_out * e_factor if energies is None else energies
            x = decode_ops.bucketize_embed_add(x.reshape(B * N, C), ev.reshape(-1), self.energy_bins, self.embed_energy.weight).view(B, N, C)
            x, out_lens = decode_ops.length_regulate(x, dur_out if durations is None else durations)          # :212-214
        if pitches is None:
            pitch_out = pitch_out * p_factor
        if energies is None:
            energy_out = energy_out * e_factor
        return x, out_lens, log_dur_out, pitch_out, energy_out


class Postnet(nn.Module):
    """fairseq/fairseq/models/text_to_speech/tacotron2.py:111-140: `n_layers` x [Conv1d(k, "same") -> BatchNorm1d -> tanh (all but the
    last) -> dropout], channels in_dim -> n_channels ... -> in_dim; same parameter names (`convolutions.{i}.0` conv, `.1` batch norm), same
    initialisation.  Plain torch (MIOpen): the branch is off in the released recipe (README.md:288-323 has no --add-postnet)."""

    def __init__(self, in_dim: int, n_channels: int, kernel_size: int, n_layers: int, dropout: float):
        super().__init__()
        assert kernel_size % 2 == 1
        self.convolutions = nn.ModuleList()
        for i in range(n_layers):
            last = i == n_layers - 1
            layers = [nn.Conv1d(in_dim if i == 0 else n_channels, in_dim if last else n_channels, kernel_size, padding=(kernel_size - 1) // 2),
                      nn.BatchNorm1d(in_dim if last else n_channels)] + ([] if last else [nn.Tanh()]) + [nn.Dropout(dropout)]
            nn.init.xavier_uniform_(layers[0].weight, nn.init.calculate_gain("linear" if last else "tanh"))
            self.convolutions.append(nn.Sequential(*layers))

    def forward(self, x: Tensor) -> Tensor:
        x = x.transpose(1, 2)
        for conv in self.convolutions:
            x = conv(x)
        return x.transpose(1, 2)


def sinusoidal_table(n: int, dim: int, padding_idx: int = 1) -> Tensor:
    """modules/sinusoidal_positional_embedding.py:36-58: [sin | cos] halves, row `padding_idx` zeroed."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(n, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n, 1)], dim=1)
    tab[padding_idx] = 0
    return tab


def positions_from_padding_mask(padding_mask: Tensor, padding_idx: int = 1) -> Tensor:
    """The reference feeds the padding MASK to the positional embedding as if it were tokens with pad=1
    (fastspeech2_noemb.py:150,166 + fairseq/utils.py:256-266): non-pad frames get 2,3,..., pads get 1 (the zero row)."""
    keep = (~padding_mask).int()
    return (torch.cumsum(keep, dim=1) * keep).long() + padding_idx


class FastSpeech2NoEmb(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        a = SimpleNamespace(**{**DEFAULT_TTS_ARGS, **kw})
        self.args = a
        self.ragged = True                          # eval-mode GPU inference skips the tiles of padding frames no valid frame depends on
        self.pos_emb_alpha = nn.Parameter(torch.ones(1))
        self.dec_pos_emb_alpha = nn.Parameter(torch.ones(1))
        self.register_buffer("pos_table", sinusoidal_table(a.max_positions + 2, a.embed_dim), persistent=False)
        fft = lambda: FFTLayer(a.embed_dim, a.heads, a.fft_hidden_dim, a.fft_kernel_size, a.dropout, a.attention_dropout)   # noqa: E731
        self.encoder_fft_layers = nn.ModuleList(fft() for _ in range(a.enc_layers))
        self.var_adaptor = VarianceAdaptor(a.embed_dim, a.var_pred_hidden_dim, a.var_pred_kernel_size, a.var_pred_n_bins,
                                           a.pitch_min, a.pitch_max, a.energy_min, a.energy_max, a.var_pred_dropout)
        self.decoder_fft_layers = nn.ModuleList(fft() for _ in range(a.dec_layers))
        self.out_proj = nn.Linear(a.embed_dim, a.out_dim)
        self.out_dim = a.out_dim
        self.postnet = Postnet(a.out_dim, a.postnet_conv_dim, a.postnet_conv_kernel_size, a.postnet_layers, a.postnet_dropout) \
            if a.add_postnet else None                                                                       # fastspeech2_noemb.py:128-136

    def _pos(self, padding_mask: Tensor) -> Tensor:
        idx = positions_from_padding_mask(padding_mask).clamp(max=self.pos_table.shape[0] - 1)
        return self.pos_table[idx]

    def forward(self, x: Tensor, padding_mask: Tensor, durations=None, pitches=None, energies=None):
        """x [B,N,256] adaptor output, padding_mask [B,N] bool -> (mel [B,F,80], mel_post or None, out_lens, log_dur, pitch, energy):
        the reference's six values in its order (fastspeech2_noemb.py:140-174; `mel_post = mel + postnet(mel)` only with --add-postnet)."""
        if x.shape[1] == 0:                       # nothing decoded (the reference would fail in torch.cat([]))
            z = x.new_zeros(x.shape[0], 0)
            mel0 = x.new_zeros(x.shape[0], 0, self.args.out_dim)
            return mel0, (mel0 if self.postnet is not None else None), x.new_zeros(x.shape[0], dtype=torch.long), z, z, z
        x = _drop(x + self.pos_emb_alpha * self._pos(padding_mask), self.args.dropout, self.training)       # fastspeech2_noemb.py:150-151
        for layer in self.encoder_fft_layers:
            x = layer(x, padding_mask)
        x, out_lens, log_dur, pitch, energy = self.var_adaptor(x, padding_mask, durations, pitches, energies)
        F_ = x.shape[1]
        dec_mask = torch.arange(F_, device=x.device).unsqueeze(0) >= out_lens.unsqueeze(1)
        x = x + self.dec_pos_emb_alpha * self._pos(dec_mask)
        # Ragged batch: the reference computes every padded frame (fastspeech2.py:86-98 masks attention keys only, the two K=9 convolutions
        # of an FFT layer read 8 frames past an utterance's end), so a valid frame depends on padding frames up to 8 per remaining layer
        # (+ the postnet's 5 x 2).  Frames beyond that bound influence nothing valid: their tiles are skipped (zeros).  Valid frames keep
        # exactly the bits of the dense computation.
        lens = None
        if not self.training and not torch.is_grad_enabled() and x.is_cuda and self.ragged:
            lens = out_lens.to(torch.int32).contiguous()
        nl = len(self.decoder_fft_layers)
        reach = (self.args.fft_kernel_size - 1) // 2 * 2                                   # frames one layer's two convolutions reach
        post = 0 if self.postnet is None else sum((c.kernel_size[0] - 1) // 2 for c in self.postnet.modules() if isinstance(c, nn.Conv1d))
        for i, layer in enumerate(self.decoder_fft_layers):
            x = layer(x, dec_mask, lens, reach * (nl - i) + post)
        from ..decode_ops import linear as _lin
        mel = _lin(x.contiguous(), self.out_proj)
        mel_post = mel + self.postnet(mel) if self.postnet is not None else None                             # :171-173
        return mel, mel_post, out_lens, log_dur, pitch, energy
