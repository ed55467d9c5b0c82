"""Inference-side HIP ops of the DASpeech hot path (C ABI: include/daspeech_decode.h).

Host-side mirror of the reference code these replace:
  graph_decode        DASpeech/models/s2s_conformer_dag_fastspeech2.py:201-243 (lookahead / greedy branch of forward_decoder)
  posterior / expect_features   DASpeech/criterions/s2s_dag_fastspeech2_loss.py:259-263
  predicted_durations / bucketize_embed_add / length_regulate   fairseq/fairseq/models/text_to_speech/fastspeech2.py:98-114,169-210
No CPU fallback: GPU tensors only.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def _gpu(name, *ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"{name}: expected GPU tensors (HIP op, no CPU path)")


def _code(t: Tensor) -> int:
    c = _lib.DTYPE_CODES.get(str(t.dtype))
    if c is None:
        raise RuntimeError(f"unsupported dtype {t.dtype}")
    return c


def argmax_logp(logits: Tensor) -> Tuple[Tensor, Tensor]:
    """tok[b,j] = argmax_v logits (first max), score[b,j] = max_v log_softmax(logits)   (:207-208)."""
    _gpu("argmax_logp", logits)
    x = logits.detach().contiguous()
    B, L, V = x.shape
    lib = _lib.load()
    with torch.cuda.device(x.device):
        tok = torch.empty((B, L), dtype=torch.int32, device=x.device)
        score = torch.empty((B, L), dtype=torch.float32, device=x.device)
        _lib.check(lib.dsp_argmax_logp(_lib.ptr(x), _code(x), _lib.ptr(tok), _lib.ptr(score), B, L, V,
                                       _lib.current_stream_handle()), "dsp_argmax_logp")
    return tok, score


def lookahead_next(links: Tensor, score: Optional[Tensor], decode_beta: float = 1.0, greedy: bool = False) -> Tensor:
    """next[b,i] = argmax_j(links[b,i,j-i-1] + decode_beta*score[b,j]) on the compact layout (:209-217)."""
    _gpu("lookahead_next", links, score)
    k = links.detach().to(torch.float32).contiguous()
    B, L, TR = k.shape
    sc = None if greedy else score.detach().to(torch.float32).contiguous()
    lib = _lib.load()
    with torch.cuda.device(k.device):
        nxt = torch.empty((B, L), dtype=torch.int32, device=k.device)
        _lib.check(lib.dsp_lookahead_next(_lib.ptr(k), _lib.ptr(sc), float(decode_beta), 1 if greedy else 0, _lib.ptr(nxt),
                                          B, L, TR, _lib.current_stream_handle()), "dsp_lookahead_next")
    return nxt


def graph_decode(logits: Tensor, links: Tensor, features: Tensor, output_length: Tensor, pad: int,
                 decode_beta: float = 1.0, strategy: str = "lookahead"):
    """Lookahead / greedy graph decode.  Returns (output_tokens [B,N] int64 pad-filled, features [B,F,D] zero padded,
    features_padding_mask [B,F] bool, feature_lengths [B] int64).  One host sync (the output shapes), vs. three `.tolist()`
    round trips + per-sample Python loops in the reference (:208-243)."""
    if strategy not in ("lookahead", "greedy"):
        raise NotImplementedError(f"decode strategy {strategy}")
    _gpu("graph_decode", logits, links, features, output_length)
    B, L, V = logits.shape
    tok, score = argmax_logp(logits)
    nxt = lookahead_next(links, score, decode_beta, greedy=(strategy == "greedy"))
    feats = features.detach().contiguous()
    D = feats.shape[2]
    ol = output_length.to(torch.long).contiguous()
    lib = _lib.load()
    dev = logits.device
    with torch.cuda.device(dev):
        cap = L
        out_tok = torch.empty((B, cap), dtype=torch.long, device=dev)
        keep = torch.empty((B, cap), dtype=torch.int32, device=dev)
        nfeat = torch.empty((B,), dtype=torch.int32, device=dev)
        _lib.check(lib.dsp_follow_path(_lib.ptr(nxt), _lib.ptr(tok), _lib.ptr(ol), int(pad), _lib.ptr(out_tok), _lib.ptr(keep),
                                       _lib.ptr(nfeat), B, L, cap, _lib.current_stream_handle()), "dsp_follow_path")
        fmax = int(nfeat.max().item()) if B else 0          # the one sync: output shapes depend on it
        out_feat = torch.empty((B, fmax, D), dtype=feats.dtype, device=dev)
        if fmax:
            _lib.check(lib.dsp_gather_rows(_lib.ptr(feats), _code(feats), _lib.ptr(keep), _lib.ptr(nfeat), _lib.ptr(out_feat),
                                           B, L, D, cap, fmax, _lib.current_stream_handle()), "dsp_gather_rows")
    lens = nfeat.to(torch.long)
    mask = torch.arange(fmax, device=dev).unsqueeze(0) >= lens.unsqueeze(1)      # lengths_to_padding_mask
    return out_tok[:, : fmax + 1].contiguous(), out_feat, mask, lens


def posterior(alpha: Tensor, beta: Tensor) -> Tensor:
    """score = exp(alpha + beta - logsumexp_j(alpha + beta)), NaN -> 0   (s2s_dag_fastspeech2_loss.py:259-260)."""
    _gpu("posterior", alpha, beta)
    a = alpha.detach().to(torch.float32).contiguous()
    b = beta.detach().to(torch.float32).contiguous()
    B, T, L = a.shape
    lib = _lib.load()
    with torch.cuda.device(a.device):
        score = torch.empty_like(a)
        _lib.check(lib.dsp_posterior(_lib.ptr(a), _lib.ptr(b), _lib.ptr(score), B, T, L, _lib.current_stream_handle()),
                   "dsp_posterior")
    return score


def expect_features(alpha: Tensor, beta: Tensor, features: Tensor) -> Tensor:
    """Expected hidden states of the "expect" strategy: (score @ features)[:, 1:]   (:259-263).  The posterior is a HIP
    kernel; the [T x L] x [L x D] product is a library MFMA GEMM (hipBLASLt through torch.matmul)."""
    score = posterior(alpha, beta).to(features.dtype)
    return torch.matmul(score, features)[:, 1:, :]


def predicted_durations(log_dur: Tensor, padding_mask: Tensor, d_factor: float = 1.0) -> Tensor:
    """clamp(round((exp(log_dur)-1)*d_factor), 0).long(), 0 at pads   (fastspeech2.py:202-205)."""
    _gpu("predicted_durations", log_dur, padding_mask)
    ld = log_dur.detach().to(torch.float32).contiguous()
    pm = padding_mask.to(torch.uint8).contiguous()
    lib = _lib.load()
    with torch.cuda.device(ld.device):
        dur = torch.empty(ld.shape, dtype=torch.long, device=ld.device)
        _lib.check(lib.dsp_durations(_lib.ptr(ld), _lib.ptr(pm), float(d_factor), _lib.ptr(dur), ld.numel(),
                                     _lib.current_stream_handle()), "dsp_durations")
    return dur


def bucketize_embed_add(x: Tensor, values: Tensor, bins: Tensor, emb_weight: Tensor) -> Tensor:
    """x + Embedding(bucketize(values, bins))  — pitch / energy embedding of the variance adaptor (fastspeech2.py:169-177,207-210).
    Returns a new tensor (x is not modified)."""
    _gpu("bucketize_embed_add", x, values, bins, emb_weight)
    out = x.detach().to(torch.float32).contiguous().clone()
    v = values.detach().to(torch.float32).contiguous()
    bn = bins.detach().to(torch.float32).contiguous()
    em = emb_weight.detach().to(torch.float32).contiguous()
    C = out.shape[-1]
    n = v.numel()
    assert out.numel() == n * C and em.shape[0] == bn.numel() + 1 and em.shape[1] == C
    lib = _lib.load()
    with torch.cuda.device(out.device):
        _lib.check(lib.dsp_bucketize_embed_add(_lib.ptr(out), _lib.ptr(v), _lib.ptr(bn), bn.numel(), _lib.ptr(em), n, C,
                                               _lib.current_stream_handle()), "dsp_bucketize_embed_add")
    return out.to(x.dtype)


def length_regulate(x: Tensor, durations: Tensor) -> Tuple[Tensor, Tensor]:
    """LengthRegulator.forward (fastspeech2.py:98-114): rows of x [B,N,C] repeated durations[b,t] times, zero padded to the
    batch maximum; returns (out [B,max,C], out_lens [B] int64).  One sync for the output shape (the reference: B*N)."""
    _gpu("length_regulate", x, durations)
    xx = x.detach().contiguous()
    dur = durations.to(torch.long).contiguous()
    B, N, C = xx.shape
    lib = _lib.load()
    dev = xx.device
    with torch.cuda.device(dev):
        cum = torch.empty((B, N), dtype=torch.long, device=dev)
        lens = torch.empty((B,), dtype=torch.long, device=dev)
        _lib.check(lib.dsp_length_regulator_lens(_lib.ptr(dur), _lib.ptr(cum), _lib.ptr(lens), B, N,
                                                 _lib.current_stream_handle()), "dsp_length_regulator_lens")
        maxlen = int(lens.max().item()) if B else 0
        out = torch.empty((B, maxlen, C), dtype=xx.dtype, device=dev)
        if maxlen:
            _lib.check(lib.dsp_length_regulator_expand(_lib.ptr(xx), _code(xx), _lib.ptr(cum), _lib.ptr(out), B, N, C, maxlen,
                                                       _lib.current_stream_handle()), "dsp_length_regulator_expand")
    return out, lens
