"""Inference-side HIP ops of the DASpeech hot path (C ABI: include/daspeech_decode.h).

Host-side mirror of the reference code these replace:
  graph_decode        DASpeech/models/s2s_conformer_dag_fastspeech2.py:201-243 (lookahead / greedy branch of forward_decoder)
  posterior / expect_features   DASpeech/criterions/s2s_dag_fastspeech2_loss.py:259-263
  predicted_durations / bucketize_embed_add / length_regulate   fairseq/fairseq/models/text_to_speech/fastspeech2.py:98-114,169-210
No CPU fallback: GPU tensors only.
"""
import ctypes
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def _gpu(name, *ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"{name}: expected GPU tensors (HIP op, no CPU path)")


def _mask_bytes(m: Optional[Tensor]) -> Optional[Tensor]:
    """a [B,T] padding mask as bytes for the C ABI: contiguous bool tensors are reinterpreted (no launch), anything else is converted"""
    if m is None:
        return None
    if m.dtype == torch.bool and m.is_contiguous():
        return m.view(torch.uint8)
    return m.to(torch.uint8).contiguous()


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


def extract_links(q: Tensor, k: Tensor, log_gates: Tensor, output_length: Tensor, TR: int,
                  dist_bias: Optional[Tensor] = None) -> Tensor:
    """Fused compact transition log-probabilities [B, L, TR] fp32 from the link predictor's q / k [B,L,H,CK] and
    log_gates [B,L,H] (s2t_conformer_dag.py:171-212); inference only (no gradient)."""
    _gpu("extract_links", q, k, log_gates, output_length)
    qf = q.detach().to(torch.float32).contiguous()
    kf = k.detach().to(torch.float32).contiguous()
    gf = log_gates.detach().to(torch.float32).contiguous()
    ol = output_length.to(torch.long).contiguous()
    B, L, H, CK = qf.shape
    bias = None if dist_bias is None else dist_bias.detach().to(device=qf.device, dtype=torch.float32).contiguous()
    lib = _lib.load()
    with torch.cuda.device(qf.device):
        links = torch.empty((B, L, TR), dtype=torch.float32, device=qf.device)
        ws = _links_workspace(lib, B, L, H, CK, TR, 0, qf.device)
        if ws is not None:                # the matrix-core kernels (csrc/extract_links_mfma.hip)
            _lib.check(lib.dsp_extract_links_ws(_lib.ptr(qf), _lib.ptr(kf), _lib.ptr(gf), _lib.ptr(ol), _lib.ptr(bias), _lib.ptr(links), None,
                                                B, L, H, CK, TR, float(CK) ** -0.5, _lib.ptr(ws), ws.numel(), _lib.current_stream_handle()),
                       "dsp_extract_links_ws")
        else:
            _lib.check(lib.dsp_extract_links(_lib.ptr(qf), _lib.ptr(kf), _lib.ptr(gf), _lib.ptr(ol), _lib.ptr(bias), _lib.ptr(links),
                                             B, L, H, CK, TR, float(CK) ** -0.5, _lib.current_stream_handle()), "dsp_extract_links")
    return links


def _links_workspace(lib, B: int, L: int, H: int, CK: int, TR: int, phase: int, device) -> Optional[Tensor]:
    """Scratch for the matrix-core link kernels (phase 0 inference / 1 training forward / 2 backward), or None when the library serves this
    shape with the fp32-FMA kernels (dsp_extract_links_workspace reports 0 bytes)."""
    import ctypes
    n = ctypes.c_size_t(0)
    _lib.check(lib.dsp_extract_links_workspace(B, L, H, CK, TR, phase, ctypes.byref(n)), "dsp_extract_links_workspace")
    return torch.empty(n.value, dtype=torch.uint8, device=device) if n.value else None


class _ExtractLinksFn(torch.autograd.Function):
    """Fused link producer under autograd: forward = dsp_extract_links_train (compact band + per-row soft-max state), backward =
    dsp_extract_links_bwd (scores recomputed per tile; no [B,L,L,H] tensor in either direction)."""

    @staticmethod
    def forward(ctx, q, k, log_gates, output_length, TR, dist_bias):
        qf = q.detach().to(torch.float32).contiguous()
        kf = k.detach().to(torch.float32).contiguous()
        gf = log_gates.detach().to(torch.float32).contiguous()
        ol = output_length.to(torch.long).contiguous()
        B, L, H, CK = qf.shape
        bias = None if dist_bias is None else dist_bias.detach().to(device=qf.device, dtype=torch.float32).contiguous()
        lib = _lib.load()
        with torch.cuda.device(qf.device):
            links = torch.empty((B, L, TR), dtype=torch.float32, device=qf.device)
            stats = torch.empty((B, L, H, 2), dtype=torch.float32, device=qf.device)
            ws = _links_workspace(lib, B, L, H, CK, TR, 1, qf.device)
            if ws is not None:
                _lib.check(lib.dsp_extract_links_ws(_lib.ptr(qf), _lib.ptr(kf), _lib.ptr(gf), _lib.ptr(ol), _lib.ptr(bias), _lib.ptr(links),
                                                    _lib.ptr(stats), B, L, H, CK, TR, float(CK) ** -0.5, _lib.ptr(ws), ws.numel(),
                                                    _lib.current_stream_handle()), "dsp_extract_links_ws")
            else:
                _lib.check(lib.dsp_extract_links_train(_lib.ptr(qf), _lib.ptr(kf), _lib.ptr(gf), _lib.ptr(ol), _lib.ptr(bias), _lib.ptr(links),
                                                       _lib.ptr(stats), B, L, H, CK, TR, float(CK) ** -0.5, _lib.current_stream_handle()),
                           "dsp_extract_links_train")
        ctx.save_for_backward(qf, kf, gf, ol, links, stats, bias if bias is not None else qf.new_empty(0))
        ctx.TR, ctx.has_bias, ctx.in_dtypes = TR, bias is not None, (q.dtype, k.dtype, log_gates.dtype)
        ctx.mark_non_differentiable(output_length)
        return links

    @staticmethod
    def backward(ctx, grad_links):
        qf, kf, gf, ol, links, stats, bias = ctx.saved_tensors
        B, L, H, CK = qf.shape
        g = grad_links.detach().to(torch.float32).contiguous()
        lib = _lib.load()
        with torch.cuda.device(qf.device):
            dq, dk = torch.empty_like(qf), torch.empty_like(kf)
            dg = torch.empty_like(gf)
            ws = _links_workspace(lib, B, L, H, CK, ctx.TR, 2, qf.device)
            if ws is not None:
                _lib.check(lib.dsp_extract_links_bwd_ws(_lib.ptr(qf), _lib.ptr(kf), _lib.ptr(gf), _lib.ptr(ol), _lib.ptr(bias if ctx.has_bias else None),
                                                        _lib.ptr(links), _lib.ptr(g), _lib.ptr(stats), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dg),
                                                        B, L, H, CK, ctx.TR, float(CK) ** -0.5, _lib.ptr(ws), ws.numel(),
                                                        _lib.current_stream_handle()), "dsp_extract_links_bwd_ws")
            else:
                _lib.check(lib.dsp_extract_links_bwd(_lib.ptr(qf), _lib.ptr(kf), _lib.ptr(gf), _lib.ptr(ol), _lib.ptr(bias if ctx.has_bias else None),
                                                     _lib.ptr(links), _lib.ptr(g), _lib.ptr(stats), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dg),
                                                     B, L, H, CK, ctx.TR, float(CK) ** -0.5, _lib.current_stream_handle()), "dsp_extract_links_bwd")
        dt = ctx.in_dtypes
        return dq.to(dt[0]), dk.to(dt[1]), dg.to(dt[2]), None, None, None


def extract_links_autograd(q: Tensor, k: Tensor, log_gates: Tensor, output_length: Tensor, TR: int,
                           dist_bias: Optional[Tensor] = None) -> Tensor:
    """`extract_links` with gradients w.r.t. q, k and log_gates (training: the step in front of dag_loss).  `dist_bias` is a constant."""
    _gpu("extract_links", q, k, log_gates, output_length)
    return _ExtractLinksFn.apply(q, k, log_gates, output_length, int(TR), dist_bias)


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


class _PosteriorFeaturesFn(torch.autograd.Function):
    """score @ features with the row soft-max of alpha + beta fused in (no [B,T,L] score tensor in either direction); gradient w.r.t.
    the features only — alpha / beta are detached, as the reference's are (dag_loss.py:180-186)."""

    @staticmethod
    def forward(ctx, alpha, beta, features):
        a = alpha.detach().to(torch.float32).contiguous()
        b = beta.detach().to(torch.float32).contiguous()
        f = features.detach().to(torch.float32).contiguous()
        B, T, L = a.shape
        D = f.shape[2]
        lib = _lib.load()
        with torch.cuda.device(a.device):
            out = torch.empty((B, T, D), dtype=torch.float32, device=a.device)
            lse = torch.empty((B, T), dtype=torch.float32, device=a.device)
            _lib.check(lib.dsp_posterior_features(_lib.ptr(a), _lib.ptr(b), _lib.ptr(f), _lib.ptr(out), _lib.ptr(lse), B, T, L, D,
                                                  _lib.current_stream_handle()), "dsp_posterior_features")
        ctx.save_for_backward(a, b, lse)
        ctx.fdtype, ctx.L = features.dtype, L
        return out.to(features.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        a, b, lse = ctx.saved_tensors
        B, T, L = a.shape
        g = grad_out.detach().to(torch.float32).contiguous()
        D = g.shape[2]
        lib = _lib.load()
        with torch.cuda.device(a.device):
            df = torch.empty((B, L, D), dtype=torch.float32, device=a.device)
            _lib.check(lib.dsp_posterior_features_bwd(_lib.ptr(a), _lib.ptr(b), _lib.ptr(lse), _lib.ptr(g), _lib.ptr(df), B, T, L, D,
                                                      _lib.current_stream_handle()), "dsp_posterior_features_bwd")
        return None, None, df.to(ctx.fdtype)


def posterior_features(alpha: Tensor, beta: Tensor, features: Tensor) -> Tensor:
    """[B,T,D] = softmax_j(alpha + beta) @ features, fused (dsp_posterior_features): the posterior never exists in HBM; differentiable
    w.r.t. `features`.  Falls back to the two-step form when the shape does not fit the kernel (odd D, rows beyond LDS)."""
    _gpu("posterior_features", alpha, beta, features)
    B, T, L = alpha.shape
    if features.shape[2] % 2 or 8 * max(L, T) * 4 > 150 * 1024:
        return torch.matmul(posterior(alpha, beta).to(features.dtype), features)
    return _PosteriorFeaturesFn.apply(alpha, beta, features)


def expect_features(alpha: Tensor, beta: Tensor, features: Tensor) -> Tensor:
    """Expected hidden states of the "expect" strategy: (score @ features)[:, 1:]   (:259-263), fused: see posterior_features."""
    return posterior_features(alpha, beta, features)[:, 1:, :]


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


def restore_valid_links(links: Tensor) -> Tensor:
    """compact [B,L,TR] -> dense [B,L,L] with -inf elsewhere (s2t_conformer_dag.py:157-169); only the Viterbi strategies need
    it — lookahead / greedy run on the compact layout."""
    B, L, TR = links.shape
    idx = torch.arange(L, device=links.device).unsqueeze(1) + torch.arange(TR, device=links.device).unsqueeze(0) + 1
    idx = idx.masked_fill(idx >= L, L)
    res = torch.full((B, L, L + 1), float("-inf"), dtype=torch.float, device=links.device)
    res.scatter_(2, idx.unsqueeze(0).expand(B, -1, -1), links.float())
    return res[:, :, :L]


@torch.no_grad()
def viterbi_decode(logits: Tensor, links: Tensor, features: Tensor, output_length: Tensor, pad: int, decode_beta: float = 1.0,
                   decode_viterbibeta: float = 1.0, joint: bool = True, src_upsample_scale: float = 0.5):
    """`viterbi` / `jointviterbi` strategies of forward_decoder (s2s_conformer_dag_fastspeech2.py:244-304) on the HIP max-DP.

    The reference's loop `alpha, index = max(alpha[:, :, None] + dense_links, dim=1) [+ score * beta]` (:258-262) is the alignment
    DP K6 with an emission row that does not depend on the step: row 0 of the DP is the start vertex alone (emission
    `score[0]*beta` for jointviterbi, 0 otherwise), rows 1.. carry `score[j]*beta` (every row for jointviterbi, row 1 only for
    viterbi, :252), and the emission of each sample's FINAL vertex is 0, so that `alpha_max[m+2, L_b-1]` is exactly the
    reference's `max_j(scores[m][j] + links[j, L_b-1])` (:267-269) and `trace[m+2, L_b-1]` its arg-max (:278).  Same float
    operations in the same order (one add per transition, max, one add per cell) and the same tie rule (smallest index), so
    tokens and lengths are those of `viterbi_decode_torch`, which restates the reference loop.  The DP runs on the compact
    links — the dense `[B, L, L]` restore and the L/4 torch launches per batch are gone."""
    B, L, V = logits.shape
    TR = links.shape[2]
    dev = logits.device
    tok32, sc = argmax_logp(logits)                                       # unreduced_logits / unreduced_tokens (:207-208), one pass over the logits
    tok32 = tok32.contiguous()
    max_length = max(1, int(L / 8 / src_upsample_scale))                 # (:256)
    T = max_length + 2
    olen = output_length.to(torch.int64).contiguous()
    scb = (sc * decode_beta).contiguous()
    match = (scb if joint else torch.zeros_like(scb)).unsqueeze(1).repeat(1, T, 1)
    match[:, 1] = scb
    if not joint:
        match[:, 0] = 0
    ar = torch.arange(B, device=dev)
    match[ar, :, (olen - 1).clamp(min=0)] = 0
    links_c = links.float().contiguous()
    rows = torch.full((B,), T, dtype=torch.int64, device=dev)
    amax = torch.empty((B, T, L), dtype=torch.float32, device=dev)
    lib = _lib.load()
    # windows wider than 32 (the model's default): the blocked max-plus DP + block trace of the dense-window alignment kernels; narrow windows
    # keep the row-sequential DP with its arg-max trace.  alpha_max, tie rule and therefore tokens / lengths are the same either way.
    blocks = bool(lib.dsp_dag_max_alpha_blocks_supported(L, TR))
    trace = torch.empty((B, T, L), dtype=torch.int16 if blocks else torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.current_stream_handle()
        if blocks:
            _lib.check(lib.dsp_dag_max_alpha_blocks(_lib.ptr(match), _lib.ptr(links_c), _lib.ptr(olen), _lib.ptr(rows), _lib.ptr(amax), _lib.ptr(trace),
                                                    B, T, L, TR, st), "dsp_dag_max_alpha_blocks")
        else:
            _lib.check(lib.dsp_dag_max_alpha(_lib.ptr(match), _lib.ptr(links_c), _lib.ptr(olen), _lib.ptr(rows), _lib.ptr(amax), _lib.ptr(trace),
                                             B, T, L, TR, st), "dsp_dag_max_alpha")
        best = amax[ar, 2:, (olen - 1).clamp(min=0)]                     # [B, M]: best score of every length (:267-269)
        lengths = (torch.arange(max_length, device=dev) + 1).unsqueeze(0).float()
        _, pred_length = torch.max(best / lengths ** decode_viterbibeta, dim=1)
        pred_length = pred_length + 1                                    # (:275-276)
        path = torch.empty((B, L), dtype=torch.int64, device=dev)
        start_rows = (pred_length + 2).contiguous()                       # bound to a name: must outlive the launch
        if blocks:
            _lib.check(lib.dsp_dag_backtrace_blocks(_lib.ptr(amax), _lib.ptr(trace), _lib.ptr(links_c), _lib.ptr(olen), _lib.ptr(start_rows), _lib.ptr(path),
                                                    B, T, L, TR, st), "dsp_dag_backtrace_blocks")
        else:
            _lib.check(lib.dsp_dag_backtrace(_lib.ptr(trace), _lib.ptr(olen), _lib.ptr(start_rows), _lib.ptr(path), B, T, L, st),
                       "dsp_dag_backtrace")
    # the token pass (:283-299) in one kernel: vertices visited at DP rows 1 .. pred_length in graph order, a token kept if it is the last
    # visited one or (not pad and different from the next visited token).  The final vertex out of reach within max_length steps (a window far
    # narrower than the model's): every candidate score is -inf, the reference's arg-maxes all return index 0 (:267-278) and it emits the one
    # token of vertex 0 — reproduced on the device, not "fixed" (no host-side branch: it would synchronise every decode batch for a corner case)
    unreach = torch.isneginf(best).all(dim=1).to(torch.uint8).contiguous()
    feats = features.detach().contiguous()
    D = feats.shape[2]
    with torch.cuda.device(dev):
        cap = L
        out_tok = torch.empty((B, cap), dtype=torch.long, device=dev)
        keep = torch.empty((B, cap), dtype=torch.int32, device=dev)
        nk = torch.empty((B,), dtype=torch.int32, device=dev)
        pl = pred_length.to(torch.int64).contiguous()
        _lib.check(lib.dsp_viterbi_collect(_lib.ptr(path), _lib.ptr(pl), _lib.ptr(unreach), _lib.ptr(tok32), int(pad), _lib.ptr(out_tok), _lib.ptr(keep),
                                           _lib.ptr(nk), B, L, cap, _lib.current_stream_handle()), "dsp_viterbi_collect")
        fmax = int(nk.max().item()) if B else 0              # the one sync: output shapes depend on it
        out_feat = torch.empty((B, fmax, D), dtype=feats.dtype, device=dev)
        if fmax:
            _lib.check(lib.dsp_gather_rows(_lib.ptr(feats), _code(feats), _lib.ptr(keep), _lib.ptr(nk), _lib.ptr(out_feat),
                                           B, L, D, cap, fmax, _lib.current_stream_handle()), "dsp_gather_rows")
    n_keep = nk.to(torch.long)
    mask = torch.arange(fmax, device=dev).unsqueeze(0) >= n_keep.unsqueeze(1)
    return out_tok[:, :fmax].contiguous(), out_feat, mask, n_keep


def viterbi_decode_torch(logits: Tensor, links: Tensor, features: Tensor, output_length: Tensor, pad: int, decode_beta: float = 1.0,
                         decode_viterbibeta: float = 1.0, joint: bool = True, src_upsample_scale: float = 0.5):
    """The same decode as a restatement of the reference loop in torch (device-agnostic; the checker of `viterbi_decode`):
    max_length = int(L / 8 / scale) max-product steps over the dense links, length-normalised best end, back-trace by batched
    gathers (the reference back-traces per sample on the host).  Returns like graph_decode."""
    B, L, V = logits.shape
    dense = restore_valid_links(links)
    logp = torch.log_softmax(logits.float(), dim=-1)
    sc, tok = logp.max(dim=-1)                                           # unreduced_logits / unreduced_tokens (:207-208)
    alpha = dense[:, 0].clone()                                          # (:248) — the reference aliases links[:,0]; harmless
    if joint:
        alpha = alpha + sc[:, 0].unsqueeze(1) * decode_beta              # (:249-250)
    alpha = alpha + sc * decode_beta                                     # (:252) applied for both strategies
    max_length = max(1, int(L / 8 / src_upsample_scale))                 # (:256)
    scores, indexs = [alpha], []
    for _ in range(max_length - 1):
        alpha, index = torch.max(alpha.unsqueeze(-1) + dense, dim=1)     # (:258)
        if joint:
            alpha = alpha + sc * decode_beta
        scores.append(alpha); indexs.append(index)
    scores = torch.stack(scores, 0)                                      # [M,B,L]
    ar = torch.arange(B, device=logits.device)
    link_last = dense[ar, :, (output_length - 1)].unsqueeze(0)           # link of every vertex into the final vertex (:267)
    best, max_idx = torch.max(scores + link_last, dim=-1)                # [M,B]
    lengths = (torch.arange(max_length, device=logits.device) + 1).unsqueeze(-1).float()
    _, pred_length = torch.max(best / lengths ** decode_viterbibeta, dim=0)
    pred_length = pred_length + 1                                        # (:275-276)
    j = max_idx.gather(0, (pred_length - 1).unsqueeze(0)).squeeze(0)     # end vertex per sample (:278)
    # batched back-trace: path[b, k] = k-th vertex from the END
    path = torch.full((B, max_length), -1, dtype=torch.long, device=logits.device)
    path[:, 0] = j
    if indexs:
        idx_all = torch.stack(indexs, 0)                                 # [M-1,B,L]
        for k in range(max_length - 1):
            step = pred_length - k - 2                                   # (:287) indexs[length-k-2]
            live = step >= 0
            nj = idx_all[step.clamp(min=0), ar, j]
            j = torch.where(live, nj, j)
            path[:, k + 1] = torch.where(live, j, torch.full_like(j, -1))
    valid = path >= 0
    ptok = tok.gather(1, path.clamp(min=0))
    nxt_tok = torch.cat([torch.full((B, 1), -12345, device=ptok.device, dtype=ptok.dtype), ptok[:, :-1]], dim=1)   # token visited just before (backward order)
    keep = valid & ((torch.arange(max_length, device=ptok.device).unsqueeze(0) == 0) | ((ptok != pad) & (ptok != nxt_tok)))
    n_keep = keep.sum(1)
    fmax = int(n_keep.max().item())
    # forward order = reversed backward order; compact the kept entries to the left
    order = torch.argsort((~keep.flip(1)).to(torch.int8), dim=1, stable=True)
    fwd_path = path.flip(1).gather(1, order)[:, :fmax]
    fwd_tok = ptok.flip(1).gather(1, order)[:, :fmax]
    mask = torch.arange(fmax, device=ptok.device).unsqueeze(0) >= n_keep.unsqueeze(1)
    out_tok = fwd_tok.masked_fill(mask, pad)
    out_feat = features.gather(1, fwd_path.clamp(min=0).unsqueeze(-1).expand(-1, -1, features.shape[-1])).masked_fill(mask.unsqueeze(-1), 0)
    return out_tok, out_feat, mask, n_keep


def dwconv_bn_silu(x: Tensor, conv_weight: Tensor, bn: "torch.nn.BatchNorm1d") -> Tensor:
    """SiLU(BatchNorm_eval(depthwise_conv1d(x))) on channels-last x [B,T,C] — the middle of the Conformer convolution module in
    eval mode (include/daspeech_decode.h: dsp_dwconv_bn_silu).  conv_weight is the Conv1d(C, C, K, groups=C) weight [C,1,K]."""
    _gpu("dwconv_bn_silu", x, conv_weight)
    xf = x.detach().to(torch.float32).contiguous()
    B, T, C = xf.shape
    wt = conv_weight.detach().to(torch.float32).reshape(C, -1).contiguous()
    K = wt.shape[1]
    f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    bw, bb, bm, bv = f(bn.weight), f(bn.bias), f(bn.running_mean), f(bn.running_var)
    lib = _lib.load()
    with torch.cuda.device(xf.device):
        y = torch.empty_like(xf)
        _lib.check(lib.dsp_dwconv_bn_silu(_lib.ptr(xf), _lib.ptr(wt), _lib.ptr(bw), _lib.ptr(bb), _lib.ptr(bm), _lib.ptr(bv), float(bn.eps),
                                          _lib.ptr(y), B, T, C, K, _lib.current_stream_handle()), "dsp_dwconv_bn_silu")
    return y.to(x.dtype)


class SplitConv1d:
    """fp32-accurate Conv1d(Cin, Cout, K, padding=(K-1)//2) on the fp16 matrix cores (include/daspeech_decode.h: dsp_conv1d_split).
    Packs the weight once (hi / lo fp16 in MFMA fragment order, per 512-channel input slice); call with channels-last x [B,T,Cin]."""

    def __init__(self, weight: Tensor, bias: Optional[Tensor]):
        _gpu("SplitConv1d", weight)
        lib = _lib.load()
        Cout, Cin, K = weight.shape
        self.Cout, self.Cin, self.K = Cout, Cin, K
        self.bias = None if bias is None else bias.detach().float().contiguous()
        step = Cin if Cin <= 512 else 512          # (256-channel slices with 128-row tiles: 125 vs 95 us for 2048 -> 256; r04 for the 512-wide decoder GEMMs: 3 - 10 % faster, not taken)
        assert Cin % step == 0 and step in (128, 256, 512) and Cout % 4 == 0 and K % 2 == 1, (Cin, Cout, K)
        self.step, self.nslices = step, Cin // step
        with torch.cuda.device(weight.device):
            st = _lib.current_stream_handle()
            n = lib.dsp_conv1d_split_packed_elems(K, Cout, step)
            self.hi = torch.empty((self.nslices * n,), dtype=torch.float16, device=weight.device); self.lo = torch.empty_like(self.hi)
            for sl in range(self.nslices):
                wt = weight.detach().float()[:, sl * step:(sl + 1) * step, :].permute(2, 0, 1).contiguous()        # [K][Cout][step]
                _lib.check(lib.dsp_conv1d_split_pack(_lib.ptr(wt), ctypes.c_void_p(self.hi.data_ptr() + 2 * sl * n),
                                                     ctypes.c_void_p(self.lo.data_ptr() + 2 * sl * n), K, Cout, step, st), "dsp_conv1d_split_pack")

    ACT = {None: 0, "relu": 1, "silu": 2, "gelu": 3}
    KSPLIT = True              # short sequences: split deep reductions over workgroups (set False to time / compare the single-launch form)

    def _tap_groups(self, B: int, T: int) -> int:
        """tap groups for the split-K form, 0 = one launch: only when the layer would occupy under half of the 256 CUs with a reduction
        of >= 4 slice-taps per workgroup (the FastSpeech2 encoder's second FFT convolution: 1024 -> 256, K = 9 over ~60 positions)"""
        if not SplitConv1d.KSPLIT or self.Cout % 4:
            return 0
        wgs = B * ((T + 63) // 64) * ((self.Cout + 255) // 256)
        if wgs >= 128 or self.nslices * self.K < 4:
            return 0
        tg = 1
        while tg < self.K and wgs * self.nslices * tg < 256 and tg < 3:
            tg += 1
        return tg if self.nslices * tg >= 2 else 0


    def __call__(self, x: Tensor, relu: bool = False, act: Optional[str] = None, residual: Optional[Tensor] = None, alpha: float = 1.0,
                 lens: Optional[Tensor] = None, slack: int = 0) -> Tensor:
        """act(conv(x) + bias), or residual + alpha * that when a residual [B,T,Cout] is given.  lens [B] int32 (ragged batch): time tiles
        starting at or after lens[b] + slack are padding and are not computed: they come back as zeros (dsp_conv1d_split_ragged) or, in
        the split-K form of short sequences, as residual + alpha * act(bias) — finite either way."""
        _gpu("SplitConv1d", x)
        code = 1 if relu else self.ACT[act]
        assert x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == self.Cin and x.stride(2) == 1 and x.stride(0) == x.shape[1] * x.stride(1)
        B, T, _ = x.shape
        lib = _lib.load()
        with torch.cuda.device(x.device):
            st = _lib.current_stream_handle()
            out = torch.empty((B, T, self.Cout), dtype=torch.float32, device=x.device)
            tg = self._tap_groups(B, T)
            if lens is not None:
                assert lens.dtype == torch.int32 and lens.is_cuda and lens.is_contiguous() and lens.numel() == B
            if tg:
                # short sequence: split the reduction over slices x tap groups so that the launch fills the chip (dsp_conv1d_split_ksplit)
                r = None
                if residual is not None:
                    r = residual if (residual.dtype == torch.float32 and residual.is_contiguous()) else residual.float().contiguous()
                    assert tuple(r.shape) == (B, T, self.Cout)
                nws = lib.dsp_conv1d_split_ksplit_workspace_bytes(B, T, self.Cout, self.nslices, tg)
                ws = torch.empty((nws // 4,), dtype=torch.float32, device=x.device)
                _lib.check(lib.dsp_conv1d_split_ksplit(_lib.ptr(x), x.stride(1), _lib.ptr(self.hi), _lib.ptr(self.lo), _lib.ptr(self.bias), _lib.ptr(r), self.Cout,
                                                       float(alpha), _lib.ptr(out), self.Cout, B, T, self.step, self.nslices, self.Cout, self.K, code, tg,
                                                       _lib.ptr(ws), nws, _lib.ptr(lens), int(slack), st), "dsp_conv1d_split_ksplit")
            elif lens is not None:
                r = None
                if residual is not None:
                    r = residual if (residual.dtype == torch.float32 and residual.is_contiguous()) else residual.float().contiguous()
                    assert tuple(r.shape) == (B, T, self.Cout)
                assert lens.dtype == torch.int32 and lens.is_cuda and lens.is_contiguous() and lens.numel() == B
                _lib.check(lib.dsp_conv1d_split_ragged(_lib.ptr(x), x.stride(1), _lib.ptr(self.hi), _lib.ptr(self.lo), _lib.ptr(self.bias), _lib.ptr(r),
                                                       self.Cout, float(alpha), _lib.ptr(out), self.Cout, B, T, self.step, self.nslices, self.Cout,
                                                       self.K, code, _lib.ptr(lens), int(slack), st), "dsp_conv1d_split_ragged")
            elif residual is not None or alpha != 1.0:
                r = None
                if residual is not None:
                    r = residual if (residual.dtype == torch.float32 and residual.is_contiguous()) else residual.float().contiguous()
                    assert tuple(r.shape) == (B, T, self.Cout)
                _lib.check(lib.dsp_conv1d_split_residual(_lib.ptr(x), x.stride(1), _lib.ptr(self.hi), _lib.ptr(self.lo), _lib.ptr(self.bias), _lib.ptr(r),
                                                         self.Cout, float(alpha), _lib.ptr(out), self.Cout, B, T, self.step, self.nslices, self.Cout,
                                                         self.K, code, st), "dsp_conv1d_split_residual")
            else:
                _lib.check(lib.dsp_conv1d_split(_lib.ptr(x), x.stride(1), _lib.ptr(self.hi), _lib.ptr(self.lo), _lib.ptr(self.bias), _lib.ptr(out),
                                                self.Cout, B, T, self.step, self.nslices, self.Cout, self.K, code, 0, st), "dsp_conv1d_split")
        return out


SPLIT_GEMM = True          # set_split_gemm(False): every Linear / FFT convolution goes back to torch (hipBLASLt / MIOpen fp32)


def set_split_gemm(on: bool) -> bool:
    """Switch the fp32-accurate matrix-core GEMM / conv path of eval-mode inference on or off; returns the previous setting."""
    global SPLIT_GEMM
    old, SPLIT_GEMM = SPLIT_GEMM, bool(on)
    return old


def valid_lengths(pad_mask: Optional[Tensor]) -> Optional[Tensor]:
    """[B] int32: one past the last position that is not padding (a ragged batch's row bounds for the `lens` arguments below)"""
    if pad_mask is None or not pad_mask.is_cuda:
        return None
    T = pad_mask.shape[1]
    return ((~pad_mask) * torch.arange(1, T + 1, device=pad_mask.device, dtype=torch.int32)).amax(1).to(torch.int32).contiguous()


def split_linear(x: Tensor, lin, act: Optional[str] = None, residual: Optional[Tensor] = None, alpha: float = 1.0,
                 lens: Optional[Tensor] = None, slack: int = 0) -> Optional[Tensor]:
    """act(x @ W^T + b) at fp32 accuracy on the fp16 matrix cores (a SplitConv1d with one tap), or None when the shape / mode is not
    served (caller falls back to F.linear): eval-mode inference in fp32 on the GPU, in_features 128 / 256 / 512 or a multiple of 512,
    out_features % 4 == 0, at least 128 rows.  The packed weight is cached on the module."""
    if (not SPLIT_GEMM or torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32 or torch.is_autocast_enabled() or lin.weight.dtype != torch.float32
            or x.dim() != 3 or not x.is_contiguous() or x.shape[0] * x.shape[1] < 128):
        return None
    wt = lin.weight                                                # nn.Linear [out,in] or a kernel-1 nn.Conv1d [out,in,1]
    if wt.dim() == 3 and wt.shape[2] != 1:
        return None
    cout, cin = wt.shape[0], wt.shape[1]
    if not ((cin in (128, 256, 512) or (cin > 512 and cin % 512 == 0)) and cout % 4 == 0 and cout >= 128):
        return None                                                # narrow outputs (gates, mel projection) waste the 256-row weight tile
    bias = getattr(lin, "bias", None)                              # nn.Embedding used as a tied output projection has none
    key = (lin.weight.data_ptr(), lin.weight._version, None if bias is None else bias._version)
    cache = getattr(lin, "_dsp_split", None)
    if cache is None or cache[0] != key:
        cache = (key, SplitConv1d(wt if wt.dim() == 3 else wt.unsqueeze(-1), bias))
        lin._dsp_split = cache
    return cache[1](x, act=act, residual=residual, alpha=alpha, lens=lens, slack=slack)


def linear(x: Tensor, lin, act: Optional[str] = None, residual: Optional[Tensor] = None, alpha: float = 1.0, lens: Optional[Tensor] = None,
           slack: int = 0) -> Tensor:
    """[residual + alpha *] act(lin(x)): through split_linear where it applies (eval-mode fp32 inference on the GPU), torch otherwise.
    `lin` is an nn.Linear or a kernel-1 nn.Conv1d (applied on the channels-last x).  lens [B] int32: rows from lens[b] + slack on are padding
    that may come back as zeros (whole tiles are skipped on the split path; the torch path computes them)."""
    if not lin.training:
        y = split_linear(x, lin, act, residual, alpha, lens, slack)
        if y is not None:
            return y
    y = torch.nn.functional.linear(x, lin.weight if lin.weight.dim() == 2 else lin.weight.squeeze(-1), getattr(lin, "bias", None))
    if act is not None:
        y = {"relu": torch.relu, "silu": torch.nn.functional.silu, "gelu": torch.nn.functional.gelu}[act](y)
    if alpha != 1.0:
        y = alpha * y
    return y if residual is None else residual + y


class _CatLinear:
    """the weights / biases of several nn.Linear over the same input stacked into one [sum(out), in] projection"""

    def __init__(self, lins):
        self.weight = torch.cat([l.weight.detach() for l in lins], 0)
        self.bias = torch.cat([(l.bias.detach() if l.bias is not None else torch.zeros(l.weight.shape[0], dtype=l.weight.dtype, device=l.weight.device))
                               for l in lins], 0)
        self.training = False


def linear_fused(x: Tensor, lins, lens: Optional[Tensor] = None, slack: int = 0) -> tuple:
    """(lin(x) for lin in lins) for nn.Linear modules sharing the input x: in eval-mode fp32 inference on the GPU ONE split GEMM over the
    stacked weights (cached on the first module) whose output columns are returned as row-strided views — every output column sees the
    same reduction as in its own GEMM, so the values are bit-identical to separate calls; separate `linear` calls otherwise."""
    first = lins[0]
    if (SPLIT_GEMM and not first.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and x.dim() == 3 and x.is_contiguous() and x.shape[0] * x.shape[1] >= 128 and all(l.weight.dim() == 2 and l.weight.dtype == torch.float32 for l in lins)):
        key = tuple((l.weight.data_ptr(), l.weight._version, None if l.bias is None else l.bias._version) for l in lins)
        cache = getattr(first, "_dsp_cat", None)
        if cache is None or cache[0] != key:
            cache = (key, _CatLinear(lins))
            first._dsp_cat = cache
        y = split_linear(x, cache[1], lens=lens, slack=slack)
        if y is not None:
            outs, o = [], 0
            for l in lins:
                outs.append(y[..., o:o + l.weight.shape[0]]); o += l.weight.shape[0]
            return tuple(outs)
    return tuple(linear(x, l, lens=lens, slack=slack) for l in lins)


def _packed(lin) -> Optional["SplitConv1d"]:
    """the SplitConv1d (packed hi / lo weights) split_linear caches on an nn.Linear, built on first use"""
    wt = lin.weight
    bias = getattr(lin, "bias", None)
    key = (wt.data_ptr(), wt._version, None if bias is None else bias._version)
    cache = getattr(lin, "_dsp_split", None)
    if cache is None or cache[0] != key:
        cache = (key, SplitConv1d(wt if wt.dim() == 3 else wt.unsqueeze(-1), bias))
        lin._dsp_split = cache
    return cache[1]


def ffn_fused(x: Tensor, ln: Optional["torch.nn.LayerNorm"], lin1, lin2, act: str, residual: Optional[Tensor] = None, alpha: float = 1.0,
              post_ln: Optional["torch.nn.LayerNorm"] = None, need_out: bool = True):
    """[residual +] alpha * lin2(act(lin1(ln(x)))) in one matrix-core launch at fp32 accuracy (dsp_ffn_split: the hidden activations stay in
    LDS, LayerNorm runs while the tile is staged), or None when the shape / mode is not served: eval-mode fp32 inference on the GPU,
    256 channels in and out, hidden width a multiple of 512.  With post_ln the reduction also applies that LayerNorm to its result (the
    next block of a pre-norm layer starts with one) and the call returns (out, post_ln(out)); need_out=False: (None, post_ln(out))."""
    if (not SPLIT_GEMM or torch.is_grad_enabled() or lin1.training or not x.is_cuda or x.dtype != torch.float32 or torch.is_autocast_enabled() or x.dim() != 3
            or not x.is_contiguous() or lin1.weight.dtype != torch.float32 or lin1.weight.dim() != 2 or lin2.weight.dim() != 2
            or lin2.weight.dtype != torch.float32 or x.data_ptr() % 16):
        return None                                   # (null biases are handled by the kernels: `if (p.b1)`, `if (p.b2)`)
    B, T, C = x.shape
    H = lin1.weight.shape[0]
    if C != 256 or tuple(lin1.weight.shape) != (H, C) or tuple(lin2.weight.shape) != (C, H) or H % 512 or B * T < 128:
        return None
    for n_ in (ln, post_ln):
        if n_ is not None and (tuple(n_.normalized_shape) != (C,) or n_.weight is None or n_.bias is None or n_.weight.dtype != torch.float32
                               or n_.weight.data_ptr() % 16 or n_.bias.data_ptr() % 16):
            return None
    lib = _lib.load()
    p1, p2 = _packed(lin1), _packed(lin2)
    r = None
    if residual is not None:
        r = residual if (residual.dtype == torch.float32 and residual.is_contiguous()) else residual.float().contiguous()
        assert tuple(r.shape) == (B, T, C)
    with torch.cuda.device(x.device):
        nws = lib.dsp_ffn_split_workspace_bytes(B, T, C, H)
        ws = torch.empty((nws // 4,), dtype=torch.float32, device=x.device)
        out = torch.empty_like(x) if (need_out or post_ln is None) else None
        out_ln = None if post_ln is None else torch.empty_like(x)
        _lib.check(lib.dsp_ffn_split(_lib.ptr(x), x.stride(1), _lib.ptr(None if ln is None else ln.weight), _lib.ptr(None if ln is None else ln.bias),
                                     float(ln.eps) if ln is not None else 0.0, _lib.ptr(p1.hi), _lib.ptr(p1.lo), _lib.ptr(p1.bias), _lib.ptr(p2.hi),
                                     _lib.ptr(p2.lo), _lib.ptr(p2.bias), _lib.ptr(r), C, float(alpha), _lib.ptr(out), C, _lib.ptr(ws), nws, B, T, C, H,
                                     SplitConv1d.ACT[act], _lib.ptr(None if post_ln is None else post_ln.weight),
                                     _lib.ptr(None if post_ln is None else post_ln.bias), float(post_ln.eps) if post_ln is not None else 0.0,
                                     _lib.ptr(out_ln), _lib.current_stream_handle()), "dsp_ffn_split")
    return out if post_ln is None else (out, out_ln)


def linear_ln(x: Tensor, ln: "torch.nn.LayerNorm", lins, act: Optional[str] = None, lens: Optional[Tensor] = None, slack: int = 0) -> tuple:
    """(lin(ln(x)) for lin in lins): for 256-channel inputs in eval-mode fp32 inference on the GPU ONE split GEMM over the stacked weights with
    the LayerNorm applied while the row tile is staged (dsp_linear_ln_split) — no LayerNorm launch, no normalised copy of x; otherwise
    layer_norm followed by linear_fused."""
    first = lins[0]
    if (SPLIT_GEMM and not first.training and not ln.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
            and not torch.is_autocast_enabled() and x.dim() == 3 and x.is_contiguous() and x.shape[2] == 256 and x.shape[0] * x.shape[1] >= 128
            and tuple(ln.normalized_shape) == (256,) and ln.weight is not None and ln.bias is not None and ln.weight.dtype == torch.float32
            and all(l.weight.dim() in (2, 3) and l.weight.shape[1] == 256 and l.weight.dtype == torch.float32 and (l.weight.dim() == 2 or l.weight.shape[2] == 1)
                    for l in lins)):
        if len(lins) == 1:
            pk = _packed(first)
        else:
            key = tuple((l.weight.data_ptr(), l.weight._version, None if l.bias is None else l.bias._version) for l in lins)
            cache = getattr(first, "_dsp_cat", None)
            if cache is None or cache[0] != key:
                cache = (key, _CatLinear(lins))
                first._dsp_cat = cache
            pk = _packed(cache[1])
        if pk.Cout % 4 == 0 and pk.Cout >= 128:
            B, T, _ = x.shape
            lib = _lib.load()
            with torch.cuda.device(x.device):
                y = torch.empty((B, T, pk.Cout), dtype=torch.float32, device=x.device)
                _lib.check(lib.dsp_linear_ln_split(_lib.ptr(x), x.stride(1), _lib.ptr(ln.weight), _lib.ptr(ln.bias), float(ln.eps), _lib.ptr(pk.hi), _lib.ptr(pk.lo),
                                                   _lib.ptr(pk.bias), None, 0, 1.0, _lib.ptr(y), pk.Cout, B, T, pk.Cout, SplitConv1d.ACT[act], _lib.ptr(lens),
                                                   int(slack), _lib.current_stream_handle()), "dsp_linear_ln_split")
            outs, o = [], 0
            for l in lins:
                outs.append(y[..., o:o + l.weight.shape[0]]); o += l.weight.shape[0]
            return tuple(outs)
    xn = layer_norm(x, ln)
    if len(lins) == 1:
        return (linear(xn, first, act=act, lens=lens, slack=slack),)
    assert act is None
    return linear_fused(xn, lins, lens=lens, slack=slack)


def layer_norm(x: Tensor, ln: "torch.nn.LayerNorm") -> Tensor:
    """ln(x): the one-wave-per-row HIP kernel (dsp_layer_norm) in eval-mode fp32 inference on the GPU, torch otherwise."""
    C = x.shape[-1]
    if (SPLIT_GEMM and not ln.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and x.is_contiguous() and tuple(ln.normalized_shape) == (C,) and C % 4 == 0 and C <= 2048 and (ln.weight is None or ln.weight.dtype == torch.float32)):
        lib = _lib.load()
        with torch.cuda.device(x.device):
            y = torch.empty_like(x)
            _lib.check(lib.dsp_layer_norm(_lib.ptr(x), _lib.ptr(ln.weight), _lib.ptr(ln.bias), float(ln.eps), _lib.ptr(y), x.numel() // C, C,
                                          _lib.current_stream_handle()), "dsp_layer_norm")
        return y
    return ln(x)


def relpos_attention(q: Tensor, k: Tensor, v: Tensor, p: Tensor, bias_u: Tensor, bias_v: Tensor, pad_mask: Optional[Tensor], heads: int) -> Optional[Tensor]:
    """Fused Conformer relative-position attention (dsp_relpos_attention): q, k, v [B,T,C] fp32 (C = heads * 64) with one common row stride
    (contiguous tensors or the three column slices of a fused projection), p [1 or none, 2T-1, C], bias_u / bias_v [heads, 64], pad_mask
    [B,T] bool.  Returns [B,T,C] contiguous, or None when the shape / mode is not served."""
    B, T, C = q.shape
    ld = q.stride(1)
    if (not SPLIT_GEMM or torch.is_grad_enabled() or not q.is_cuda or q.dtype != torch.float32 or torch.is_autocast_enabled() or C != heads * 64
            or any(t.stride(2) != 1 or t.stride(1) != ld or t.stride(0) != T * ld or t.data_ptr() % 16 or t.shape != q.shape for t in (q, k, v)) or ld % 4):
        return None
    pp = p.reshape(-1, C).contiguous()
    if pp.shape[0] != 2 * T - 1 or pp.dtype != torch.float32 or tuple(bias_u.shape) != (heads, 64) or tuple(bias_v.shape) != (heads, 64):
        return None
    lib = _lib.load()
    pm = _mask_bytes(pad_mask)
    bu, bv = bias_u.detach().float().contiguous(), bias_v.detach().float().contiguous()      # named: must outlive the launch
    # the kernel reads all three with 16-byte loads: a view at an odd storage offset gets an aligned copy instead of a C-side error
    pp, bu, bv = (t if t.data_ptr() % 16 == 0 else t.clone() for t in (pp, bu, bv))
    with torch.cuda.device(q.device):
        out = torch.empty((B, T, C), dtype=torch.float32, device=q.device)
        _lib.check(lib.dsp_relpos_attention(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), ld, _lib.ptr(pp), _lib.ptr(bu), _lib.ptr(bv), _lib.ptr(pm),
                                            _lib.ptr(out), B, T, heads, 64, _lib.current_stream_handle()), "dsp_relpos_attention")
    return out


def attention(q: Tensor, k: Tensor, v: Tensor, key_pad_mask: Optional[Tensor], heads: int, q_lens: Optional[Tensor] = None,
              q_slack: int = 0) -> Optional[Tensor]:
    """softmax(q k^T / sqrt(dk) + key padding) v per head at fp32 accuracy on the fp16 matrix cores (dsp_attention_split): q [B,N,C],
    k / v [B,M,C] fp32, possibly column slices of a wider projection output (row-strided views; unit stride inside a row), C = heads * dk
    with dk 64 or 128, key_pad_mask [B,M] bool or None.  Returns [B,N,C] contiguous, or None when the shape / mode is not served
    (training, autocast, other head widths: the caller keeps torch's scaled_dot_product_attention).  q_lens [B] int32: queries from
    q_lens[b] + q_slack on are padding; their 32-query groups are skipped and come back as zeros."""
    B, N, C = q.shape
    M = k.shape[1]
    dk = C // heads
    if (not SPLIT_GEMM or torch.is_grad_enabled() or not q.is_cuda or torch.is_autocast_enabled() or dk * heads != C or dk not in (64, 128)
            or any(t.dtype != torch.float32 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1) or t.stride(1) % 4 or t.data_ptr() % 16
                   for t in (q, k, v)) or k.shape != v.shape or k.shape[0] != B or k.shape[2] != C or N < 1 or M < 1):
        return None
    lib = _lib.load()
    pm = _mask_bytes(key_pad_mask)
    with torch.cuda.device(q.device):
        out = torch.empty((B, N, C), dtype=torch.float32, device=q.device)
        _lib.check(lib.dsp_attention_split(_lib.ptr(q), q.stride(1), _lib.ptr(k), k.stride(1), _lib.ptr(v), v.stride(1), _lib.ptr(pm), _lib.ptr(out),
                                           B, N, M, heads, dk, float(dk) ** -0.5, _lib.ptr(q_lens), int(q_slack), _lib.current_stream_handle()),
                   "dsp_attention_split")
    return out
