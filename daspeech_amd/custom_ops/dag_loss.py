"""Operator API of the DAG dynamic-programming ops — drop-in for DASpeech/custom_ops/dag_loss.py.

Same eight public names, argument order, return types and side effects as the reference
(DASpeech/custom_ops/__init__.py:1):

    dag_loss(match_all, links, output_length, target_length) -> loss[B]                 (dag_loss.py:66-121,187)
    dag_loss_with_alpha_beta(...) -> (loss[B], (alpha, beta))                            (dag_loss.py:123-188)
    dag_best_alignment(...) -> path[B,L] int64                                           (dag_loss.py:190-236)
    dag_logsoftmax_gather_inplace(word_ins_out, select_idx) -> (word_ins_out, match)     (dag_loss.py:238-299)
    torch_dag_loss / torch_dag_best_alignment / torch_dag_logsoftmax_gather_inplace / logsumexp_keepdim
        the pure-torch variants selected by the criteria's --torch-dag-* flags             (dag_loss.py:303-425)

The first four run hand-written HIP kernels (gfx950) through the C ABI of include/daspeech_dag.h; they raise if
the shared library is missing or the tensors are not on a GPU — there is NO silent fallback to the torch path.
Differences from the reference, all deliberate and documented in DESIGN.md:
  * kernels run on torch's CURRENT stream (the reference used legacy stream 0 + private streams);
  * `match` returned by dag_logsoftmax_gather_inplace is a [B,L,S]-shaped *view* of a contiguous [B,S,L] buffer, so
    the caller's `.transpose(1, 2)` (nat_dag_loss.py:128) is already contiguous and dag_loss's `.contiguous()` copies
    nothing; values are identical;
  * invalid samples (unreachable end, bad lengths) give -inf / zero gradients instead of device asserts
    (dag_loss.cu:68-69) — the criteria already zero non-finite losses (nat_dag_loss.py:143-145);
  * Viterbi ties follow the torch implementation's rule (smallest predecessor index), see SURVEY.md §7.
"""
import os
from typing import Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from .. import _lib
from . import dag_double

__all__ = ["dag_loss", "dag_loss_with_alpha_beta", "dag_best_alignment", "dag_logsoftmax_gather_inplace",
           "torch_dag_loss", "torch_dag_best_alignment", "torch_dag_logsoftmax_gather_inplace", "logsumexp_keepdim"]


# ----------------------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------------------

def _require_gpu(name: str, *tensors: Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(f"{name}: expected GPU tensors (the HIP ops have no CPU path; use torch_{name} on CPU)")
        dev = dev or t.device
        if t.device != dev:
            raise RuntimeError(f"{name}: tensors are on different devices")
    return dev


def _check_dp_args(name, match_all, links, output_length, target_length):
    if match_all.dim() != 3 or links.dim() != 3:
        raise RuntimeError(f"{name}: match_all and links must be 3-D (got {match_all.dim()}-D and {links.dim()}-D)")
    B, T, L = match_all.shape
    if links.shape[0] != B or links.shape[1] != L:
        raise RuntimeError(f"{name}: links must be [batch, prelen, translen] = [{B}, {L}, *], got {tuple(links.shape)}")
    if output_length.shape != (B,) or target_length.shape != (B,):
        raise RuntimeError(f"{name}: output_length / target_length must have shape [{B}]")
    if links.shape[2] < 1 or T < 1 or L < 1:
        raise RuntimeError(f"{name}: empty dimension")
    if output_length.dtype != torch.long or target_length.dtype != torch.long:
        raise RuntimeError(f"{name}: output_length / target_length must be int64")
    return B, T, L, links.shape[2]


def _f32c(t: Tensor) -> Tensor:
    """fp32 working copy (a no-op for the fp32 contiguous tensors the criteria pass).  fp16 / bf16 inputs are WIDENED — the DP runs in
    fp32 where the reference's half instantiation accumulates in half (dag_loss.cu:160), results go back in the caller's dtype.  float64
    never gets here: dag_loss / dag_loss_with_alpha_beta / dag_best_alignment route it to dag_double.py (narrowing silently would hand
    back fp32 accuracy in a double tensor)."""
    if t.dtype == torch.float64:
        raise RuntimeError("internal: float64 reached the fp32 HIP launch path (dag_double.py serves double inputs)")
    return t.detach().to(torch.float32).contiguous()


# Scratch of the DP launches comes from torch's caching allocator PER CALL (like the reference's per-call ATen scratch,
# dag_loss.cu:154): stream-ordered, returned to the cache when the call returns (the C ABI moves its status words out of it on the
# launch stream), subject to the allocator's OOM retry, and graph-capturable.  The C ABI zeroes what it uses on the launch stream,
# so nothing is allocated, freed or kept by the library.  (r02 kept a grow-only tensor per (device, stream) alive for the life of
# the process — 2.1 GB at C2 with the README's dense window, pinned outside the allocator.)
def _workspace(dev: torch.device, nbytes: int) -> Tensor:
    return torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)


# The banded fast paths (exp-space strips, values-only max-DP, fused gradients; windows <= 128) read rows with 16-byte loads.  A graph length is
# floor(src_upsample * frames) — any integer; three graphs in four are not a multiple of 4.  r05 padded such inputs to the next multiple with
# F.pad copies of match / links and cut alpha / beta / the gradients back (~4 extra passes over [B,T,L]).  r06: the C ABI takes a ROW PITCH
# (dsp_dag_loss_fwd_ld & co.), dag_logsoftmax_gather_inplace writes `match` with a pitch rounded up to 4, alpha / beta / grad_match are
# allocated with that pitch and handed on as [:, :, :L] views: no copy on the training path.  (The reference's contract is `.contiguous()`,
# dag_loss.py:103-104.)
def _round4(n: int) -> int:
    return (n + 3) & ~3


def _row_pitch(t: Tensor):
    """ld if `t` [B,T,L] fp32 is laid out as rows of pitch ld (a multiple of 4 covering L, 16-byte aligned base, batch stride T*ld), else 0."""
    if t.dtype != torch.float32 or t.dim() != 3:
        return 0
    B, T, L = t.shape
    sb, st, sl = t.stride()
    ld = st if T > 1 else (sb if B > 1 else _round4(L))
    if sl != 1 or ld % 4 or ld < _round4(L) or (B > 1 and sb != T * ld) or t.data_ptr() % 16:
        return 0
    if t.storage_offset() + ((B - 1) * T + (T - 1)) * ld + _round4(L) > t.untyped_storage().nbytes() // 4:
        return 0                                     # the last row's 16-byte tail must lie inside the allocation
    return ld


_PITCH_FILL = os.environ.get("DSP_PITCH_FILL")          # debug: "nan" poisons the pitch padding (nothing may ever read it)


def _pitched_empty(B: int, T: int, L: int, dev, fill=None) -> Tensor:
    """[B,T,L] fp32 view of a [B,T,round4(L)] buffer."""
    if fill is None and _PITCH_FILL:
        fill = float(_PITCH_FILL)
    buf = torch.empty((B, T, _round4(L)), dtype=torch.float32, device=dev) if fill is None else \
        torch.full((B, T, _round4(L)), fill, dtype=torch.float32, device=dev)
    return buf[:, :, :L] if buf.shape[2] != L else buf


def _as_pitched(match_all: Tensor):
    """(fp32 [B,T,L] with 16-byte aligned rows, its pitch): the caller's tensor when it already is laid out so (the gather's output, any
    dense tensor whose L is a multiple of 4), else ONE copy into a pitched buffer."""
    m = match_all.detach()
    ld = _row_pitch(m)
    if ld:
        return m, ld
    B, T, L = m.shape
    out = _pitched_empty(B, T, L, m.device)
    out.copy_(m)                                     # widens fp16 / bf16 on the way (the DP runs in fp32, see _f32c)
    return out, _round4(L)


def _dag_forward(match_all, links, output_length, target_length, need_beta: bool):
    """-> (m, k, ol, tl, alpha, beta, loss, (ld_match, ld_ab)); alpha / beta are [B,T,L] (views of pitched buffers on the strip kernels: windows <= 128)."""
    dev = _require_gpu("dag_loss", match_all, links, output_length, target_length)
    B, T, L, TR = _check_dp_args("dag_loss", match_all, links, output_length, target_length)
    if match_all.dtype == torch.float64 or links.dtype == torch.float64:
        raise RuntimeError("internal: float64 reached the fp32 HIP launch path (dag_double.py serves double inputs)")
    k = _f32c(links)
    ol = output_length.contiguous()
    tl = target_length.contiguous()
    lib = _lib.load()
    pitched = bool(lib.dsp_dag_pitch_supported(0, L, TR)) and k.data_ptr() % 16 == 0
    with torch.cuda.device(dev):
        if pitched:
            m, ldm = _as_pitched(match_all)
            lda = _round4(L)
            alpha = _pitched_empty(B, T, L, dev)
            beta = _pitched_empty(B, T, L, dev) if need_beta else None
        else:
            m, ldm, lda = _f32c(match_all), L, L
            alpha = torch.empty((B, T, L), dtype=torch.float32, device=dev)
            beta = torch.empty((B, T, L), dtype=torch.float32, device=dev) if need_beta else None
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        wsz = lib.dsp_dag_workspace_bytes(B, T, _round4(L) if pitched else L, TR)
        ws = _workspace(dev, wsz)
        rc = lib.dsp_dag_loss_fwd_ld(_lib.ptr(m), ldm, _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _lib.ptr(beta), lda,
                                     _lib.ptr(loss), B, T, L, TR, _lib.ptr(ws), ws.numel(), _lib.current_stream_handle())
        _lib.check(rc, "dsp_dag_loss_fwd")
    return m, k, ol, tl, alpha, beta, loss, (ldm, lda)


def _dag_backward(grad_output, alpha, beta, m, k, ol, tl, need_match: bool, need_links: bool, lds=None):
    B, T, L = m.shape
    TR = k.shape[2]
    dev = m.device
    lib = _lib.load()
    ldm, lda = lds if lds is not None else (L, L)
    with torch.cuda.device(dev):
        go = grad_output.detach().to(torch.float32).contiguous()
        gm, ldg = None, L
        if need_match:
            if lda != L or lda % 4 == 0:                         # (the fused gradient kernel wants 16-byte rows for grad_match too)
                gm, ldg = _pitched_empty(B, T, L, dev), _round4(L)
            else:
                gm = torch.empty((B, T, L), dtype=torch.float32, device=dev)
        gl = torch.empty_like(k) if need_links else None
        wsz, ws = 0, None                       # the gradient kernels keep no scratch
        rc = lib.dsp_dag_loss_bwd_ld(_lib.ptr(go), _lib.ptr(alpha), _lib.ptr(beta), lda, _lib.ptr(m), ldm, _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl),
                                     _lib.ptr(gm), ldg, _lib.ptr(gl), B, T, L, TR, _lib.ptr(ws), wsz, _lib.current_stream_handle())
        _lib.check(rc, "dsp_dag_loss_bwd")
    return gm, gl


# ----------------------------------------------------------------------------------------------------------------
# HIP-backed autograd functions (class names and tunable class attributes kept from the reference)
# ----------------------------------------------------------------------------------------------------------------

class DagLossFunc(Function):
    # launch-config knobs of the CUDA tuner (dag_loss.py:67-69); kept so `DagLossFunc.config = n` does not break
    # callers — the HIP kernels pick their own geometry.
    config = 1
    config1 = 2
    config2 = 2

    @staticmethod
    def forward(ctx, match_all, links, output_length, target_length):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        m, k, ol, tl, alpha, beta, loss, ctx.lds = _dag_forward(match_all, links, output_length, target_length, need)
        ctx.save_for_backward(alpha, beta if need else alpha, m, k, ol, tl)
        ctx.in_dtypes = (match_all.dtype, links.dtype)
        return loss.to(match_all.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None
        alpha, beta, m, k, ol, tl = ctx.saved_tensors
        gm, gl = _dag_backward(grad_output, alpha, beta, m, k, ol, tl, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.lds)
        gm = gm.to(ctx.in_dtypes[0]) if gm is not None else None
        gl = gl.to(ctx.in_dtypes[1]) if gl is not None else None
        return gm, gl, None, None


class DagLossWithAlphaBetaFunc(Function):
    config = 1
    config1 = 2
    config2 = 2

    @staticmethod
    def forward(ctx, match_all, links, output_length, target_length):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        m, k, ol, tl, alpha, beta, loss, ctx.lds = _dag_forward(match_all, links, output_length, target_length, need)
        ctx.save_for_backward(alpha, beta if need else alpha, m, k, ol, tl)
        ctx.in_dtypes = (match_all.dtype, links.dtype)
        if beta is None:
            # no gradient required: the reference launches no beta kernel and hands back the table as allocated, all zeros
            # (dag_loss.cu:339-340,355-371) — callers (the expect strategy in validation / while the DAG is frozen) compute with it
            beta = torch.zeros_like(alpha)
        if not alpha.is_contiguous():                                # (graph length not a multiple of 4: the tables sit in pitched buffers; the
            alpha, beta = alpha.contiguous(), beta.contiguous()      #  reference hands out dense tensors, and so does this operator)
        ctx.mark_non_differentiable(alpha, beta)
        return loss.to(match_all.dtype), (alpha, beta)

    @staticmethod
    def backward(ctx, grad_output, unused):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None
        alpha, beta, m, k, ol, tl = ctx.saved_tensors
        gm, gl = _dag_backward(grad_output, alpha, beta, m, k, ol, tl, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.lds)
        gm = gm.to(ctx.in_dtypes[0]) if gm is not None else None
        gl = gl.to(ctx.in_dtypes[1]) if gl is not None else None
        return gm, gl, None, None


def _any_double(*ts) -> bool:
    return any(t.dtype == torch.float64 for t in ts)


def dag_loss(match_all, links, output_length, target_length):
    """`DagLossFunc.apply` (dag_loss.py:188) — float64 inputs take the double-precision band DP of dag_double.py (the reference
    dispatches a double instantiation, dag_loss.cu:160; the HIP kernels compute in fp32)."""
    if _any_double(match_all, links):
        _require_gpu("dag_loss", match_all, links, output_length, target_length)
        _check_dp_args("dag_loss", match_all, links, output_length, target_length)
        return dag_double.dag_loss(match_all.double(), links.double(), output_length, target_length)
    return DagLossFunc.apply(match_all, links, output_length, target_length)


def dag_loss_with_alpha_beta(match_all, links, output_length, target_length):
    if _any_double(match_all, links):
        _require_gpu("dag_loss", match_all, links, output_length, target_length)
        _check_dp_args("dag_loss", match_all, links, output_length, target_length)
        return dag_double.dag_loss_with_alpha_beta(match_all.double(), links.double(), output_length, target_length)
    return DagLossWithAlphaBetaFunc.apply(match_all, links, output_length, target_length)


class DagBestAlignmentFunc(Function):
    config = 1

    @staticmethod
    def forward(ctx, match_all, links, output_length, target_length):
        dev = _require_gpu("dag_best_alignment", match_all, links, output_length, target_length)
        B, T, L, TR = _check_dp_args("dag_best_alignment", match_all, links, output_length, target_length)
        k = _f32c(links)
        ol = output_length.contiguous()
        tl = target_length.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            # the banded fast path keeps values only and back-traces lazily from alpha_max: no [B,T,L] trace tensor (the reference always writes
            # one); it reads 16-byte rows, so match / alpha_max go in with a row pitch (see _dag_forward)
            pitched = bool(lib.dsp_dag_pitch_supported(1, L, TR)) and k.data_ptr() % 16 == 0
            path = torch.empty((B, L), dtype=torch.long, device=dev)
            if pitched:
                m, ldm = _as_pitched(match_all)
                alpha = _pitched_empty(B, T, L, dev)
                ws = _workspace(dev, lib.dsp_dag_alignment_workspace_bytes(B, T, _round4(L), TR))
                rc = lib.dsp_dag_best_alignment_ld(_lib.ptr(m), ldm, _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _round4(L), None,
                                                   _lib.ptr(path), B, T, L, TR, _lib.ptr(ws), ws.numel(), _lib.current_stream_handle())
                _lib.check(rc, "dsp_dag_best_alignment_ld")
            else:
                m = _f32c(match_all)
                alpha = torch.empty((B, T, L), dtype=torch.float32, device=dev)
                lazy_ok = bool(lib.dsp_dag_alignment_trace_optional(L, TR)) and m.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0
                trace = None if lazy_ok else torch.empty((B, T, L), dtype=torch.int32, device=dev)
                ws = _workspace(dev, lib.dsp_dag_alignment_workspace_bytes(B, T, L, TR))
                rc = lib.dsp_dag_best_alignment_ws(_lib.ptr(m), _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha),
                                                   _lib.ptr(trace), _lib.ptr(path), B, T, L, TR, _lib.ptr(ws), ws.numel(),
                                                   _lib.current_stream_handle())
                _lib.check(rc, "dsp_dag_best_alignment_ws")
        ctx.mark_non_differentiable(path)
        return path

    @staticmethod
    def backward(ctx, grad_output):
        assert False, "no backward function for best alignment"


def dag_best_alignment(match_all, links, output_length, target_length):
    if _any_double(match_all, links):
        _require_gpu("dag_best_alignment", match_all, links, output_length, target_length)
        _check_dp_args("dag_best_alignment", match_all, links, output_length, target_length)
        return dag_double.dag_best_alignment(match_all.double(), links.double(), output_length, target_length)
    return DagBestAlignmentFunc.apply(match_all, links, output_length, target_length)


def _elem_strides(t: Tensor):
    return [int(s) for s in t.stride()]


# The reference keeps the softmax IN the logits buffer between forward and backward and forbids any other use of that
# buffer (dag_loss.py:249-251).  LAZY_SOFTMAX = True keeps two floats per row instead and leaves the logits alone until the
# backward overwrites them with the gradient: same match, same gradient, no B*L*V store in the forward.  Off by default —
# the default reproduces the reference's side effect bit for bit; daspeech_amd.criterions turns it on.
LAZY_SOFTMAX = False


def set_lazy_softmax(flag: bool) -> bool:
    """Select how dag_logsoftmax_gather_inplace keeps its backward state; returns the previous setting."""
    global LAZY_SOFTMAX
    prev, LAZY_SOFTMAX = LAZY_SOFTMAX, bool(flag)
    return prev


def _lsg_check(word_ins_out: Tensor, select_idx: Tensor):
    dev = _require_gpu("dag_logsoftmax_gather_inplace", word_ins_out, select_idx)
    if word_ins_out.dim() != 3 or select_idx.dim() != 3:
        raise RuntimeError("dag_logsoftmax_gather_inplace: word_ins_out and select_idx must be 3-D")
    if not word_ins_out.is_contiguous():
        raise RuntimeError("dag_logsoftmax_gather_inplace: word_ins_out must be contiguous (it is modified in place)")
    code = _lib.DTYPE_CODES.get(str(word_ins_out.dtype))
    if code is None:
        raise RuntimeError(f"dag_logsoftmax_gather_inplace: unsupported dtype {word_ins_out.dtype}")
    if select_idx.dtype != torch.long:
        raise RuntimeError("dag_logsoftmax_gather_inplace: select_idx must be int64")
    B, L, V = word_ins_out.shape
    if select_idx.shape[0] != B or select_idx.shape[1] != L:
        raise RuntimeError("dag_logsoftmax_gather_inplace: select_idx must be [batch, prelen, slen]")
    return dev, code, B, L, V, select_idx.shape[2]


def _lsg_forward(word_ins_out: Tensor, select_idx: Tensor, write_softmax: bool) -> Tensor:
    """K1 launch: returns the [B,S,L] match buffer (contiguous when L is a multiple of 4, else a view of rows pitched to the next multiple);
    word_ins_out becomes softmax if write_softmax."""
    dev, code, B, L, V, S = _lsg_check(word_ins_out, select_idx)
    lib = _lib.load()
    with torch.cuda.device(dev):
        buf = _pitched_empty(B, S, L, dev)                                 # "match_all" layout, rows on 16-byte boundaries (pitch = L rounded up to 4)
        ld = _round4(L)
        isb, isj, iss = _elem_strides(select_idx)
        rc = lib.dsp_logsoftmax_gather(_lib.ptr(word_ins_out), code, _lib.ptr(select_idx), isb, isj, iss,
                                       _lib.ptr(buf), S * ld, 1, ld, B, L, V, S, 1 if write_softmax else 0,
                                       _lib.current_stream_handle())
        _lib.check(rc, "dsp_logsoftmax_gather")
    return buf


def _lsg_forward_lazy(word_ins_out: Tensor, select_idx: Tensor):
    """K1 launch that leaves the logits untouched: returns (match buffer [B,S,L], row statistics [B,L,2] = (max, 1/sum-exp))."""
    dev, code, B, L, V, S = _lsg_check(word_ins_out, select_idx)
    lib = _lib.load()
    with torch.cuda.device(dev):
        buf = _pitched_empty(B, S, L, dev)
        ld = _round4(L)
        stats = torch.empty((B, L, 2), dtype=torch.float32, device=dev)
        isb, isj, iss = _elem_strides(select_idx)
        rc = lib.dsp_logsoftmax_gather_stats(_lib.ptr(word_ins_out), code, _lib.ptr(select_idx), isb, isj, iss,
                                             _lib.ptr(buf), S * ld, 1, ld, _lib.ptr(stats), B, L, V, S, _lib.current_stream_handle())
        _lib.check(rc, "dsp_logsoftmax_gather_stats")
    return buf, stats


def _lsg_backward(softmax_inout: Tensor, select_idx: Tensor, grad_match_bls: Tensor, stats: Tensor = None) -> Tensor:
    """K1 backward launch: softmax_inout [B,L,V] (softmax, or the logits when `stats` is given) -> d/d logits in place;
    grad_match_bls is [B,L,S]-shaped (any strides)."""
    B, L, V = softmax_inout.shape
    S = select_idx.shape[2]
    g = grad_match_bls.detach()
    if g.dtype != torch.float32:
        g = g.float()
    code = _lib.DTYPE_CODES[str(softmax_inout.dtype)]
    lib = _lib.load()
    with torch.cuda.device(softmax_inout.device):
        isb, isj, iss = _elem_strides(select_idx)
        gsb, gsj, gss = _elem_strides(g)
        if stats is None:
            rc = lib.dsp_logsoftmax_gather_bwd(_lib.ptr(softmax_inout), code, _lib.ptr(select_idx), isb, isj, iss,
                                               _lib.ptr(g), gsb, gsj, gss, B, L, V, S, _lib.current_stream_handle())
            _lib.check(rc, "dsp_logsoftmax_gather_bwd")
        else:
            rc = lib.dsp_logsoftmax_gather_bwd_lazy(_lib.ptr(softmax_inout), code, _lib.ptr(select_idx), isb, isj, iss,
                                                    _lib.ptr(g), gsb, gsj, gss, _lib.ptr(stats), B, L, V, S,
                                                    _lib.current_stream_handle())
            _lib.check(rc, "dsp_logsoftmax_gather_bwd_lazy")
    return softmax_inout


class DagLogsoftmaxGatherFunc(Function):

    @staticmethod
    def forward(ctx, word_ins_out, select_idx):
        need = ctx.needs_input_grad[0]
        ctx.lazy = bool(need and LAZY_SOFTMAX)
        if ctx.lazy:
            buf, stats = _lsg_forward_lazy(word_ins_out, select_idx)
        else:
            buf, stats = _lsg_forward(word_ins_out, select_idx, need), None
        selected = buf.transpose(1, 2)                                         # [B, L, S] view
        if need:
            # (lazy: written by the backward.)  Without a gradient nothing is ever written: the buffer is NOT marked dirty, so a caller may
            # run the gather on a `.detach()` of a tensor another graph still needs — the argmax strategy does (the reference, whose
            # operator marks dirty unconditionally, dag_loss.py:272, pays a B*L*V clone for it, s2s_dag_fastspeech2_loss.py:215)
            ctx.mark_dirty(word_ins_out)
        ctx.set_materialize_grads(False)
        if need:
            if ctx.lazy:
                ctx.save_for_backward(word_ins_out, select_idx, stats)
            else:
                ctx.save_for_backward(word_ins_out, select_idx)
            ctx.has_backward = False
        return word_ins_out, selected

    @staticmethod
    def backward(ctx, grad_word_ins_out, grad_output):
        if not ctx.needs_input_grad[0]:
            return None, None
        assert grad_word_ins_out is None, "Cannot reuse word_ins_out after logsoftmax_gather"
        if grad_output is None:
            return None, None
        assert not ctx.has_backward, "Cannot backward twice in logsoftmax_gather"
        ctx.has_backward = True
        if ctx.lazy:
            grad_input, select_idx, stats = ctx.saved_tensors     # holds the logits, becomes the gradient in place
            return _lsg_backward(grad_input, select_idx, grad_output, stats).detach(), None
        grad_input, select_idx = ctx.saved_tensors        # holds softmax, becomes the gradient in place
        return _lsg_backward(grad_input, select_idx, grad_output).detach(), None


dag_logsoftmax_gather_inplace = DagLogsoftmaxGatherFunc.apply


# ----------------------------------------------------------------------------------------------------------------
# torch variants (device-agnostic).  Behavioural twins of dag_loss.py:303-425: dense links[b,i,j] = i -> j.
# ----------------------------------------------------------------------------------------------------------------

def logsumexp_keepdim(x: Tensor, dim: int) -> Tensor:
    """logsumexp that returns -inf (not nan) where every entry along `dim` is -inf (dag_loss.py:303-311)."""
    top = x.max(dim=dim, keepdim=True)[0]
    dead = top == float("-inf")
    shift = top.detach().masked_fill(dead, 0.0)
    total = (x - shift).exp().sum(dim=dim, keepdim=True)
    return total.masked_fill(dead, 1.0).log() + shift.masked_fill(dead, float("-inf"))


def _dp_dense(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor, use_max: bool) -> Tensor:
    if links.dim() != 3 or links.shape[1] != links.shape[2]:
        raise AssertionError("links should be batch_size * prelen * prelen")
    B, T, L = match_all.shape
    emit = match_all.transpose(1, 2)                       # [B, L, T]
    col = torch.full((B, L, 1), float("-inf"), dtype=match_all.dtype, device=match_all.device)
    col[:, 0, 0] = emit[:, 0, 0]
    cols = [col]
    for t in range(1, T):
        scores = cols[-1] + links                          # [B, L(from), L(to)]
        if use_max:
            nxt = scores.max(dim=1)[0].unsqueeze(-1)
        else:
            nxt = logsumexp_keepdim(scores, 1).transpose(1, 2)
        cols.append(nxt + emit[:, :, t:t + 1])
    table = torch.cat(cols, -1)                            # [B, L, T]
    return table[torch.arange(B, device=table.device), output_length - 1, target_length - 1]


def torch_dag_loss(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor) -> Tensor:
    """Marginal log-likelihood over all paths; dense links (dag_loss.py:325-366)."""
    return _dp_dense(match_all, links, output_length, target_length, use_max=False)


def _torch_max_loss(match_all, links, output_length, target_length):
    return _dp_dense(match_all, links, output_length, target_length, use_max=True)


def torch_dag_best_alignment(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor) -> Tensor:
    """Viterbi path through autograd of the max-DP (dag_loss.py:388-419).  Like the reference this turns on
    requires_grad on the caller's match_all."""
    with torch.enable_grad():
        match_all.requires_grad_()
        best = _torch_max_loss(match_all, links, output_length, target_length)
        (hits,) = torch.autograd.grad(best.sum(), [match_all])          # 1 on the chosen (t, j) cells
    val, path = hits.max(dim=1)
    return path.masked_fill(val < 0.5, -1)


def torch_dag_logsoftmax_gather_inplace(word_ins_out: Tensor, select_idx: Tensor) -> Tuple[Tensor, Tensor]:
    """log_softmax + gather without the in-place side effect (dag_loss.py:421-425)."""
    logp = torch.log_softmax(word_ins_out, -1, dtype=torch.float32)
    return word_ins_out, logp.gather(dim=-1, index=select_idx)
