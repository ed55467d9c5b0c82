"""float64 inputs to the DAG operators.

The reference instantiates its kernels for float and double (AT_DISPATCH_FLOATING_TYPES_AND_HALF: dag_loss.cu:160,294,415,499,
dag_best_alignment.cu:143).  The fast HIP kernels of this package compute in fp32 (exp-space strips, fp32 matrix cores) — narrowing a
double tensor silently would hand back fp32 accuracy under a float64 dtype.  Double inputs are routed HERE:

  * GPU tensors (r06): the double-precision HIP kernels of csrc/dag_dp_f64.hip (dsp_dag_loss_fwd_f64 / _bwd_f64 /
    dsp_dag_best_alignment_f64) behind autograd Functions shaped like the fp32 ones — alpha / beta tables, no autograd through T steps;
  * the torch band DP below (alpha_table / beta_table / ...): the same recurrences on the COMPACT band `links[B, L, TR]` written with
    torch ops, device-agnostic and differentiable by autograd — what r05 ran on the GPU too (T tensors of [B, L, TR] doubles kept alive
    for autograd: out of memory on dense windows); kept as the CPU-tensor form the tests pin to the fp64 oracle and cross-check the
    kernels against.

Semantics follow the kernels (dag_loss.cu:40-140 alpha, :178-274 beta, dag_best_alignment.cu:39-253): cells outside
{t < T_b, t <= j < L_b} are -inf, an all -inf predecessor set stays -inf (no emission added), Viterbi ties take the smallest predecessor
index, vertices off the path are -1."""
import torch
from torch import Tensor

NEG = float("-inf")


def _band_index(L: int, TR: int, device):
    """pred[j, d] = j - d - 1 (clamped) and its validity: the predecessor vertex of edge (j-d-1) -> j, as links[b, j-d-1, d]."""
    j = torch.arange(L, device=device).view(L, 1)
    d = torch.arange(TR, device=device).view(1, TR)
    pred = j - d - 1
    return pred.clamp(min=0), pred >= 0


def _links_by_target(links: Tensor):
    """lt[b, j, d] = links[b, j-d-1, d] (-inf where j-d-1 < 0): the transition scores grouped by TARGET vertex."""
    B, L, TR = links.shape
    pred, ok = _band_index(L, TR, links.device)
    d = torch.arange(TR, device=links.device).view(1, TR).expand(L, TR)
    lt = links[:, pred, d]                                              # [B, L, TR]
    return lt.masked_fill(~ok.unsqueeze(0), NEG), pred, ok


def _lse(x: Tensor, dim: int) -> Tensor:
    top = x.max(dim=dim, keepdim=True)[0]
    dead = top == NEG
    s = (x - top.masked_fill(dead, 0.0)).exp().sum(dim=dim, keepdim=True)
    return (s.masked_fill(dead, 1.0).log() + top).squeeze(dim)        # -inf where every entry is -inf (top carries it)


def alpha_table(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor, use_max: bool = False) -> Tensor:
    """alpha[B, T, L] (log-sum-exp DP, or max-DP with use_max)."""
    B, T, L = match_all.shape
    TR = links.shape[2]
    lt, pred, _ = _links_by_target(links)
    jj = torch.arange(L, device=match_all.device).view(1, L)
    live_col = jj < output_length.view(B, 1)
    row = torch.full((B, L), NEG, dtype=match_all.dtype, device=match_all.device)
    row = torch.where(jj == 0, match_all[:, 0, :], row)
    rows = [row]
    for t in range(1, T):
        scores = rows[-1][:, pred] + lt                                  # [B, L, TR]
        acc = scores.max(dim=2)[0] if use_max else _lse(scores, 2)
        nxt = torch.where(acc == NEG, acc, acc + match_all[:, t, :])
        keep = live_col & (jj >= t) & (t < target_length.view(B, 1))
        rows.append(torch.where(keep, nxt, torch.full_like(nxt, NEG)))
    return torch.stack(rows, 1)


def beta_table(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor) -> Tensor:
    """beta[B, T, L]: beta[b, T_b-1, L_b-1] = match there; beta[t, j] = match[t, j] + LSE_d(links[j, d] + beta[t+1, j+d+1])."""
    B, T, L = match_all.shape
    TR = links.shape[2]
    dev = match_all.device
    jj = torch.arange(L, device=dev).view(1, L)
    succ = jj.view(L, 1) + torch.arange(TR, device=dev).view(1, TR) + 1                       # [L, TR]
    ok = succ.unsqueeze(0) < output_length.view(B, 1, 1)                                     # successor inside this sample's graph
    succ_c = succ.clamp(max=L - 1)
    lk = links.masked_fill(~ok, NEG)
    last_t = (target_length - 1).view(B, 1)
    last_j = (output_length - 1).view(B, 1)
    neg_row = torch.full((B, L), NEG, dtype=match_all.dtype, device=dev)
    rows = [None] * T
    nxt = neg_row
    for t in range(T - 1, -1, -1):
        acc = _lse(nxt[:, succ_c] + lk, 2)
        cur = torch.where(acc == NEG, acc, acc + match_all[:, t, :])
        keep = (jj < output_length.view(B, 1)) & (jj >= t) & (t < last_t)
        cur = torch.where(keep, cur, neg_row)
        cur = torch.where((t == last_t) & (jj == last_j), match_all[:, t, :], cur)
        rows[t] = cur
        nxt = cur
    return torch.stack(rows, 1)


def _pick_loss(a: Tensor, output_length: Tensor, target_length: Tensor) -> Tensor:
    """alpha[b, T_b-1, L_b-1]; an unreachable sample (-inf) is cut out of the autograd graph — zero gradients, as the kernels give
    (the reference asserts on the device instead, dag_loss.cu:68-69)."""
    B = a.shape[0]
    loss = a[torch.arange(B, device=a.device), target_length - 1, output_length - 1]
    return torch.where(torch.isfinite(loss), loss, loss.detach())


# ---- GPU: native double-precision kernels ------------------------------------------------------------------------------------------

def _native_forward(match_all, links, output_length, target_length, need_beta):
    from .. import _lib
    lib = _lib.load()
    m = match_all.detach().contiguous(); k = links.detach().contiguous()
    ol = output_length.contiguous(); tl = target_length.contiguous()
    B, T, L = m.shape
    TR = k.shape[2]
    with torch.cuda.device(m.device):
        alpha = torch.empty_like(m)
        beta = torch.empty_like(m) if need_beta else None
        loss = torch.empty((B,), dtype=torch.float64, device=m.device)
        _lib.check(lib.dsp_dag_loss_fwd_f64(_lib.ptr(m), _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _lib.ptr(beta), _lib.ptr(loss),
                                            B, T, L, TR, _lib.current_stream_handle()), "dsp_dag_loss_fwd_f64")
    return m, k, ol, tl, alpha, beta, loss


class _DagLossF64(torch.autograd.Function):
    """dag_loss / dag_loss_with_alpha_beta for float64 CUDA tensors (the contract of custom_ops.dag_loss.DagLossWithAlphaBetaFunc)."""

    @staticmethod
    def forward(ctx, match_all, links, output_length, target_length):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        m, k, ol, tl, alpha, beta, loss = _native_forward(match_all, links, output_length, target_length, need)
        ctx.save_for_backward(alpha, beta if need else alpha, m, k, ol, tl)
        if beta is None:
            beta = torch.zeros_like(alpha)                   # the reference's un-launched beta table (dag_loss.cu:339-340)
        ctx.mark_non_differentiable(alpha, beta)
        return loss, alpha, beta

    @staticmethod
    def backward(ctx, grad_output, _ga, _gb):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None
        from .. import _lib
        lib = _lib.load()
        alpha, beta, m, k, ol, tl = ctx.saved_tensors
        B, T, L = m.shape
        TR = k.shape[2]
        with torch.cuda.device(m.device):
            go = grad_output.detach().to(torch.float64).contiguous()
            gm = torch.empty_like(m) if ctx.needs_input_grad[0] else None
            gl = torch.empty_like(k) if ctx.needs_input_grad[1] else None
            _lib.check(lib.dsp_dag_loss_bwd_f64(_lib.ptr(go), _lib.ptr(alpha), _lib.ptr(beta), _lib.ptr(m), _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl),
                                                _lib.ptr(gm), _lib.ptr(gl), B, T, L, TR, _lib.current_stream_handle()), "dsp_dag_loss_bwd_f64")
        return gm, gl, None, None


def _native(match_all: Tensor, links: Tensor) -> bool:
    return match_all.is_cuda and links.is_cuda and match_all.shape[2] <= 10240


def dag_loss(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor) -> Tensor:
    if _native(match_all, links):
        return _DagLossF64.apply(match_all, links, output_length, target_length)[0]
    a = alpha_table(match_all, links, output_length, target_length)
    return _pick_loss(a, output_length, target_length)


def dag_loss_with_alpha_beta(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor):
    if _native(match_all, links):
        loss, a, b = _DagLossF64.apply(match_all, links, output_length, target_length)
        return loss, (a, b)
    a = alpha_table(match_all, links, output_length, target_length)
    loss = _pick_loss(a, output_length, target_length)
    need = match_all.requires_grad or links.requires_grad
    with torch.no_grad():
        # without a gradient the reference launches no beta kernel and returns the zero-initialised table (dag_loss.cu:339-340)
        b = beta_table(match_all, links, output_length, target_length) if need else torch.zeros_like(a)
    return loss, (a.detach(), b)


@torch.no_grad()
def dag_best_alignment(match_all: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor) -> Tensor:
    """path[B, L]: the target position each vertex of the best path emits, -1 off the path (dag_best_alignment.cu:182-253)."""
    B, T, L = match_all.shape
    TR = links.shape[2]
    if _native(match_all, links):
        from .. import _lib
        lib = _lib.load()
        m = match_all.detach().contiguous(); k = links.detach().contiguous()
        ol = output_length.contiguous(); tl = target_length.contiguous()
        with torch.cuda.device(m.device):
            amax = torch.empty_like(m)
            trace = torch.empty((B, T, L), dtype=torch.int32, device=m.device)
            path = torch.empty((B, L), dtype=torch.long, device=m.device)
            _lib.check(lib.dsp_dag_best_alignment_f64(_lib.ptr(m), _lib.ptr(k), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(amax), _lib.ptr(trace),
                                                      _lib.ptr(path), B, T, L, TR, _lib.current_stream_handle()), "dsp_dag_best_alignment_f64")
        return path
    a = alpha_table(match_all, links, output_length, target_length, use_max=True)
    lt, pred, ok = _links_by_target(links)
    path = torch.full((B, L), -1, dtype=torch.long, device=match_all.device)
    bi = torch.arange(B, device=match_all.device)
    pos = (output_length - 1).clone()
    alive = torch.ones(B, dtype=torch.bool, device=match_all.device)
    for step in range(T):
        t = target_length - 1 - step                                      # per-sample row
        act = alive & (t >= 0)
        tc = t.clamp(min=0)
        cur = path[bi, pos]
        path[bi, pos] = torch.where(act, tc, cur)
        # predecessor of (t, pos): arg-max over d of alpha[t-1, pos-d-1] + links[pos-d-1, d]; smallest predecessor index on ties
        tp = (t - 1).clamp(min=0)
        cand = a[bi.view(B, 1), tp.view(B, 1), pred[pos]] + lt[bi, pos]   # [B, TR], d ascending = predecessor DEscending
        best = cand.max(dim=1, keepdim=True)[0]
        is_best = (cand == best) & (best > NEG)
        d_idx = torch.arange(TR, device=match_all.device).view(1, TR)
        d_pick = torch.where(is_best, d_idx, torch.full_like(d_idx, -1)).max(dim=1)[0]      # largest d = smallest predecessor
        has = d_pick >= 0
        new_pos = pos - d_pick - 1
        go = act & (t >= 1)
        alive = alive & ~(go & ~has)                                      # unreachable: stop (the rest of the path stays -1)
        pos = torch.where(go & has, new_pos, pos)
    return path
