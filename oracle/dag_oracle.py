"""TEST INFRASTRUCTURE ONLY — numpy front-end of ``oracle/dag_oracle.c``.

Each wrapper takes / returns numpy arrays and calls the C restatement (built by ``oracle/Makefile`` into
``oracle/_build/libdag_oracle.so``).  ``dtype`` selects the f32 or the f64 instantiation.  Reference lines
are cited in the C source next to each function.
Parity status: PINNED — checked against the golden vectors in ``tests/golden/`` that were produced by
importing the reference's own torch implementations (``tests/golden/make_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdag_oracle.so")
_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("dag_oracle.c", "dag_oracle_impl.inc", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32"
    if dtype == np.float64:
        return "_f64"
    raise TypeError(dtype)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _idx_strides(idx):
    """(array kept alive, data pointer, element strides) for an int64 [B,L,S] index that may be a stride-0 view."""
    idx = np.asarray(idx)
    assert idx.dtype == np.int64 and idx.ndim == 3
    st = [s // 8 for s in idx.strides]
    return idx, ctypes.c_void_p(idx.ctypes.data), st


def logsoftmax_gather(x, idx, dtype=np.float32, want_softmax=False):
    x = _c(x, dtype)
    B, L, V = x.shape
    idx, ip, (sb, sj, ss) = _idx_strides(idx)
    S = idx.shape[2]
    out = np.empty((B, L, S), dtype)
    sm = np.empty_like(x) if want_softmax else None
    getattr(lib(), "orc_logsoftmax_gather" + _suffix(dtype))(
        _p(x), ip, ctypes.c_int64(sb), ctypes.c_int64(sj), ctypes.c_int64(ss), _p(out),
        _p(sm) if want_softmax else None, B, L, V, S)
    return (out, sm) if want_softmax else out


def logsoftmax_gather_bwd(softmax, idx, g, dtype=np.float32):
    sm = _c(softmax, dtype)
    g = _c(g, dtype)
    B, L, V = sm.shape
    idx, ip, (sb, sj, ss) = _idx_strides(idx)
    S = idx.shape[2]
    gx = np.empty_like(sm)
    getattr(lib(), "orc_logsoftmax_gather_bwd" + _suffix(dtype))(
        _p(sm), ip, ctypes.c_int64(sb), ctypes.c_int64(sj), ctypes.c_int64(ss), _p(g), _p(gx), B, L, V, S)
    return gx


def _dp_args(match, links, out_len, tgt_len, dtype):
    match = _c(match, dtype)
    links = _c(links, dtype)
    ol = _c(out_len, np.int64)
    tl = _c(tgt_len, np.int64)
    B, T, L = match.shape
    TR = links.shape[2]
    assert links.shape[:2] == (B, L)
    return match, links, ol, tl, B, T, L, TR


def dag_alpha(match, links, out_len, tgt_len, dtype=np.float32):
    match, links, ol, tl, B, T, L, TR = _dp_args(match, links, out_len, tgt_len, dtype)
    a = np.empty((B, T, L), dtype)
    getattr(lib(), "orc_dag_alpha" + _suffix(dtype))(_p(match), _p(links), _p(ol), _p(tl), _p(a), B, T, L, TR)
    return a


def dag_beta(match, links, out_len, tgt_len, dtype=np.float32):
    match, links, ol, tl, B, T, L, TR = _dp_args(match, links, out_len, tgt_len, dtype)
    b = np.empty((B, T, L), dtype)
    getattr(lib(), "orc_dag_beta" + _suffix(dtype))(_p(match), _p(links), _p(ol), _p(tl), _p(b), B, T, L, TR)
    return b


def dag_loss(match, links, out_len, tgt_len, dtype=np.float32, from_beta=True):
    """loss[b] as the reference returns it: beta[b,0,0] with grad, alpha[b,T_b-1,L_b-1] without (dag_loss.py:107-110)."""
    if from_beta:
        return dag_beta(match, links, out_len, tgt_len, dtype)[:, 0, 0].copy()
    a = dag_alpha(match, links, out_len, tgt_len, dtype)
    B = a.shape[0]
    return a[np.arange(B), np.asarray(tgt_len) - 1, np.asarray(out_len) - 1].copy()


def dag_grad(g_out, alpha, beta, match, links, out_len, tgt_len, dtype=np.float32):
    match, links, ol, tl, B, T, L, TR = _dp_args(match, links, out_len, tgt_len, dtype)
    alpha = _c(alpha, dtype)
    beta = _c(beta, dtype)
    g_out = _c(g_out, dtype)
    gm = np.empty((B, T, L), dtype)
    gl = np.empty((B, L, TR), dtype)
    getattr(lib(), "orc_dag_grad" + _suffix(dtype))(
        _p(g_out), _p(alpha), _p(beta), _p(match), _p(links), _p(ol), _p(tl), _p(gm), _p(gl), B, T, L, TR)
    return gm, gl


def dag_best_alignment(match, links, out_len, tgt_len, dtype=np.float32, want_internals=False):
    match, links, ol, tl, B, T, L, TR = _dp_args(match, links, out_len, tgt_len, dtype)
    a = np.empty((B, T, L), dtype)
    tr = np.empty((B, T, L), np.int32)
    path = np.empty((B, L), np.int64)
    getattr(lib(), "orc_dag_best_alignment" + _suffix(dtype))(
        _p(match), _p(links), _p(ol), _p(tl), _p(a), _p(tr), _p(path), B, T, L, TR)
    return (path, a, tr) if want_internals else path


# ---------------- decode / TTS glue ----------------

def argmax_logp(logits):
    x = _c(logits, np.float32)
    B, L, V = x.shape
    tok = np.empty((B, L), np.int32)
    sc = np.empty((B, L), np.float32)
    lib().orc_argmax_logp(_p(x), _p(tok), _p(sc), B, L, V)
    return tok, sc


def lookahead_next(links, sc, beta=1.0, greedy=False):
    links = _c(links, np.float32)
    B, L, TR = links.shape
    sc = _c(sc if sc is not None else np.zeros((B, L)), np.float32)
    nxt = np.empty((B, L), np.int32)
    lib().orc_lookahead_next(_p(links), _p(sc), ctypes.c_float(beta), 0 if greedy else 1, _p(nxt), B, L, TR)
    return nxt


def follow_path(nxt, tok, out_len, pad, nmax=None):
    nxt = _c(nxt, np.int32)
    tok = _c(tok, np.int32)
    ol = _c(out_len, np.int64)
    B, L = nxt.shape
    nmax = nmax or L
    toks = np.empty((B, nmax), np.int64)
    keep = np.empty((B, nmax), np.int32)
    nf = np.empty((B,), np.int32)
    lib().orc_follow_path(_p(nxt), _p(tok), _p(ol), int(pad), _p(toks), _p(keep), _p(nf), B, L, nmax)
    return toks, keep, nf


def length_regulate(x, dur):
    x = _c(x, np.float32)
    dur = _c(dur, np.int64)
    B, N, C = x.shape
    lens = dur.sum(1)
    maxlen = int(lens.max()) if B else 0
    out = np.empty((B, maxlen, C), np.float32)
    ol = np.empty((B,), np.int64)
    lib().orc_length_regulate(_p(x), _p(dur), _p(out), _p(ol), B, N, C, maxlen)
    return out, ol


def durations(log_dur, pad_mask, factor=1.0):
    ld = _c(log_dur, np.float32)
    pm = _c(pad_mask, np.uint8)
    out = np.empty(ld.shape, np.int64)
    lib().orc_durations(_p(ld), _p(pm), ctypes.c_float(factor), _p(out), ld.size)
    return out


def bucketize(v, bins):
    v = _c(v, np.float32)
    bins = _c(bins, np.float32)
    out = np.empty(v.shape, np.int64)
    lib().orc_bucketize(_p(v), _p(bins), bins.size, _p(out), v.size)
    return out


def posterior_expect(alpha, beta, feat):
    a = _c(alpha, np.float32)
    b = _c(beta, np.float32)
    f = _c(feat, np.float32)
    B, T, L = a.shape
    D = f.shape[2]
    score = np.empty((B, T, L), np.float32)
    ex = np.empty((B, T, D), np.float32)
    lib().orc_posterior_expect(_p(a), _p(b), _p(f), _p(score), _p(ex), B, T, L, D)
    return score, ex


def restore_valid_links(links):
    """compact [B,L,TR] -> dense [B,L,L] (-inf elsewhere); DASpeech/custom_ops/dag_loss.py:439-448."""
    links = np.asarray(links)
    B, L, TR = links.shape
    dense = np.full((B, L, L), -np.inf, links.dtype)
    for d in range(TR):
        n = L - d - 1
        if n <= 0:
            break
        i = np.arange(n)
        dense[:, i, i + d + 1] = links[:, :n, d]
    return dense
