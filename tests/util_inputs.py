"""Seeded synthetic inputs shaped like SURVEY.md §8(d): logits=randn, links = masked log_softmax(randn) with all -inf
rows for vertices without successors, ragged lengths L_b = L - randint(0,5), T_b = T - randint(0,5)."""
import numpy as np


def make_dag_inputs(seed, B, T, L, TR, V=None, ragged=True, match_scale=2.0):
    rng = np.random.default_rng(seed)
    out_len = np.full(B, L, np.int64)
    tgt_len = np.full(B, T, np.int64)
    if ragged:
        out_len -= rng.integers(0, min(5, L - 1), B)
        tgt_len -= rng.integers(0, min(5, T - 1), B)
    tgt_len = np.minimum(tgt_len, out_len)
    raw = rng.standard_normal((B, L, TR)).astype(np.float32)
    i = np.arange(L)[None, :, None]
    d = np.arange(TR)[None, None, :]
    valid = (i + d + 1) < out_len[:, None, None]
    raw = np.where(valid, raw, -np.inf)
    mx = np.max(np.where(valid, raw, -1e30), axis=-1, keepdims=True)
    e = np.where(valid, np.exp(raw - mx), 0.0)
    s = e.sum(-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        links = np.where(valid, raw - mx - np.log(np.where(s > 0, s, 1.0)), -np.inf).astype(np.float32)
    match = (rng.standard_normal((B, T, L)) * match_scale - 5.0).astype(np.float32)
    return match, links, out_len, tgt_len


def seeded_weights(shapes, seed, gain=1.0):
    """Deterministic weights keyed by PARAMETER NAME (not by order): the same dict is produced wherever it is called with the same
    names / shapes / seed — the golden generators (authoring container, reference modules) and the GPU tests (product modules with
    the reference's parameter names) rebuild identical full-width weights from a seed instead of shipping them as fixtures.
    Rules: *alpha -> 1 + 0.1 n; LayerNorm weight -> 1 + 0.1 n, bias -> 0.05 n; matrices / conv kernels -> n * gain / sqrt(fan_in) with
    fan_in = prod(shape[1:]); other vectors -> 0.05 n."""
    import zlib
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        rng = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
        n = rng.standard_normal(shape)
        is_ln = ("layer_norm" in name) or (".ln1." in name) or (".ln2." in name)
        if name.endswith("alpha"):
            v = 1.0 + 0.1 * n
        elif is_ln and name.endswith("weight"):
            v = 1.0 + 0.1 * n
        elif len(shape) >= 2:
            v = n * (gain / np.sqrt(np.prod(shape[1:])))
        else:
            v = 0.05 * n
        out[name] = v.astype(np.float32)
    return out
