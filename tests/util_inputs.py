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


def seeded_model_state(shapes, seed):
    """`seeded_weights` for a WHOLE S2ST model (reference parameter names), post-processed by name so that random weights decode a
    non-degenerate graph and speak for a few frames per token — applied identically by the golden generator (to the reference model)
    and by the tests (to the product model):
      * BatchNorm running_var -> 1 + 0.1 |n| (a variance), num_batches_tracked untouched;
      * the decoder's residual branches (self_attn / encoder_attn out_proj, fc2) x 0.1 and its learned positions x 40: vertex features
        then follow the vertex POSITION, not a common mode, so neighbouring vertices emit different tokens;
      * the four special-token embeddings x 0.05: with tied input / output embeddings the all-<unk> graph skeleton would otherwise
        score <unk> highest everywhere;
      * link positions x 20; duration predictor: proj.weight x 0.05, proj.bias = ln 4.5 (3-4 mel frames per token)."""
    fl = {k: v for k, v in shapes.items() if not k.endswith(("num_batches_tracked", "_float_tensor", ".version"))}
    w = seeded_weights(fl, seed, gain=1.0)
    for k in list(w):
        if k.endswith("running_var"):
            w[k] = (1.0 + 0.1 * np.abs(w[k] / 0.05)).astype(np.float32)
        if k.startswith("decoder.layers.") and k.endswith(("self_attn.out_proj.weight", "encoder_attn.out_proj.weight", "fc2.weight")):
            w[k] = w[k] * np.float32(0.1)
    if "decoder.embed_positions.weight" in w:
        w["decoder.embed_positions.weight"] = w["decoder.embed_positions.weight"] * np.float32(40.0)
    if "decoder.embed_tokens.weight" in w:
        w["decoder.embed_tokens.weight"][:4] *= np.float32(0.05)
    if "decoder.link_positional.weight" in w:
        w["decoder.link_positional.weight"] = w["decoder.link_positional.weight"] * np.float32(20.0)
    dp = "tts.var_adaptor.duration_predictor.proj."
    if dp + "weight" in w:
        w[dp + "weight"] = w[dp + "weight"] * np.float32(0.05)
        w[dp + "bias"] = np.full_like(w[dp + "bias"], np.log(4.5))
    if "decoder.output_projection.weight" in fl and "decoder.embed_tokens.weight" in w:       # tied (--share-decoder-input-output-embed)
        w["decoder.output_projection.weight"] = w["decoder.embed_tokens.weight"]
    return w


def seeded_fbank(seed, frames):
    """[B, max(frames), 80] float32 synthetic filter-bank batch, zero padded, from a seed (golden generator and tests)."""
    rng = np.random.default_rng(seed)
    B, F = len(frames), int(max(frames))
    x = rng.standard_normal((B, F, 80)).astype(np.float32)
    for b, n in enumerate(frames):
        x[b, int(n):] = 0
    return x
