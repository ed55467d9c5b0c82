#!/usr/bin/env python3
"""Golden vectors for the Python-side rows of the hot path, produced by RUNNING THE
REFERENCE'S OWN FUNCTIONS in the authoring container.

The DASpeech package cannot be imported here (its __init__ pulls fairseq -> omegaconf), so the functions are
lifted out of the reference's source files AT RUN TIME: the file is parsed with `ast`, the wanted `def` node is compiled on its
own and executed in a namespace that supplies the few names it uses (torch, F, logsumexp, a fake `self` carrying `args` /
`pad` / `tgt_dict`, the reference's importable torch DAG ops).  Nothing of the reference's text is stored in this repo — only
inputs and outputs (`*.npz`).

  graph_links.npz     S2TConformerDAGModel.extract_links / extract_valid_links / restore_valid_links
                      (DASpeech/models/s2t_conformer_dag.py:140-212), banded (TR < L-1) and full (TR = L-1) windows, ragged lengths
  graph_decode.npz    S2SConformerDAGFastSpeech2Model.forward_decoder (DASpeech/models/s2s_conformer_dag_fastspeech2.py:194-304):
                      lookahead, greedy, viterbi, jointviterbi on the same graphs
  glat.npz            the glat_function closure of NATDAGLoss.forward (DASpeech/criterions/nat_dag_loss.py:202-264), strategies
                      None and number-random, with the torch DAG ops of DASpeech/custom_ops/dag_loss.py; the random draws are
                      stored so that a consumer can replay them;  + parse_anneal_argument / get_anneal_value
                      (DASpeech/criterions/utilities.py:17-37) samples
"""
import ast
import importlib.util
import os
import random
import types
from typing import List  # noqa: F401  (used by the lifted _collate_frames)

import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util_inputs import seeded_weights  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def lift(path, chain, ns):
    """Compile the (possibly nested) def named by `chain` from `path` and return the function object, globals = ns."""
    tree = ast.parse(open(path).read())
    node = tree
    for name in chain:
        node = next(n for n in ast.iter_child_nodes(node) if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name == name)
    node.decorator_list = []
    code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
    exec(code, ns)
    return ns[chain[-1]]


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


S2T = f"{REF}/DASpeech/models/s2t_conformer_dag.py"
S2S = f"{REF}/DASpeech/models/s2s_conformer_dag_fastspeech2.py"
CRIT = f"{REF}/DASpeech/criterions/nat_dag_loss.py"
PAD = 1


def lengths_to_padding_mask(lens):            # fairseq/data/data_utils.py (3-line helper, restated)
    m = int(lens.max())
    return torch.arange(m).unsqueeze(0) >= lens.unsqueeze(1)


def make_self(max_transition_length, heads, dim):
    ns = {"torch": torch, "F": F, "logsumexp": torch.logsumexp}
    s = types.SimpleNamespace()
    s.args = types.SimpleNamespace(max_transition_length=max_transition_length, decoder_attention_heads=heads, decoder_embed_dim=dim)
    s.pad = PAD
    for name in ("extract_valid_links", "restore_valid_links", "extract_links"):
        setattr(s, name, types.MethodType(lift(S2T, ["S2TConformerDAGModel", name], ns), s))
    return s


def links_golden():
    store = {}
    rng = np.random.default_rng(21)
    for tag, (B, L, heads, dim, TR) in {"band": (3, 14, 8, 32, 5), "full": (2, 9, 8, 16, 99999), "wide": (2, 40, 8, 64, 33)}.items():
        s = make_self(TR, heads, dim)
        torch.manual_seed(3)
        lp = nn.Embedding(L + 2, dim, padding_idx=PAD)
        ql, kl, gl = nn.Linear(2 * dim, dim), nn.Linear(2 * dim, dim), nn.Linear(2 * dim, heads)
        with torch.no_grad():
            for m in (lp, ql, kl, gl):
                for p in m.parameters():
                    p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32) * (0.35 if p.dim() > 1 else 0.1)))
            lp.weight[PAD] = 0
        feats = torch.from_numpy(rng.standard_normal((B, L, dim)).astype(np.float32))
        lens = torch.tensor([L, L - 3, 2][:B]) if tag == "band" else torch.tensor([L, L - 2][:B])
        prev = torch.full((B, L), 3, dtype=torch.long)
        prev[torch.arange(L).unsqueeze(0) >= lens.unsqueeze(1)] = PAD

        # link_positional(prev_output_tokens) in the reference = learned positional embedding over make_positions (fairseq utils.py:256-266)
        class LinkPos(nn.Module):
            def forward(self, toks):
                keep = toks.ne(PAD).int()
                return lp((torch.cumsum(keep, 1) * keep).long() + PAD)
        with torch.no_grad():
            links = s.extract_links(feats, prev, LinkPos(), ql, kl, gl)
            dense = s.restore_valid_links(links)
        store.update({f"{tag}_feats": feats.numpy(), f"{tag}_prev": prev.numpy(), f"{tag}_pos_w": lp.weight.detach().numpy(),
                      f"{tag}_q_w": ql.weight.detach().numpy(), f"{tag}_q_b": ql.bias.detach().numpy(),
                      f"{tag}_k_w": kl.weight.detach().numpy(), f"{tag}_k_b": kl.bias.detach().numpy(),
                      f"{tag}_g_w": gl.weight.detach().numpy(), f"{tag}_g_b": gl.bias.detach().numpy(),
                      f"{tag}_max_transition_length": np.int64(TR), f"{tag}_heads": np.int64(heads),
                      f"{tag}_links": links.numpy(), f"{tag}_dense": dense.numpy()})
        print("links", tag, tuple(links.shape), "finite", int(torch.isfinite(links).sum()))
    np.savez_compressed(os.path.join(HERE, "graph_links.npz"), **store)
    # released head geometry (8 heads x 32 / 64 channels: what the fused HIP kernel serves); weights rebuilt from a seed by the tests
    store = {}
    for tag, (B, L, heads, dim, TR, seed) in {"h32": (2, 48, 8, 256, 20, 901), "h64": (2, 40, 8, 512, 99999, 902)}.items():
        s = make_self(TR, heads, dim)
        shapes = {"pos.weight": (L + 2, dim), "q.weight": (dim, 2 * dim), "q.bias": (dim,), "k.weight": (dim, 2 * dim), "k.bias": (dim,),
                  "g.weight": (heads, 2 * dim), "g.bias": (heads,)}
        w = {k: torch.from_numpy(v) for k, v in seeded_weights(shapes, seed, gain=2.0).items()}
        w["pos.weight"] = w["pos.weight"] * 4.0
        w["pos.weight"][PAD] = 0
        lp = nn.Embedding(L + 2, dim, padding_idx=PAD); ql, kl, gl = nn.Linear(2 * dim, dim), nn.Linear(2 * dim, dim), nn.Linear(2 * dim, heads)
        with torch.no_grad():
            lp.weight.copy_(w["pos.weight"]); ql.weight.copy_(w["q.weight"]); ql.bias.copy_(w["q.bias"]); kl.weight.copy_(w["k.weight"])
            kl.bias.copy_(w["k.bias"]); gl.weight.copy_(w["g.weight"]); gl.bias.copy_(w["g.bias"])
        feats = torch.from_numpy(rng.standard_normal((B, L, dim)).astype(np.float32))
        lens = torch.tensor([L, L - 7])
        prev = torch.full((B, L), 3, dtype=torch.long)
        prev[torch.arange(L).unsqueeze(0) >= lens.unsqueeze(1)] = PAD

        class LinkPos2(nn.Module):
            def forward(self, toks):
                keep = toks.ne(PAD).int()
                return lp((torch.cumsum(keep, 1) * keep).long() + PAD)
        with torch.no_grad():
            links = s.extract_links(feats, prev, LinkPos2(), ql, kl, gl)
        store.update({f"{tag}_feats": feats.numpy(), f"{tag}_prev": prev.numpy(), f"{tag}_seed": np.int64(seed), f"{tag}_heads": np.int64(heads),
                      f"{tag}_max_transition_length": np.int64(TR), f"{tag}_links": links.numpy()})
        fin = links[torch.isfinite(links)]
        print("links", tag, tuple(links.shape), "finite", int(torch.isfinite(links).sum()), "range", float(fin.min()), float(fin.max()))
    np.savez_compressed(os.path.join(HERE, "graph_links_released_heads.npz"), **store)


def make_links(rng, B, L, TR, lens):
    raw = rng.standard_normal((B, L, TR)).astype(np.float32)
    i = np.arange(L)[None, :, None]; d = np.arange(TR)[None, None, :]
    valid = (i + d + 1) < lens[:, None, None]
    raw = np.where(valid, raw, -np.inf)
    mx = np.max(np.where(valid, raw, -1e30), -1, keepdims=True)
    e = np.where(valid, np.exp(raw - mx), 0)
    ssum = e.sum(-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(valid, raw - mx - np.log(np.where(ssum > 0, ssum, 1)), -np.inf).astype(np.float32)


def decode_golden():
    ns = {"torch": torch, "random": random, "np": np, "lengths_to_padding_mask": lengths_to_padding_mask, "List": List}
    ns["_collate_frames"] = lift(f"{REF}/fairseq/fairseq/data/audio/speech_to_text_dataset.py", ["_collate_frames"], ns)
    fwd = lift(S2S, ["S2SConformerDAGFastSpeech2Model", "forward_decoder"], ns)
    restore = lift(S2T, ["S2TConformerDAGModel", "restore_valid_links"], {"torch": torch})
    # the Viterbi branch calls `.cuda(scores.get_device())` on a small host tensor (:273); on CPU tensors that is the identity
    torch.Tensor.cuda = lambda self, *a, **k: self
    DecOut = __import__("collections").namedtuple("DecOut", "output_tokens output_scores features features_padding_mask attn step max_step history")
    store = {}
    rng = np.random.default_rng(33)
    cases = {"a": (3, 24, 7, 12, 6, 1.0, 1.0), "b": (2, 40, 39, 9, 5, 0.7, 1.3), "q": (3, 32, 8, 6, 4, 1.0, 1.0)}
    for tag, (B, L, TR, V, D, dbeta, vbeta) in cases.items():
        lens = np.array([L, L - 5, L - 1][:B])
        logits = (rng.standard_normal((B, L, V)) * 2).astype(np.float32)
        links = make_links(rng, B, L, min(TR, L - 1), lens)
        if tag == "q":          # quantised scores: ties everywhere, pins the first-max rules
            logits = np.round(logits)
            links = np.where(np.isfinite(links), np.round(links * 2) / 2, links).astype(np.float32)
        logits[:, :, PAD] += 0.5                                    # make <pad> an occasional argmax (dropped by the decode)
        feats = rng.standard_normal((B, L, D)).astype(np.float32)
        prev = np.full((B, L), 3, np.int64)
        prev[np.arange(L)[None, :] >= lens[:, None]] = PAD
        store.update({f"{tag}_logits": logits, f"{tag}_links": links, f"{tag}_feats": feats, f"{tag}_prev": prev,
                      f"{tag}_decode_beta": np.float32(dbeta), f"{tag}_viterbibeta": np.float32(vbeta)})
        for strat in ("lookahead", "greedy", "viterbi", "jointviterbi"):
            s = types.SimpleNamespace()
            s.args = types.SimpleNamespace(max_transition_length=TR, decode_strategy=strat, decode_beta=dbeta, decode_viterbibeta=vbeta,
                                           src_upsample_scale=0.5)
            s.tgt_dict = types.SimpleNamespace(pad_index=PAD)
            s.restore_valid_links = types.MethodType(restore, s)
            s.extract_features = lambda toks, enc, seed, require_links=True: (torch.from_numpy(logits.copy()), torch.from_numpy(links.copy()),
                                                                              torch.from_numpy(feats.copy()))
            d0 = DecOut(torch.from_numpy(prev), None, None, None, None, 0, 0, None)
            with torch.no_grad():
                out = fwd(s, d0, None)
            store[f"{tag}_{strat}_tokens"] = out.output_tokens.numpy()
            store[f"{tag}_{strat}_features"] = out.features.numpy()
            store[f"{tag}_{strat}_mask"] = out.features_padding_mask.numpy()
            print("decode", tag, strat, out.output_tokens.tolist())
    np.savez_compressed(os.path.join(HERE, "graph_decode.npz"), **store)


def glat_golden():
    ref_ops = load_by_path("ref_dag", f"{REF}/DASpeech/custom_ops/dag_loss.py")
    util = load_by_path("ref_util", f"{REF}/DASpeech/criterions/utilities.py")
    restore = lift(S2T, ["S2TConformerDAGModel", "restore_valid_links"], {"torch": torch})
    store = {}
    rng = np.random.default_rng(44)
    for tag, (B, L, T, TR, V, strategy, p) in {"none": (3, 18, 6, 5, 11, None, 0.5), "nr": (4, 22, 7, 21, 9, "number-random", 0.5),
                                               "nr0": (2, 12, 5, 11, 7, "number-random", 0.01),
                                               "cmlm": (4, 20, 8, 19, 9, "cmlm", 0.5)}.items():
        lens = np.array([L, L - 2, L - 4, L - 1][:B]); tlens = np.array([T, T - 1, T - 2, T][:B])
        logits = (rng.standard_normal((B, L, V)) * 1.5).astype(np.float32)
        links = make_links(rng, B, L, min(TR, L - 1), lens)
        prev = np.full((B, L), 3, np.int64); prev[np.arange(L)[None, :] >= lens[:, None]] = PAD
        tgt = rng.integers(4, V, (B, T)); tgt[np.arange(T)[None, :] >= tlens[:, None]] = PAD
        # make some vertices predict their aligned token so that same_num > 0 (argmax == target on a few positions)
        for b in range(B):
            for j in range(0, int(lens[b]), 3):
                logits[b, j, tgt[b, min(j // 3, int(tlens[b]) - 1)]] += 6.0
        cfg = types.SimpleNamespace(torch_dag_logsoftmax_gather=True, torch_dag_best_alignment=True)
        crit = types.SimpleNamespace(cfg=cfg, glance_strategy=strategy)
        ns = {"torch": torch, "self": crit, "torch_dag_logsoftmax_gather_inplace": ref_ops.torch_dag_logsoftmax_gather_inplace,
              "torch_dag_best_alignment": ref_ops.torch_dag_best_alignment, "dag_logsoftmax_gather_inplace": None, "dag_best_alignment": None}
        glat_fn = lift(CRIT, ["NATDAGLoss", "forward", "glat_function"], ns)
        model = types.SimpleNamespace(pad=PAD, args=types.SimpleNamespace(max_transition_length=TR))
        model.restore_valid_links = types.MethodType(restore, model)
        seed = 1000 + len(tag)
        torch.manual_seed(seed)
        with torch.enable_grad():
            gp, gt, info = glat_fn(model, torch.from_numpy(logits.copy()), torch.from_numpy(tgt), torch.from_numpy(prev), {"context_p": p},
                                   links=torch.from_numpy(links.copy()))
        # replay of the draws, in the reference's order: randn(oracle.shape) [number-random / cmlm], rand_like(target_length) [cmlm
        # only: the glance count], then rand(prev.shape)
        torch.manual_seed(seed)
        noise = torch.randn(B, L) if strategy is not None else torch.zeros(B, L)
        unif_n = torch.rand(B) if strategy == "cmlm" else torch.zeros(B)
        unif = torch.rand(B, L)
        store.update({f"{tag}_logits": logits, f"{tag}_links": links, f"{tag}_prev": prev, f"{tag}_tgt": tgt, f"{tag}_p": np.float32(p),
                      f"{tag}_strategy": np.array(str(strategy)), f"{tag}_noise": noise.numpy(), f"{tag}_unif": unif.numpy(), f"{tag}_unif_n": unif_n.numpy(),
                      f"{tag}_glat_prev": gp.numpy(), f"{tag}_matchmask": info["matchmask"].numpy(), f"{tag}_keep_word_mask": info["keep_word_mask"].numpy(),
                      f"{tag}_glat_accu": np.float32(info["glat_accu"]), f"{tag}_glat_keep": np.float32(info["glat_keep"])})
        print("glat", tag, "kept", info["keep_word_mask"].sum(1).tolist(), "accu", float(info["glat_accu"]))
    # annealing schedule samples
    arg_strs = ["0.5:0.1@200k", "0.3", "0.5@10:0.25@1k:0.1@4000"]
    ups = [0, 5, 10, 999, 1000, 2500, 100000, 200000, 300000]
    store["anneal_args"] = np.array(arg_strs)
    store["anneal_updates"] = np.array(ups)
    store["anneal_values"] = np.array([[util.get_anneal_value(util.parse_anneal_argument(a), u) for u in ups] for a in arg_strs], np.float64)
    np.savez_compressed(os.path.join(HERE, "glat.npz"), **store)


if __name__ == "__main__":
    links_golden()
    decode_golden()
    glat_golden()
