"""hipGraph capture of the shape-static front of eval-mode inference: encoder -> graph skeleton -> NAT decoder -> output projection ->
fused links (`model.forward_encoder` + `initialize_output_tokens_by_src` + `decode_graph`), ~300 of the acoustic stage's ~450 launches.

Everything up to the transition graph has shapes fixed by (batch, padded source frames); the graph decode behind it returns
data-dependent lengths (one host sync, `decode_ops.graph_decode`) and stays eager, as does the TTS half.  One graph per input shape, LRU of
`max_graphs`; a replay copies the batch into the graph's static inputs and returns the graph's static outputs — valid until the next
replay of the SAME shape on that stream (the generator consumes them before it issues the next batch).  The HIP operators take their
scratch from torch's caching allocator and launch on torch's current stream, so they capture like torch's own kernels.

The reference has no counterpart (fairseq issues every kernel eagerly); this is host-side plumbing of the MI355X path: the B = 32 stage is
host-bound when it runs alone (12.0 ms of issue for 10.3 ms of kernels), a replay costs one launch."""
from collections import OrderedDict
from typing import Dict, Tuple

import torch
from torch import Tensor


class CapturedGraphStage:
    def __init__(self, model, max_graphs: int = 16, warmup: int = 2):
        self.model, self.max_graphs, self.warmup = model, max_graphs, warmup
        self._cache: "OrderedDict[tuple, tuple]" = OrderedDict()
        self.captures = 0
        self.replays = 0

    def _run(self, src: Tensor, lens: Tensor):
        m = self.model
        enc = m.forward_encoder(src, lens)
        prev = m.initialize_output_tokens_by_src(lens, max_src_len=src.shape[1])
        logits, links, feats = m.decode_graph(prev, enc)
        return prev, enc, logits, links, feats

    def _capture(self, key, src: Tensor, lens: Tensor):
        assert src.is_cuda and not self.model.training, "CapturedGraphStage serves eval-mode inference on a GPU"
        s_src, s_len = src.clone(), lens.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=src.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                       # warm-up off the capture: weight packing, position tables, allocator pools
            for _ in range(self.warmup):
                self._run(s_src, s_len)
        cur.wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = self._run(s_src, s_len)
        self.captures += 1
        ent = (s_src, s_len, g, outs)
        self._cache[key] = ent
        while len(self._cache) > self.max_graphs:
            self._cache.popitem(last=False)
        return ent

    @torch.no_grad()
    def __call__(self, src_tokens: Tensor, src_lengths: Tensor) -> Tuple[Tensor, Dict[str, Tensor], Tensor, Tensor, Tensor]:
        """-> (prev_output_tokens, encoder dict, logits [B,L,V], links [B,L,TR], features [B,L,D]) — static tensors of the graph."""
        key = (tuple(src_tokens.shape), src_tokens.dtype, str(src_tokens.device), tuple(src_lengths.shape), src_lengths.dtype)
        ent = self._cache.get(key)
        if ent is None:
            ent = self._capture(key, src_tokens, src_lengths)
        else:
            self._cache.move_to_end(key)
        s_src, s_len, g, outs = ent
        s_src.copy_(src_tokens); s_len.copy_(src_lengths)
        g.replay()
        self.replays += 1
        return outs
