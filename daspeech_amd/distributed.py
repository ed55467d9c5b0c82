"""Data-parallel gradient exchange of the training step: ONE flat-bucket all-reduce per optimizer step, like the reference's
legacy DDP (fairseq/fairseq/distributed/legacy_distributed_data_parallel.py:76-165, buffer_size 2**28, called from
trainer.py:926-930).  Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
A ~190 MB fp16 / 374 MB fp32 bucket is link-bound at ~2 ms on a one-link ring and ~0.3 ms as reduce-scatter + all-gather over
all 7 xGMI links — negligible next to the step, so nothing is overlapped."""
from typing import Iterable

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], world_size: int = None, bucket_elems: int = 2 ** 28, force: bool = False):
    """grad <- sum over ranks of grad / world_size, in place, through flat buckets of at most `bucket_elems` elements.
    force: run the flatten / all-reduce / unflatten path even in a world of one rank (single_rank_self_test: the bucket code and the
    collective backend are exercised on a 1-GPU box, where the call is otherwise a no-op)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    world_size = world_size or dist.get_world_size()
    if world_size == 1 and not force:
        return
    # every rank must flatten the SAME layout: a parameter without a gradient on this rank (unused branch, zero_grad(set_to_none))
    # gets a zero gradient, as the reference's legacy DDP does (legacy_distributed_data_parallel.py:134-137)
    plist = [p for p in params if p.requires_grad]
    for p in plist:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    by_dtype = {}
    for p in plist:
        by_dtype.setdefault(p.grad.dtype, []).append(p.grad)       # _flatten_dense_tensors needs one dtype per bucket

    def reduce_bucket(bucket):
        flat = _flatten_dense_tensors(bucket)
        flat.div_(world_size)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        for g, s in zip(bucket, _unflatten_dense_tensors(flat, bucket)):
            g.copy_(s)
    for dtype in sorted(by_dtype, key=str):
        bucket, n = [], 0
        for g in by_dtype[dtype]:
            if bucket and n + g.numel() > bucket_elems:
                reduce_bucket(bucket)
                bucket, n = [], 0
            bucket.append(g)
            n += g.numel()
        if bucket:
            reduce_bucket(bucket)


def single_rank_self_test(device, backend: str = "nccl", bucket_elems: int = 1000) -> dict:
    """A process group of ONE rank on `device` (backend "nccl" = RCCL) and the flat-bucket exchange forced through it: several buckets, two
    dtypes, a parameter without a gradient.  In a world of one the all-reduce is the identity, so every gradient must come back bit-identical
    (and the missing one as zeros).  What a 1-GPU box can verify of the multi-GPU path: RCCL initialises on this device, the bucket layout
    round-trips, the collective runs on the device's stream.  Returns what it did; raises on any mismatch."""
    import socket
    own_group = not dist.is_initialized()
    if own_group:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
    try:
        g = torch.Generator().manual_seed(5)
        shapes = [(300,), (17, 40), (900,), (5, 5, 5), (64,)]
        params = [torch.nn.Parameter(torch.randn(s, generator=g).to(device)) for s in shapes]
        params.append(torch.nn.Parameter(torch.randn(33, generator=g).to(device).half()))
        for p in params[:-2] + params[-1:]:
            p.grad = torch.randn(p.shape, generator=g).to(device).to(p.dtype)
        want = [None if p.grad is None else p.grad.clone() for p in params]
        all_reduce_gradients(params, bucket_elems=bucket_elems, force=True)
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)
        for p, w in zip(params, want):
            assert p.grad is not None
            assert torch.equal(p.grad, torch.zeros_like(p) if w is None else w), "flat-bucket round trip changed a gradient"
        probe = torch.ones(4, device=device)
        dist.all_reduce(probe)
        assert float(probe.sum()) == 4.0
        return {"backend": dist.get_backend(), "world": dist.get_world_size(), "params": len(params), "bucket_elems": bucket_elems}
    finally:
        if own_group:
            dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------------------
# Length-balanced sharding of utterances over ranks ("balance by src_frames — sort + round-robin").
# Every kernel of this path treats utterances independently, so ranks own disjoint utterances; what has to be balanced is the
# COST: the DAG DP is T·L·TR cells per utterance (L ∝ src_frames, T ∝ src_frames / 13, TR fixed or ∝ L) and the acoustic /
# vocoder stages are linear in the frame count — all monotone in src_frames, which is why the reference's batcher sorts by it
# (fairseq/fairseq/data/data_utils.py batch_by_size on num_tokens = frames) before legacy DDP hands whole batches to ranks.
# ----------------------------------------------------------------------------------------------------------------------------
def dag_cost(src_frames, upsample_ratio: float = 0.5, frames_per_target: float = 13.0, trans_len: int = None):
    """T·L·TR of one utterance as the DAG DP sees it: L = ratio·frames (`--src-upsample-ratio`), T = frames / 13 + 2 (CVSS-C
    statistics, README.md:165-167), TR = min(trans_len, L - 1) (None: the README's dense window L - 1)."""
    f = torch.as_tensor(src_frames, dtype=torch.float64)
    L = (f * upsample_ratio).floor().clamp(min=2)
    T = (f / frames_per_target).round().clamp(min=1) + 2
    TR = L - 1 if trans_len is None else torch.minimum(L - 1, torch.full_like(L, float(trans_len)))
    return T * L * TR


def balanced_shards(src_frames, world_size: int, cost=None):
    """Index lists, one per rank: utterances sorted by descending cost and dealt out in boustrophedon order (0..W-1, W-1..0, ...),
    the "sort + round-robin" split of SURVEY §8e with the serpentine turn that keeps rank 0 from always taking the longest of
    every round.  Deterministic (ties by index), every utterance appears exactly once, shard sizes differ by at most one.
    `cost`: per-utterance cost (default: dag_cost(src_frames) with the dense window)."""
    f = torch.as_tensor(src_frames)
    n = int(f.numel())
    c = dag_cost(f) if cost is None else torch.as_tensor(cost, dtype=torch.float64)
    assert c.numel() == n and world_size >= 1
    order = sorted(range(n), key=lambda i: (-float(c[i]), i))
    shards = [[] for _ in range(world_size)]
    for k, i in enumerate(order):
        rnd, pos = divmod(k, world_size)
        shards[pos if rnd % 2 == 0 else world_size - 1 - pos].append(i)
    return shards


def shard_spread(src_frames, shards, cost=None):
    """(max - min) / mean of the per-rank cost sums — what the max-over-ranks timing of a data-parallel step pays for."""
    f = torch.as_tensor(src_frames)
    c = dag_cost(f) if cost is None else torch.as_tensor(cost, dtype=torch.float64)
    sums = torch.tensor([float(c[torch.as_tensor(s, dtype=torch.long)].sum()) if len(s) else 0.0 for s in shards], dtype=torch.float64)
    return float((sums.max() - sums.min()) / sums.mean())


def shard_sample(sample: dict, idx, device=None):
    """Rows `idx` of a batch dict as make_s2st_batch / the reference's collater build it (nested dicts of [B, ...] tensors), trimmed to
    the shard's own longest utterance on the padded axes the length fields name."""
    idx = torch.as_tensor(idx, dtype=torch.long)

    def take(x):
        if isinstance(x, dict):
            return {k: take(v) for k, v in x.items()}
        if torch.is_tensor(x) and x.dim() >= 1:
            y = x.index_select(0, idx.to(x.device))
            return y.to(device) if device is not None else y
        return x
    out = take(sample)
    ni = out.get("net_input", {})
    if "src_tokens" in ni and "src_lengths" in ni and ni["src_lengths"].numel():
        ni["src_tokens"] = ni["src_tokens"][:, : int(ni["src_lengths"].max())].contiguous()
    if "target_text" in out and "target_text_lengths" in out and out["target_text_lengths"].numel():
        Tm = int(out["target_text_lengths"].max())
        out["target_text"] = out["target_text"][:, :Tm].contiguous()
        for k in ("durations", "pitches", "energies"):
            if k in out:
                out[k] = out[k][:, : Tm - 1].contiguous()
    if "target_audio" in out and "target_audio_lengths" in out and out["target_audio_lengths"].numel():
        out["target_audio"] = out["target_audio"][:, : int(out["target_audio_lengths"].max())].contiguous()
    return out
