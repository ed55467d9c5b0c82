"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange used by the training step."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from daspeech_amd.distributed import all_reduce_gradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    x = torch.full((3, 8), float(rank + 1))
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    all_reduce_gradients(model.parameters(), bucket_elems=100)          # small bucket: exercises several flushes
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for g, out in zip(local, gathered):
        dist.all_gather(out, g)
    ok = all(torch.allclose(p.grad, sum(o) / world, atol=1e-6) for p, o in zip(model.parameters(), gathered))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_flat_bucket_all_reduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_balanced_shards_spread_on_cvss_shaped_sample():
    """sort by src_frames + round-robin.  Per-rank sum of T*L*TR within 5 % on a CVSS-C shaped sample (300-800 frames, 32 per
    rank) for 2 / 4 / 8 ranks, dense and banded window; every utterance exactly once; shard sizes within one."""
    from daspeech_amd.distributed import balanced_shards, shard_spread, dag_cost
    g = torch.Generator().manual_seed(3)
    for world in (2, 4, 8):
        for trial in range(5):
            frames = torch.randint(300, 801, (32 * world,), generator=g)
            shards = balanced_shards(frames, world)
            assert sorted(i for s in shards for i in s) == list(range(32 * world))
            assert max(map(len, shards)) - min(map(len, shards)) <= 1
            for tr in (None, 32):
                cost = dag_cost(frames, trans_len=tr)
                spread = shard_spread(frames, shards, cost)
                assert spread <= 0.05, (world, trial, tr, spread)
            # frames themselves (the acoustic / vocoder cost) too
            assert shard_spread(frames, shards, frames.double()) <= 0.05
    # what it replaces: a strided split of the same pool is 5-50x worse
    frames = torch.randint(300, 801, (256,), generator=g)
    naive = [list(range(r, 256, 8)) for r in range(8)]
    assert shard_spread(frames, naive) > 3 * shard_spread(frames, balanced_shards(frames, 8))
    # ragged pool (not a multiple of the world size), ties, one rank
    assert balanced_shards([5, 5, 5], 2) == [[0], [1, 2]] or sum(map(len, balanced_shards([5, 5, 5], 2))) == 3
    assert balanced_shards([7, 3, 9], 1) == [[2, 0, 1]]
    assert balanced_shards([], 4) == [[], [], [], []]


def test_shard_sample_trims_to_the_shard():
    from daspeech_amd.distributed import balanced_shards, shard_sample
    from daspeech_amd.synthetic import make_s2st_batch
    pool = make_s2st_batch(12, "cpu", seed=5)
    shards = balanced_shards(pool["net_input"]["src_lengths"], 3)
    seen = []
    for r in range(3):
        sub = shard_sample(pool, shards[r])
        idx = torch.tensor(shards[r])
        n = sub["net_input"]["src_lengths"]
        assert torch.equal(n, pool["net_input"]["src_lengths"][idx])
        assert sub["net_input"]["src_tokens"].shape[1] == int(n.max())
        for j, i in enumerate(shards[r]):
            f = int(n[j])
            assert torch.equal(sub["net_input"]["src_tokens"][j, :f], pool["net_input"]["src_tokens"][i, :f])
            t = int(sub["target_text_lengths"][j])
            assert torch.equal(sub["target_text"][j, :t], pool["target_text"][i, :t])
            assert torch.equal(sub["durations"][j, : t - 1], pool["durations"][i, : t - 1])
            m = int(sub["target_audio_lengths"][j])
            assert torch.equal(sub["target_audio"][j, :m], pool["target_audio"][i, :m])
        seen += shards[r]
    assert sorted(seen) == list(range(12))


def test_single_rank_self_test_gloo():
    """distributed.single_rank_self_test on CPU (gloo): the forced flat-bucket round trip in a world of one (the GPU twin runs it over RCCL)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import torch; from daspeech_amd.distributed import single_rank_self_test; "
            "r = single_rank_self_test(torch.device('cpu'), backend='gloo'); assert r['world'] == 1 and r['params'] == 6, r; print('ok')" % root)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
