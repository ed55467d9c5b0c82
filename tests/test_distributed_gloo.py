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
