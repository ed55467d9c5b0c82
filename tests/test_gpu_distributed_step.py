"""The data-parallel training step on REAL ranks (BASELINE C5): every rank runs s2s_dag_fastspeech2_loss on its own
utterances (HIP DAG ops, fused links, whole model), the gradients go through the flat-bucket all-reduce of daspeech_amd/distributed.py,
and the result must be what the reference's legacy DDP + trainer produce: grad = sum_r grad_r / world * (world / sum_r sample_size_r) =
the mean of the per-rank gradients (legacy_distributed_data_parallel.py:107-110, trainer.py:932-946; sample_size is 1 per rank).

  * `gloo`, 2 processes sharing ONE GPU: runs on the single-GPU box of the driver (gloo reduces CUDA tensors through the host);
  * `nccl` (= RCCL over xGMI), one process per GPU: skipped unless >= 2 GPUs are visible — the rehearsal for the 8-GPU node."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PICK = ("decoder.gate_linear.weight", "encoder.conformer_layers.1.ffn1.w_1.weight", "decoder.layers.0.fc1.weight", "adaptor.fc1.weight",
        "tts.out_proj.weight", "tts.var_adaptor.embed_pitch.weight", "decoder.embed_tokens.weight")


def _model_and_halves(device):
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
    from daspeech_amd.synthetic import make_s2st_batch
    torch.manual_seed(0)
    m = S2SConformerDAGFastSpeech2Model(encoder_layers=2, decoder_layers=1, tts=dict(enc_layers=1, dec_layers=1)).to(device).eval()
    halves = [make_s2st_batch(3, device, seed=11 + r, min_frames=120, max_frames=200) for r in range(2)]
    return m, halves


def _step(m, batch, seed):
    from daspeech_amd.criterions import S2SDAGFastSpeech2Loss
    crit = S2SDAGFastSpeech2Loss(glat_p="0.5", glance_strategy="number-random", tts_loss_weight=5.0)
    m.zero_grad(set_to_none=True)
    torch.manual_seed(seed)                                   # the glancing draws
    s = dict(batch); s["update_num"] = 10
    loss, sample_size, log = crit(m, s)
    loss.backward()
    return float(loss), sample_size


def _worker(rank, world, port, backend, one_gpu, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0 if one_gpu else rank)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from daspeech_amd.distributed import all_reduce_gradients
    m, halves = _model_and_halves(dev)
    loss, sample_size = _step(m, halves[rank], 100 + rank)
    all_reduce_gradients(m.parameters(), world)
    # trainer.py:932-946: multiply_grads(world / total sample_size); sample_size is 1 on every rank
    ss = torch.tensor([float(sample_size)], device=dev)
    dist.all_reduce(ss)
    for p in m.parameters():
        if p.grad is not None:
            p.grad.mul_(world / float(ss))
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    out = {"rank": rank, "loss": loss, "world": dist.get_world_size(), "n_grads": sum(p.grad is not None for p in m.parameters()),
           "total": float(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in m.parameters() if p.grad is not None))),
           "grads": {k: named[k].grad.float().cpu().numpy() for k in PICK}}      # (numpy: a tensor in the queue is a shared-memory handle that dies with the rank)
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _run(backend, one_gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, one_gpu, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=120)
    # single process, same weights: each half alone, then the mean of the two gradients
    dev = torch.device("cuda", 0)
    m, halves = _model_and_halves(dev)
    want, losses = None, []
    for r in range(2):
        loss, _ = _step(m, halves[r], 100 + r)
        losses.append(loss)
        g = {k: (p.grad.detach().float().clone() if p.grad is not None else torch.zeros_like(p, dtype=torch.float32)) for k, p in m.named_parameters()}
        want = g if want is None else {k: want[k] + g[k] for k in g}
    want = {k: v / 2 for k, v in want.items()}
    total = float(torch.sqrt(sum(v.double().pow(2).sum() for v in want.values())))
    for o in res:
        assert o["world"] == 2
        assert o["loss"] == pytest.approx(losses[o["rank"]], rel=1e-5)
        assert o["n_grads"] == len(want)                      # a parameter without a local gradient still takes part (zero), as legacy DDP
        # Tolerances (r06).  The step's stock-torch layers (MIOpen / hipBLASLt kernels with atomics and run-time algorithm choice) are not
        # run-to-run reproducible on this stack: tools/train_step_determinism.py finds identical single-process steps up to 1e-3 of a gradient's
        # maximum apart (1e-2 on one TTS convolution weight), in discrete alternatives; the DAG ops themselves are bit-reproducible
        # (tools/determinism_stress.py).  The ranks here are fresh processes, the reference below runs in a process earlier tests have warmed.
        # What this test pins is the EXCHANGE — mean over ranks, every parameter taking part: a sum instead of the mean is a factor 2, a
        # missing rank ~50 %.
        assert o["total"] == pytest.approx(total, rel=5e-3)
        for k in PICK:
            w = want[k].cpu().numpy().astype(np.float64)
            d = o["grads"][k].astype(np.float64) - w
            if "embed_pitch" in k or "embed_energy" in k:
                # rows of these tables are addressed by torch.bucketize(value, bins): a value within an ulp of a bin edge lands in either bucket
                # from run to run (3 of 10 runs move ONE frame's gradient to the neighbouring row: 2.4 % of this tensor's norm, always the same
                # alternative).  The exchange is pinned on what a bucket flip cannot change: the sum over the table's rows.
                w, d = w.sum(0), d.sum(0)
            assert np.linalg.norm(d) <= 2e-2 * np.linalg.norm(w) + 1e-7 * total, (k, np.linalg.norm(d) / max(np.linalg.norm(w), 1e-30))
            assert float(np.abs(d).max()) <= 3e-2 * float(np.abs(w).max()) + 1e-7 * total, k
    assert all(np.array_equal(res[0]["grads"][k], res[1]["grads"][k]) for k in PICK)       # both ranks hold the same reduced gradient


def test_two_ranks_one_gpu_gloo_training_step_gradients():
    _run("gloo", one_gpu=True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")
def test_two_ranks_nccl_training_step_gradients():
    _run("nccl", one_gpu=False)


def test_bench_dag_workload_two_ranks_through_its_own_launcher():
    """`python bench.py --gpus 2 --workload dag` from a plain shell: the self-spawn path (torch.distributed.run, 127.0.0.1 rendezvous), two
    ranks running the REAL DAG ops with their own inputs, barrier + max-over-ranks timing, one JSON line from rank 0 whose utt/s counts
    both ranks.  Backend gloo with both ranks on this box's one GPU (DSP_BENCH_BACKEND — rehearsal mode; RCCL needs one GPU per rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["DSP_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "dag", "--steps", "2", "--warmup", "1",
                          "--dag-batch", "4", "--graph-len", "1024", "--tgt-len", "128", "--vocab", "1024", "--no-cpu-baseline", "--no-peaked",
                          "--no-c1"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dag"]["finite_losses"] == 4 and d["dag"]["launch_status"] == 0
    assert abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # whole-job: both ranks' utterances
    assert d["roofline"]["frac"] > 0


def test_single_rank_rccl_flat_bucket_self_test():
    """r06: what a 1-GPU box can verify of the multi-GPU gradient exchange — a one-rank "nccl" (= RCCL) process group on cuda:0 and the
    flat-bucket all-reduce forced through it (several buckets, two dtypes, a parameter without a gradient): gradients come back bit-identical.
    Run in a child process: the test session itself must not keep a process group."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import torch; from daspeech_amd.distributed import single_rank_self_test; "
            "r = single_rank_self_test(torch.device('cuda:0')); assert r['backend'] == 'nccl' and r['world'] == 1, r; print('ok', r)" % ROOT)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
