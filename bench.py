#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: "utterances/sec end-to-end S2ST (fbank→waveform) + dag_loss fwd+bwd ms/batch".

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under torch.distributed.run with N ranks (one per
GPU, RCCL); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  Every rank owns its own utterances (weak scaling, no
data-path collective — SURVEY.md §8e); the only collectives are the contract's barrier and the max-over-ranks of the timings.

The default run (`--workload headline`) measures BOTH halves of the metric and prints them in ONE JSON line:

  A. C4 (BASELINE configs[3]): the full S2ST pipeline fbank -> waveform, B=32 per GPU, lookahead decode, at the REFERENCE's
     precision: fp32 acoustic model and fp32 vocoder (HIP kernels; the vocoder and the FastSpeech2 convolutions multiply fp32 operands
     split hi/lo on the fp16 matrix cores with fp32 accumulation — within 2^-22 of an fp32 convolution, waveform <= 1e-4 of the
     reference generator's).  One STEP = one batch through S2SNATGenerator.  K steps timed between barrier + synchronize on both
     sides.  `value` = utterances/s over all ranks, `ms_per_step` = ms per batch.  `s2st_fp16_vocoder` is the same pipeline with the
     fp16-STORAGE vocoder (narrower than the reference: 1.5e-3 off its waveform) — reported beside the headline, never as it.
     `s2st_sustained` repeats the headline leg for 200 more batches (a longer GPU phase for the driver's utilisation sampler).
  B. C2 (configs[1]): the DAG training hot path, B=32 per GPU, graph_len 4096, tgt_len 512, vocab 8192, TR=32, fp32:
     dag_logsoftmax_gather_inplace -> dag_loss fwd (alpha || beta) -> dag_loss bwd -> gather bwd -> dag_best_alignment.
     K passes, each phase bracketed by HIP events on the launch stream; `dag.dag_loss_fwd_bwd_ms` is the metric's second half and
     `roofline` prices the DP forward launch against SURVEY.md §8(d)'s algorithmic bytes.  Every pass starts from FRESH logits
     (restored from a master copy between the event brackets), and a second input set with peaked, trained-model-like scores
     (`dag.peaked`) exercises the exactness-guard paths the random inputs never enter.
  C. C1 (configs[0]: B=4, T=256, L=2048, V=512, TR=L-1, the reference's CPU-runnable case) on the HIP ops, next to
     `cpu_baseline`: the reference's torch CPU path (twin in oracle/torch_port.py) MEASURED at C1 on this box's host cores
     — thread count swept, warm-up + median, no extrapolation (rank 0, N=1 only).

Other workloads (`--workload dag|s2st|s2tt|train`) run one of the parts alone (train = C5 per-GPU step with the flat-bucket
gradient all-reduce).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
METRIC = "utterances/sec end-to-end S2ST (fbank→waveform) + dag_loss fwd+bwd ms/batch"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="headline", choices=["headline", "dag", "s2tt", "s2st", "train", "plumbing"],
                    help="headline (default) = C4 S2ST utt/s + C2 DAG ops with the DP roofline + C1 vs the CPU baseline, one JSON line; "
                         "dag / s2st / s2tt / train = that part alone")
    ap.add_argument("--tr", type=int, default=32, help="C2 transition window (32 = banded fast path, 4095 = README's --max-transition-length 99999)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch of the model workloads (default 32; 64 for s2tt, BASELINE configs[2])")
    ap.add_argument("--dag-batch", type=int, default=32)
    ap.add_argument("--graph-len", type=int, default=4096)
    ap.add_argument("--tgt-len", type=int, default=512)
    ap.add_argument("--vocab", type=int, default=8192)
    ap.add_argument("--vocoder-backend", default="hip", choices=["torch", "hip", "hip_fp16"],
                    help="hip = fp32 activations / weights as the reference (split operands on the fp16 matrix cores), hip_fp16 = fp16 storage")
    ap.add_argument("--sustain-steps", type=int, default=200, help="headline: extra S2ST batches after the K timed ones (reported separately)")
    ap.add_argument("--amp", default="none", choices=["none", "bf16", "fp16"],
                    help="autocast dtype of the dense Conformer / Transformer / FastSpeech2 layers (default fp32, the mode the mel parity is stated for)")
    ap.add_argument("--decode-strategy", default="lookahead", choices=["lookahead", "greedy", "viterbi", "jointviterbi"])
    ap.add_argument("--vocoder-group", type=int, default=None,
                    help="s2st: utterances per vocoder call (length-sorted groups; default: the whole batch for the fp32 vocoder, 8 for hip_fp16)")
    ap.add_argument("--no-overlap", action="store_true", help="s2st: one batch at a time (generator.generate) instead of the two-deep batch pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="seconds the cpu_baseline leg may take")
    ap.add_argument("--no-peaked", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC record instead of two rocprofv3 --pmc child passes of this run")
    ap.add_argument("--no-c1", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline: skip the C3 (s2tt, B=64) and C5 (train, B=32, fp16 model) legs")
    ap.add_argument("--extra-steps", type=int, default=6, help="headline: timed steps of the C3 / C5 legs")
    ap.add_argument("--torch-links", action="store_true", help="train: the torch [B,L,L,h] formulation of extract_links instead of the fused band kernels")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 64 if args.workload == "s2tt" else 32
    return args


def vocoder_group(args):
    """Utterances per vocoder call.  The kernels skip every tile past an utterance's own length, so a padded group costs the sum of its
    lengths; the fp32 vocoder's launches are long enough that one call per batch is fastest (r03: 36.5 vs 38.3 ms per batch of 32), the
    fp16-storage one keeps the groups of 8 its two-deep pipeline was tuned with."""
    if args.vocoder_group is not None:
        return args.vocoder_group
    return 8 if args.vocoder_backend == "hip_fp16" else args.batch


# ======================================================================================================================
# launcher: --gpus N spawns N ranks itself when it was not started by one
# ======================================================================================================================
def maybe_spawn(args):
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return False
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


class Ctx:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # DSP_BENCH_BACKEND=gloo: the rehearsal mode of the multi-rank plumbing (tests/): ranks may share a device, collectives on host
        # tensors.  The default, and the only mode a measurement may use, is "nccl" (= RCCL), one rank per GPU.
        self.backend = os.environ.get("DSP_BENCH_BACKEND", "nccl")
        self.plumbing = args.workload == "plumbing"
        if not torch.cuda.is_available():
            if not self.plumbing:
                raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
            self.backend = "gloo"
        if args.gpus != self.world:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch one rank per GPU")
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev:
            if self.backend == "nccl" and self.local_rank >= ndev:
                raise SystemExit(f"rank {self.rank}: local rank {self.local_rank} but only {ndev} GPUs are visible")
            torch.cuda.set_device(self.local_rank % ndev)
            self.dev = torch.device("cuda", self.local_rank % ndev)
        else:
            self.dev = torch.device("cpu")
        self.cdev = self.dev if self.backend == "nccl" else torch.device("cpu")       # where collective payloads live
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=self.dev)
            else:
                dist.init_process_group(backend="gloo")
            self.world = dist.get_world_size()            # n_gpus of the line comes from the communicator's world, not from the flag
            assert self.world == args.gpus, f"world size {self.world} != --gpus {args.gpus}"
            # one collective before anything is timed: every rank really is in the communicator, on its own device
            probe = torch.tensor([1.0, float(torch.cuda.current_device() if ndev else self.rank)], device=self.cdev)
            dist.all_reduce(probe)
            assert int(probe[0].item()) == self.world, f"all-reduce over {self.world} ranks summed to {probe[0].item()}"
            if self.backend == "nccl" or not ndev:
                assert int(probe[1].item()) == self.world * (self.world - 1) // 2, "ranks share a device"
            name = torch.cuda.get_device_name(self.dev) if ndev else "cpu"
            print(f"[bench] rank {self.rank}/{self.world} on {self.dev} ({name}), {self.backend} all-reduce ok", file=sys.stderr, flush=True)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if self.dev.type == "cuda":
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world > 1:
            t = self.torch.tensor([seconds], device=self.cdev, dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return seconds

    def timed(self, step, steps, warmup, flush=None):
        """`flush` drains a pipelined step (work of the last submitted batch still queued on the host side): called at the end of the
        warm-up and INSIDE the timed region after the K-th step, so the K timed steps contain all the work of exactly K batches."""
        for i in range(warmup):
            step(i)
        if flush is not None:
            flush()
        self.barrier()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            out = step(warmup + i)
        if flush is not None:
            flush()
        self.barrier()
        self.last_local_s = time.perf_counter() - t0      # this rank's own clock (scaling_diag); the reported time is the max over ranks
        return self.max_over_ranks(self.last_local_s), out

    def gather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (a list of one without a process group)."""
        if self.world > 1:
            out = [None] * self.world
            self.dist.all_gather_object(out, obj)
            return out
        return [obj]


# ======================================================================================================================
# C1 CPU baseline (rank 0, N = 1): the reference's torch path, measured — SURVEY.md §8(d)
# ======================================================================================================================
def cpu_baseline_c1(budget_s):
    """torch_dag_logsoftmax_gather_inplace -> torch_dag_loss (dense [B,L,L] links) fwd + autograd bwd at C1 (B=4, T=256, L=2048,
    V=512) on the host cores.  Thread count swept on a short slice (T=9), then full-size runs — one warm-up on the slice, then as
    many full repetitions (<= 3) as the budget allows; the median is reported.  Nothing is extrapolated: if not even ONE full C1
    pass fits the budget, T is halved until it does and the sample says so."""
    import torch
    from oracle import torch_port
    B, T, L, V = 4, 256, 2048, 512
    cores = os.cpu_count() or 1
    t_start = time.perf_counter()
    sweep = {}
    cand = sorted({n for n in (8, 16, 32, 64, 128, cores) if n <= cores})
    torch_port.time_cpu_dag_path(1, 5, 512, V, 511, threads=cand[0], with_alignment=False)           # page in, warm the allocator
    for n in cand:
        r = torch_port.time_cpu_dag_path(B, 9, L, V, L - 1, threads=n, with_alignment=False)
        sweep[n] = r["fwd_s"] + r["bwd_s"]
        if time.perf_counter() - t_start > 0.35 * budget_s:
            break
    best = min(sweep, key=sweep.get)
    est_full = sweep[best] * (T - 1) / 8.0
    Tm = T
    while est_full > 0.6 * budget_s and Tm > 16:                      # does one full pass fit?  (it does on every box seen so far)
        Tm //= 2
        est_full /= 2
    runs = []
    while len(runs) < 3 and (not runs or time.perf_counter() - t_start + est_full < budget_s):
        r = torch_port.time_cpu_dag_path(B, Tm, L, V, L - 1, threads=best, with_alignment=False)
        runs.append((r["fwd_s"] + r["bwd_s"], r["fwd_s"], r["bwd_s"]))
        est_full = runs[-1][0]
    runs.sort()
    med = runs[len(runs) // 2]
    return {
        "value": B / med[0], "unit": "utt/s", "cores": best, "kind": "port", "host_cores": cores,
        "sample": f"C1 (BASELINE configs[0]) B={B}, T={Tm}, L={L}, V={V}, dense links: torch_dag_logsoftmax_gather_inplace + torch_dag_loss forward + "
                  f"autograd backward, {best} threads (best of sweep {dict((k, round(v, 3)) for k, v in sweep.items())} s on a T=9 slice), "
                  f"median of {len(runs)} full runs after a warm-up" + ("" if Tm == T else f" — T reduced from {T} to fit the time budget"),
        "seconds_per_batch": med[0], "fwd_s": med[1], "bwd_s": med[2], "runs": len(runs), "tgt_len": Tm,
        "leg_seconds": time.perf_counter() - t_start,
    }


# ======================================================================================================================
# DAG ops (C2 / C1)
# ======================================================================================================================
def make_dag_inputs(torch, dev, B, L, T, V, TR, seed, peaked=False):
    gen = torch.Generator(device=dev).manual_seed(seed)
    cg = torch.Generator().manual_seed(seed)
    out_len = (L - torch.randint(0, 5, (B,), generator=cg)).to(dev)
    tgt_len = (T - torch.randint(0, 5, (B,), generator=cg)).to(dev)
    tgt = torch.randint(4, V, (B, T), generator=cg).to(dev)
    logits = torch.randn(B, L, V, device=dev, generator=gen)
    raw = torch.randn(B, L, TR, device=dev, generator=gen)
    if peaked:
        # what a trained model produces (tools/peaked_bench.py): the aligned token of a band of vertices around the diagonal scores
        # ~+14 nats over the rest of the vocabulary (log-prob ~ -0.3 on the band, ~ -12 - 14 elsewhere), 4-sigma transition logits
        raw = raw * 4.0
        j = torch.arange(L, device=dev).view(1, L)
        centre = (j.float() * (T - 1) / (L - 1)).round().long().clamp(0, T - 1)            # target index a vertex most likely emits
        for off in (-1, 0, 1):
            t_idx = (centre + off).clamp(0, T - 1).expand(B, L)
            tok = tgt.gather(1, t_idx)
            logits.scatter_add_(2, tok.unsqueeze(-1), torch.full((B, L, 1), 9.0 if off else 14.0, device=dev))
    i = torch.arange(L, device=dev).view(1, L, 1)
    d = torch.arange(TR, device=dev).view(1, 1, TR)
    valid = (i + d + 1) < out_len.view(B, 1, 1)
    dead = ~valid.any(-1, keepdim=True)
    links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(dead, 0.0), -1)
    links = links.masked_fill(~valid, float("-inf")).contiguous()
    return logits, links, out_len, tgt_len, tgt


def run_links_ops(ctx, B, L, TR, steps, warmup, seed):
    """The transition producer in front of the DP (DAGDecoder.extract_links, s2t_conformer_dag.py:171-212) at the DP leg's graph: the released
    link predictor's geometry (8 heads x 64), random q / k / gates.  HIP-event ms of the inference call and of forward + backward under autograd."""
    torch = ctx.torch
    from daspeech_amd import decode_ops
    dev = ctx.dev
    g = torch.Generator(device=dev); g.manual_seed(seed)
    q0 = torch.randn(B, L, 8, 64, device=dev, generator=g) * 0.5
    k0 = torch.randn(B, L, 8, 64, device=dev, generator=g) * 0.5
    lg = torch.log_softmax(torch.randn(B, L, 8, device=dev, generator=g), -1)
    olen = torch.full((B,), L, device=dev, dtype=torch.long)
    w = torch.randn(B, L, TR, device=dev, generator=g)

    def timed(fn):
        for _ in range(warmup):
            fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / steps

    def infer():
        with torch.no_grad():
            decode_ops.extract_links(q0, k0, lg, olen, TR)

    def train():
        q, k, gt = q0.detach().requires_grad_(), k0.detach().requires_grad_(), lg.detach().requires_grad_()
        decode_ops.extract_links_autograd(q, k, gt, olen, TR).backward(w)

    return {"links_fwd_ms": timed(infer), "links_fwd_bwd_ms": timed(train)}


def run_dag_ops(ctx, B, L, T, V, TR, steps, warmup, seed, peaked=False, lazy=False, fresh=True):
    """K passes of the DAG hot path; returns per-phase HIP-event times (ms) and the wall time of the passes."""
    torch = ctx.torch
    from daspeech_amd import custom_ops as ops
    mod = sys.modules["daspeech_amd.custom_ops.dag_loss"]
    lsg_fwd, lsg_bwd, lsg_fwd_lazy = mod._lsg_forward, mod._lsg_backward, mod._lsg_forward_lazy
    dev = ctx.dev
    logits, links, out_len, tgt_len, tgt = make_dag_inputs(torch, dev, B, L, T, V, TR, seed, peaked)
    master = logits.clone() if fresh else None
    idx = tgt.unsqueeze(1).expand(-1, L, -1)
    names = ["gather_fwd", "dag_fwd", "dag_bwd", "gather_bwd", "best_alignment"]
    ev = {n: [] for n in names}

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()              # current stream == the stream the C ABI launches on
        return e

    def step(record):
        if master is not None:
            logits.copy_(master)                                     # fresh logits every pass; outside every event bracket
        k = links.detach().requires_grad_()
        e0 = mark()
        # K1 through the launch wrappers the autograd Function uses (the logits buffer is a recycled leaf here, so the Function's
        # mark_dirty contract cannot be exercised on it; tests cover the Function itself)
        if lazy:
            match_all, stats = lsg_fwd_lazy(logits, idx)
            match_all.requires_grad_()
        else:
            match_all, stats = lsg_fwd(logits, idx, True).requires_grad_(), None           # [B,T,L] contiguous; logits <- softmax
        e1 = mark()
        loss = ops.dag_loss(match_all, k, out_len, tgt_len)
        e2 = mark()
        obj = -(loss / tgt_len).mean()
        go = torch.autograd.grad(obj, [loss], retain_graph=True)[0]
        e2b = mark()
        gm, gk = torch.autograd.grad(loss, [match_all, k], grad_outputs=go)
        e3 = mark()
        gx = lsg_bwd(logits, idx, gm.transpose(1, 2), stats)
        e4 = mark()
        with torch.no_grad():
            path = ops.dag_best_alignment(match_all.detach(), links, out_len, tgt_len)
        e5 = mark()
        if record:
            for n, (a, b) in zip(names, [(e0, e1), (e1, e2), (e2b, e3), (e3, e4), (e4, e5)]):
                ev[n].append((a, b))
        return loss, gx, gk, path

    for _ in range(warmup):
        step(False)
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(True)
    ctx.barrier()
    wall = ctx.max_over_ranks(time.perf_counter() - t0)
    finite = bool(torch.isfinite(out[0]).all())
    if not peaked:
        assert finite, "non-finite loss in the benchmark batch"
    phases = {n: sum(a.elapsed_time(b) for a, b in ev[n]) / max(1, len(ev[n])) for n in names}
    from daspeech_amd import _lib
    status = _lib.last_launch_status()
    del logits, master
    torch.cuda.empty_cache()
    return phases, wall, {"finite_losses": int(torch.isfinite(out[0]).sum()), "launch_status": status}


def live_dp_traffic(args, TR):
    """HBM bytes per launch of the DP forward kernel, OBSERVED by this run: two short child runs of this script's `dag` workload under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes), per-dispatch
    averages of the DP kernel, FETCH doubled (the guide's gfx950 correction for 16-byte-per-lane streams).  None if rocprofv3 is not there or
    a pass fails — the committed record of tools/refresh_profiles.sh is used then."""
    import csv, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or os.environ.get("DSP_BENCH_CHILD"):
        return None
    kern = "dag_strip4g_kernel" if TR <= 32 else "dag_dense_mfma_kernel"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dsp_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", "dag", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-c1", "--no-peaked", "--tr", str(args.tr),
                   "--dag-batch", str(args.dag_batch), "--graph-len", str(args.graph_len), "--tgt-len", str(args.tgt_len), "--vocab", str(args.vocab)]
            env = dict(os.environ, DSP_BENCH_CHILD="1", TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            files = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not files:
                return None
            tot, disp = 0.0, set()
            for row in csv.DictReader(open(files[0])):
                if kern in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    tot += float(row["Counter_Value"]); disp.add(row["Dispatch_Id"])
            if not disp:
                return None
            vals[counter] = tot / len(disp)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"hbm_bytes_per_launch": int(round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)), "FETCH_SIZE_KB_raw": vals["FETCH_SIZE"],
            "WRITE_SIZE_KB": vals["WRITE_SIZE"], "source": "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over `bench.py --workload dag` "
                                                            f"(per-dispatch mean of {kern}, FETCH doubled per MI355X_MICROARCH.md)"}


def dag_report(ctx, args, steps, warmup):
    B, L, T, V = args.dag_batch, args.graph_len, args.tgt_len, args.vocab
    TR = min(args.tr, L - 1)
    phases, wall, info = run_dag_ops(ctx, B, L, T, V, TR, steps, warmup, 1234 + ctx.rank)
    BTL, BLV, BLTR = B * T * L * 4.0, B * L * V * 4.0, B * L * TR * 4.0
    # 2 * (B*T*L*4 [read match] + B*L*TR*4 [read links] + B*T*L*4 [write alpha/beta])
    alg_bytes = 2.0 * (BTL + BLTR + BTL)
    phase_bytes = {"gather_fwd": 2 * BLV + BTL, "gather_bwd": 2 * BLV + BTL, "dag_fwd": alg_bytes,
                   "dag_bwd": 3 * BTL + BLTR + BTL + BLTR, "best_alignment": 2 * BTL + BLTR}
    ms = phases["dag_fwd"]
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    traffic, src = None, None
    live = None
    if ctx.rank == 0 and ctx.world == 1 and not getattr(args, "no_live_traffic", False):
        ctx.torch.cuda.synchronize()
        live = live_dp_traffic(args, TR)
    pmc = os.path.join(ROOT, "profiles", "pmc_dag_fwd.json")
    if live is not None:
        traffic, src = live["hbm_bytes_per_launch"], live["source"]
    elif os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get(f"tr{TR}", {})
            if rec.get("shape") in (None, [B, T, L, TR]):
                traffic, src = rec.get("hbm_bytes_per_launch"), rec.get("source", "profiles/pmc_dag_fwd.json")
        except Exception:
            pass
    roofline = {"bound": "hbm", "kernel": "dag_loss forward DP (alpha||beta, one launch)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                "algorithmic_bytes": alg_bytes, "avg_launch_ms": ms,
                # the banded DP is T dependent rows per launch — latency per row beside the bandwidth figure (the launch is paced by
                # this chain, not by HBM: profiles/r04_dp_coresidency.txt)
                "us_per_row": ms * 1e3 / T, "rows_per_launch": T}
    step_ms = sum(phases.values())
    rep = {
        "workload": f"C2 DAG training hot path: logsoftmax_gather + dag_loss fwd+bwd + dag_best_alignment, B={B}/GPU, graph_len={L}, "
                    f"tgt_len={T}, vocab={V}, TR={TR}, fp32, fresh logits every pass",
        "dag_loss_fwd_bwd_ms": phases["dag_fwd"] + phases["dag_bwd"], "step_ms": step_ms, "utt_per_s": ctx.world * B / (step_ms * 1e-3),
        "phases_ms": phases, "wall_ms_per_pass_incl_restore": wall * 1e3 / steps,
        "roofline_phases": {n: {"algorithmic_bytes": phase_bytes[n], "ms": phases[n], "achieved_GBps": phase_bytes[n] / (phases[n] * 1e-3) / 1e9,
                                "frac_of_hbm_peak": phase_bytes[n] / (phases[n] * 1e-3) / 1e9 / HBM_PEAK_GBS} for n in phases},
        **info,
    }
    if not args.no_peaked:
        pp, _, pinfo = run_dag_ops(ctx, B, L, T, V, TR, max(3, steps // 2), 2, 4321 + ctx.rank, peaked=True)
        rep["peaked"] = {"note": "trained-model-like scores (aligned tokens +14 nats on a band around the diagonal, 4-sigma transition logits)",
                         "phases_ms": pp, "dag_loss_fwd_bwd_ms": pp["dag_fwd"] + pp["dag_bwd"],
                         "dag_fwd_vs_random": pp["dag_fwd"] / phases["dag_fwd"], **pinfo}
    lp, _, _ = run_dag_ops(ctx, B, L, T, V, TR, max(3, steps // 2), 2, 1234 + ctx.rank, lazy=True)
    rep["lazy_softmax"] = {"note": "gather backward state = (max, 1/sum-exp) per row instead of the in-place softmax (the mode daspeech_amd.criterions "
                                   "uses): same match and gradients", "phases_ms": lp, "step_ms": sum(lp.values())}
    return rep, roofline


def c1_report(ctx, steps):
    """BASELINE configs[0] on the HIP ops: B=4, T=256, L=2048, V=512, TR=L-1 (the reference's dense training window)."""
    B, T, L, V = 4, 256, 2048, 512
    phases, _, info = run_dag_ops(ctx, B, L, T, V, L - 1, max(3, steps // 2), 2, 99 + ctx.rank)
    rep = {"workload": f"C1 B={B}, T={T}, L={L}, V={V}, TR={L - 1} on the HIP ops", "phases_ms": phases,
           "dag_loss_fwd_bwd_ms": phases["dag_fwd"] + phases["dag_bwd"],
           "gather_plus_dag_fwd_bwd_ms": phases["gather_fwd"] + phases["dag_fwd"] + phases["dag_bwd"] + phases["gather_bwd"], **info}
    # the same forward on trained-model-like scores (emissions near 0 on a band around the alignment, -20 nats elsewhere, 4-sigma
    # transition logits).  r02: the matrix-core DP spent its exact-redo budget on them and the stand-by log-space kernels finished the
    # batch (7.1 ms); r03: the diagonal block's per-lane-reference pass keeps them in exp space — reported beside the
    # friendly case with the ratio, and with what the launch itself says about hand-overs and exact-redo cells
    try:
        torch = ctx.torch
        from daspeech_amd import custom_ops as ops, _lib
        d = ctx.dev; g = torch.Generator(device=d).manual_seed(7); TR = L - 1
        ol = torch.full((B,), L, device=d); tl = torch.full((B,), T, device=d)
        i = torch.arange(L, device=d).view(1, L, 1); dd = torch.arange(TR, device=d).view(1, 1, TR); valid = (i + dd + 1) < L
        raw = 4.0 * torch.randn(B, L, TR, device=d, generator=g)
        links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
        j = torch.arange(L, device=d).view(1, 1, L).float(); c = (torch.arange(T, device=d).float() * (L - 1) / (T - 1)).view(1, T, 1)
        match = torch.where((j - c).abs() < 6, -0.5 + 0.3 * torch.randn(B, T, L, device=d, generator=g), -20.0 + 3.0 * torch.randn(B, T, L, device=d, generator=g))
        del raw
        for _ in range(2): loss = ops.dag_loss(match, links, ol, tl)
        status = _lib.last_launch_status(); gave_up = _lib.last_dense_gave_up(); redo_cells = _lib.last_fallback_count()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): loss = ops.dag_loss(match, links, ol, tl)
        e1.record(); torch.cuda.synchronize()
        rep["peaked"] = {"note": "trained-model-like scores: forward only", "dag_fwd_ms": e0.elapsed_time(e1) / 3, "launch_status": int(status),
                         "handed_to_standby_kernels": bool(gave_up), "exact_redo_cells": int(redo_cells), "finite_losses": int(torch.isfinite(loss).sum())}
        rep["peaked"]["dag_fwd_vs_friendly"] = rep["peaked"]["dag_fwd_ms"] / phases["dag_fwd"]
    except Exception as e:          # noqa: the friendly-case report must not depend on this leg
        rep["peaked"] = {"error": repr(e)[:200]}
    return rep


# ======================================================================================================================
# model workloads (C3 / C4 / C5)
# ======================================================================================================================
VOCODER_ARITH = {
    "hip": "fp32 activations + weights as the reference, operands split hi/lo on the fp16 matrix cores, fp32 accumulate (waveform <= 1e-4 of the reference generator)",
    "hip_fp16": "fp16-STORAGE activations + weights, fp32 accumulate (narrower than the reference: waveform 1.5e-3 off)",
    "torch": "torch / MIOpen fp32 convolutions",
}


def build_model_step(ctx, args, workload):
    torch = ctx.torch
    from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
    from daspeech_amd.distributed import all_reduce_gradients
    from daspeech_amd.generator import S2SNATGenerator
    from daspeech_amd.models import HiFiGANGenerator
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model, S2TConformerDAGModel
    from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
    dev, world, rank = ctx.dev, ctx.world, ctx.rank
    torch.manual_seed(1234)
    B = args.batch
    model = calibrate_synthetic_weights(S2TConformerDAGModel() if workload == "s2tt" else S2SConformerDAGFastSpeech2Model()).to(dev)
    model.args.decode_strategy = args.decode_strategy
    if getattr(args, "torch_links", False):
        model.decoder.fused_links = False
    if world > 1:
        # ONE global pool of world * B utterances per step (same seed on every rank), split by balanced_shards on src_frames: the per-rank
        # sum of T*L*TR (and of frames) differs by < 1 % instead of the 10-50 % of an arbitrary split — the max-over-ranks timing pays for it
        from daspeech_amd.distributed import balanced_shards, shard_sample
        batches = []
        for i in range(2):
            pool = make_s2st_batch(B * world, "cpu", seed=9000 + i)
            shards = balanced_shards(pool["net_input"]["src_lengths"], world)
            batches.append(shard_sample(pool, shards[rank], dev))
    else:
        batches = [make_s2st_batch(B, dev, seed=100 * rank + i) for i in range(2)]
    amp_dtype = {"none": None, "bf16": torch.bfloat16, "fp16": torch.float16}[args.amp]
    state = {"frames": 0, "model": model, "batches": batches, "B": B}
    prec = "fp32" if args.amp == "none" else f"{args.amp} autocast dense layers / fp32 graph + TTS glue ops"
    if workload == "s2tt":
        model.eval()

        @torch.no_grad()
        def step(i):
            ni = batches[i % len(batches)]["net_input"]
            with torch.autocast("cuda", dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
                enc = model.forward_encoder(ni["src_tokens"], ni["src_lengths"])
                prev = model.initialize_output_tokens_by_src(ni["src_lengths"], max_src_len=ni["src_tokens"].shape[1])
                return model.forward_decoder(prev, enc)["output_tokens"]
        wl = (f"C3 S2TT full forward (s2t_conformer_dag): Conformer(12L,256) -> DA-Transformer(4L,512) + fused links -> {args.decode_strategy} "
              f"graph decode to tokens, B={B}/GPU, fbank80 300-800 frames, {prec}")
    elif workload == "s2st":
        model.eval()
        voc = HiFiGANGenerator(conv_backend=args.vocoder_backend).to(dev).eval()
        gen = S2SNATGenerator(voc, torch.zeros(80, device=dev), torch.ones(80, device=dev), vocoder_group=vocoder_group(args))
        state["voc"] = voc

        def count(out):
            if out is not None:
                state["frames"] += sum(o["feature"].shape[0] for o in out)
            return out

        if args.no_overlap:
            def step(i):
                with torch.autocast("cuda", dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
                    return count(gen.generate(model, batches[i % len(batches)]))
        else:
            # two-deep pipeline over consecutive batches (generator.submit / flush): the vocoder of batch i-1 runs on a second stream
            # under the launch-bound acoustic model of batch i; ctx.timed flushes inside the timed region
            def step(i):
                with torch.autocast("cuda", dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
                    return count(gen.submit(model, batches[i % len(batches)]))
            state["flush"] = lambda: count(gen.flush())
        wl = (f"C4 full S2ST pipeline (s2s_conformer_dag_fastspeech2 + HiFi-GAN V1), {args.decode_strategy} decode: Conformer(12L,256) -> "
              f"DA-Transformer(4L,512) + links -> HIP graph decode -> FFN adapter -> FastSpeech2-NoEmb (HIP variance-adaptor glue + length regulator) "
              f"-> HiFi-GAN V1 (vocoder arithmetic: {VOCODER_ARITH[args.vocoder_backend]}; groups of {vocoder_group(args)} with per-utterance lengths"
              + ("" if args.no_overlap else "; vocoder of batch i-1 on a second stream under the acoustic model of batch i") + f"), B={B}/GPU, fbank80 300-800 frames, {prec}")
    else:
        model.train()
        if args.amp == "bf16":
            # torch.autocast(bf16) over an fp32 model — NOT the reference's scheme, kept as a comparison leg
            opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, fused=True)

            def step(i):
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss, log = s2s_dag_fastspeech2_loss(model, batches[i % len(batches)], glat_p="0.5:0.1@200k", update_num=100000 + i)
                loss.backward()
                all_reduce_gradients(model.parameters(), world)
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
                return loss
            scheme = "bf16 autocast over fp32 weights, Adam"
        else:
            # the reference's --fp16 (README.md:241,274): model.half(), fp16 batch, flat fp32 master + Adam, dynamic loss scaling
            from daspeech_amd.fp16_trainer import FP16FlatOptimizer, half_sample
            model.half()
            hb = [half_sample(b) for b in batches]
            state["batches"] = hb
            opt = FP16FlatOptimizer(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, clip_norm=1.0, init_scale=2.0 ** 7)
            state["opt"] = opt

            def step(i):
                opt.zero_grad()
                loss, log = s2s_dag_fastspeech2_loss(model, hb[i % len(hb)], glat_p="0.5:0.1@200k", update_num=100000 + i)
                opt.backward(loss)
                all_reduce_gradients(model.parameters(), world)            # ONE flat fp16 bucket: sum / world
                opt.step()                                                 # (x world / total sample_size = 1: sample_size is 1 per rank)
                return loss
            scheme = "fp16 model + fp16 batch, flat fp32 master weights, Adam (decoupled decay), dynamic loss scaling, clip-norm 1 — fairseq's --fp16"
        wl = (f"C5 DASpeech training step: s2s_dag_fastspeech2_loss (GLAT two-pass with number-random glancing, HIP DAG ops in fp32, expect strategy, dropout "
              f"0.1 / 0.1 / 0.1 as README) fwd+bwd + flat-bucket gradient all-reduce + optimizer, B={B}/GPU (global {B * world}), {scheme}")
    return step, wl, state


def vocoder_roofline(ctx, args, state):
    """The pipeline's dominant hand-written kernel family, the HiFi-GAN conv stack (MFMA bound, 0.614 GFLOP per mel frame):
    one vocoder call of the pipeline's group shape, timed with events on the launch stream."""
    torch = ctx.torch
    voc = state.get("voc")
    if voc is None or args.vocoder_backend not in ("hip", "hip_fp16"):
        return None
    Tm = max(8, int(round(state["mel_frames_per_utt"])))
    vg = vocoder_group(args)
    mel = torch.randn(vg, 80, Tm, device=ctx.dev)
    with torch.no_grad():
        for _ in range(2):
            voc(mel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            voc(mel)
        e1.record(); torch.cuda.synchronize()
    v_ms = e0.elapsed_time(e1) / 5
    tf = 0.614e9 * vg * Tm / (v_ms * 1e-3) / 1e12
    f32 = args.vocoder_backend == "hip"
    kern = ("hifigan_conv_f32 kernels: three fp16 MFMAs per fragment pair, so `achieved` counts the convolution's FLOPs and the matrix cores issue 3x that"
            if f32 else "hifigan_conv / hifigan_resunit kernels")
    return {"bound": "mfma", "kernel": f"HiFi-GAN V1 generator conv stack ({kern}), one call of {vg} x {Tm} frames",
            "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS, "traffic": None, "avg_call_ms": v_ms,
            **({"mfma_issue_factor": 3, "frac_of_issue_peak": 3 * tf / MFMA_F16_PEAK_TFLOPS} if f32 else {})}


def scaling_diag(ctx, batches, steps):
    """Per-rank record for the 1 -> 8 GPU runs the driver times (r06): which device every rank ran on, how many utterances / source frames /
    DAG cells (sum of T*L*TR, distributed.dag_cost) its shard of each step's pool held, and its OWN wall time next to the reported maximum — so
    that a scaling efficiency below 1 can be read as imbalance (spread of cost or of time) or as interference.  No efficiency is computed here."""
    from daspeech_amd.distributed import dag_cost
    torch = ctx.torch
    lens = [b["net_input"]["src_lengths"].detach().cpu() for b in batches]
    mine = {"rank": ctx.rank, "device": str(ctx.dev),
            "device_name": torch.cuda.get_device_name(ctx.dev) if ctx.dev.type == "cuda" else "cpu",
            "utterances_per_step": [int(x.numel()) for x in lens], "src_frames_per_step": [int(x.sum()) for x in lens],
            "dag_cells_per_step": [float(dag_cost(x).sum()) for x in lens],
            "local_ms_per_step": getattr(ctx, "last_local_s", 0.0) * 1e3 / max(1, steps)}
    ranks = ctx.gather_objects(mine)
    cost = [sum(r["dag_cells_per_step"]) for r in ranks]
    frames = [sum(r["src_frames_per_step"]) for r in ranks]
    tms = [r["local_ms_per_step"] for r in ranks]
    spread = lambda v: (max(v) - min(v)) / max(v) if max(v) > 0 else 0.0
    return {"ranks": ranks, "dag_cells_spread": spread(cost), "src_frames_spread": spread(frames), "local_time_spread": spread(tms),
            "backend": ctx.backend if ctx.world > 1 else "none (single rank)"}


def run_model(ctx, args, workload, steps, warmup, sustain=0):
    step, wl, state = build_model_step(ctx, args, workload)
    warmup = max(warmup, 2 * len(state["batches"]))      # MIOpen / hipBLASLt pick algorithms per new shape: keep that out of the timing
    elapsed, _ = ctx.timed(step, steps, warmup, flush=state.get("flush"))
    B = state["B"]
    rep = {"workload": wl, "value": ctx.world * B * steps / elapsed, "unit": "utt/s", "ms_per_step": elapsed * 1e3 / steps, "steps": steps,
           "warmup": warmup, "batch_per_gpu": B}
    rep["scaling_diag"] = scaling_diag(ctx, state["batches"], steps)
    if sustain > 0:                                      # the same step, many more times (not the contract's K: reported beside it)
        el2, _ = ctx.timed(step, sustain, 0, flush=state.get("flush"))
        rep["sustained"] = {"steps": sustain, "value": ctx.world * B * sustain / el2, "unit": "utt/s", "ms_per_step": el2 * 1e3 / sustain}
        steps += sustain
    if workload == "s2st":
        state["mel_frames_per_utt"] = state["frames"] / max(1, (steps + warmup) * B)
        rep["mel_frames_per_utt"] = state["mel_frames_per_utt"]
        if ctx.rank == 0:
            rep["roofline"] = vocoder_roofline(ctx, args, state)
    if workload == "train":
        rep["peak_memory_GB"] = ctx.torch.cuda.max_memory_allocated() / 2 ** 30
        if state.get("opt") is not None:
            rep["loss_scale"] = state["opt"].scaler.loss_scale
            rep["last_grad_norm"] = state["opt"].last_grad_norm
        rep["workload"] += ("; extract_links: torch [B,L,L,h] formulation" if getattr(args, "torch_links", False)
                            else "; extract_links: fused compact-band HIP forward + backward")
    state.clear()
    ctx.torch.cuda.empty_cache()
    return rep


# ======================================================================================================================
def main():
    args = parse()
    maybe_spawn(args)
    ctx = Ctx(args)
    # library selection switches for the stock-torch part of the training step (experiments; defaults are what the numbers are quoted with)
    if os.environ.get("DSP_TRAIN_BLAS"):
        ctx.torch.backends.cuda.preferred_blas_library(os.environ["DSP_TRAIN_BLAS"])          # "cublas" (= rocBLAS) | "cublaslt" (= hipBLASLt)
    if os.environ.get("DSP_CUDNN_BENCHMARK"):
        ctx.torch.backends.cudnn.benchmark = os.environ["DSP_CUDNN_BENCHMARK"] == "1"
    from daspeech_amd import _lib
    _lib.load()                                       # fails loudly if the HIP extension is missing
    world, rank = ctx.world, ctx.rank
    if ctx.plumbing:
        # launcher / rendezvous / barrier / max-over-ranks / sharding rehearsal: K trivial "steps" (a sleep proportional to this rank's
        # shard cost), no kernels — value is NOT a measurement and the line says so
        from daspeech_amd.distributed import balanced_shards, shard_spread
        g = ctx.torch.Generator().manual_seed(17)
        frames = ctx.torch.randint(300, 801, (32 * world,), generator=g)
        shards = balanced_shards(frames, world)
        mine = float(frames[ctx.torch.tensor(shards[rank])].sum())
        el, _ = ctx.timed(lambda i: time.sleep(mine * 1e-8), args.steps, args.warmup)
        diag = scaling_diag(ctx, [{"net_input": {"src_lengths": frames[ctx.torch.tensor(shards[rank])]}}], args.steps)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": el * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "none", "data": "none", "config": {"workload": "plumbing rehearsal: no kernels, NOT a measurement",
                                                                           "backend": ctx.backend, "shard_spread": shard_spread(frames, shards)},
                              "roofline": None, "cpu_baseline": None, "scaling_diag": diag}))
        if world > 1:
            ctx.dist.destroy_process_group()
        return
    base = {"metric": METRIC, "unit": "utt/s", "n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "data": "synthetic"}
    par = f"dp{world} (independent utterances per rank, no data-path collective)"
    if args.workload in ("s2tt", "s2st", "train"):
        rep = run_model(ctx, args, args.workload, args.steps, args.warmup)
        dtype = ("f32" if args.amp == "none" else args.amp) if args.workload != "train" else ("bf16" if args.amp == "bf16" else "fp16")
        result = {**base, "value": rep["value"], "steps": rep["steps"], "warmup": rep["warmup"], "ms_per_step": rep["ms_per_step"], "dtype": dtype,
                  "config": {"workload": rep["workload"], "batch_per_gpu": rep["batch_per_gpu"], "parallelism": par,
                             **({"mel_frames_per_utt": rep["mel_frames_per_utt"]} if "mel_frames_per_utt" in rep else {}),
                             **({"peak_memory_GB": rep["peak_memory_GB"]} if "peak_memory_GB" in rep else {})},
                  "roofline": rep.get("roofline"), "cpu_baseline": None, "scaling_diag": rep.get("scaling_diag")}
    elif args.workload == "dag":
        rep, roofline = dag_report(ctx, args, args.steps, args.warmup)
        result = {**base, "value": rep["utt_per_s"], "steps": args.steps, "warmup": args.warmup, "ms_per_step": rep["step_ms"], "dtype": "f32",
                  "config": {"workload": rep["workload"], "batch_per_gpu": args.dag_batch, "parallelism": par}, "roofline": roofline, "dag": rep,
                  "cpu_baseline": None}
        if not args.no_c1:
            result["c1"] = c1_report(ctx, args.steps)
    else:
        # headline: A (C4 S2ST, carries `value`), B (C2 DAG ops, carries `roofline`), C (C1 HIP vs the measured CPU baseline)
        import copy
        s2st = run_model(ctx, args, "s2st", args.steps, args.warmup, sustain=args.sustain_steps if args.vocoder_backend == "hip" else 0)
        fast = None
        if args.vocoder_backend == "hip":                 # the fp16-storage vocoder beside the headline
            a16 = copy.copy(args); a16.vocoder_backend = "hip_fp16"
            fast = run_model(ctx, a16, "s2st", args.steps, args.warmup)
        dag, roofline = dag_report(ctx, args, args.steps, args.warmup)
        # the README's training window (--max-transition-length 99999 -> TR = L-1) at C2, timed by THIS run
        tr_full = None
        if min(args.tr, args.graph_len - 1) != args.graph_len - 1:
            try:
                pf, _, inf = run_dag_ops(ctx, args.dag_batch, args.graph_len, args.tgt_len, args.vocab, args.graph_len - 1, 3, 1, 77 + rank)
                tr_full = {"workload": f"C2 with the README's dense window: B={args.dag_batch}, graph_len={args.graph_len}, tgt_len={args.tgt_len}, TR={args.graph_len - 1}",
                           "dag_fwd_ms": pf["dag_fwd"], "dag_bwd_ms": pf["dag_bwd"], "best_alignment_ms": pf["best_alignment"],
                           "dag_loss_fwd_bwd_ms": pf["dag_fwd"] + pf["dag_bwd"], "phases_ms": pf, **inf}
                # dense window: every DP row is a triangular [1 x L].[L x L] product in both directions -> 2 dirs * B * T * L^2/2 * 2 FLOP on the
                # fp32 matrix cores (v_mfma_f32_16x16x4_f32, 157 TFLOP/s dense peak); the alignment is the same count of (max, +) pairs on the VALU
                Lf, fl = float(args.graph_len), 2.0 * args.dag_batch * args.tgt_len * float(args.graph_len) ** 2
                tr_full["roofline"] = {"bound": "mfma", "kernel": "dag_dense_mfma forward (alpha||beta)", "flop": fl, "achieved": fl / (pf["dag_fwd"] * 1e-3) / 1e12,
                                       "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / (pf["dag_fwd"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None}
                tr_full["alignment_roofline"] = {"bound": "valu", "kernel": "dag_dense_max (max-plus products) + block back-trace", "pair_ops": fl / 4.0,
                                                 "achieved_Gpairs_per_s": fl / 4.0 / (pf["best_alignment"] * 1e-3) / 1e9}
                # the producer of that window's transitions (extract_links, 8 heads x 64) on the same graph: inference call, forward + backward
                try:
                    ctx.torch.cuda.empty_cache()
                    lk = run_links_ops(ctx, args.dag_batch, args.graph_len, args.graph_len - 1, 3, 1, 91 + rank)
                    tr_full["extract_links"] = {"workload": f"extract_links (8 heads x 64) B={args.dag_batch}, graph_len={args.graph_len}, TR={args.graph_len - 1}: "
                                                            "inference call / forward + backward under autograd", **lk,
                                                "links_bytes": 4.0 * args.dag_batch * args.graph_len * (args.graph_len - 1),
                                                "score_flop": 2.0 * args.dag_batch * 8 * 64 * Lf * (Lf - 1) / 2}
                except Exception as e:      # noqa: the DP numbers of this leg do not depend on it
                    tr_full["extract_links"] = {"error": repr(e)[:200]}
            except Exception as e:      # noqa: a leg of its own, the headline does not depend on it
                tr_full = {"error": repr(e)[:200]}
        # 32 < TR <= 64: exp-space strips with two vertices per lane (forward), values-only max-DP strips (alignment), the TR <= 32 gradient kernel in two planes of 32 transitions (backward)
        tr64 = None
        try:
            p64, _, i64 = run_dag_ops(ctx, args.dag_batch, args.graph_len, args.tgt_len, args.vocab, 64, 3, 1, 55 + rank)
            b64 = 2.0 * (2 * args.dag_batch * args.tgt_len * args.graph_len * 4.0 + args.dag_batch * args.graph_len * 64 * 4.0)
            tr64 = {"workload": "C2 with TR=64 (windows 33..64: dag_dp_strip2g forward, dag_dp_maxstripw alignment, backward = the exp-space gradient kernel in two planes of 32 transitions, one launch)", "phases_ms": p64,
                    "dag_loss_fwd_bwd_ms": p64["dag_fwd"] + p64["dag_bwd"], "dag_fwd_frac_of_hbm_peak": b64 / (p64["dag_fwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, **i64}
        except Exception as e:      # noqa
            tr64 = {"error": repr(e)[:200]}
        result = {**base, "value": s2st["value"], "steps": s2st["steps"], "warmup": s2st["warmup"], "ms_per_step": s2st["ms_per_step"],
                  "dtype": "f32" if args.amp == "none" else args.amp,
                  "dag_loss_fwd_bwd_ms_per_batch": dag["dag_loss_fwd_bwd_ms"],
                  "config": {"workload": "value / ms_per_step: " + s2st["workload"] + "  ||  dag_loss_fwd_bwd_ms_per_batch / roofline: " + dag["workload"],
                             "batch_per_gpu": s2st["batch_per_gpu"], "dag_batch_per_gpu": args.dag_batch, "graph_len": args.graph_len,
                             "tgt_len": args.tgt_len, "vocab": args.vocab, "trans_len": min(args.tr, args.graph_len - 1),
                             "mel_frames_per_utt": s2st.get("mel_frames_per_utt"), "parallelism": par},
                  "roofline": roofline, "s2st_vocoder_roofline": s2st.get("roofline"), "dag": dag, "cpu_baseline": None,
                  "scaling_diag": s2st.get("scaling_diag")}
        result["config"]["vocoder_arithmetic"] = VOCODER_ARITH[args.vocoder_backend]
        if "sustained" in s2st:
            result["s2st_sustained"] = s2st["sustained"]
        if fast is not None:
            result["s2st_fp16_vocoder"] = {"value": fast["value"], "unit": "utt/s", "ms_per_step": fast["ms_per_step"], "steps": fast["steps"],
                                           "note": "same pipeline, " + VOCODER_ARITH["hip_fp16"] + " — not the reference's precision, not the headline",
                                           "vocoder_roofline": fast.get("roofline")}
        if tr_full is not None:
            result["dag_tr%d" % (args.graph_len - 1)] = tr_full
        result["dag_tr64"] = tr64
        # C3 and C5 in the same driver-timed line (short legs; `--workload s2tt|train` run them alone at the contract's K)
        if not args.no_extra_legs:
            for key, wl, nb in (("c3_s2tt", "s2tt", 64), ("c5_train", "train", 32)):
                try:
                    ax = copy.copy(args); ax.batch = nb
                    rr = run_model(ctx, ax, wl, args.extra_steps, 2)
                    result[key] = {"workload": rr["workload"], "value": rr["value"], "unit": "utt/s", "ms_per_step": rr["ms_per_step"],
                                   "steps": rr["steps"], "warmup": rr["warmup"], "batch_per_gpu": rr["batch_per_gpu"],
                                   "dtype": "f32" if wl == "s2tt" else "fp16",
                                   **({k: rr[k] for k in ("peak_memory_GB", "loss_scale", "last_grad_norm") if k in rr})}
                except Exception as e:      # noqa: legs of their own
                    result[key] = {"error": repr(e)[:300]}
        if not args.no_c1:
            result["c1"] = c1_report(ctx, args.steps)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("headline", "dag"):
        cb = cpu_baseline_c1(args.cpu_budget)
        result["cpu_baseline"] = cb
        if "c1" in result and cb["tgt_len"] == 256:
            gpu_s = result["c1"]["gather_plus_dag_fwd_bwd_ms"] * 1e-3
            result["c1"]["speedup_vs_cpu_baseline"] = cb["seconds_per_batch"] / gpu_s
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
