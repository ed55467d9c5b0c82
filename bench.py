#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): per GPU a batch of B=32 utterance graphs, graph_len L=4096, target length
T=512, vocab V=8192, fp32, transition window TR (default 32, `--tr 4095` = README's --max-transition-length 99999).
One STEP = one pass of the DAG training hot path over the batch with inputs resident in HBM:
    dag_logsoftmax_gather_inplace (K1, softmax stored for backward)
 -> dag_loss forward (K2 alpha || K3 beta) -> dag_loss backward (K4, K5) -> K1 backward
 -> dag_best_alignment (K6 + K7, the GLAT alignment).
`value` = utterances/s over all ranks (weak scaling: every rank owns its own 32 utterances, no data-path collective;
SURVEY.md §8e); `ms_per_step` is the "dag_loss fwd+bwd ms/batch" of BASELINE.json's metric for this step definition.
The logits buffer is recycled in place between steps (step s reads what step s-1's K1-backward left there: identical
bytes, flops and control flow, no extra 8.6 GB restore copy inside the timed region).

Extra objects on the JSON line: `roofline` (the DAG DP forward launch, HIP-event timed on the launch stream, against
SURVEY.md §8d's algorithmic bytes), `cpu_baseline` (the reference's torch CPU path, twin in oracle/torch_port.py, on a
bounded sample, rank 0 / N=1 only), `phases_ms` (per-op event timings).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "utterances/sec end-to-end S2ST (fbank→waveform) + dag_loss fwd+bwd ms/batch"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tr", type=int, default=32, help="transition window (32 = tuner default, 4095 = README flag)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 32; 64 for s2tt, BASELINE configs[2])")
    ap.add_argument("--graph-len", type=int, default=4096)
    ap.add_argument("--tgt-len", type=int, default=512)
    ap.add_argument("--vocab", type=int, default=8192)
    ap.add_argument("--workload", default="dag", choices=["dag", "s2tt", "s2st", "train"],
                    help="dag = C2 DAG-op hot path (default, the roofline-carrying line); s2tt = C3 speech-to-text forward + graph decode; s2st = C4 full fbank->waveform pipeline; "
                         "train = C5 DASpeech training step (s2s_dag_fastspeech2_loss + flat-bucket gradient all-reduce)")
    ap.add_argument("--vocoder-backend", default="hip", choices=["torch", "hip"])
    ap.add_argument("--amp", default="none", choices=["none", "bf16", "fp16"],
                    help="s2st only: autocast dtype of the dense Conformer / Transformer / FastSpeech2 layers (the reference runs --fp16); "
                         "default fp32, the mode the mel parity (<= 1e-4) is stated for")
    ap.add_argument("--decode-strategy", default="lookahead", choices=["lookahead", "greedy", "viterbi", "jointviterbi"],
                    help="s2tt / s2st: graph decode mode (the reference's test_scripts run lookahead and jointviterbi)")
    ap.add_argument("--vocoder-group", type=int, default=8, help="s2st: utterances per vocoder call (length-sorted groups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="1,12", help="B,T of the bounded CPU sample")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 64 if args.workload == "s2tt" else 32
    return args


def cpu_baseline(args):
    """Reference CPU path (torch_dag_logsoftmax_gather_inplace -> torch_dag_loss fwd+bwd -> torch_dag_best_alignment,
    dense [B,L,L] links) on a bounded sample of the same workload; cost is linear in B and in T."""
    import torch
    from oracle import torch_port
    cores = os.cpu_count() or 1
    sb, stt = [int(v) for v in args.cpu_sample.split(",")]
    L, V = args.graph_len, min(args.vocab, 8192)
    # dense links make the reference CPU path's cost independent of TR; the sample uses TR=L-1 so the end is reachable
    r = torch_port.time_cpu_dag_path(sb, stt, L, V, L - 1, threads=cores)
    sample_s = r["fwd_s"] + r["bwd_s"] + r.get("align_s", 0.0)
    # linear extrapolation to the full batch: B/sb samples, (T-1)/(stt-1) DP rows (fwd, bwd and alignment all scale so)
    full_s = sample_s * (args.batch / sb) * ((args.tgt_len - 1) / (stt - 1))
    return {
        "value": args.batch / full_s, "unit": "utt/s", "cores": cores, "kind": "port",
        "threads": torch.get_num_threads(),
        "sample": f"B={sb},T={stt} of B={args.batch},T={args.tgt_len} at L={L},V={V} (dense links); "
                  f"measured {sample_s:.2f}s (fwd {r['fwd_s']:.2f} bwd {r['bwd_s']:.2f} align {r.get('align_s', 0):.2f}), "
                  f"extrapolated linearly in B and T to {full_s:.0f}s per batch",
        "sample_seconds": sample_s, "extrapolated_batch_seconds": full_s,
    }


def run_model_workload(args, torch, dist, dev, world, rank):
    """C4 (s2st) / C5 (train): released architecture (93.6 M parameters), random weights, synthetic CVSS-C shaped batches."""
    from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
    from daspeech_amd.distributed import all_reduce_gradients
    from daspeech_amd.generator import S2SNATGenerator
    from daspeech_amd.models import HiFiGANGenerator
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model, S2TConformerDAGModel
    from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
    torch.manual_seed(1234)
    B = args.batch
    model = calibrate_synthetic_weights(S2TConformerDAGModel() if args.workload == "s2tt" else S2SConformerDAGFastSpeech2Model()).to(dev)
    model.args.decode_strategy = args.decode_strategy
    batches = [make_s2st_batch(B, dev, seed=100 * rank + i) for i in range(2)]
    args.warmup = max(args.warmup, 2 * len(batches))     # MIOpen/hipBLASLt pick algorithms per new shape: keep that out of the timing

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    roof = None
    amp_dtype = {"none": None, "bf16": torch.bfloat16, "fp16": torch.float16}[args.amp]
    if args.workload == "s2tt":
        model.eval()

        @torch.no_grad()
        def step(i):
            ni = batches[i % len(batches)]["net_input"]
            with torch.autocast("cuda", dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
                enc = model.forward_encoder(ni["src_tokens"], ni["src_lengths"])
                prev = model.initialize_output_tokens_by_src(ni["src_lengths"], max_src_len=ni["src_tokens"].shape[1])
                return model.forward_decoder(prev, enc)["output_tokens"]
        wl = (f"C3 S2TT full forward (s2t_conformer_dag): Conformer(12L,256) -> DA-Transformer(4L,512) + fused links -> {args.decode_strategy} "
              f"graph decode to tokens, B={B}/GPU, fbank80 300-800 frames, " + ("fp32" if args.amp == "none" else f"{args.amp} autocast"))
    elif args.workload == "s2st":
        model.eval()
        voc = HiFiGANGenerator(conv_backend=args.vocoder_backend).to(dev).eval()
        gen = S2SNATGenerator(voc, torch.zeros(80, device=dev), torch.ones(80, device=dev), vocoder_group=args.vocoder_group)
        frames = [0]

        def step(i):
            with torch.autocast("cuda", dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
                out = gen.generate(model, batches[i % len(batches)])
            frames[0] += sum(o["feature"].shape[0] for o in out)
            return out
        wl = (f"C4 full S2ST pipeline, {args.decode_strategy} decode: Conformer(12L,256) -> DA-Transformer(4L,512) + links -> HIP graph decode -> "
              f"FFN adapter -> FastSpeech2-NoEmb (HIP variance-adaptor glue + length regulator) -> HiFi-GAN V1 ({args.vocoder_backend} convs), "
              f"B={B}/GPU, fbank80 300-800 frames, " + ("fp32" if args.amp == "none" else f"{args.amp} autocast dense layers / fp32 graph + TTS glue ops"))
    else:
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01,
                               fused=True)        # one multi-tensor kernel (fairseq uses its FusedAdam the same way); foreach: 73 ms, fused: 68.5 ms per step
        # the reference trains with fairseq's --fp16 (README.md:241,274: fp16 compute, dynamic loss scaling); --amp bf16 selects
        # bf16 autocast instead (no loss scaling; MIOpen falls back to a naive bf16 weight-gradient conv: 83 vs 72 ms per step)
        use_fp16 = args.amp != "bf16"
        train_dtype = torch.float16 if use_fp16 else torch.bfloat16
        scaler = torch.amp.GradScaler("cuda", enabled=use_fp16, init_scale=2.0 ** 7)

        def step(i):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=train_dtype):
                loss, log = s2s_dag_fastspeech2_loss(model, batches[i % len(batches)])
            scaler.scale(loss).backward()
            all_reduce_gradients(model.parameters(), world)            # ONE flat bucket
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            scaler.step(opt)
            scaler.update()
            return loss
        wl = (f"C5 DASpeech training step: s2s_dag_fastspeech2_loss (GLAT two-pass, HIP DAG ops, expect strategy) fwd+bwd + "
              f"flat-bucket gradient all-reduce + Adam, B={B}/GPU (global {B * world}), " + ("fp16 autocast + loss scaling" if use_fp16 else "bf16 autocast") + " dense layers / fp32 DAG ops")
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if args.workload == "s2tt" and rank == 0 and amp_dtype is None:
        # roofline of the workload's dominant kernel family, the fp32-accurate split GEMM on the fp16 matrix cores:
        # the Conformer feed-forward's first GEMM at this batch's row count; 3 MFMAs per product, priced against the dense fp16 peak
        from daspeech_amd import decode_ops
        lin = model.encoder.conformer_layers[0].ffn1["w_1"]
        ni = batches[0]["net_input"]
        with torch.no_grad():
            Tenc = int(model.forward_encoder(ni["src_tokens"], ni["src_lengths"])["encoder_out"].shape[1])
            xg = torch.randn(B, Tenc, lin.in_features, device=dev)
            if decode_ops.split_linear(xg, lin, act="silu") is not None:
                for _ in range(3):
                    decode_ops.split_linear(xg, lin, act="silu")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    decode_ops.split_linear(xg, lin, act="silu")
                e1.record(); torch.cuda.synchronize()
                g_ms = e0.elapsed_time(e1) / 20
                mf = 3 * 2.0 * B * Tenc * lin.in_features * lin.out_features / (g_ms * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": f"conv1d_split_kernel (Linear {lin.in_features}->{lin.out_features} + SiLU on {B * Tenc} rows, fp32-accurate: "
                                                   "3 fp16 MFMAs per product)", "achieved": mf, "peak": 2500.0, "unit": "TFLOP/s", "frac": mf / 2500.0,
                        "traffic": None, "avg_call_ms": g_ms, "effective_fp32_TFLOPs": mf / 3}
    if args.workload == "s2st":
        extra["mel_frames_per_utt"] = frames[0] / max(1, (args.steps + args.warmup) * B)
        # roofline of the pipeline's dominant hand-written kernel family, the HiFi-GAN conv stack (MFMA bound, 0.614 GFLOP per mel
        # frame): one vocoder call of the pipeline's group shape, timed with events on the launch stream
        if rank == 0 and args.vocoder_backend == "hip":
            Tm = max(8, int(round(extra["mel_frames_per_utt"])))
            mel = torch.randn(args.vocoder_group, 80, Tm, device=dev)
            with torch.no_grad():
                for _ in range(2):
                    voc(mel)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    voc(mel)
                e1.record(); torch.cuda.synchronize()
            v_ms = e0.elapsed_time(e1) / 5
            tf = 0.614e9 * args.vocoder_group * Tm / (v_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "HiFi-GAN V1 generator conv stack (hifigan_conv / hifigan_resunit kernels), one call of "
                                               f"{args.vocoder_group} x {Tm} frames", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": tf / 2500.0, "traffic": None, "avg_call_ms": v_ms}
    result = {
        "metric": METRIC, "value": world * B * args.steps / elapsed, "unit": "utt/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": ("f32" if args.amp == "none" else args.amp) if args.workload in ("s2st", "s2tt") else ("bf16" if args.amp == "bf16" else "fp16"), "data": "synthetic",
        "config": {"workload": wl, "batch_per_gpu": B, "parallelism": f"dp{world}", **extra},
        "roofline": roof, "cpu_baseline": None,
    }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
    from daspeech_amd import custom_ops as ops
    from daspeech_amd import _lib
    _lib.load()
    if args.workload != "dag":
        return run_model_workload(args, torch, dist, dev, world, rank)
    import sys as _sys
    _mod = _sys.modules["daspeech_amd.custom_ops.dag_loss"]
    lsg_fwd, lsg_bwd, lsg_fwd_lazy = _mod._lsg_forward, _mod._lsg_backward, _mod._lsg_forward_lazy

    B, L, T, V = args.batch, args.graph_len, args.tgt_len, args.vocab
    TR = min(args.tr, L - 1)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    cg = torch.Generator().manual_seed(1234 + rank)
    logits = torch.randn(B, L, V, device=dev, generator=gen)
    out_len = (L - torch.randint(0, 5, (B,), generator=cg)).to(dev)
    tgt_len = (T - torch.randint(0, 5, (B,), generator=cg)).to(dev)
    tgt = torch.randint(4, V, (B, T), generator=cg).to(dev)
    raw = torch.randn(B, L, TR, device=dev, generator=gen)
    i = torch.arange(L, device=dev).view(1, L, 1)
    d = torch.arange(TR, device=dev).view(1, 1, TR)
    valid = (i + d + 1) < out_len.view(B, 1, 1)
    dead = ~valid.any(-1, keepdim=True)
    links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(dead, 0.0), -1)
    links = links.masked_fill(~valid, float("-inf")).contiguous()
    del raw, valid, dead
    idx = tgt.unsqueeze(1).expand(-1, L, -1)

    names = ["gather_fwd", "dag_fwd", "dag_bwd", "gather_bwd", "best_alignment"]
    ev = {n: [] for n in names}

    def step(record, lazy=False, store=None):
        store = ev if store is None else store

        def mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record()              # current stream == the stream the C ABI launches on
            return e
        k = links.detach().requires_grad_()
        e0 = mark()
        # K1 through the same launch wrappers the autograd Function uses (the logits buffer is a recycled leaf here,
        # so the Function's mark_dirty contract cannot be exercised on it; tests cover the Function itself)
        if lazy:        # backward state = two floats per row; the logits are only overwritten by the backward (custom_ops.set_lazy_softmax)
            match_all, stats = lsg_fwd_lazy(logits, idx)
            match_all.requires_grad_()
        else:
            match_all, stats = lsg_fwd(logits, idx, True).requires_grad_(), None   # [B,T,L] contiguous; logits <- softmax
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
                store[n].append((a, b))
        return loss, gx, gk, path

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out[0]).all(), "non-finite loss in the benchmark batch"

    phases = {n: sum(a.elapsed_time(b) for a, b in ev[n]) / max(1, len(ev[n])) for n in names}

    # the same step with the gather's backward state kept as row statistics (the mode daspeech_amd.criterions uses; the
    # reference forbids reading the logits buffer after the op, so the in-place softmax is not observable): reported beside
    # the headline numbers, never as `value`
    ev_lazy = {n: [] for n in names}
    logits.normal_(generator=gen)
    for _ in range(max(2, args.warmup)):
        step(False, lazy=True)
    barrier()
    t0l = time.perf_counter()
    for _ in range(args.steps):
        out_l = step(True, lazy=True, store=ev_lazy)
    barrier()
    elapsed_l = time.perf_counter() - t0l
    if world > 1:
        tt = torch.tensor([elapsed_l], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_l = float(tt.item())
    assert torch.isfinite(out_l[0]).all()
    lazy_report = {"ms_per_step": elapsed_l * 1e3 / args.steps, "value": world * B * args.steps / elapsed_l, "unit": "utt/s",
                   "phases_ms": {n: sum(a.elapsed_time(b) for a, b in ev_lazy[n]) / max(1, len(ev_lazy[n])) for n in names},
                   "note": "gather backward state = (max, 1/sum-exp) per row instead of the in-place softmax: same match and gradients"}
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * B * args.steps / elapsed

    # roofline of the DAG DP forward launch (alpha + beta): SURVEY.md §8(d)
    #   2 * (B*T*L*4 [read match] + B*L*TR*4 [read links] + B*T*L*4 [write alpha/beta])
    alg_bytes = 2.0 * (B * T * L * 4 + B * L * TR * 4 + B * T * L * 4)
    dag_fwd_ms = phases["dag_fwd"]
    achieved = alg_bytes / (dag_fwd_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_dag_fwd.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(f"tr{TR}", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "dag_loss forward DP (alpha||beta, one launch)", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes": alg_bytes, "avg_launch_ms": dag_fwd_ms}

    # the same accounting for every phase of the step (SURVEY.md §8(d) byte counts; a phase = the launches of one operator call)
    BTL, BLV, BLTR = B * T * L * 4.0, B * L * V * 4.0, B * L * TR * 4.0
    phase_bytes = {
        "gather_fwd": 2 * BLV + BTL,                     # logits read, softmax written in place, match written
        "gather_bwd": 2 * BLV + BTL,                     # softmax read, gradient written in place, grad_match read
        "dag_fwd": alg_bytes,
        "dag_bwd": 3 * BTL + BLTR + BTL + BLTR,          # alpha, beta, match + links read; grad_match + grad_links written
        "best_alignment": 2 * BTL + BLTR,                # match read, alpha_max written (no trace tensor), links read
    }
    roofline_phases = {n: {"algorithmic_bytes": phase_bytes[n], "ms": phases[n],
                           "achieved_GBps": phase_bytes[n] / (phases[n] * 1e-3) / 1e9,
                           "frac_of_hbm_peak": phase_bytes[n] / (phases[n] * 1e-3) / 1e9 / HBM_PEAK_GBS} for n in names}
    result = {
        "metric": METRIC, "value": value, "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C2 DAG training hot path: logsoftmax_gather + dag_loss fwd+bwd + dag_best_alignment, "
                               f"B={B}/GPU, graph_len={L}, tgt_len={T}, vocab={V}, TR={TR}, fp32",
                   "batch_per_gpu": B, "graph_len": L, "tgt_len": T, "vocab": V, "trans_len": TR,
                   "parallelism": f"dp{world} (independent utterances per rank, no data-path collective)"},
        "roofline": roofline, "phases_ms": phases, "roofline_phases": roofline_phases, "lazy_softmax": lazy_report,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
