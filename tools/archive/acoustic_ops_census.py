#!/usr/bin/env python3
"""Which torch ops (and from which source lines) the acoustic stage still launches beside the HIP kernels: torch.profiler with stacks."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from daspeech_amd.generator import S2SNATGenerator
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
dev = torch.device("cuda"); torch.manual_seed(1234)
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).eval()
gen = S2SNATGenerator(None, torch.zeros(80, device=dev), torch.ones(80, device=dev))
b = make_s2st_batch(32, dev, seed=0)
with torch.no_grad():
    for _ in range(3): gen._acoustic(model, b)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        gen._acoustic(model, b); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    src = next((s for s in (ev.stack or []) if "daspeech_amd" in s), "?")
    k = (ev.name, src.split("daspeech_amd/")[-1][:70])
    agg[k][0] += 1; agg[k][1] += ev.self_device_time_total
tot = sum(v[1] for v in agg.values())
print(f"torch ops with device time: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1] / 1e3:7.3f} ms {v[0]:4d} x  {k[0]:28s} {k[1]}")
