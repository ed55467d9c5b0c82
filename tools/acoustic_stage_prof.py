#!/usr/bin/env python3
"""The acoustic stage of the S2ST step (fbank -> encoder -> DAG decoder -> links -> graph decode -> adaptor -> FastSpeech2 -> mel, no
vocoder) on its own: host time to ISSUE a batch vs wall time per batch (is it launch-bound?), and — when run under
`rocprofv3 --kernel-trace --stats` — the kernels it consists of.  usage: acoustic_stage_prof.py [batches]
    cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ac -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/acoustic_stage_prof.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.generator import S2SNATGenerator
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda"); torch.manual_seed(1234)
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).eval()
gen = S2SNATGenerator(None, torch.zeros(80, device=dev), torch.ones(80, device=dev))
batches = [make_s2st_batch(32, dev, seed=i) for i in range(2)]
if os.environ.get("RAGGED", "1") == "0":                       # compute every padded row
    model.decoder.ragged = model.tts.ragged = False
with torch.no_grad():
    for i in range(3): gen._acoustic(model, batches[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): gen._acoustic(model, batches[i % 2])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
with torch.no_grad():
    ac = gen._acoustic(model, batches[0])
    gl = batches[0]["net_input"]["src_lengths"] // 2
    print(f"graph lengths: mean {gl.float().mean().item():.0f} of max {gl.max().item()}; mel frames: mean {ac['out_lens'].float().mean().item():.0f} of max "
          f"{ac['out_lens'].max().item()}")
print(f"acoustic stage, B=32: host issue {(t1 - t0) / N * 1e3:.2f} ms/batch, wall {(t2 - t0) / N * 1e3:.2f} ms/batch ({N} batches + 3 warm-up)")
