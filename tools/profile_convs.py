import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.generator import S2SNATGenerator
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).eval()
gen = S2SNATGenerator(None, torch.zeros(80, device=dev), torch.ones(80, device=dev))
b = make_s2st_batch(32, dev, seed=0)
for _ in range(3): gen.generate(model, b, generate_waveform=False)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    gen.generate(model, b, generate_waveform=False); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if "convolution" in e.key and "miopen" in e.key]
for e in sorted(rows, key=lambda e: -e.device_time_total):
    print(e.key, e.count, round(e.device_time_total / 1e3, 3), "ms", [s for s in e.input_shapes[:2]])
