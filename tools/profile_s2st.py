import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.generator import S2SNATGenerator
from daspeech_amd.models import HiFiGANGenerator
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).eval()
voc = HiFiGANGenerator(conv_backend="hip").to(dev).eval()
gen = S2SNATGenerator(voc, torch.zeros(80, device=dev), torch.ones(80, device=dev))
b = make_s2st_batch(32, dev, seed=0)
for _ in range(3): gen.generate(model, b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity, record_function
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    gen.generate(model, b); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by=os.environ.get("SORT", "self_cuda_time_total"), row_limit=28, max_name_column_width=50))
