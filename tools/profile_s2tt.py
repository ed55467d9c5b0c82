import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.models.daspeech import S2TConformerDAGModel
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = calibrate_synthetic_weights(S2TConformerDAGModel()).to(dev).eval()
model.args.decode_strategy = os.environ.get("DECODE", "lookahead")
b = make_s2st_batch(64, dev, seed=0)["net_input"]
@torch.no_grad()
def step():
    enc = model.forward_encoder(b["src_tokens"], b["src_lengths"])
    prev = model.initialize_output_tokens_by_src(b["src_lengths"], max_src_len=b["src_tokens"].shape[1])
    return model.forward_decoder(prev, enc)["output_tokens"]
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by=os.environ.get("SORT", "self_cuda_time_total"), row_limit=int(os.environ.get("ROWS", "26")), max_name_column_width=60))
