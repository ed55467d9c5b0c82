import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
dev = torch.device("cuda:0")
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).train()
b = make_s2st_batch(32, dev, seed=0)
def step():
    with torch.autocast("cuda", dtype=torch.float16 if os.environ.get("AMP", "fp16") == "fp16" else torch.bfloat16):
        loss, log = s2s_dag_fastspeech2_loss(model, b)
    loss.backward()
    model.zero_grad(set_to_none=True)
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by=os.environ.get("SORT", "self_cuda_time_total"), row_limit=int(os.environ.get("ROWS", "40")), max_name_column_width=90))
