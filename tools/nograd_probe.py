import sys; sys.path.insert(0, "/root/repo")
import torch, collections
from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
torch.manual_seed(3)
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).cuda().train()
batch = make_s2st_batch(32, "cuda", seed=4)
with torch.autocast("cuda", dtype=torch.float16):
    loss, log = s2s_dag_fastspeech2_loss(model, batch, glat_p="0.5:0.1@200k", update_num=100000)
loss.backward()
c = collections.Counter()
for n, p in model.named_parameters():
    if p.grad is None: c[".".join(n.split(".")[:3])] += 1
print(c)
print([n for n, p in model.named_parameters() if n.startswith("decoder.layers.0") and p.grad is None])
print([n for n, p in model.named_parameters() if n.startswith("decoder") and p.grad is not None])
print({k: (float(v) if torch.is_tensor(v) and v.numel() == 1 else v) for k, v in log.items() if not (torch.is_tensor(v) and v.numel() > 1)})
