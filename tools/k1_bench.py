import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
mod = sys.modules.get("daspeech_amd.custom_ops.dag_loss") or __import__("daspeech_amd.custom_ops.dag_loss", fromlist=["x"])
import daspeech_amd.custom_ops
mod = sys.modules["daspeech_amd.custom_ops.dag_loss"]
B, L, V, T = 32, 4096, 8192, 512
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(B, L, V, device=dev).to(dt)
    tgt = torch.randint(0, V, (B, T), device=dev)
    idx = tgt.unsqueeze(1).expand(-1, L, -1)
    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
    es = x.element_size()
    rd = B * L * V * es / 1e9
    t_ro = t(lambda: mod._lsg_forward(x, idx, False))
    t_rw = t(lambda: mod._lsg_forward(x, idx, True))
    g = torch.randn(B, T, L, device=dev)
    t_bw = t(lambda: mod._lsg_backward(x, idx, g.transpose(1, 2)))
    print(f"{dt}: fwd no-store {t_ro:.3f} ms ({(rd+0.27)/t_ro*1e3:.0f} GB/s) | fwd+softmax {t_rw:.3f} ms ({(2*rd+0.27)/t_rw*1e3:.0f} GB/s) | bwd {t_bw:.3f} ms ({(2*rd+0.27)/t_bw*1e3:.0f} GB/s)")
