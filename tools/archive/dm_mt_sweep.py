"""Dense-window DP forward at C2 / TR = L-1 for the chunk heights dm_mt = 2 (32 rows, product) and 4 (64 rows): time and agreement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from daspeech_amd import _lib, custom_ops as ops
B, L, T, V = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 4096, 512, 64)
dev = torch.device("cuda:0")
logits, links, ol, tl, tgt = bench.make_dag_inputs(torch, dev, B, L, T, V, L - 1, 77)
match = torch.log_softmax(torch.randn(B, T, L, device=dev) * 2, -1).contiguous()
del logits
ref = None
for mt in ([int(a) for a in sys.argv[5:]] or [2, 4, 2, 4]):
    _lib.set_option("dm_mt", mt)
    k = links.detach().requires_grad_()
    for _ in range(2):
        loss, (a, b) = ops.dag_loss_with_alpha_beta(match, k, ol, tl)
    st = _lib.last_launch_status()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        loss, (a, b) = ops.dag_loss_with_alpha_beta(match, k, ol, tl)
    e1.record(); torch.cuda.synchronize()
    msg = ""
    if ref is None: ref = (loss.clone(), a.clone(), b.clone())
    else:
        fa = torch.isfinite(ref[1]); fb = torch.isfinite(ref[2])
        msg = (f" | vs first: loss {float((loss - ref[0]).abs().max()):.2e} alpha {float((a[fa] - ref[1][fa]).abs().max()):.2e} beta {float((b[fb] - ref[2][fb]).abs().max()):.2e}"
               f" inf-pattern {bool((torch.isfinite(a) == fa).all() and (torch.isfinite(b) == fb).all())}")
    print(f"dm_mt={mt}: fwd {e0.elapsed_time(e1) / 3:.2f} ms status {st} finite {int(torch.isfinite(loss).sum())}/{B}{msg}", flush=True)
_lib.set_option("dm_mt", 0)
