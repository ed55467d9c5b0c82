"""Timing scan of the remaining operators over shapes (r05: looking for shapes that fall off the fast paths)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, decode_ops
dev = torch.device("cuda:0")
def timeit(f, n=5):
    for _ in range(2): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("== gather forward (in-place softmax), fp32 / fp16: ms (TB/s)")
for V in (4096, 5000, 6000, 8192, 8200, 10000, 12288, 16384, 20000, 32000):
    B, L, S = 16, 2048, 256
    row = []
    for dt in (torch.float32, torch.float16):
        x = torch.randn(B, L, V, device=dev, dtype=dt)
        idx = torch.randint(0, V, (B, 1, S), device=dev).expand(-1, L, -1)
        ms = timeit(lambda: ops.dag_logsoftmax_gather_inplace(x.clone().requires_grad_() * 1.0, idx)) - timeit(lambda: x.clone().requires_grad_() * 1.0)
        row.append(f"{ms:.3f} ({2.0 * B * L * V * x.element_size() / ms / 1e9:.2f})")
        del x
    print(f"V={V:6d}  " + "   ".join(row), flush=True)
print("== extract_links inference / train fwd+bwd, ms")
for (B, L, TR) in [(32, 400, 399), (32, 1024, 1023), (32, 1200, 1199), (32, 2048, 2047), (32, 4096, 4095), (32, 4096, 32), (32, 4096, 1024)]:
    H, CK = 8, 64
    q = torch.randn(B, L, H, CK, device=dev) * 0.3; k = torch.randn(B, L, H, CK, device=dev) * 0.3
    lg = torch.log_softmax(torch.randn(B, L, H, device=dev), -1); ol = torch.full((B,), L, device=dev)
    ms_i = timeit(lambda: decode_ops.extract_links(q, k, lg, ol, TR), 3)
    def tr():
        qa, ka, ga = q.clone().requires_grad_(), k.clone().requires_grad_(), lg.clone().requires_grad_()
        l = decode_ops.extract_links_autograd(qa, ka, ga, ol, TR)
        l.masked_fill(~torch.isfinite(l), 0.0).sum().backward()
    ms_t = timeit(tr, 2)
    print(f"B={B} L={L} TR={TR}: inference {ms_i:.2f}  train fwd+bwd {ms_t:.2f}   (links {B * L * TR * 4 / 1e9:.2f} GB)", flush=True)
    del q, k, lg
    torch.cuda.empty_cache()
print("== fp32-accurate vocoder, ms per call and us per mel frame")
from daspeech_amd.models import HiFiGANGenerator
voc = HiFiGANGenerator(conv_backend="hip").cuda().eval()
with torch.no_grad():
    for (B, T) in [(1, 100), (1, 600), (4, 329), (8, 329), (16, 329), (32, 329), (32, 100), (64, 329), (3, 777)]:
        mel = torch.randn(B, 80, T, device=dev)
        ms = timeit(lambda: voc(mel), 3)
        print(f"B={B} T={T}: {ms:.2f} ms  {ms * 1e3 / (B * T):.2f} us/frame", flush=True)
