"""extract_links forward / forward + backward: the matrix-core kernels (csrc/extract_links_mfma.hip) against the fp32-FMA kernels
(csrc/extract_links.hip), ms per call on one MI355X.  python tools/xl_mfma_time.py [quick]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib, decode_ops

dev = torch.device("cuda:0")


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(32, 400, 399), (32, 400, 64), (32, 256, 255), (32, 128, 127), (32, 1024, 1023), (32, 2048, 2047), (32, 4096, 4095), (32, 4096, 1024), (32, 4096, 64),
          (8, 400, 399), (8, 1024, 1023), (16, 400, 399), (16, 640, 639), (4, 2048, 2047), (1, 4096, 4095), (64, 400, 399), (2, 1000, 999), (4, 600, 599)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:5]
print("B L TR: inference ms fp32-FMA -> matrix-core | train fwd+bwd ms fp32-FMA -> matrix-core")
for B, L, TR in shapes:
    torch.manual_seed(0)
    q0 = torch.randn(B, L, 8, 64, device=dev) * 0.5; k0 = torch.randn(B, L, 8, 64, device=dev) * 0.5
    g0 = torch.log_softmax(torch.randn(B, L, 8, device=dev), -1)
    olen = torch.full((B,), L, device=dev, dtype=torch.long)
    w = torch.randn(B, L, TR, device=dev)
    res = []
    for mode in (0, 1):
        _lib.set_option("xl_mfma", mode)
        n = 3 if (L >= 2048 and mode == 0) else 10
        with torch.no_grad():
            ti = timed(lambda: decode_ops.extract_links(q0, k0, g0, olen, TR), n)

        def step():
            q, k, lg = q0.clone().requires_grad_(), k0.clone().requires_grad_(), g0.clone().requires_grad_()
            links = decode_ops.extract_links_autograd(q, k, lg, olen, TR)
            links.backward(w)
        tt = timed(step, n)
        res.append((ti, tt))
    _lib.set_option("xl_mfma", -1)
    print(f"B={B} L={L} TR={TR}: inference {res[0][0]:.3f} -> {res[1][0]:.3f}   train fwd+bwd {res[0][1]:.3f} -> {res[1][1]:.3f}", flush=True)
