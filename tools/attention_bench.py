"""Time dsp_attention_split against torch's fp32 scaled_dot_product_attention at the acoustic stage's shapes (B=32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import decode_ops

dev = torch.device("cuda:0")
SHAPES = [("decoder self", 32, 354, 354, 8, 64), ("decoder cross", 32, 354, 177, 8, 64), ("tts encoder", 32, 64, 64, 2, 128),
          ("tts decoder", 32, 560, 560, 2, 128)]


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, B, N, M, H, dk in SHAPES:
    C = H * dk
    torch.manual_seed(0)
    q, k, v = torch.randn(B, N, C, device=dev), torch.randn(B, M, C, device=dev), torch.randn(B, M, C, device=dev)
    lens = torch.randint(int(0.45 * M), M + 1, (B,), device=dev); lens[0] = M
    pad = torch.arange(M, device=dev)[None] >= lens[:, None]
    mask = torch.zeros(B, 1, 1, M, device=dev).masked_fill(pad.view(B, 1, 1, M), float("-inf"))
    q4, k4, v4 = (t.view(B, -1, H, dk).transpose(1, 2) for t in (q, k, v))
    with torch.no_grad():
        t_sd = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, attn_mask=mask))
        t_us = timeit(lambda: decode_ops.attention(q, k, v, pad, H))
        ref = (torch.softmax((q4.double() @ k4.double().transpose(-1, -2)) * dk ** -0.5 + mask.double(), -1) @ v4.double()).transpose(1, 2).reshape(B, N, C)
        e_us = (decode_ops.attention(q, k, v, pad, H).double() - ref).abs().max().item()
        e_sd = (torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, attn_mask=mask).transpose(1, 2).reshape(B, N, C).double() - ref).abs().max().item()
    fl = 4.0 * B * H * N * M * dk
    print(f"{name:14s} B={B} N={N} M={M} H={H} dk={dk}: torch sdpa {t_sd:7.1f} us  split {t_us:7.1f} us  ({fl / t_us * 1e-6:6.1f} TFLOP/s nominal)  "
          f"max err split {e_us:.2e} sdpa {e_sd:.2e}")


# Conformer relative-position attention (r03's fp32-FMA kernel: 83.5 / 48.4 / 112.7 us at T = 177 / 120 / 200)
for B, T, H in [(32, 177, 4), (32, 120, 4), (32, 200, 4)]:
    C = H * 64
    torch.manual_seed(1)
    q, k, v = (torch.randn(B, T, C, device=dev) for _ in range(3))
    pos = torch.randn(1, 2 * T - 1, C, device=dev); bu, bv = torch.randn(H, 64, device=dev), torch.randn(H, 64, device=dev)
    lens = torch.randint(int(0.45 * T), T + 1, (B,), device=dev); lens[0] = T
    pad = torch.arange(T, device=dev)[None] >= lens[:, None]
    with torch.no_grad():
        t_us = timeit(lambda: decode_ops.relpos_attention(q, k, v, pos, bu, bv, pad, H))
    print(f"relpos B={B} T={T} H={H}: {t_us:7.1f} us")
