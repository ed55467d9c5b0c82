#!/usr/bin/env python3
"""Which cells of C1 on trained-model-like scores fail the dense DP's exactness guard, and why: torch emulation of the guard's two
references (shared per-8-column exponents vs the per-lane prefix reference of r03) on the exact alpha of the log-space kernel.
Output of the r03 run: profiles/r03e_dense_guard_emulation.txt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math
from daspeech_amd import custom_ops as ops, _lib
B, T, L = 4, 256, 2048; TR = L - 1
d = torch.device("cuda"); g = torch.Generator(device=d).manual_seed(0)
ol = torch.full((B,), L, device=d); tl = torch.full((B,), T, device=d)
i = torch.arange(L, device=d).view(1, L, 1); dd = torch.arange(TR, device=d).view(1, 1, TR); valid = (i + dd + 1) < L
raw = 4.0 * torch.randn(B, L, TR, device=d, generator=g)
k = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
j = torch.arange(L, device=d).view(1, 1, L).float(); c = (torch.arange(T, device=d).float() * (L - 1) / max(T - 1, 1)).view(1, T, 1)
m = torch.where((j - c).abs() < 6, -0.5 + 0.3 * torch.randn(B, T, L, device=d, generator=g), -20.0 + 3.0 * torch.randn(B, T, L, device=d, generator=g))
_lib.set_option("dp_path", 1)
loss, (alpha, beta) = ops.dag_loss_with_alpha_beta(m, k, ol, tl)
_lib.set_option("dp_path", 0)
LOG2E = 1.4426950408889634
a2 = (alpha * LOG2E).double(); m2 = (m * LOG2E).double()
NJ = L // 64
prev = a2[:, :-1]                                   # rows t-1
# group exponents of prev row: ceil(max over 8)
gmax = prev.view(B, T - 1, L // 8, 8).amax(-1)
gexp = torch.where(torch.isinf(gmax), torch.full_like(gmax, -1e30), gmax.ceil())
# prefix max over groups inside each 64-block
gexp_b = gexp.view(B, T - 1, NJ, 8)
ref = torch.cummax(gexp_b, dim=-1).values            # [B,T-1,NJ,8]
bexp = gexp_b.amax(-1)                               # block exponents
# ro: running reference over source blocks V < U with the jump rule
ro = torch.full((B, T - 1, NJ), -1e30, device=d, dtype=torch.float64)
cur = torch.full((B, T - 1), -1e30, device=d, dtype=torch.float64)
for U in range(NJ):
    ro[:, :, U] = cur
    sx = bexp[:, :, U]
    live = sx > -1e29; first = cur < -1e29
    jump = live & ~first & (sx > cur + 60)
    cur = torch.where(live & (first | jump), sx, cur)
rt = torch.maximum(ro.unsqueeze(-1), ref).view(B, T - 1, L // 8).repeat_interleave(8, dim=-1)      # [B,T-1,L]
P2 = a2[:, 1:] - m2[:, 1:] - rt                      # log2 P
tt = torch.arange(1, T, device=d).view(1, T - 1, 1); uu = torch.arange(L, device=d).view(1, 1, L)
fin = torch.isfinite(a2[:, 1:])
flag = fin & (P2 < -90)
print("alpha: finite cells", int(fin.sum()), "flagged", int(flag.sum()), "per row", float(flag.sum()) / (B * (T - 1)))
kdist = (uu - tt).expand_as(flag)[flag]
print("distance to the DP diagonal (u - t): quantiles", torch.quantile(kdist.double(), torch.tensor([0, .25, .5, .75, .9, .99, 1.0], device=d, dtype=torch.float64)).tolist())
bd = (uu.double() - c[:, 1:].double()).expand_as(flag)[flag]
print("distance to the band centre: quantiles", torch.quantile(bd, torch.tensor([0, .25, .5, .75, .9, .99, 1.0], device=d, dtype=torch.float64)).tolist())
print("log2 P of flagged: quantiles", torch.quantile(P2[flag], torch.tensor([0, .25, .5, .75, 1.0], device=d, dtype=torch.float64)).tolist())
# slope of alpha2 along a row left of the band
t0 = 128; row = a2[0, t0]; print("row", t0, "alpha2 at columns t0..t0+40:", [round(float(x), 1) for x in row[t0:t0 + 40:2]])
cc = int(c[0, t0, 0]); print("   around the band centre", cc, [round(float(x), 1) for x in row[cc - 40:cc + 24:4]])
for thr in (-90, -100, -110, -120):
    print("threshold", thr, "flagged", int((fin & (P2 < thr)).sum()))
# tier 2: per-lane reference = max(ro, ceil(exclusive prefix max of the previous row inside the own block))
pb = prev.view(B, T - 1, NJ, 64)
ex = torch.cat([torch.full_like(pb[..., :1], float("-inf")), torch.cummax(pb, dim=-1).values[..., :-1]], dim=-1)
ru = torch.maximum(ro.unsqueeze(-1), torch.where(torch.isinf(ex), torch.full_like(ex, -1e30), ex.ceil())).view(B, T - 1, L)
P3 = a2[:, 1:] - m2[:, 1:] - ru
flag2 = flag & (P3 < -90)
print("tier 2 (per-lane prefix reference): still flagged", int(flag2.sum()), "of", int(flag.sum()))
rows_any = flag.view(B, T - 1, NJ, 64).any(-1)
print("(row, block) pairs with a flagged lane:", int(rows_any.sum()), "of", B * (T - 1) * NJ, "| per block:", rows_any.sum((0, 1)).tolist())
if flag2.any():
    kd2 = (uu - tt).expand_as(flag)[flag2]; print("   remaining: distance to diagonal quantiles", torch.quantile(kd2.double(), torch.tensor([0, .5, 1.0], device=d, dtype=torch.float64)).tolist(), "column in block", ((uu % 64).expand_as(flag)[flag2]).unique().tolist()[:20])
