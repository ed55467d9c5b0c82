#!/usr/bin/env python3
"""DSP_DEBUG=prof accounting of strip5: s5_prof.py cpl w  (alpha only, C2)"""
import os, sys
os.environ["DSP_DEBUG"] = "prof"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
cpl, w = int(sys.argv[1]), int(sys.argv[2])
B, T, L, TR = 32, 512, 4096, 32
g = torch.Generator(device="cuda").manual_seed(0)
match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 9
raw = torch.randn(B, L, TR, device="cuda", generator=g)
ol = torch.full((B,), L, device="cuda"); tl = torch.full((B,), T, device="cuda")
i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
valid = (i + d + 1) < ol.view(B, 1, 1)
links = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf")).contiguous()
lib = _lib.load(); st = _lib.current_stream_handle()
alpha = torch.empty_like(match)
_lib.set_option("dp_path", 8); _lib.set_option("s5_cpl", cpl); _lib.set_option("s5_w", w)
for _ in range(3):
    assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), None, None, B, T, L, TR, None, 0, st) == 0
_lib.last_launch_status()
wd = lib.dsp_dag_debug_words()
ncw = w // cpl // 64
print(f"cpl={cpl} w={w}: cycles per row (ticket 0 workgroup, {T} rows)")
for wave in range(ncw + 3):
    v = [wd[7 + wave * 4 + k] / T for k in range(4)]
    role = "compute" if wave < ncw else ("loader", "fetch", "publish")[wave - ncw]
    if wave < ncw: print(f"  wave {wave:2d} {role:8s}: read-wait {v[0]:7.1f}  fma {v[1]:7.1f}  tail+stores {v[2]:7.1f}  barrier {v[3]:7.1f}  total {sum(v):7.1f}")
    else: print(f"  wave {wave:2d} {role:8s}: work {v[0]:7.1f}  barrier {v[3]:7.1f}  total {v[0] + v[3]:7.1f}")
