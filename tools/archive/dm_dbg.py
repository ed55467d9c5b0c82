import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
if os.environ.get("DSP_SO"): _lib.SO_PATH = os.path.abspath(os.environ["DSP_SO"])
B, T, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 256, 2048); TR = L - 1
g = torch.Generator(device="cuda").manual_seed(0)
match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6
ol = torch.full((B,), L, device="cuda"); tl = torch.full((B,), T, device="cuda")
links = torch.empty(B, L, TR, device="cuda")
i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
for b0 in range(0, B, 2):
    raw = torch.randn(min(2, B - b0), L, TR, device="cuda", generator=g)
    valid = (i + d + 1) < ol[b0:b0 + 2].view(-1, 1, 1)
    links[b0:b0 + 2] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
    del raw, valid
lib = _lib.load(); st = _lib.current_stream_handle()
alpha = torch.empty_like(match)
_lib.set_option("dp_path", 9); _lib.set_option("dm_depth", 1)
assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), None, None, B, T, L, TR, None, 0, st) == 0
import time
for mt in (1, 9):
    _lib.set_option("dm_depth", mt)
    beta = torch.empty_like(match)
    for _ in range(2):
        assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _lib.ptr(beta), None, B, T, L, TR, None, 0, st) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _lib.ptr(beta), None, B, T, L, TR, None, 0, st) == 0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    _lib.last_launch_status(); w = lib.dsp_dag_debug_words()
    print(f"depth={mt}: wall {dt*1e3:.3f} ms; last block of sd 0: ready-wait {w[39]*16/2.4e3:.1f} us, gemm {w[40]*16/2.4e3:.1f} us, diag {w[41]*16/2.4e3:.1f} us (at 2.4 GHz), chunks {w[42]}")
_lib.set_option("dm_depth", 0)
print("status", _lib.last_launch_status(), "exact cells", _lib.last_fallback_count())
print(_lib.debug_fallback_cells())
a = alpha[0].cpu()
for (b, t, j, P) in _lib.debug_fallback_cells()[:6]:
    b &= 0xff
    print("cell", b, t, j, "P", P, "prev row around diag:", alpha[b, t - 1, max(0, j - 12): j + 4].cpu().tolist())
