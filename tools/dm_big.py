import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
B, T, L = 32, 512, 4096; TR = L - 1
g = torch.Generator(device="cuda").manual_seed(0)
match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6
ol = torch.full((B,), L, device="cuda") - torch.arange(B, device="cuda") % 5; tl = torch.full((B,), T, device="cuda") - torch.arange(B, device="cuda") % 4
links = torch.empty(B, L, TR, device="cuda")
for b0 in range(0, B, 2):
    raw = torch.randn(2, L, TR, device="cuda", generator=g)
    i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
    valid = (i + d + 1) < ol[b0:b0 + 2].view(-1, 1, 1)
    links[b0:b0 + 2] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
    del raw, valid
lib = _lib.load(); st = _lib.current_stream_handle()
alpha = torch.empty_like(match); beta = torch.empty_like(match)
def run():
    assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _lib.ptr(beta), None, B, T, L, TR, None, 0, st) == 0
ref = None
for name, path, mt in (("mfma mt=2", 9, 2), ("mfma mt=1", 9, 1)):
    _lib.set_option("dp_path", path); _lib.set_option("dm_mt", mt)
    run(); torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"C2 TR=4095 {name}: alpha||beta {dt * 1e3:.2f} ms, status {_lib.last_launch_status()} exact-cells {_lib.last_fallback_count()}", flush=True)
    if ref is None: ref = (alpha.clone(), beta.clone())
    else:
        for w, (x, y) in enumerate(((alpha, ref[0]), (beta, ref[1]))):
            f = torch.isfinite(y)
            print(f"   vs mt=2 {'alpha' if w == 0 else 'beta'}: inf pattern equal {bool(torch.equal(torch.isneginf(x), torch.isneginf(y)))}, max diff {float((x[f] - y[f]).abs().max()):.3e}", flush=True)
