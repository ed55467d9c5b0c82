import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
lib = _lib.load(); st = _lib.current_stream_handle()
for (B, T, L) in [(4, 256, 2048), (16, 150, 1024), (32, 512, 4096)]:
    TR = L - 1
    g = torch.Generator(device="cuda").manual_seed(0)
    match = torch.randn(B, T, L, device="cuda", generator=g) * 2 - 6
    ol = torch.full((B,), L, device="cuda"); tl = torch.full((B,), T, device="cuda")
    links = torch.empty(B, L, TR, device="cuda")
    i = torch.arange(L, device="cuda").view(1, L, 1); d = torch.arange(TR, device="cuda").view(1, 1, TR)
    for b0 in range(0, B, 2):
        raw = torch.randn(min(2, B - b0), L, TR, device="cuda", generator=g)
        valid = (i + d + 1) < L
        links[b0:b0 + 2] = torch.log_softmax(raw.masked_fill(~valid, float("-inf")).masked_fill(~valid.any(-1, keepdim=True), 0.0), -1).masked_fill(~valid, float("-inf"))
        del raw, valid
    res = {}
    for name, mt, dep in (("mt2 two WGs/CU (default)", 2, 1), ("mt1", 1, 1), ("mt2 one WG/CU", 2, 9)):
        alpha = torch.empty_like(match); beta = torch.empty_like(match)
        _lib.set_option("dp_path", 9); _lib.set_option("dm_mt", mt); _lib.set_option("dm_depth", dep)
        def run(): assert lib.dsp_dag_loss_fwd(_lib.ptr(match), _lib.ptr(links), _lib.ptr(ol), _lib.ptr(tl), _lib.ptr(alpha), _lib.ptr(beta), None, B, T, L, TR, None, 0, st) == 0
        run(); run(); torch.cuda.synchronize()
        n = 3 if L >= 4096 else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / n, alpha, beta)
    _lib.set_option("dp_path", 0); _lib.set_option("dm_mt", 0); _lib.set_option("dm_depth", 0)
    same = torch.equal(res["mt2 two WGs/CU (default)"][1], res["mt2 one WG/CU"][1]) and torch.equal(res["mt2 two WGs/CU (default)"][2], res["mt2 one WG/CU"][2])
    print(f"B={B} T={T} L={L}: " + " | ".join(f"{k}: {v[0]:.3f} ms" for k, v in res.items()) + f" | two builds bitwise equal: {same}", flush=True)
    del links, match, res
