"""Diagnose exp-space DP paths on peaked scores (GPU box only): usage debug_peaked.py slope [paths]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from util_inputs import make_dag_inputs
from oracle import dag_oracle as orc
from daspeech_amd import _lib, custom_ops as ops

slope = float(sys.argv[1]); paths = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 5]
B, T, L, TR = 2, 40, 1024, 32
match, links, ol, tl = make_dag_inputs(123, B, T, L, TR, ragged=True)
jj = np.arange(L, dtype=np.float32)[None, None, :]
centre = (np.arange(T, dtype=np.float32) * (L - 1) / (T - 1))[None, :, None]
match = (match * 0.1 - slope * np.abs(jj - centre)).astype(np.float32)
a64 = orc.dag_alpha(match, links, ol, tl, np.float64); b64 = orc.dag_beta(match, links, ol, tl, np.float64)
a32 = orc.dag_alpha(match, links, ol, tl, np.float32)
print("oracle f32 vs f64 neginf mismatch:", int((np.isneginf(a32) != np.isneginf(a64)).sum()), "min finite a64", a64[np.isfinite(a64)].min())
dev = torch.device("cuda")
m = torch.from_numpy(match).to(dev).requires_grad_(); k = torch.from_numpy(links).to(dev); o = torch.from_numpy(ol).to(dev); t = torch.from_numpy(tl).to(dev)
for path in paths:
    _lib.set_option("dp_path", path)
    loss, (alpha, beta) = ops.dag_loss_with_alpha_beta(m, k, o, t)
    st = _lib.last_launch_status()
    w = _lib.load().dsp_dag_debug_words()
    a = alpha.cpu().numpy(); b = beta.cpu().numpy()
    if os.environ.get("DSP_DEBUG") == "medium":
        import struct
        n = min(14, int(w[2]))
        print("   first medium cells (sample|0x100=beta, t, vertex, S):", [(int(w[7 + 4 * i]), int(w[8 + 4 * i]), int(w[9 + 4 * i]), struct.unpack('f', struct.pack('I', w[10 + 4 * i]))[0]) for i in range(n)])
    for nm, x, r in (("alpha", a, a64), ("beta", b, b64)):
        mis = np.argwhere(np.isneginf(x) != np.isneginf(r))
        fin = np.isfinite(r) & np.isfinite(x)
        err = np.abs(x[fin] - r[fin]); rel = err / np.maximum(1.0, np.abs(r[fin]))
        print(f"path {path} {nm}: status {st} exact {w[1]} medium {w[2]} neginf mismatches {len(mis)} first {mis[:6].tolist()} max abs err {err.max():.4g} max rel {rel.max():.3g}")
        for q in mis[:4]:
            print("    cell", q.tolist(), "got", x[tuple(q)], "want", r[tuple(q)], "row neighbours want", r[q[0], q[1], max(0, q[2]-2):q[2]+3])
_lib.set_option("dp_path", 0)
