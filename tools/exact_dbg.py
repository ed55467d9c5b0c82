"""Debug: the dense matrix-core DP against the f64 oracle on one test shape; where do the cells differ, and were they exact-redo cells?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from daspeech_amd import _lib
from test_gpu_dag_ops import make_dag_inputs, to_dev, orc, ops
B, T, L, TR = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (2, 70, 400, 399)
match, links, ol, tl = make_dag_inputs(11 + L, B, T, L, TR)
m, k, o, t = to_dev(match, links, ol, tl); m.requires_grad_()
a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
b64 = orc.dag_beta(match, links, ol, tl, np.float64)
for mt in (0, 1):
    _lib.set_option("dp_path", 9); _lib.set_option("dm_mt", mt)
    loss, (alpha, beta) = ops().dag_loss_with_alpha_beta(m, k, o, t)
    print("mt", mt, "status", _lib.last_launch_status(), "exact cells", _lib.last_fallback_count(), _lib.debug_fallback_cells()[:4])
    for name, x, r in (("alpha", alpha.cpu().numpy(), a64), ("beta", beta.cpu().numpy(), b64)):
        f = np.isfinite(r)
        bad = f & (np.abs(np.where(f, x - r, 0)) > 1e-3)
        idx = np.argwhere(bad)
        print(" ", name, "bad", len(idx))
        for (b, tt, j) in idx[:20]:
            print("   ", b, tt, j, x[b, tt, j], r[b, tt, j], "prev-row live:", np.argwhere(np.isfinite(r[b, tt - 1 if name == "alpha" else tt + 1])).ravel()[[0, -1]])
x = alpha.cpu().numpy()
b, tt, j = 0, 64, 64
print("ours", x[b, tt, j], "truth", a64[b, tt, j], "prev", x[b, tt - 1, j - 1], a64[b, tt - 1, j - 1], "match", match[b, tt, j])
need = x[b, tt, j] - x[b, tt - 1, j - 1] - match[b, tt, j]
print("weight used", need, "links[63,:6]", links[b, 63, :6], "links[62,:4]", links[b, 62, :4], "links[64,:4]", links[b, 64, :4])
hit = np.argwhere(np.abs(links[b] - need) < 2e-3)
print("candidates", hit[:10])
print("links[0,0,0]", links[0, 0, 0], "all candidates within 3e-4:", np.argwhere(np.abs(links[b] - need) < 3e-4).tolist(), "other sample:", np.argwhere(np.abs(links[1] - need) < 3e-4).tolist())
for (b, tt, j) in ((1, 62, 64),):
    terms = [(v, x[b, tt - 1, v] + links[b, v, j - v - 1]) for v in range(61, 64)]
    print("terms", terms, "ours - match", x[b, tt, j] - match[b, tt, j], "truth - match", a64[b, tt, j] - match[b, tt, j])
