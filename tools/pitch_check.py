import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from daspeech_amd import custom_ops as ops, _lib
from tools.dp_microbench import inputs, timeit
from util_inputs import make_dag_inputs
from oracle import dag_oracle as orc
for L in (4096, 4098, 4097, 4095):
    m, k, ol, tl = inputs(32, 512, L, 32)
    # match as the gather hands it over: rows pitched to a multiple of 4
    buf = torch.empty(32, 512, (L + 3) // 4 * 4, device="cuda"); mv = buf[:, :, :L]; mv.copy_(m)
    mg = mv.detach().requires_grad_(); kg = k.clone().requires_grad_()
    f = timeit(lambda: ops.dag_loss(mg, kg, ol, tl))
    loss = ops.dag_loss(mg, kg, ol, tl); go = torch.ones_like(loss)
    b = timeit(lambda: torch.autograd.grad(loss, [mg, kg], grad_outputs=go, retain_graph=True))
    with torch.no_grad():
        a = timeit(lambda: ops.dag_best_alignment(mv, k, ol, tl))
    print(f"L={L}: fwd {f[0]:.3f} ms  bwd {b[0]:.3f} ms  align {a[0]:.3f} ms", flush=True)
# long graph off the grid: L = 9001 > 8192 (alignment leaves the values-only strips), TR = 32
B, T, L, TR = 2, 300, 9001, 32
match, links, ol, tl = make_dag_inputs(77, B, T, L, TR)
dev = torch.device("cuda")
mm = torch.from_numpy(match).to(dev).requires_grad_(); kk = torch.from_numpy(links).to(dev).requires_grad_()
o = torch.from_numpy(ol).to(dev); t = torch.from_numpy(tl).to(dev)
loss = ops.dag_loss(mm, kk, o, t); gm, gk = torch.autograd.grad(loss.sum(), [mm, kk])
path = ops.dag_best_alignment(mm.detach(), kk.detach(), o, t)
b64 = orc.dag_beta(match, links, ol, tl, np.float64); a64 = orc.dag_alpha(match, links, ol, tl, np.float64)
np.testing.assert_allclose(loss.detach().cpu().numpy(), b64[:, 0, 0], rtol=3e-6, atol=2e-5 * T)
gm64, gl64 = orc.dag_grad(np.ones(B), a64, b64, match, links, ol, tl, np.float64)
np.testing.assert_allclose(gm.cpu().numpy(), gm64, rtol=2e-3, atol=1e-7); np.testing.assert_allclose(gk.cpu().numpy(), gl64, rtol=2e-3, atol=1e-7)
np.testing.assert_array_equal(path.cpu().numpy(), orc.dag_best_alignment(match, links, ol, tl, np.float32))
print("L = 9001 (off the grid, beyond the values-only alignment): loss, gradients, path match the oracle; status", _lib.last_launch_status())
