#!/usr/bin/env python3
"""The DAG operators inside one training step (C5 shape, GLAT): shapes, time per call and exact-fallback cell counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import custom_ops as ops, _lib
from daspeech_amd.criterions import s2s_dag_fastspeech2_loss
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
dev = torch.device("cuda:0")
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).train()
b = make_s2st_batch(32, dev, seed=0)
import daspeech_amd.custom_ops.dag_loss as dl
orig_fwd = dl.dag_loss_with_alpha_beta if hasattr(dl, "dag_loss_with_alpha_beta") else None
calls = []
def wrap(name, fn):
    def inner(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        m, l = a[0], a[1]
        if name == "dag_loss_with_alpha_beta": wrap.saved = tuple(x.detach().clone() for x in a[:4])
        st = _lib.last_launch_status(); fb = _lib.last_fallback_count()
        calls.append((name, tuple(m.shape), tuple(l.shape), dt * 1e3, st, fb, float(torch.isneginf(m).float().mean())))
        if fb > 1000 and name == "dag_loss_with_alpha_beta":
            cells = _lib.debug_fallback_cells()
            loss_, (al, be) = out
            ol, tl = a[2], a[3]
            for (bd, tt, u, P) in cells[:10]:
                bb = (bd & 0xff) % m.shape[0]; isb = bool(bd & 0x100)
                Lb, Tb = int(ol[bb]), int(tl[bb]); L = m.shape[2]
                if not isb:
                    prev = al[bb, tt - 1]; live = torch.isfinite(prev).nonzero().flatten()
                    print(f"  alpha cell b={bb} t={tt} j={u} (L_b={Lb} T_b={Tb}) P={P:.3e}: prev row live columns: {live.numel()} first {live[:3].tolist()} last {live[-3:].tolist()}; match[t][j]={float(m[bb, tt, u]):.2f} result {float(al[bb, tt, u]):.2f}; "
                          f"links into j from first live: {[round(float(l[bb, i, u - i - 1]), 1) for i in live[:3].tolist() if i < u]}")
                else:
                    j = L - 1 - u; t = Tb - 1 - tt
                    nxt = be[bb, t + 1]; live = torch.isfinite(nxt).nonzero().flatten()
                    print(f"  beta cell b={bb} t={t} j={j} (L_b={Lb} T_b={Tb}) P={P:.3e}: next row live columns: {live.numel()} first {live[:3].tolist()} last {live[-3:].tolist()}; match={float(m[bb, t, j]):.2f} result {float(be[bb, t, j]):.2f}")
        return out
    return inner
import daspeech_amd.criterions as cr
for name in ("dag_loss", "dag_loss_with_alpha_beta", "dag_best_alignment"):
    if hasattr(cr, name): setattr(cr, name, wrap(name, getattr(cr, name)))
    if hasattr(ops, name): setattr(ops, name, wrap(name, getattr(ops, name)))
for it in range(2):
    calls.clear()
    with torch.autocast("cuda", dtype=torch.float16):
        loss, log = s2s_dag_fastspeech2_loss(model, b, glat_p="0.5:0.1@200k", update_num=100000)
    loss.backward(); model.zero_grad(set_to_none=True)
for c in calls: print(f"{c[0]}: match {c[1]} links {c[2]}: {c[3]:.3f} ms status {c[4]} exact-cells {c[5]} -inf emissions {100 * c[6]:.1f} %")

# the same call again, outside the step: as it is, on the row-sequential log-space kernels, and with the forced emissions removed
saved = getattr(wrap, "saved", None)

def timeit(label, *a):
    for _ in range(2): ops.dag_loss_with_alpha_beta(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.dag_loss_with_alpha_beta(*a)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms, exact cells {_lib.last_fallback_count()}")
if saved is None: saved = wrap.saved
m, l, ol, tl = saved
m = m.float().requires_grad_(); l = l.float()
timeit("as in the step (auto path)", m, l, ol, tl)
_lib.set_option("dp_path", 1); timeit("row-sequential log-space kernels", m, l, ol, tl); _lib.set_option("dp_path", 0)
m2 = torch.where(torch.isneginf(m), torch.full_like(m, -8.0), m).detach().requires_grad_()
timeit("forced emissions removed (-inf -> -8)", m2, l, ol, tl)
l2 = torch.where(torch.isneginf(l), l, l.clamp(min=-20.0))
timeit("transitions clamped to >= -20 nats", m, l2, ol, tl)
