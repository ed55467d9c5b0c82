#!/usr/bin/env python3
"""Is a 7- / 11-tap dilated Conv1d as a sum of Winograd F(2,3) tap groups as accurate as the direct form in fp32 arithmetic?  (CPU, numpy.)
direct: K channel-GEMMs per output position.  grouped: the K taps in groups of 3 (+ a remainder of 1 or 2 taps done directly); a 3-tap group
costs 4 channel-GEMMs per TWO outputs (2 per output instead of 3).  GEMM-taps per output: K=3: 2 (3), K=7: 5 (7), K=11: 8 (11) -> 15 / 21 of the
matrix-core work of a HiFi-GAN MRF block.  Errors are measured against an fp64 evaluation of the same convolution."""
import numpy as np

rng = np.random.default_rng(0)


def direct(x, w, d, dt):          # x [T + (K-1) d, Ci], w [K, Co, Ci] -> y [T, Co]
    K = w.shape[0]; T = x.shape[0] - (K - 1) * d
    y = np.zeros((T, w.shape[1]), dt)
    for k in range(K):
        y += x[k * d:k * d + T].astype(dt) @ w[k].astype(dt).T
    return y


def f23_group(x, w3, d, T, dt):   # 3 taps at offsets 0, d, 2d;  outputs in pairs (t, t + d) along each residue class of the dilated lattice
    # transformed weights (fp64 -> dt, as a weight packer would)
    g0, g1, g2 = (w3[i].astype(np.float64) for i in range(3))
    U = [g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2]
    U = [u.astype(dt) for u in U]
    y = np.zeros((T, w3.shape[1]), dt)
    xs = x.astype(dt)
    t = 0
    done = np.zeros(T, bool)
    for t0 in range(T):
        if done[t0]:
            continue
        t1 = t0 + d
        d0, d1, d2 = xs[t0], xs[t0 + d], xs[t0 + 2 * d]
        d3 = xs[t0 + 3 * d] if t0 + 3 * d < xs.shape[0] else np.zeros_like(d0)
        V = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
        M = [V[i] @ U[i].T for i in range(4)]
        y[t0] = M[0] + M[1] + M[2]; done[t0] = True
        if t1 < T and not done[t1]:
            y[t1] = M[1] - M[2] - M[3]; done[t1] = True
    return y


def grouped(x, w, d, dt):
    K = w.shape[0]; T = x.shape[0] - (K - 1) * d
    y = np.zeros((T, w.shape[1]), dt)
    k = 0
    while K - k >= 3:
        y += f23_group(x[k * d:], w[k:k + 3], d, T, dt); k += 3
    for kk in range(k, K):
        y += x[kk * d:kk * d + T].astype(dt) @ w[kk].astype(dt).T
    return y


for K, d in ((3, 1), (3, 5), (7, 1), (7, 3), (11, 1), (11, 5)):
    Ci = Co = 128; T = 192
    x = rng.standard_normal((T + (K - 1) * d, Ci)).astype(np.float32); x = np.where(x > 0, x, 0.1 * x)     # lrelu'd activations
    w = (rng.standard_normal((K, Co, Ci)) / np.sqrt(K * Ci)).astype(np.float32)
    ref = direct(x, w, d, np.float64)
    e_d = np.abs(direct(x, w, d, np.float32) - ref).max() / np.abs(ref).max()
    e_g = np.abs(grouped(x, w, d, np.float32) - ref).max() / np.abs(ref).max()
    assert np.abs(grouped(x, w, d, np.float64) - ref).max() < 1e-12
    print(f"K={K:2d} dilation {d}: direct fp32 {e_d:.2e}   F(2,3) tap groups fp32 {e_g:.2e}   ({e_g / e_d:.1f}x)")
