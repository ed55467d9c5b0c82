#!/usr/bin/env python3
"""dag_loss backward on windows 33 .. 128: the exp-space kernel with a plane of workgroups per 32 transitions (k5_path 3, family 6; 0 = auto) against the tiled
log-space kernel (k5_path 1) and the dense block products (k5_path 2, TR > 64) — HIP-event times through the C ABI, gradients compared,
utterance 0 against the fp64 oracle with --oracle.  GPU box only.
usage: bwd_wide_bench.py [B T L TR] [--oracle]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
import daspeech_amd.custom_ops  # noqa: F401
dl = sys.modules['daspeech_amd.custom_ops.dag_loss']
from tools.dp_microbench import inputs


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, T, L, TR = [int(v) for v in args[:4]] if len(args) >= 4 else (32, 512, 4096, 64)
    m, k, ol, tl = inputs(B, T, L, TR)
    mm, kk, ol, tl, alpha, beta, loss, (ldm, lda) = dl._dag_forward(m, k, ol, tl, True)
    go = -(1.0 / tl.float()) / B
    lib = _lib.load()
    st = _lib.current_stream_handle()
    out, fam = {}, {}
    for path in (1, 2, 3, 0):
        _lib.set_option("k5_path", path)
        gm = torch.full((B, T, ldm), float("nan"), device=mm.device); gl = torch.full_like(kk, float("nan"))

        def run():
            rc = lib.dsp_dag_loss_bwd_ld(_lib.ptr(go), _lib.ptr(alpha), _lib.ptr(beta), lda, _lib.ptr(mm), ldm, _lib.ptr(kk), _lib.ptr(ol), _lib.ptr(tl),
                                         _lib.ptr(gm), ldm, _lib.ptr(gl), B, T, L, TR, None, 0, st)
            _lib.check(rc, "bwd")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        import ctypes
        diag = (ctypes.c_uint * 4)(); lib.dsp_dag_debug_k5(diag); fam[path] = diag[3]
        ts = []
        for _ in range(10):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ts.sort()
        out[path] = (gm, gl)
        print(f"k5_path {path} (family {fam[path]}): min {ts[0]*1e3:.1f} us  median {ts[len(ts)//2]*1e3:.1f} us", flush=True)
    _lib.set_option("k5_path", 0)
    for path in (2, 3, 0):
        gm1, gl1 = out[1]; gm, gl = out[path]
        sc_m = gm1.abs().max().item(); sc_l = gl1.abs().max().item()
        print(f"k5_path {path} vs tiled: grad_match max diff {(gm - gm1).abs().max().item():.3g} (scale {sc_m:.3g}), "
              f"grad_links max diff {(gl - gl1).abs().max().item():.3g} (scale {sc_l:.3g}), "
              f"max rel on cells > 1e-6 of scale {(((gl - gl1).abs() / gl1.abs().clamp_min(1e-30))[gl1.abs() > 1e-6 * sc_l]).max().item() if sc_l > 0 else 0.0:.3g}, "
              f"nan {bool(torch.isnan(gm).any())} {bool(torch.isnan(gl).any())}")
    if "--oracle" in sys.argv:
        import numpy as np
        from oracle import dag_oracle as orc
        mm1, kk1 = m[:1].cpu().numpy().astype(np.float64), k[:1].cpu().numpy().astype(np.float64)
        o1, t1 = ol[:1].cpu().numpy(), tl[:1].cpu().numpy()
        a64, b64 = orc.dag_alpha(mm1, kk1, o1, t1, np.float64), orc.dag_beta(mm1, kk1, o1, t1, np.float64)
        gm64, gl64 = orc.dag_grad(go[:1].cpu().numpy().astype(np.float64), a64, b64, mm1, kk1, o1, t1, np.float64)
        for path in (0, 1, 2, 3):
            np.testing.assert_allclose(out[path][0][0].cpu().numpy()[:, :L], gm64[0], rtol=3e-3, atol=1e-9)
            np.testing.assert_allclose(out[path][1][0].cpu().numpy(), gl64[0], rtol=3e-3, atol=1e-9)
        print("utterance 0 of every family matches the fp64 oracle")


if __name__ == "__main__":
    main()
