#!/usr/bin/env python3
"""dag_loss backward at one shape: the fused launch (k5_fuse 1 / 2) against the two-launch form (k5_fuse 3) — HIP-event times through the C
ABI, bitwise comparison of both gradients, and utterance 0 against the fp64 oracle.  GPU box only.
usage: bwd_bench.py [B T L TR] [--oracle]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
import daspeech_amd.custom_ops  # noqa: F401
dl = sys.modules['daspeech_amd.custom_ops.dag_loss']
from tools.dp_microbench import inputs


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, T, L, TR = [int(v) for v in args[:4]] if len(args) >= 4 else (32, 512, 4096, 32)
    m, k, ol, tl = inputs(B, T, L, TR)
    mm, kk, ol, tl, alpha, beta, loss, (ldm, lda) = dl._dag_forward(m, k, ol, tl, True)
    go = -(1.0 / tl.float()) / B
    lib = _lib.load()
    st = _lib.current_stream_handle()
    out = {}
    for fuse in (3, 1, 2):
        _lib.set_option("k5_fuse", fuse)
        gm = dl._pitched_empty(B, T, L, mm.device, fill=float("nan")); gl = torch.full_like(kk, float("nan"))

        def run():
            rc = lib.dsp_dag_loss_bwd_ld(_lib.ptr(go), _lib.ptr(alpha), _lib.ptr(beta), lda, _lib.ptr(mm), ldm, _lib.ptr(kk), _lib.ptr(ol), _lib.ptr(tl),
                                         _lib.ptr(gm), dl._round4(L), _lib.ptr(gl), B, T, L, TR, None, 0, st)
            _lib.check(rc, "bwd")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ts.sort()
        out[fuse] = (gm, gl)
        nbytes = (4 * B * T * L + 2 * B * L * TR) * 4
        print(f"k5_fuse {fuse}: min {ts[0]*1e3:.1f} us  median {ts[len(ts)//2]*1e3:.1f} us  -> {nbytes/ts[len(ts)//2]/1e6:.0f} GB/s algorithmic "
              f"({nbytes/ts[len(ts)//2]/1e6/8000:.3f} of HBM)", flush=True)
    _lib.set_option("k5_fuse", 0)
    for fuse in (1, 2):
        same_m = torch.equal(out[fuse][0], out[3][0]); same_l = torch.equal(out[fuse][1], out[3][1])
        dm = (out[fuse][0] - out[3][0]).abs().max().item(); dk = (out[fuse][1] - out[3][1]).abs().max().item()
        print(f"k5_fuse {fuse} vs two launches: grad_match bitwise {same_m} (max diff {dm:.3g}), grad_links bitwise {same_l} (max diff {dk:.3g}), "
              f"nan {bool(torch.isnan(out[fuse][0]).any())} {bool(torch.isnan(out[fuse][1]).any())}")
    if "--oracle" in sys.argv:
        import numpy as np
        from oracle import dag_oracle as orc
        mm1, kk1 = m[:1].cpu().numpy().astype(np.float64), k[:1].cpu().numpy().astype(np.float64)
        o1, t1 = ol[:1].cpu().numpy(), tl[:1].cpu().numpy()
        a64, b64 = orc.dag_alpha(mm1, kk1, o1, t1, np.float64), orc.dag_beta(mm1, kk1, o1, t1, np.float64)
        gm64, gl64 = orc.dag_grad(go[:1].cpu().numpy().astype(np.float64), a64, b64, mm1, kk1, o1, t1, np.float64)
        for fuse in (1, 2, 3):
            np.testing.assert_allclose(out[fuse][0][0].cpu().numpy(), gm64[0], rtol=3e-3, atol=1e-9)
            np.testing.assert_allclose(out[fuse][1][0].cpu().numpy(), gl64[0], rtol=3e-3, atol=1e-9)
        print("utterance 0 of every variant matches the fp64 oracle")


if __name__ == "__main__":
    main()
