#!/usr/bin/env python3
"""Randomised sweep of the HIP HiFi-GAN path: for random batch sizes and per-utterance lengths, a padded batch with `lengths` equals
each utterance vocoded alone (bit-exact), is zero past the valid region, and stays within the fp16-storage tolerance of the fp32
torch backend with the same weights.  usage: fuzz_vocoder.py [n_cases] [seed]   (GPU box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.models import HiFiGANGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
hip = HiFiGANGenerator(conv_backend="hip").cuda().eval()
ref = HiFiGANGenerator(conv_backend="torch").cuda().eval()
ref.load_state_dict(hip.state_dict())
bad = 0
for case in range(n):
    B = int(torch.randint(1, 9, (1,))); Tm = int(torch.randint(1, 420, (1,)))
    lens = torch.randint(0 if case % 5 == 4 else 1, Tm + 1, (B,)); lens[int(torch.randint(0, B, (1,)))] = Tm
    mel = torch.randn(B, 80, Tm, device="cuda") * 1.5
    tag = f"case {case}: B={B} frames={Tm} lens={lens.tolist()}"
    try:
        with torch.no_grad():
            out = hip(mel, lens.cuda())
            assert out.shape == (B, 1, Tm * 256) and torch.isfinite(out).all(), "shape / finite"
            for b in range(B):
                k = int(lens[b])
                assert (out[b, 0, k * 256:] == 0).all(), f"utterance {b}: non-zero past its length"
                if k == 0: continue
                alone = hip(mel[b:b + 1, :, :k].contiguous())[0, 0]
                assert torch.equal(alone, out[b, 0, : k * 256]), f"utterance {b}: padded batch != alone (max diff {(alone - out[b, 0, :k * 256]).abs().max():.3e})"
                r = ref(mel[b:b + 1, :, :k].contiguous())[0, 0]
                err = (alone - r).abs()
                assert float(err.max()) < 2e-2 and float(err.mean()) < 2e-3, f"utterance {b}: vs fp32 torch max {float(err.max()):.3e} mean {float(err.mean()):.3e}"
    except Exception as e:   # noqa
        bad += 1; print("FAIL", tag, "->", str(e).splitlines()[0][:300] if str(e) else repr(e))
print(f"{n} cases, {bad} failures")
