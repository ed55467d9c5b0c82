#!/usr/bin/env python3
"""HIP vocoder — "hip" (fp32, split operands) and "hip_fp16" (fp16 storage, fp32 accumulation on MFMA) — vs the waveform the REFERENCE hifi-gan Generator produced for the same seeded V1
weights (tests/golden/hifigan_v1_seeded.npz): the measured errors behind the tolerance of tests/test_tts_golden.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_tts_golden import _hifigan_v1_from_seed
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hifigan_v1_seeded.npz"), allow_pickle=False))
mel, lens = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["lens"]).cuda()
for backend in ("hip", "hip_fp16", "torch"):
    m = _hifigan_v1_from_seed(g, backend, "cuda")
    with torch.no_grad():
        batch = m(mel, lengths=lens) if backend != "torch" else None
        for b, n in enumerate(g["lens"]):
            single = m(mel[b:b + 1, :, :n].contiguous())[0, 0]
            ref = g[f"wav{b}"]
            err = np.abs(single.cpu().numpy() - ref)
            line = f"{backend:8s} utterance {b} ({int(n)} frames, {len(ref)} samples, reference rms {np.sqrt((ref ** 2).mean()):.3f}): max |err| {err.max():.3e}  mean |err| {err.mean():.3e}"
            if batch is not None: line += f"  | padded batch with lengths == alone: {bool(torch.equal(batch[b, 0, : n * 256], single))}"
            print(line)
