"""HG_ABLATE sweeps of the layer-at-a-time HiFi-GAN chain (timing only; outputs are wrong by construction).
bits: 1 no staging loads, 2 no MFMA loop, 4 no epilogue."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.models import HiFiGANGenerator
from daspeech_amd.hifigan_ops import HiFiGANHipRunner
B, T = int(sys.argv[1]), int(sys.argv[2])
g = HiFiGANGenerator().cuda().eval()
g.conv_backend = "hip"; g._hip_runner = HiFiGANHipRunner(g, fuse_units=False)
mel = torch.randn(B, 80, T, device="cuda")
with torch.no_grad():
    for _ in range(2): g(mel)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g(mel)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"HG_ABLATE={os.environ.get('HG_ABLATE', '0')}: {dt*1e3:.2f} ms")
