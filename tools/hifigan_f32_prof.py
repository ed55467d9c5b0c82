"""One fp32-accurate (fused-unit) vocoder call shape under rocprofv3: python tools/hifigan_f32_prof.py B T [reps]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.models import HiFiGANGenerator
from daspeech_amd.hifigan_ops import HiFiGANHipRunner
B, T = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = HiFiGANGenerator().cuda().eval()
r = HiFiGANHipRunner(g, fuse_units=True, precision="fp32")
mel = torch.randn(B, 80, T, device="cuda")
with torch.no_grad():
    for _ in range(2): r(mel)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r(mel)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"hip-f32 fused: B={B} T={T}: {dt*1e3:.2f} ms -> {0.614e9*B*T/dt/1e12:.1f} TFLOP/s")
