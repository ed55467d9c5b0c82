"""Power-model probe: the fp32-accurate vocoder call on random data vs all-zero weights and activations (same instruction stream, no toggling
operands).  If the call is power-limited the zero run is faster (higher clock); if it is issue / latency limited the two agree."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.models import HiFiGANGenerator
from daspeech_amd.hifigan_ops import HiFiGANHipRunner
B, T, reps = 32, 329, 20
def run(tag, zero):
    g = HiFiGANGenerator().cuda().eval()
    if zero:
        with torch.no_grad():
            for p in g.parameters(): p.zero_()
    r = HiFiGANHipRunner(g, fuse_units=True, precision="fp32")
    mel = torch.zeros(B, 80, T, device="cuda") if zero else torch.randn(B, 80, T, device="cuda")
    with torch.no_grad():
        for _ in range(3): r(mel)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): r(mel)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{tag}: {dt*1e3:.2f} ms", flush=True)
run("random", False); run("zeros", True); run("random", False)
