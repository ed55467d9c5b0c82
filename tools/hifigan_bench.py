import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.models import HiFiGANGenerator
B, T = int(sys.argv[1]), int(sys.argv[2])
g = HiFiGANGenerator().cuda().eval()
mel = torch.randn(B, 80, T, device="cuda")
flops = 0.614e9 * B * T
from daspeech_amd.hifigan_ops import HiFiGANHipRunner
for backend in (("torch",) if os.environ.get("HG_TORCH") else ()) + ("hip-chain", "hip", "hip-f32"):
    g.conv_backend = "torch" if backend == "torch" else "hip"
    g._hip_runner = HiFiGANHipRunner(g, fuse_units=(backend == "hip"), precision="fp32" if backend == "hip-f32" else "fp16") if backend != "torch" else None
    g._hip_runner_key = ("hip",) + tuple((p.data_ptr(), p._version) for p in g.parameters())
    with torch.no_grad():
        for _ in range(2): g(mel)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): g(mel)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{backend}: B={B} T={T}: {dt*1e3:.2f} ms  -> {flops/dt/1e12:.1f} TFLOP/s, {B/dt:.0f} utt/s, x{B*T*256/22050/dt:.0f} real time")
