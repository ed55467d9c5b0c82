"""r05 experiment driver: the fp32-accurate vocoder call under the timing-only kernel variants of dsp_hifigan_set_experiment
(bit0 Winograd cost emulation, bit1 merged accumulators, bit2 no residual re-read, bit8 one workgroup per CU).  Wrong results for != 0."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib
from daspeech_amd.models import HiFiGANGenerator
from daspeech_amd.hifigan_ops import HiFiGANHipRunner
B, T = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
exps = [int(a) for a in sys.argv[4:]] or [0, 1, 2, 3, 4, 7, 256, 259, 0]
g = HiFiGANGenerator().cuda().eval()
r = HiFiGANHipRunner(g, fuse_units=True, precision="fp32")
mel = torch.randn(B, 80, T, device="cuda")
lib = _lib.load()
f = lib.dsp_hifigan_set_experiment; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_int]
with torch.no_grad():
    for e in exps:
        f(e)
        for _ in range(3): r(mel)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): r(mel)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(f"exp={e:4d}: B={B} T={T}: {dt*1e3:.2f} ms", flush=True)
f(0)
