"""DSP_DEBUG=prof accounting of the dense-window forward's last column block (sample 0, alpha): cycles waiting for readiness, in the products, in
the diagonal recurrence — at C2 / TR = L - 1 (or B T L on the command line)."""
import sys, os
os.environ["DSP_DEBUG"] = "prof"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import bench
from daspeech_amd import _lib, custom_ops as ops
B, T, L = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 512, 4096)
dev = torch.device("cuda:0")
_, links, ol, tl, _ = bench.make_dag_inputs(torch, dev, B, L, T, 64, L - 1, 77)
match = torch.log_softmax(torch.randn(B, T, L, device=dev) * 2, -1).contiguous()
lib = _lib.load()
for mt in ([int(a) for a in sys.argv[4:]] or [0]):
    _lib.set_option("dm_mt", mt)
    k = links.detach().requires_grad_()
    for _ in range(2): ops.dag_loss_with_alpha_beta(match, k, ol, tl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.dag_loss_with_alpha_beta(match, k, ol, tl)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    _lib.last_launch_status(); w = lib.dsp_dag_debug_words()
    print(f"dm_mt={mt}: wall {dt*1e3:.2f} ms; last block of sd 0: ready-wait {w[39]*16/100:.1f} us, gemm {w[40]*16/100:.1f} us, diag {w[41]*16/100:.1f} us "
          f"(s_memtime at 100 MHz), chunks {w[42]}", flush=True)
_lib.set_option("dm_mt", 0)
