"""Dense-window dag_best_alignment at chunk heights dx_mt = 1 (16 rows) and 2 (32 rows): time, bit-identical paths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from daspeech_amd import _lib, custom_ops as ops
B, L, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 4096, 512)
dev = torch.device("cuda:0")
_, links, ol, tl, _ = bench.make_dag_inputs(torch, dev, B, L, T, 64, L - 1, 77)
match = torch.log_softmax(torch.randn(B, T, L, device=dev) * 2, -1).contiguous()
ref = None
for mt in ([int(a) for a in sys.argv[4:]] or [1, 2, 1, 2, 0]):
    _lib.set_option("dx_mt", mt)
    for _ in range(2): path = ops.dag_best_alignment(match, links, ol, tl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): path = ops.dag_best_alignment(match, links, ol, tl)
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = path.clone()
    print(f"dx_mt={mt}: alignment {e0.elapsed_time(e1) / 3:.2f} ms, path == first: {bool((path == ref).all())}, on-path vertices {int((path >= 0).sum())}", flush=True)
_lib.set_option("dx_mt", 0)
