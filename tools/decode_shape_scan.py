"""Timing scan of the graph-decode strategies and the posterior-features op over shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from daspeech_amd import decode_ops
dev = torch.device("cuda:0")
def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("graph decode, ms: lookahead / greedy / viterbi / jointviterbi")
for (B, L, V, D) in [(32, 400, 512, 512), (64, 400, 6000, 512), (32, 1024, 512, 512), (32, 1024, 8192, 512), (8, 1023, 512, 512)]:
    TR = L - 1
    _, links, ol, tl, _ = bench.make_dag_inputs(torch, dev, B, L, 8, 16, TR, 3)
    logits = torch.randn(B, L, V, device=dev); feats = torch.randn(B, L, D, device=dev)
    row = []
    for st in ("lookahead", "greedy"):
        row.append(timeit(lambda: decode_ops.graph_decode(logits, links, feats, ol, 1, 1.0, st)))
    for joint in (False, True):
        row.append(timeit(lambda: decode_ops.viterbi_decode(logits, links, feats, ol, 1, 1.0, 1.0, joint, 0.5), 3))
    print(f"B={B} L={L} V={V}: " + " / ".join(f"{r:.2f}" for r in row), flush=True)
    del logits, feats, links
print("posterior_features fwd, ms (and GB/s over alpha+beta)")
for (B, T, L, D) in [(32, 60, 400, 512), (32, 128, 1024, 512), (32, 512, 4096, 512), (32, 60, 398, 512)]:
    a = torch.randn(B, T, L, device=dev); b = torch.randn(B, T, L, device=dev); f = torch.randn(B, L, D, device=dev)
    ms = timeit(lambda: decode_ops.posterior_features(a, b, f))
    print(f"B={B} T={T} L={L}: {ms:.3f} ms  ({2 * B * T * L * 4 / ms / 1e6:.0f} GB/s, {2.0 * B * T * L * D / ms / 1e9:.1f} TFLOP/s)", flush=True)
