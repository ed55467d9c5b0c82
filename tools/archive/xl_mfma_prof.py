"""One forward + backward of extract_links on the matrix-core kernels, three times (for rocprofv3: tools/xl_mfma_prof.sh L TR)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd import _lib, decode_ops
dev = torch.device("cuda:0")
B, L, TR = 32, int(sys.argv[1]), int(sys.argv[2])
q0 = torch.randn(B, L, 8, 64, device=dev) * 0.5; k0 = torch.randn(B, L, 8, 64, device=dev) * 0.5
g0 = torch.log_softmax(torch.randn(B, L, 8, device=dev), -1)
olen = torch.full((B,), L, device=dev, dtype=torch.long)
w = torch.randn(B, L, TR, device=dev)
_lib.set_option("xl_mfma", 1)
for _ in range(3):
    q, k, lg = q0.clone().requires_grad_(), k0.clone().requires_grad_(), g0.clone().requires_grad_()
    links = decode_ops.extract_links_autograd(q, k, lg, olen, TR)
    links.backward(w)
torch.cuda.synchronize()
