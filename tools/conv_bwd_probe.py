import torch, time
dev = torch.device("cuda:0")
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
B, T = 32, 330
for (ci, co, k) in [(256, 1024, 9), (1024, 256, 9), (256, 256, 3), (80, 1024, 5)]:
    conv = torch.nn.Conv1d(ci, co, k, padding=k // 2).to(dev)
    x = torch.randn(B, ci, T, device=dev, requires_grad=True)
    def f32():
        y = conv(x); y.sum().backward()
    def bf16():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        y.float().sum().backward()
    def unfold_bf16():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xu = torch.nn.functional.unfold(x.unsqueeze(-1), (k, 1), padding=(k // 2, 0))      # [B, ci*k, T]
            y = torch.matmul(conv.weight.view(co, -1), xu) + conv.bias.view(1, -1, 1)
        y.float().sum().backward()
    print(f"conv {ci}->{co} k{k}: fp32 {bench(f32):.2f} ms | bf16 autocast {bench(bf16):.2f} ms | unfold+bf16 GEMM {bench(unfold_bf16):.2f} ms")
