import torch, time
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(96600, 256, device=dev); w = torch.randn(1024, 256, device=dev) * 0.06; b = torch.randn(1024, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / n
f1 = lambda: torch.relu(torch.nn.functional.linear(x, w, b))
f2 = lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)
r1, r2 = f1(), f2()
print("max diff", (r1 - r2).abs().max().item())
print("linear+relu ms", timeit(f1), " addmm_activation ms", timeit(f2), " linear only", timeit(lambda: torch.nn.functional.linear(x, w, b)))
