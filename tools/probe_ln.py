import torch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / n
for rows, C in ((294400, 96), (73600, 192), (18400, 384), (4600, 768), (96600, 256)):
    x = torch.randn(rows, C, device=dev); y = torch.randn(rows, C, device=dev)
    w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    t_ln = timeit(lambda: torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5))
    t_add = timeit(lambda: x + y)
    t_gelu = timeit(lambda: torch.nn.functional.gelu(x))
    byts = rows * C * 4
    print(f"rows {rows} C {C}: LN {t_ln*1e3:.1f} us ({2*byts/t_ln/1e9:.0f} GB/s)  add {t_add*1e3:.1f} us ({3*byts/t_add/1e9:.0f} GB/s)  gelu {t_gelu*1e3:.1f} us ({2*byts/t_gelu/1e9:.0f} GB/s)")
