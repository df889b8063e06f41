import torch, time
for n in [29491200, 58982400, 117964800]:
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(50): y.copy_(x)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 50 * 1e-3
    print(f"copy {n*4/1e6:.0f} MB: {t*1e6:.1f} us  {2*n*4/t/1e12:.2f} TB/s (read+write)")
    s.record()
    for _ in range(50): z = x.sum()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 50 * 1e-3
    print(f"read-only sum {n*4/1e6:.0f} MB: {t*1e6:.1f} us  {n*4/t/1e12:.2f} TB/s")
