import torch, time
for mb in (16, 64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
    x.random_()
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("copy %5d MB: %.1f us, %.2f TB/s (r+w)" % (mb, ms * 1e3, 2 * mb / 1024 / 1024 / (ms * 1e-3) * 1.048576))
    for _ in range(5): y.add_(1)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): y.add_(1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("inplace add %5d MB: %.1f us, %.2f TB/s (r+w)" % (mb, ms * 1e3, 2 * mb / 1024 / 1024 / (ms * 1e-3) * 1.048576))
