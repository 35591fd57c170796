import torch, time
n = 1 << 30  # 4 GiB fp32
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
a.fill_(1.0); torch.cuda.synchronize()
def t(fn, bytes_, label):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{label}: {bytes_ / dt / 1e12:.2f} TB/s")
t(lambda: b.copy_(a), 8 * n, "copy (read + write)")
t(lambda: a.fill_(2.0), 4 * n, "fill (write only)")
t(lambda: a.sum(), 4 * n, "sum (read only)")
