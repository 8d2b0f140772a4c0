"""Practical HBM bandwidth of this box: a device-to-device copy (read + write) and a read-only reduction, 1 GiB each."""
import torch, time
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
x.normal_()
for name, fn, byt in (("copy (read+write)", lambda: y.copy_(x), 2 * x.numel() * 4), ("sum (read)", lambda: x.sum(), x.numel() * 4),
                      ("fill (write)", lambda: y.fill_(1.0), x.numel() * 4)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("%-20s %.2f TB/s" % (name, byt / dt / 1e12))
