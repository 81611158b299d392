"""Practical HBM ceiling of the box: torch copy / read-only / write-only streams over 1 GiB (tuning aid for the roofline talk
in DESIGN.md: the 8 TB/s peak is never reached by a mixed read+write stream)."""
import torch
dev = "cuda:0"
n = 1 << 29                                   # 2^29 bf16 = 1 GiB
x = torch.randn(n, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gib = x.numel() * 2
t = timed(lambda: y.copy_(x)); print("copy   %.2f TB/s (read+write)" % (2 * gib / t / 1e12))
t = timed(lambda: y.fill_(1.0)); print("fill   %.2f TB/s (write)" % (gib / t / 1e12))
xf = x.view(torch.int16)
t = timed(lambda: torch.sum(xf, dtype=torch.int64)); print("sum    %.2f TB/s (read)" % (gib / t / 1e12))
t = timed(lambda: torch.add(x, x, out=y)); print("add    %.2f TB/s (read+write, 1 stream in)" % (2 * gib / t / 1e12))
