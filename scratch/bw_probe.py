"""What this GPU streams with stock kernels at the bench sizes (134 MB tensors): copy, add, sum."""
import torch
dev = torch.device("cuda:0")
n = 16384 * 1024 * 2
a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); c = torch.empty(n, device=dev)
def t(fn, nbytes, label, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)
    print("%-28s median %.4f ms = %.2f TB/s   min %.4f ms = %.2f TB/s" % (label, ms[len(ms)//2], nbytes / ms[len(ms)//2] / 1e9, ms[0], nbytes / ms[0] / 1e9), flush=True)
t(lambda: c.copy_(a), 2 * 4 * n, "copy (read + write)")
t(lambda: torch.add(a, b, out=c), 3 * 4 * n, "add (2 reads + write)")
t(lambda: a.sum(), 4 * n, "sum (read)")
t(lambda: torch.mul(a, 2.0, out=a), 2 * 4 * n, "scale in place")
big = torch.randn(4 * n, device=dev)
t(lambda: big.sum(), 16 * n, "sum of 537 MB (read)")
