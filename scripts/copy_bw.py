"""Achievable HBM copy bandwidth on this box (torch device-to-device copy), for calibrating the roofline discussion."""
import torch
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        b.copy_(a)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("copy %5d MB: %.3f ms  read+write %.2f TB/s" % (mb, ms, 2 * mb * 1.048576e6 / ms / 1e9))
    r = torch.empty(1, dtype=torch.float64, device="cuda")
    s.record()
    for _ in range(20):
        r = a.sum()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("sum  %5d MB: %.3f ms  read %.2f TB/s" % (mb, ms, mb * 1.048576e6 / ms / 1e9))
