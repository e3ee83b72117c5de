"""HBM calibration: what a plain streaming read reaches on this box at 2 GB and 200 MB (torch kernels)."""
import torch, time
for n in (1000000, 100000):
    x = torch.randn((n, 512), device="cuda")
    for name, fn in (("sum(0)", lambda: x.sum(0)), ("sum()", lambda: x.sum()), ("abs().max()", lambda: x.abs().max()), ("copy", lambda: x.clone())):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = x.numel() * 4 / 1e9 * (2 if name == "copy" else 1)
        print(f"n={n} {name:12s} {ms*1e3:8.1f} us  {gb/ms*1e3/1e3:6.2f} TB/s")
