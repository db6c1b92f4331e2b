"""What the HBM of this box sustains for pure writes, pure reads and copies (torch kernels on 2 GiB fp32 tensors, cold):
the denominators behind the 'fraction of 8 TB/s' figures of write-heavy kernels (norm1 + qkv writes 3x what it reads).
python tools/hbm_rw.py"""
import torch
dev = torch.device("cuda:0")
n = 512 * 1024 * 1024
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev); c = torch.empty(n, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timed(lambda: a.fill_(1.0)); print(f"write only (fill_ 2 GiB):            {4 * n / t / 1e12:5.2f} TB/s")
t = timed(lambda: a.sum());       print(f"read only (sum of 2 GiB):            {4 * n / t / 1e12:5.2f} TB/s")
t = timed(lambda: b.copy_(a));    print(f"copy (2 GiB read + 2 GiB written):   {8 * n / t / 1e12:5.2f} TB/s total")
t = timed(lambda: torch.add(a, b, out=c)); print(f"add (4 GiB read + 2 GiB written):    {12 * n / t / 1e12:5.2f} TB/s total")
