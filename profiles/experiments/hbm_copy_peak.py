#!/usr/bin/env python
"""Achievable HBM rates of the box ('the build must measure its own achievable peaks'): device-to-device
copy, fill and read-reduce of buffers far larger than the 256 MB infinity cache."""
import torch


def rate(fn, nbytes, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e12


def main():
    n = 1 << 30                                            # 4 GiB of fp32 per buffer
    x = torch.empty(n, device="cuda").normal_()
    y = torch.empty_like(x)
    print(f"copy  (read 4 GiB + write 4 GiB): {rate(lambda: y.copy_(x), 8 * n):.2f} TB/s moved")
    print(f"fill  (write 4 GiB):              {rate(lambda: y.zero_(), 4 * n):.2f} TB/s")
    print(f"read  (sum of 4 GiB):             {rate(lambda: x.sum(), 4 * n):.2f} TB/s")
    m = 1 << 24                                            # 64 MiB: infinity-cache resident
    xs, ys = x[:m], y[:m]
    print(f"copy of 64 MiB (cache resident):  {rate(lambda: ys.copy_(xs), 8 * m, 50):.2f} TB/s moved")


if __name__ == "__main__":
    main()
