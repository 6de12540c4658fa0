#!/usr/bin/env python3
"""Calibration: what do plain device copies / reads reach in the same back-to-back-launch harness as tools/bench_kernels.py?
(torch elementwise kernels, cold buffers from a ring; sizes of the 4K HDR frame and multiples)"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from tools.bench_kernels import time_events
    for mb in (33, 66, 132, 264, 1056):
        n = mb * 1024 * 1024 // 4
        ring = max(2, min(8, 2048 // mb))
        src = [torch.rand(n, device="cuda") for _ in range(ring)]
        dst = [torch.empty(n, device="cuda") for _ in range(ring)]
        iters = 1000 if mb <= 264 else 200
        med, mn = time_events(torch, lambda k: dst[k % ring].copy_(src[k % ring]), iters)
        print(f"[copy] {mb} MB -> {mb} MB: median {med:.1f} us  {2 * mb * 1.048576 / med * 1e3:.0f} GB/s moved ({2 * mb * 1.048576 / med * 1e3 / 80:.1f}% of 8 TB/s)", flush=True)
        med, mn = time_events(torch, lambda k: dst[k % ring].fill_(1.0), iters)
        print(f"[fill] {mb} MB: median {med:.1f} us  {mb * 1.048576 / med * 1e3:.0f} GB/s", flush=True)
        acc = torch.zeros(1, device="cuda")
        med, mn = time_events(torch, lambda k: torch.sum(src[k % ring], dim=0, keepdim=True, out=acc), iters)
        print(f"[sum]  {mb} MB: median {med:.1f} us  {mb * 1.048576 / med * 1e3:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
