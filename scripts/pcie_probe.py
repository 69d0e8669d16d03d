#!/usr/bin/env python3
"""What the box's PCIe link carries (pinned host memory, torch copies): one direction at a time and both at once,
for the chunk sizes mc_process_batch uses."""
import time

import torch

dev = torch.device("cuda:0")
for mb in (6.6, 33, 132):
    n = int(mb * 1e6 / 4)
    h_in = torch.empty(n, dtype=torch.float32).pin_memory()
    h_out = torch.empty(n, dtype=torch.float32).pin_memory()
    d_a = torch.empty(n, dtype=torch.float32, device=dev)
    d_b = torch.zeros(n, dtype=torch.float32, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = max(3, int(400 / mb))

    def run(h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    d_a.copy_(h_in, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_out.copy_(d_b, non_blocking=True)
        torch.cuda.synchronize()
        return reps * n * 4 / (time.perf_counter() - t0) / 1e9

    run(True, True)
    print(f"{mb:6.1f} MB copies: H2D alone {run(True, False):5.1f} GB/s, D2H alone {run(False, True):5.1f} GB/s, both at once {run(True, True):5.1f} GB/s each way")
