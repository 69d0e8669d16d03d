#!/usr/bin/env python3
"""The dependency pattern of mc_process_batch's pinned path (H2D stream / compute stream / D2H stream, two chunks in
flight, two copies per direction and chunk) with a trivial kernel in place of the engine: what the runtime gives."""
import time

import torch

dev = torch.device("cuda:0")
T = 32320
n = T * 256
nchunks = 10
hin = [torch.empty(nchunks * n, dtype=torch.float32).pin_memory() for _ in range(2)]
hout = [torch.empty(nchunks * n, dtype=torch.float32).pin_memory() for _ in range(2)]
NB = 3
d = [[torch.zeros(n, device=dev) for _ in range(4)] for _ in range(NB)]
s_in, s_c, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev_h2d = [torch.cuda.Event() for _ in range(NB)]
ev_comp = [torch.cuda.Event() for _ in range(NB)]
ev_d2h = [torch.cuda.Event() for _ in range(NB)]


def run(dep_in=True, dep_c_in=True, dep_c_out=True, dep_out=True, nb=2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(nchunks):
        b = k % nb
        o = k * n
        with torch.cuda.stream(s_in):
            if k >= nb and dep_in:
                s_in.wait_event(ev_comp[b])
            d[b][0].copy_(hin[0][o:o + n], non_blocking=True)
            d[b][1].copy_(hin[1][o:o + n], non_blocking=True)
            ev_h2d[b].record(s_in)
        with torch.cuda.stream(s_c):
            if dep_c_in:
                s_c.wait_event(ev_h2d[b])
            if k >= nb and dep_c_out:
                s_c.wait_event(ev_d2h[b])
            torch.add(d[b][0], d[b][1], out=d[b][2])
            torch.sub(d[b][0], d[b][1], out=d[b][3])
            ev_comp[b].record(s_c)
        with torch.cuda.stream(s_out):
            if dep_out:
                s_out.wait_event(ev_comp[b])
            hout[0][o:o + n].copy_(d[b][2], non_blocking=True)
            hout[1][o:o + n].copy_(d[b][3], non_blocking=True)
            ev_d2h[b].record(s_out)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run()
for name, kw in (("all dependencies", {}), ("all dependencies, three staging sets", dict(nb=3)), ("no dependencies", dict(dep_in=False, dep_c_in=False, dep_c_out=False, dep_out=False)),
                 ("only compute waits for copy-in", dict(dep_in=False, dep_c_out=False, dep_out=False)),
                 ("only copy-out waits for compute", dict(dep_in=False, dep_c_in=False, dep_c_out=False)),
                 ("compute waits for copy-in, copy-out waits for compute", dict(dep_in=False, dep_c_out=False)),
                 ("without copy-in waiting for compute", dict(dep_in=False)),
                 ("without compute waiting for copy-out", dict(dep_c_out=False))):
    dt = run(**kw)
    print(f"{name:60s}: {dt / nchunks * 1e3:.3f} ms per chunk, {nchunks * n * 8 / dt / 1e9:.1f} GB/s each way")
