"""Per-block error of the sharded path after predelay changes (development aid)."""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import oracle  # noqa: E402
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.sharded import partitions_for, shard_bounds  # noqa: E402
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402
from helpers import BASE, apply_params  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
single = world == 0
world = max(world, 1)
Tcfg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nb, n_ref, T = 96, 16384, Tcfg
x = make_input(nb * 256)
irs = [make_ir(12000, seed=11, norm=0.05), make_ir(9000, seed=22, norm=0.05)]
P = partitions_for(12000, n_ref)
p0, p1 = dict(BASE, predelay=500, wet=0.7), dict(BASE, select=1)
ref = oracle.RefCompat(n_ref, True)
shards = []
for r in range(world):
    pb, pe = shard_bounds(P, world, r)
    print("shard", r, pb, pe)
    s = Convolution("t", n_ref, max_batch=T, part_begin=pb, part_end=pe if world > 1 else 0, stream_threshold=8)
    s.use_torch_stream()
    shards.append(s)
for i, ir in enumerate(irs):
    ref.prepare(i, ir)
    for s in shards:
        s.prepare(i, ir)
apply_params(ref, p0, p1, True)
for s in shards:
    apply_params(s, p0, p1, False)
events = {24: 3000, 56: 0, 64: 1024} if len(sys.argv) <= 3 else {}
dx = torch.from_numpy(x).cuda()
out = torch.zeros(2, nb * 256, device="cuda")
want = np.zeros((2, nb * 256))
parts = [torch.zeros(2 * T * 256, device="cuda") for _ in range(world)]
for b in range(0, nb, T):
    if b in events:
        ref.set(0, predelay=events[b])
        for s in shards:
            s.cc[0].value.predelay = events[b]
    sl = slice(b * 256, (b + T) * 256)
    want[:, sl] = ref.process(x[0, sl], x[1, sl])
    xin = dx[:, sl].contiguous()
    if single:
        o = torch.zeros(2, T * 256, device="cuda")
        shards[0].process_device(xin[0].data_ptr(), xin[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
        out[:, sl] = o
        continue
    for s, p in zip(shards, parts):
        s.partial_device(xin[0].data_ptr(), xin[1].data_ptr(), p.data_ptr(), T)
    total = sum(parts[1:], parts[0].clone())
    o = torch.zeros(2, T * 256, device="cuda")
    shards[0].finish_device(xin[0].data_ptr(), xin[1].data_ptr(), total.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
    for s in shards[1:]:
        s.finish_device(None, None, None, None, None, T)
    out[:, sl] = o
torch.cuda.synchronize()
d = out.cpu().numpy() - want
for b in range(nb):
    s = slice(b * 256, (b + 1) * 256)
    e = np.sqrt(np.mean(d[:, s] ** 2))
    if e > 1e-6:
        print(b, f"{e:.3e}", "sig", f"{np.sqrt(np.mean(want[:, s] ** 2)):.3e}")
print("total", np.sqrt(np.mean(d ** 2)))
