"""Multi-rank path on CPU: world_size-2 gloo, the product's sharding driver
(cuda_audio_amd.sharded) with oracle-backed shards injected in place of the
HIP engine.  Checks: shard bounds tile the partition range, and
all-reduce(partials) + finish == the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cuda_audio_amd.sharded import ShardedConvolution, partitions_for, shard_bounds


def test_shard_bounds_tile_the_range():
    for P in (1, 15, 16, 17, 345, 1723, 5168):
        for world in (1, 2, 3, 4, 8):
            total = (P + 15) // 16 * 16
            got = [shard_bounds(P, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            for (a, b), (c, d) in zip(got, got[1:]):
                assert b == c and a <= b
            assert all(a % 16 == 0 and b % 16 == 0 for a, b in got)
    assert partitions_for(441000, 524288) == 1723
    assert partitions_for(10**7, 131072) == 508  # truncation n_ref - 1024 (conv.cu:239)


class OracleShard:
    """Test double with HipShard's interface, float64, one block at a time."""

    def __init__(self, n_ref, pb, pe, ir):
        import oracle

        self.o = oracle.Upols(n_ref, True, pb, pe)
        self.o.prepare(0, ir)

    def partial(self, x, part, nblocks):
        xn = x.numpy()
        p = part.view(2, -1).numpy()
        self._tmp = []
        # the oracle keeps per-block state between partial and finish, so blocks alternate
        assert nblocks == 1
        p[:] = self.o.partial(xn[0], xn[1])

    def finish(self, x, wet_sum, out, nblocks):
        xn = x.numpy()
        out.numpy()[:] = self.o.finish(xn[0], xn[1], wet_sum.view(2, -1).numpy())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuda_audio_amd.synth import make_input, make_ir

        nb, n_ref = 24, 16384
        x = make_input(nb * 256)
        ir = make_ir(9000, seed=3, norm=0.05)  # 36 partitions -> 2 shards of 32 / 4... aligned 16
        P = partitions_for(len(ir), n_ref)
        pb, pe = shard_bounds(P, world, rank)
        drv = ShardedConvolution(OracleShard(n_ref, pb, pe, ir), world=world)
        out = np.zeros((2, nb * 256))
        for b in range(nb):
            xb = torch.from_numpy(x[:, b * 256:(b + 1) * 256].copy())
            ob = torch.zeros(2, 256, dtype=torch.float64)
            part = torch.zeros(2 * 256, dtype=torch.float64)
            drv.process(xb, ob, part)
            out[:, b * 256:(b + 1) * 256] = ob.numpy()
        if rank == 0:
            import oracle

            full = oracle.Upols(n_ref, True)
            full.prepare(0, ir)
            want = full.process(x[0], x[1])
            ret["err"] = float(np.sqrt(np.mean((out - want) ** 2)))
            ret["sig"] = float(np.sqrt(np.mean(want ** 2)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sum_of_partials_equals_unsharded():
    import oracle

    oracle.lib()  # build once before forking workers
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["sig"] > 1e-3
    assert ret["err"] < 1e-14, ret["err"]
