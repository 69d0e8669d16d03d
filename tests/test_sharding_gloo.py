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

from cuda_audio_amd.sharded import BlockSlicedConvolution, ShardedConvolution, partitions_for, shard_bounds, slice_bounds


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


class OracleBatchShard:
    """Test double for the reduce-scatter driver: HipShard's partial / finish_slice over a batch of blocks.  The oracle
    advances block by block (partial, then finish), so one instance runs ahead to produce the batch's partials and a
    second one replays the batch and finishes the rank's run of blocks with the summed partial."""

    def __init__(self, n_ref, pb, pe, ir):
        import oracle

        self.a = oracle.Upols(n_ref, True, pb, pe)
        self.b = oracle.Upols(n_ref, True, pb, pe)
        self.a.prepare(0, ir)
        self.b.prepare(0, ir)

    def partial(self, x, part, nblocks):
        xn, p = x.numpy(), part.view(2, -1).numpy()
        for t in range(nblocks):
            s = slice(t * 256, (t + 1) * 256)
            p[:, s] = self.a.partial(xn[0, s], xn[1, s])
            self.a.finish(xn[0, s], xn[1, s], p[:, s].copy())  # (advances the state; its output is not used)

    def finish_slice(self, x, wet_sum_slice, out_slice, nblocks, first, count):
        xn, w, o = x.numpy(), wet_sum_slice.numpy(), out_slice.numpy()
        for t in range(nblocks):
            s = slice(t * 256, (t + 1) * 256)
            own = self.b.partial(xn[0, s], xn[1, s])
            if first <= t < first + count:
                d = slice((t - first) * 256, (t - first + 1) * 256)
                o[:, d] = self.b.finish(xn[0, s], xn[1, s], w[:, d].copy())
            else:
                self.b.finish(xn[0, s], xn[1, s], own)


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


def _scatter_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuda_audio_amd.synth import make_input, make_ir

        nbat, T, n_ref = 3, 8, 16384
        x = make_input(nbat * T * 256)
        x[0] += 0.04  # (the Q1/Q2 sums matter: every rank must keep their history to finish its run)
        ir = make_ir(9000, seed=3, norm=0.05)
        P = partitions_for(len(ir), n_ref)
        pb, pe = shard_bounds(P, world, rank)
        drv = ShardedConvolution(OracleBatchShard(n_ref, pb, pe, ir), world=world, rank=rank)
        count = T // world
        mine = np.zeros((2, nbat * count * 256))
        for k in range(nbat):
            xb = torch.from_numpy(x[:, k * T * 256:(k + 1) * T * 256].copy())
            ob = torch.zeros(2, count * 256, dtype=torch.float64)
            f, c = drv.process_scattered(xb, ob, torch.zeros(2 * T * 256, dtype=torch.float64))
            assert (f, c) == (rank * count, count)
            mine[:, k * count * 256:(k + 1) * count * 256] = ob.numpy()
        import oracle

        full = oracle.Upols(n_ref, True)
        full.prepare(0, ir)
        want = full.process(x[0], x[1]).reshape(2, nbat, T * 256)[:, :, rank * count * 256:(rank + 1) * count * 256].reshape(2, -1)
        ret[rank] = (float(np.sqrt(np.mean((mine - want) ** 2))), float(np.sqrt(np.mean(want ** 2))))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_reduce_scatter_and_slice_finish_equals_unsharded():
    """The north-star layout's exchange as a reduce-scatter: each rank sums its partition shard
    over the whole batch, receives the sum for its half of the blocks and finishes them; every rank keeps the Q1/Q2
    history.  Both ranks' runs equal the unsharded oracle's (gloo has no reduce-scatter: the driver falls back to
    all-reduce + slice, the same sums)."""
    import oracle

    oracle.lib()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_scatter_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        err, sig = ret[r]
        assert sig > 1e-3 and err < 1e-14, (r, err, sig)


def test_slice_bounds_tile_the_batch():
    for T in (8, 48, 8192, 65536):
        for world in (1, 2, 3, 4, 8):
            for pm in (1, 2, 4):
                if T % pm or T // pm < world:
                    continue
                got = [slice_bounds(T, world, r, pm) for r in range(world)]
                assert got[0][0] == 0 and sum(c for _, c in got) == T
                pos = 0
                for f, c in got:
                    assert f == pos and c > 0 and f % pm == 0 and c % pm == 0
                    pos += c
    with pytest.raises(ValueError):
        slice_bounds(6, 4, 0, 2)


class OracleSlicer:
    """Test double with HipSlicer's interface: runs the whole batch through the float64 restatement (every rank
    sees the whole input, as the product does) and hands back the requested slice."""

    def __init__(self, n_ref, ir):
        import oracle

        self.o = oracle.RefCompat(n_ref, True)
        self.o.prepare(0, ir)

    def process_slice(self, x, mine, nblocks, first, count):
        xn = x.numpy()
        full = self.o.process(xn[0], xn[1])
        mine.numpy()[:] = full[:, first * 256:(first + count) * 256]


def _slice_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuda_audio_amd.synth import make_input, make_ir

        nbat, T, n_ref = 3, 10, 4096
        x = make_input(nbat * T * 256)
        ir = make_ir(2500, seed=3, norm=0.05)
        drv = BlockSlicedConvolution(OracleSlicer(n_ref, ir), world=world, rank=rank)
        out = torch.zeros(2, nbat * T * 256, dtype=torch.float64)
        for k in range(nbat):
            sl = slice(k * T * 256, (k + 1) * T * 256)
            drv.process(torch.from_numpy(x[:, sl].copy()), out[:, sl])
        import oracle

        full = oracle.RefCompat(n_ref, True)
        full.prepare(0, ir)
        want = full.process(x[0], x[1])
        ret[rank] = float(np.sqrt(np.mean((out.numpy() - want) ** 2)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_block_slices_gather_to_the_full_output():
    import oracle

    oracle.lib()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slice_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0] < 1e-15 and ret[1] < 1e-15
