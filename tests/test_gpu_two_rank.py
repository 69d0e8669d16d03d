"""Two processes, one card: each rank runs the real HIP engine on its IR-partition shard and the partial wet
blocks are summed with gloo (the box has one GPU, so RCCL cannot be used between the ranks; the engine entry
points and the driver code are the ones the RCCL path uses).  Rank 0's output is checked against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import RMS_TOL

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from cuda_audio_amd.sharded import HipShard, ShardedConvolution, partitions_for, shard_bounds
    from cuda_audio_amd.synth import make_input, make_ir

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nb, n_ref, T = 96, 32768, 32
        x = make_input(nb * 256)
        irs = [make_ir(20000, seed=3, norm=0.05), make_ir(15000, seed=4, norm=0.05)]
        P = partitions_for(20000, n_ref)  # 79 partitions -> [0,48) and [48,80)
        pb, pe = shard_bounds(P, world, rank)
        shard = HipShard(n_ref, pb, pe, T, 0)
        for i, ir in enumerate(irs):
            shard.prepare(i, ir)
        shard.set_params(1, select=1)
        shard.set_params(0, predelay=300, panWet=-0.25)
        drv = ShardedConvolution(shard, world=world)
        dx = torch.from_numpy(x).cuda()
        out = torch.zeros(2, nb * 256, device="cuda")
        for b in range(0, nb, T):
            s = slice(b * 256, (b + T) * 256)
            drv.process(dx[:, s].contiguous(), out[:, s])
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        if rank == 0:
            import oracle

            ref = oracle.RefCompat(n_ref, True)
            for i, ir in enumerate(irs):
                ref.prepare(i, ir)
            ref.set(1, select=1)
            ref.set(0, predelay=300, panWet=-0.25)
            want = ref.process(x[0], x[1])
            ret["err"] = float(np.sqrt(np.mean((got - want) ** 2)))
            ret["sig"] = float(np.sqrt(np.mean(want ** 2)))
        shard.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_card_sum_of_partials_matches_oracle():
    import oracle

    oracle.lib()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["sig"] > 1e-3
    assert ret["err"] <= RMS_TOL, ret["err"]


def _slice_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from cuda_audio_amd.sharded import BlockSlicedConvolution, HipSlicer
    from cuda_audio_amd.synth import make_input, make_ir

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nb, n_ref, T = 96, 32768, 32
        x = make_input(nb * 256)
        irs = [make_ir(20000, seed=3, norm=0.05), make_ir(15000, seed=4, norm=0.05)]
        sl = HipSlicer(n_ref, T, 0)
        for i, ir in enumerate(irs):
            sl.prepare(i, ir)
        sl.set_params(1, select=1)
        sl.set_params(0, predelay=300, panWet=-0.25)
        drv = BlockSlicedConvolution(sl, world=world, rank=rank)
        dx = torch.from_numpy(x).cuda()
        out = torch.zeros(2, nb * 256, device="cuda")
        for b in range(0, nb, T):
            s = slice(b * 256, (b + T) * 256)
            drv.process(dx[:, s].contiguous(), out[:, s])
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        import oracle

        ref = oracle.RefCompat(n_ref, True)
        for i, ir in enumerate(irs):
            ref.prepare(i, ir)
        ref.set(1, select=1)
        ref.set(0, predelay=300, panWet=-0.25)
        want = ref.process(x[0], x[1])
        ret[rank] = float(np.sqrt(np.mean((got - want) ** 2)))
        sl.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_block_slices_gathered_on_every_rank():
    """Block-sliced operation with two processes on one card: each finishes half of the output blocks of every
    batch; after the gather both hold the reference's output."""
    import oracle

    oracle.lib()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_slice_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] <= RMS_TOL and ret[1] <= RMS_TOL, dict(ret)


def _scatter_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from cuda_audio_amd.sharded import HipShard, ShardedConvolution, partitions_for, shard_bounds
    from cuda_audio_amd.synth import make_input, make_ir

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nb, n_ref, T = 96, 32768, 32
        x = make_input(nb * 256)
        x[0] += 0.04
        irs = [make_ir(20000, seed=3, norm=0.05), make_ir(15000, seed=4, norm=0.05)]
        P = partitions_for(20000, n_ref)
        pb, pe = shard_bounds(P, world, rank)
        shard = HipShard(n_ref, pb, pe, T, 0)
        for i, ir in enumerate(irs):
            shard.prepare(i, ir)
        shard.set_params(1, select=1)
        shard.set_params(0, predelay=300, panWet=-0.25)
        drv = ShardedConvolution(shard, world=world, rank=rank)
        count = T // world
        # the gloo group works on host tensors: the partial is summed through the host, as in bench.py's rehearsal
        dx = torch.from_numpy(x).cuda()
        mine = np.zeros((2, nb // T * count * 256), np.float32)
        for k, b in enumerate(range(0, nb, T)):
            xs = dx[:, b * 256:(b + T) * 256].contiguous()
            part = torch.zeros(2 * T * 256, device="cuda")
            shard.partial(xs, part, T)
            torch.cuda.synchronize()
            h = part.cpu()
            dist.all_reduce(h)
            ssum = h.view(2, T * 256)[:, rank * count * 256:(rank + 1) * count * 256].contiguous().cuda()
            out = torch.zeros(2, count * 256, device="cuda")
            shard.finish_slice(xs, ssum, out, T, rank * count, count)
            torch.cuda.synchronize()
            mine[:, k * count * 256:(k + 1) * count * 256] = out.cpu().numpy()
        assert drv.rank == rank
        import oracle

        ref = oracle.RefCompat(n_ref, True)
        for i, ir in enumerate(irs):
            ref.prepare(i, ir)
        ref.set(1, select=1)
        ref.set(0, predelay=300, panWet=-0.25)
        want = ref.process(x[0], x[1]).reshape(2, nb // T, T * 256)[:, :, rank * count * 256:(rank + 1) * count * 256].reshape(2, -1)
        ret[rank] = (float(np.sqrt(np.mean((mine - want) ** 2))), float(np.sqrt(np.mean(want ** 2))))
        shard.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_finish_their_runs_after_a_reduce_scatter():
    """The north-star layout's reduce-scatter form with two processes on one card (real engines, the sum through gloo on
    the host): each rank finishes its half of every batch's blocks from the summed partial - mc_finish_batch_slice_device
    - and both halves are the reference's samples."""
    import oracle

    oracle.lib()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_scatter_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in range(2):
        err, sig = ret[r]
        assert sig > 1e-3 and err <= RMS_TOL, (r, err, sig)
