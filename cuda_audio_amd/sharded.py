"""IR-partition sharding across the GPUs of one node.

The output of the convolution is a sum over IR partitions, so partitions shard
cleanly: rank g owns partitions [pb_g, pe_g) of every path.  Every rank sees the
same input, runs the (cheap) forward FFT redundantly, its own share of the
partition x bin MAC and the inverse FFT, and produces a partial wet block in
the time domain.  The one exchange step is a sum of those partial blocks —
an RCCL all-reduce over xGMI (torch.distributed backend "nccl") — after which
every rank applies predelay, the Q1/Q2 terms, clamp and dry mix to the sum.
The payload is 2 KB per block and rank, so batches are long (the message is
nblocks * 2 KB) to keep the collective bandwidth- rather than latency-bound.

The reference has no multi-GPU path (single device); this file is new.
"""
import torch

BLOCK = 256
ALIGN = 16  # shard bounds are multiples of 16 partitions (4 waves x 4-partition steps)


def partitions_for(frames, fft_size, nframes=1024):
    """Partitions of an IR after the reference's truncation (conv.cu:239)."""
    n = min(int(frames), int(fft_size) - int(nframes))
    return max(1, (n + BLOCK - 1) // BLOCK)


def shard_bounds(n_partitions, world, rank):
    """Contiguous, ALIGN-aligned partition range of `rank`; the union over ranks
    covers [0, round_up(n_partitions, ALIGN)) exactly once.  May be empty for
    trailing ranks when world * ALIGN > n_partitions."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    total = (n_partitions + ALIGN - 1) // ALIGN * ALIGN
    per = ((total // ALIGN + world - 1) // world) * ALIGN
    pb = min(rank * per, total)
    pe = min(pb + per, total)
    return pb, pe


def slice_bounds(nblocks, world, rank, multiple=1):
    """Output blocks [first, first + count) of a batch that `rank` finishes in block-sliced operation:
    contiguous, `multiple`-aligned (the JACK period in blocks), covering the batch exactly once."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    units = nblocks // multiple
    if units * multiple != nblocks or units < world:
        raise ValueError("batch of %d blocks does not split over %d ranks in multiples of %d" % (nblocks, world, multiple))
    lo = rank * units // world
    hi = (rank + 1) * units // world
    return lo * multiple, (hi - lo) * multiple


class HipSlicer:
    """One rank's engine for block-sliced operation: the whole IR set, launches ordered with torch's stream."""

    def __init__(self, fft_size, max_batch, device, compat=True, period=256):
        from .engine import Convolution

        self.conv = Convolution("slicer", fft_size, max_batch=max_batch, device=device, compat=compat, period=period)
        self.conv.use_torch_stream()

    def prepare(self, idx, lr, nframes=1024):
        self.conv.prepare(idx, lr, nframes)

    def set_params(self, half, **kw):
        self.conv.cc[half].value.update(**kw)

    def process_slice(self, x, mine, nblocks, first, count):
        self.conv.process_slice_device(x[0].data_ptr(), x[1].data_ptr(), mine[0].data_ptr(), mine[1].data_ptr(), nblocks, first, count)

    def close(self):
        self.conv.close()


class BlockSlicedConvolution:
    """Throughput scaling without a data-path collective: every rank holds the whole IR set and is fed the same
    batch; rank r finishes its slice of the output blocks (mc_process_batch_slice_device).  `slicer` is a HipSlicer
    (product) or any object with the same process_slice interface (the gloo CPU tests inject an oracle-backed one).
    `gather=True` collects the slices on every rank so that `out` is the full [2, n] result, as a single engine
    would give; otherwise only this rank's slice of `out` is written."""

    def __init__(self, slicer, world=1, rank=0, group=None, period_blocks=1):
        self.slicer, self.world, self.rank, self.group, self.pm = slicer, world, rank, group, period_blocks

    def process(self, x, out, gather=True):
        n = x.shape[1]
        if n % BLOCK:
            raise ValueError("length must be a multiple of 256")
        T = n // BLOCK
        first, count = slice_bounds(T, self.world, self.rank, self.pm)
        mine = torch.empty(2, count * BLOCK, dtype=out.dtype, device=x.device)
        self.slicer.process_slice(x, mine, T, first, count)
        out[:, first * BLOCK:(first + count) * BLOCK] = mine
        if self.world == 1 or not gather:
            return out
        import torch.distributed as dist

        for r in range(self.world):  # slices may differ in length: one broadcast per owner
            f, c = slice_bounds(T, self.world, r, self.pm)
            buf = mine if r == self.rank else torch.empty(2, c * BLOCK, dtype=out.dtype, device=x.device)
            dist.broadcast(buf, src=r, group=self.group)
            out[:, f * BLOCK:(f + c) * BLOCK] = buf
        return out


class HipShard:
    """One rank's engine: cuda_audio_amd.Convolution restricted to its partition range."""

    def __init__(self, fft_size, pb, pe, max_batch, device, compat=True):
        from .engine import Convolution

        if pe <= pb:
            raise ValueError("empty shard: use fewer ranks than partitions / 16")
        self.conv = Convolution("shard", fft_size, max_batch=max_batch, device=device, compat=compat,
                                part_begin=pb, part_end=pe)
        self.conv.use_torch_stream()

    def prepare(self, idx, lr, nframes=1024):
        self.conv.prepare(idx, lr, nframes)

    def set_params(self, half, **kw):
        self.conv.cc[half].value.update(**kw)

    def partial(self, x, part, nblocks):
        self.conv.partial_device(x[0].data_ptr(), x[1].data_ptr(), part.data_ptr(), nblocks)

    def finish(self, x, wet_sum, out, nblocks):
        self.conv.finish_device(x[0].data_ptr(), x[1].data_ptr(), wet_sum.data_ptr(), out[0].data_ptr(),
                                out[1].data_ptr(), nblocks)

    def finish_slice(self, x, wet_sum_slice, out_slice, nblocks, first, count):
        """wet_sum_slice / out_slice: [2, count * 256], blocks [first, first + count) of the batch (after a reduce-scatter)."""
        self.conv.finish_slice_device(x[0].data_ptr(), x[1].data_ptr(), wet_sum_slice.data_ptr(), out_slice[0].data_ptr(),
                                      out_slice[1].data_ptr(), nblocks, first, count)

    def close(self):
        self.conv.close()


def reduce_scatter_channels(part2, ssum, rank, world, group=None):
    """Sum over ranks of the [2, n] partials, scattered by runs of blocks: rank r receives [2, n / world] = its run of
    both channel halves (one reduce-scatter per channel: a channel half is the concatenation of the ranks' runs).
    The path is chosen by the group's backend, the same on every rank, and errors propagate (a rank that fell back on its
    own would issue a different collective from the others): nccl (= RCCL) takes reduce_scatter_tensor; gloo, which has
    no reduce-scatter, an all-reduce of which every rank keeps its run.  The RCCL branch has run on one rank only so far
    (no multi-GPU node has been available: DESIGN section 6)."""
    import torch.distributed as dist

    n = part2.shape[1] // world
    backend = str(dist.get_backend(group)).lower()
    for c in range(2):
        if "nccl" in backend:
            dist.reduce_scatter_tensor(ssum[c], part2[c], op=dist.ReduceOp.SUM, group=group)
        else:
            tmp = part2[c].clone()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
            ssum[c].copy_(tmp[rank * n:(rank + 1) * n])


class ShardedConvolution:
    """Drives one shard per rank.  `shard` is a HipShard (product) or any object
    with the same partial/finish interface (the gloo CPU tests inject an
    oracle-backed one); `group` is a torch.distributed process group or None for
    a single rank."""

    def __init__(self, shard, world=1, group=None, rank=0):
        self.shard = shard
        self.world = world
        self.group = group
        self.rank = rank

    def process_scattered(self, x, out_slice, part=None):
        """The north-star layout with a reduce-scatter as its one exchange: every rank sums its shard of the partitions
        over the whole batch, receives the sum over ranks for ITS run of nblocks / world blocks, and finishes that run
        (mc_finish_batch_slice_device) into out_slice [2, n / world].  No rank is a root; each link carries 1/world of
        what a reduce to one root funnels into it.  Returns (first, count) of the blocks in out_slice."""
        n = x.shape[1]
        nblocks = n // BLOCK
        if n % BLOCK or nblocks % self.world:
            raise ValueError("the batch must be whole blocks and split evenly over the ranks")
        count = nblocks // self.world
        first = self.rank * count
        if part is None:
            part = torch.empty(2 * n, dtype=out_slice.dtype, device=x.device)
        self.shard.partial(x, part, nblocks)
        p2 = part.view(2, n)
        if self.world > 1:
            ssum = torch.empty(2, count * BLOCK, dtype=part.dtype, device=part.device)
            reduce_scatter_channels(p2, ssum, self.rank, self.world, self.group)
        else:
            ssum = p2
        self.shard.finish_slice(x, ssum.contiguous(), out_slice, nblocks, first, count)
        return first, count

    def process(self, x, out, part=None):
        """x: [2, n] input (same on every rank), out: [2, n] output, n = nblocks * 256.
        `part` is an optional preallocated [2 * n] scratch tensor for the partial."""
        n = x.shape[1]
        if n % BLOCK:
            raise ValueError("length must be a multiple of 256")
        nblocks = n // BLOCK
        if part is None:
            part = torch.empty(2 * n, dtype=out.dtype, device=x.device)
        self.shard.partial(x, part, nblocks)
        if self.world > 1:
            import torch.distributed as dist

            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        self.shard.finish(x, part, out, nblocks)
        return out
