"""Batch-axis sharding of the Melspectrogram path over one process per GPU (RCCL over xGMI).

The path is embarrassingly parallel over rows (batch x channel): no op mixes rows
(functional.py:89-91 of the reference flattens them into the FFT batch), the window and filterbank
are tiny replicated constants, so the only communication is ONE optional all-gather of the output
shards when the caller wants the whole batch on every rank.  ``backend='nccl'`` is RCCL on ROCm;
the same code runs on ``gloo`` for CPU tests of the control flow.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Contiguous, balanced [begin, end) slice of dim 0 owned by ``rank`` (sizes differ by <= 1)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('bad world_size/rank: %r/%r' % (world_size, rank))
    base, extra = divmod(int(n_rows), world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(x, world_size=None, rank=None):
    """This rank's slice of a whole-batch tensor along dim 0 (a view, no copy)."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    b, e = shard_bounds(x.shape[0], world_size, rank)
    return x[b:e]


def _row_major(t):
    """Strides are those of a C-contiguous tensor of this shape (size-1 dims aside) — also meaningful for 0 rows."""
    expect = 1
    for n, st in zip(reversed(t.shape[1:]), reversed(t.stride()[1:])):
        if n != 1 and st != expect:
            return False
        expect *= n
    return t.shape[0] <= 1 or t.stride(0) == expect


def all_gather_batch(local, total_rows=None, group=None):
    """Concatenate every rank's output shard along dim 0 with a single collective.

    ``local`` may be one of the strided (…, M, T) views the layers return: the gather is done on the
    physical frame-major buffers (a dim-0 concat commutes with the trailing transpose) with one
    ``all_gather_into_tensor``; uneven shards are padded to the largest and trimmed afterwards.
    """
    from ._lazy import realize
    local = realize(local)
    world = dist.get_world_size(group)
    if world == 1:
        return local
    # dim 0 must stay dim 0 of the physical buffer: the (.., M, T) views of the layers qualify from 3-D upwards; for a
    # 2-D (M, T) view the transpose IS dim 0, so that one is gathered through a contiguous copy instead
    # (decided from the strides alone, so that a rank holding zero rows — for which torch reports every layout as
    # contiguous — takes the same branch as its peers)
    transposed = local.dim() >= 3 and not _row_major(local) and _row_major(local.transpose(-2, -1))
    phys = local.transpose(-2, -1) if transposed else (local if _row_major(local) else local.contiguous())
    rows = phys.shape[0]
    if total_rows is None:
        cnt = torch.tensor([rows], dtype=torch.int64, device=phys.device)
        dist.all_reduce(cnt, group=group)
        total_rows = int(cnt.item())
    sizes = [shard_bounds(total_rows, world, r) for r in range(world)]
    sizes = [e - b for b, e in sizes]
    if all(s == sizes[0] for s in sizes) and rows == sizes[0]:
        out = torch.empty((total_rows,) + tuple(phys.shape[1:]), dtype=phys.dtype, device=phys.device)
        dist.all_gather_into_tensor(out, phys, group=group)
    else:
        # uneven tail (sizes differ by at most one row, a rank may even own none): still ONE collective — every rank
        # contributes `biggest` rows (only the short ranks copy theirs into a padded buffer), then the padding rows
        # are dropped with one compacting copy
        biggest = max(sizes)
        if rows == biggest:
            padded = phys.contiguous()
        else:
            padded = phys.new_empty((biggest,) + tuple(phys.shape[1:]))
            padded[:rows] = phys
        buf = torch.empty((world * biggest,) + tuple(phys.shape[1:]), dtype=phys.dtype, device=phys.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        out = torch.cat([buf[r * biggest:r * biggest + n] for r, n in enumerate(sizes) if n], dim=0)
    return out.transpose(-2, -1) if transposed else out


class ShardedPipeline(torch.nn.Module):
    """Wrap a feature pipeline: run it on this rank's batch shard, optionally all-gather the result."""

    def __init__(self, pipeline, gather=True, group=None):
        super(ShardedPipeline, self).__init__()
        self.pipeline = pipeline
        self.gather = gather
        self.group = group

    def forward(self, whole_batch):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        shard = shard_batch(whole_batch, world, rank)
        if shard.shape[0] == 0 and whole_batch.shape[0] > 0:
            # fewer rows than ranks: this rank owns none.  It must still enter the collective (the others would hang
            # until the RCCL timeout), so it runs the pipeline on one borrowed row to learn the output layout and
            # contributes zero rows of it.
            local = self.pipeline(whole_batch[:1])[:0]
        else:
            local = self.pipeline(shard)
        if not self.gather:
            from ._lazy import realize
            return realize(local)
        return all_gather_batch(local, total_rows=whole_batch.shape[0], group=self.group)
