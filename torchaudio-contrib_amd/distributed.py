"""Batch-axis sharding of the Melspectrogram path over one process per GPU (RCCL over xGMI).

The path is embarrassingly parallel over rows (batch x channel): no op mixes rows
(functional.py:89-91 of the reference flattens them into the FFT batch), the window and filterbank
are tiny replicated constants, so the only communication is ONE optional all-gather of the output
shards when the caller wants the whole batch on every rank.  ``backend='nccl'`` is RCCL on ROCm;
the same code runs on ``gloo`` for CPU tests of the control flow.

Algorithm of that one collective (``TAC_ALLGATHER`` / the ``method`` argument of :func:`all_gather_batch`):

``rccl`` (default)
    ``all_gather_into_tensor`` — RCCL picks rings / trees and the protocol from the message size.  What it picks can be
    steered from the environment without touching this file: ``NCCL_ALGO=Ring|Tree``, ``NCCL_PROTO=Simple|LL|LL128``,
    ``NCCL_MIN_NCHANNELS`` / ``NCCL_MAX_NCHANNELS`` (channels = concurrent rings; on the fully connected 8-GPU xGMI
    mesh RCCL needs >= 7 of them to drive all seven links of a GPU at once), ``NCCL_DEBUG=INFO`` prints the choice.
``p2p``
    one-shot direct exchange: every rank posts its shard to each of its ``world - 1`` peers in ONE batched group of
    point-to-point sends / receives (``batch_isend_irecv`` — ``ncclGroupStart`` / ``ncclSend`` / ``ncclRecv`` on RCCL)
    straight into the receiver's slot of the gathered buffer.  On a fully connected xGMI mesh every transfer has its own
    link, nothing is forwarded (a ring moves each shard over ``world - 1`` hops), and the per-rank wire time is
    ``shard_bytes / link_bandwidth`` instead of ``(world - 1) * shard_bytes / (rings * link_bandwidth)``.

DESIGN.md §6 holds the expected numbers for BASELINE configs[2] (338.7 MB per rank) the first 8-GPU run is judged by.
"""
import os

import torch
import torch.distributed as dist

METHODS = ('rccl', 'p2p')


def default_method():
    m = os.environ.get('TAC_ALLGATHER', 'rccl').lower()
    if m not in METHODS:
        raise ValueError('TAC_ALLGATHER must be one of %r, got %r' % (METHODS, m))
    return m


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


def _exchange(out_slots, mine, rank, world, group, method):
    """The one collective: ``mine`` (this rank's contiguous block) into slot ``rank`` of every rank's ``out_slots``
    (a (world * n, ...) buffer whose r-th block of n rows is rank r's)."""
    n = mine.shape[0]
    if method == 'rccl':
        dist.all_gather_into_tensor(out_slots, mine, group=group)
        return
    # one-shot direct exchange: my block to every peer, every peer's block into its slot here — one batched group
    out_slots[rank * n:(rank + 1) * n].copy_(mine)
    ops = []
    for step in range(1, world):
        to, frm = (rank + step) % world, (rank - step) % world          # staggered so that no two ranks start on the same peer
        ops.append(dist.P2POp(dist.isend, mine, dist.get_global_rank(group, to) if group is not None else to, group))
        ops.append(dist.P2POp(dist.irecv, out_slots[frm * n:(frm + 1) * n],
                              dist.get_global_rank(group, frm) if group is not None else frm, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def all_gather_batch(local, total_rows=None, group=None, method=None, force_collective=False):
    """Concatenate every rank's output shard along dim 0 with a single collective.

    ``local`` may be one of the strided (…, M, T) views the layers return: the gather is done on the
    physical frame-major buffers (a dim-0 concat commutes with the trailing transpose) in one exchange
    (``method``: ``'rccl'`` = ``all_gather_into_tensor``, ``'p2p'`` = direct sends to every peer; default from
    ``TAC_ALLGATHER``, see the module docstring); uneven shards are padded to the largest and trimmed afterwards.
    ``force_collective`` runs the collective even in a group of one (the RCCL smoke test of a 1-GPU box: the
    communicator, the strided-view handling and the output buffer are exercised exactly as at world > 1).
    """
    from ._lazy import realize
    local = realize(local)
    world = dist.get_world_size(group)
    method = default_method() if method is None else method
    if method not in METHODS:
        raise ValueError('method must be one of %r, got %r' % (METHODS, method))
    if world == 1 and not force_collective:
        return local
    rank = dist.get_rank(group)
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
        _exchange(out, phys if phys.is_contiguous() else phys.contiguous(), rank, world, group, method)
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
        _exchange(buf, padded, rank, world, group, method)
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
