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


def _physical(local):
    """(frame-major contiguous buffer, transposed?) of a (.., M, T) output view or a plain row-major tensor — the layout rule of
    :func:`all_gather_batch`, decided from the strides alone so that every rank takes the same branch."""
    transposed = local.dim() >= 3 and not _row_major(local) and _row_major(local.transpose(-2, -1))
    phys = local.transpose(-2, -1) if transposed else (local if _row_major(local) else local.contiguous())
    return (phys if phys.is_contiguous() else phys.contiguous()), transposed


class ChunkedAllGather(object):
    """The one all-gather of :func:`all_gather_batch`, cut into row chunks so that it OVERLAPS the computation.

    At BASELINE configs[2] a rank computes its 338.7 MB shard in ~0.9 ms and the exchange takes 5.5 - 8 ms over xGMI
    (DESIGN.md §6): run back to back the step is their sum.  Rows are independent (reference functional.py:89-91), so the shard
    is produced in ``chunks`` pieces and the exchange of piece k is posted — asynchronously, on the communicator's own stream —
    as soon as piece k exists, while piece k + 1 is being computed: the step becomes ~max(compute, exchange) + one piece.

    Every rank cuts ``[0, biggest)`` (``biggest`` = the largest shard) into the same ``chunks`` pieces (:func:`shard_bounds`);
    a rank whose shard is a row shorter simply has a shorter (or empty) last piece.  ``add(k, local)`` takes this rank's rows of
    piece k (the pipeline's output view for them), ``finish()`` waits and returns the gathered batch in rank order — the same
    tensor :func:`all_gather_batch` returns.  Methods: ``p2p`` posts the piece straight into its final place in every peer's
    output (no staging; sizes are known to all ranks, empty messages are skipped on both sides); ``rccl`` gathers the piece with
    ``all_gather_into_tensor`` into a staging block (pieces of short ranks padded) and moves it to its place with one strided
    copy on a side stream.
    """

    def __init__(self, total_rows, chunks, group=None, method=None, force_collective=False):
        self.group = group
        self.force = force_collective            # a group of one still goes through the communicator (RCCL smoke on a 1-GPU box)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.method = default_method() if method is None else method
        if self.method not in METHODS:
            raise ValueError('method must be one of %r, got %r' % (METHODS, self.method))
        self.total_rows = int(total_rows)
        bounds = [shard_bounds(self.total_rows, self.world, r) for r in range(self.world)]
        self.sizes = [e - b for b, e in bounds]
        self.offsets = [b for b, _ in bounds]
        self.biggest = max(self.sizes) if self.sizes else 0
        self.chunks = max(1, min(int(chunks), max(1, self.biggest)))
        self.pieces = [shard_bounds(self.biggest, self.chunks, k) for k in range(self.chunks)]
        self.out = None
        self.transposed = False
        self._pending = []
        self._side = None
        self._next = 0

    def local_rows(self, k, rank=None):
        """[begin, end) of piece k inside the shard of ``rank`` (default: this rank) — clipped to the rows it owns"""
        n = self.sizes[self.rank if rank is None else rank]
        b, e = self.pieces[k]
        return min(b, n), min(e, n)

    def _setup(self, phys, transposed):
        self.transposed = transposed
        self.out = torch.empty((self.total_rows,) + tuple(phys.shape[1:]), dtype=phys.dtype, device=phys.device)
        if phys.is_cuda and self.method == 'rccl':
            self._side = torch.cuda.Stream(device=phys.device)

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def add(self, k, local):
        from ._lazy import realize
        if k != self._next:
            raise ValueError('pieces must be added in order: expected %d, got %d' % (self._next, k))
        self._next += 1
        phys, transposed = _physical(realize(local))
        if self.out is None:
            self._setup(phys, transposed)
        elif tuple(phys.shape[1:]) != tuple(self.out.shape[1:]) or phys.dtype != self.out.dtype or transposed != self.transposed \
                or phys.device != self.out.device:
            # (the first piece fixed the gathered tensor's trailing shape, dtype and layout: a later piece that differs would be
            # placed wrongly without a word)
            raise ValueError('piece %d has trailing shape %r, dtype %s, transposed=%r on %s; the first piece had %r, %s, %r on %s'
                             % (k, tuple(phys.shape[1:]), phys.dtype, transposed, phys.device, tuple(self.out.shape[1:]),
                                self.out.dtype, self.transposed, self.out.device))
        lb, le = self.local_rows(k)
        if phys.shape[0] != le - lb:
            raise ValueError('piece %d of rank %d has %d rows, expected %d' % (k, self.rank, phys.shape[0], le - lb))
        if self.world == 1 and not self.force:
            self.out[lb:le].copy_(phys)
            return
        if self.method == 'p2p':
            me = self.offsets[self.rank]
            if le > lb:
                self.out[me + lb:me + le].copy_(phys)
            ops = []
            for step in range(1, self.world):
                to, frm = (self.rank + step) % self.world, (self.rank - step) % self.world
                if le > lb:
                    ops.append(dist.P2POp(dist.isend, phys, self._peer(to), self.group))
                fb, fe = self.local_rows(k, frm)
                if fe > fb:
                    ops.append(dist.P2POp(dist.irecv, self.out[self.offsets[frm] + fb:self.offsets[frm] + fe], self._peer(frm), self.group))
            if ops:
                self._pending.append((dist.batch_isend_irecv(ops), phys))          # (phys stays referenced until the sends are done)
            return
        # rccl: one all_gather_into_tensor per piece into a staging block, then to its place
        b, e = self.pieces[k]
        c = e - b
        if c == 0:
            return
        if le - lb == c:
            send = phys
        else:
            send = phys.new_empty((c,) + tuple(phys.shape[1:]))
            send[:le - lb] = phys
        stage = torch.empty((self.world * c,) + tuple(phys.shape[1:]), dtype=phys.dtype, device=phys.device)
        work = dist.all_gather_into_tensor(stage, send, group=self.group, async_op=True)
        self._pending.append((work, stage, send, k))
        if self._side is not None:                     # the move runs behind the collective on a side stream: the compute stream never waits
            stage.record_stream(self._side)
            with torch.cuda.stream(self._side):
                work.wait()
                self._place(stage, k)

    def _place(self, stage, k):
        b, e = self.pieces[k]
        c = e - b
        if all(n == self.sizes[0] for n in self.sizes):
            n = self.sizes[0]
            trail = tuple(self.out.shape[1:])
            self.out.view((self.world, n) + trail)[:, b:e].copy_(stage.view((self.world, c) + trail))
        else:
            for r in range(self.world):
                rb, re_ = self.local_rows(k, r)
                if re_ > rb:
                    self.out[self.offsets[r] + rb:self.offsets[r] + re_].copy_(stage[r * c:r * c + (re_ - rb)])

    def finish(self):
        if self._next != self.chunks:
            raise ValueError('%d of %d pieces were added' % (self._next, self.chunks))
        for item in self._pending:
            if self.method == 'p2p':
                for req in item[0]:
                    req.wait()
            elif self._side is None:
                item[0].wait()
                self._place(item[1], item[3])
        if self._side is not None:
            torch.cuda.current_stream(self.out.device).wait_stream(self._side)
        self._pending = []
        return self.out.transpose(-2, -1) if self.transposed else self.out


class ShardedPipeline(torch.nn.Module):
    """Wrap a feature pipeline: run it on this rank's batch shard, optionally all-gather the result."""

    def __init__(self, pipeline, gather=True, group=None, overlap=False, chunks=8, method=None, force_collective=False):
        """``overlap=True``: the shard is computed in ``chunks`` row pieces and the exchange of each piece runs while the
        next one is computed (:class:`ChunkedAllGather`); the result is the same tensor."""
        super(ShardedPipeline, self).__init__()
        self.pipeline = pipeline
        self.gather = gather
        self.group = group
        self.overlap = overlap
        self.chunks = chunks
        self.method = method
        self.force_collective = force_collective

    def forward(self, whole_batch):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        if self.gather and self.overlap and whole_batch.shape[0] > 0:
            return self._forward_overlapped(whole_batch, world, rank)
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
        return all_gather_batch(local, total_rows=whole_batch.shape[0], group=self.group, method=self.method,
                                force_collective=self.force_collective)

    def _forward_overlapped(self, whole_batch, world, rank):
        shard = shard_batch(whole_batch, world, rank)
        g = ChunkedAllGather(whole_batch.shape[0], self.chunks, group=self.group, method=self.method, force_collective=self.force_collective)
        from ._lazy import realize
        last = None
        for k in range(g.chunks):
            lb, le = g.local_rows(k)
            if le > lb:
                piece = last = realize(self.pipeline(shard[lb:le]))
            elif last is not None:
                # a short shard has nothing in this piece: zero rows laid out like the pieces before (no further pipeline call — a
                # stateful module must not see rows this rank does not own, nor more calls than the rows it owns explain)
                piece = last[:0]
            else:
                # an EMPTY shard has no piece to take the layout from: ONE call on a borrowed row, its zero rows reused for every piece
                piece = last = realize(self.pipeline(whole_batch[:1]))[:0]
            g.add(k, piece)
        return g.finish()
