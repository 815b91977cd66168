"""Multi-GPU plumbing: env sharding (no data-path collective) + the one real exchange step of the AMP path, the
all-gather of the discriminator-update minibatch rows (RCCL over xGMI through torch.distributed; gloo on CPU in tests).

The reference has no first-party collective: ``train.py:54-58,183-196`` pins one env shard + one agent replica per GPU and
skrl [third-party] all-reduces the gradients of every minibatch.  BASELINE.json's north_star replaces that by ONE exchange of
rows: every rank contributes ``batch / world`` rows of each group of every training step of an agent update, the ranks
all-gather them, and every replica then takes the SAME optimizer steps on the SAME global minibatches -- the replicas stay
bit-identical with no gradient traffic at all (:class:`UpdateExchange` is the collective, ``engine.AmpDiscriminatorUpdate(...,
group=...)`` its consumer).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(num_envs: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous env block [lo, hi) of ``rank``; the first ``num_envs % world_size`` ranks get one extra env."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, extra = divmod(int(num_envs), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_env_ids(local_ids: torch.Tensor, num_envs: int, world_size: int, rank: int) -> torch.Tensor:
    """Reset ids of a shard are local; the global id is local + the shard's first env."""
    return local_ids + shard_bounds(num_envs, world_size, rank)[0]


def _world(group) -> tuple[int, int]:
    if group is None or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def allgather_minibatch(shard: torch.Tensor, out: torch.Tensor | None = None, group=None,
                        force_collective: bool = False) -> torch.Tensor:
    """[B_loc, C] per rank -> [world * B_loc, C], rank-major.  One collective, no staging copy: 2.72 MB per rank at
    B_loc = 4096, C = 166.  The 8 GPUs of a node are fully connected over xGMI, so the message is small enough
    that RCCL's direct algorithm (one link per peer) applies; nothing here is ring-specific."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if shard.dim() != 2 or not shard.is_contiguous():
        raise ValueError("shard must be a contiguous [rows, cols] tensor")
    if out is None:
        out = torch.empty((world * shard.shape[0], shard.shape[1]), dtype=shard.dtype, device=shard.device)
    elif tuple(out.shape) != (world * shard.shape[0], shard.shape[1]) or not out.is_contiguous():
        raise ValueError("out has the wrong shape")
    if world == 1 and not (force_collective and dist.is_initialized()):
        out.copy_(shard)
    elif shard.is_cuda and dist.get_backend(group) != "nccl":
        # rehearsal path only (gloo has no device collectives): stage through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, shard.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, shard, group=group)
    return out


class UpdateExchange:
    """The ONE collective of a discriminator update (``steps`` = learning_epochs x mini_batches training steps of ``groups``
    row groups each: policy, replay, motion).

    Every rank fills :attr:`contrib` ``[steps, groups, rows_per_rank, C]`` with its share of every minibatch -- the rows
    ``[rank * rows_per_rank, (rank + 1) * rows_per_rank)`` of each -- :meth:`start` launches one all-gather of the whole block
    (asynchronous on RCCL's stream; xGMI links are point-to-point, so one ``steps * groups * rows_per_rank * C * 4``-byte
    message per peer uses them better than ``steps * groups`` small ones) and :meth:`finish` returns :attr:`batches`
    ``[steps, groups, world * rows_per_rank, C]``: the global minibatches, identical on every rank.  Row ``j`` of a global
    minibatch comes from rank ``j // rows_per_rank``, local row ``j % rows_per_rank`` (:meth:`source_of`).

    The gathered block arrives rank-major (``[world, steps, groups, r, C]``); ``finish`` is the one strided copy that makes every
    (step, group) batch a contiguous ``[world * r, C]`` tensor the trainer can take by pointer."""

    def __init__(self, steps: int, groups: int, rows_per_rank: int, cols: int, device, dtype=torch.float32, group=None,
                 force_collective: bool = False):
        """``force_collective``: run the real collective even for a group of one rank (how the RCCL path is exercised on a
        one-GPU box); by default a world of one hands the contribution back as the batch set, with no copy."""
        self.group = group
        self.world, self.rank = _world(group)
        self._collective = self.world > 1 or (bool(force_collective) and group is not None and dist.is_initialized())
        self.steps, self.groups, self.rows_per_rank, self.cols = int(steps), int(groups), int(rows_per_rank), int(cols)
        shape = (self.steps, self.groups, self.rows_per_rank, self.cols)
        self.contrib = torch.empty(shape, dtype=dtype, device=device)
        if self._collective:
            self.batches = torch.empty((self.steps, self.groups, self.world * self.rows_per_rank, self.cols), dtype=dtype, device=device)
            self._gathered = torch.empty((self.world,) + shape, dtype=dtype, device=device)
        else:  # world 1: the contribution IS the batch set (no gather buffer, no copy)
            self.batches, self._gathered = self.contrib, None
        self._work = None
        self._pending = False

    @property
    def first_row(self) -> int:
        """Where this rank's rows sit in every global minibatch."""
        return self.rank * self.rows_per_rank

    def source_of(self, row: int) -> tuple[int, int]:
        """(rank, local row) a row of a global minibatch came from."""
        return divmod(int(row), self.rows_per_rank)

    @property
    def bytes_per_rank(self) -> int:
        return self.contrib.numel() * self.contrib.element_size()

    def start(self) -> None:
        """Launch the all-gather of :attr:`contrib` (whatever the current stream has enqueued into it is ordered first)."""
        if self._pending:
            raise RuntimeError("UpdateExchange.start(): the previous exchange was not finished")
        self._pending = True
        if not self._collective:
            return
        if self.contrib.is_cuda and dist.get_backend(self.group) != "nccl":
            # rehearsal path only (gloo has no device collectives): stage through the host, synchronously
            host = torch.empty(self._gathered.shape, dtype=self._gathered.dtype)
            dist.all_gather_into_tensor(host.view(-1), self.contrib.cpu().view(-1), group=self.group)
            self._gathered.copy_(host)
            return
        self._work = dist.all_gather_into_tensor(self._gathered.view(-1), self.contrib.view(-1), group=self.group, async_op=True)

    def finish(self) -> torch.Tensor:
        """Join the collective and return the global minibatches ``[steps, groups, world * rows_per_rank, C]``."""
        if not self._pending:
            raise RuntimeError("UpdateExchange.finish() without start()")
        self._pending = False
        if not self._collective:
            return self.batches
        if self._work is not None:
            self._work.wait()  # device tensors: a stream dependency, not a host block
            self._work = None
        S, G, W, r, C = self.steps, self.groups, self.world, self.rows_per_rank, self.cols
        self.batches.view(S, G, W, r, C).copy_(self._gathered.permute(1, 2, 0, 3, 4))
        return self.batches
