"""Multi-GPU plumbing: env sharding (no data-path collective) + the one real exchange step of the AMP path,
the all-gather of the AMP replay minibatch (RCCL over xGMI through torch.distributed; gloo on CPU in tests).

The reference has no first-party collective: ``train.py:54-58,183-196`` only pins one env shard + agent replica
per GPU and skrl all-reduces gradients.  What the sharded *hot path* needs (BASELINE.json north_star) is that each
rank's discriminator minibatch sees replay rows from every rank's envs.
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


class ReplayAllGather:
    """Draw ``rows`` random rows of this rank's AMP observations and all-gather them (one discriminator minibatch).

    ``slots`` independent (shard, output) buffer pairs let that many gathers be in flight: ``start()`` enqueues the
    row draw on the current stream and launches the collective asynchronously on RCCL's stream, so it runs under the
    following env steps / GEMMs; ``wait_all()`` (or the next ``start()`` on a busy slot) joins them.  ``__call__`` is
    the blocking form.

    ``minibatches`` > 1 fuses that many minibatches of one agent update into ONE collective (same bytes, one launch:
    xGMI links are point-to-point, so fewer and larger messages use them better than twelve 2.7-MB ones): each rank
    contributes ``[minibatches, rows, C]`` and the gathered tensor is ``[world, minibatches, rows, C]``;
    :meth:`minibatch_blocks` returns minibatch i as its ``world`` contiguous ``[rows, C]`` blocks."""

    def __init__(self, amp_obs: torch.Tensor, rows: int, seed: int = 0, group=None, slots: int = 1, minibatches: int = 1):
        self.minibatches, self.rows_per_minibatch = int(minibatches), int(rows)
        rows = int(rows) * self.minibatches
        self.amp_obs, self.rows, self.group = amp_obs, int(rows), group
        self.gen = torch.Generator(device=amp_obs.device).manual_seed(seed)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        mk = lambda n: torch.empty((n, amp_obs.shape[1]), dtype=amp_obs.dtype, device=amp_obs.device)  # noqa: E731
        self.shards = [mk(self.rows) for _ in range(slots)]
        self.outs = [mk(world * self.rows) for _ in range(slots)]
        self.works = [None] * slots
        self._next = 0
        self._async = dist.is_initialized() and world > 1 and (not amp_obs.is_cuda or dist.get_backend(group) == "nccl")

    def _draw(self, slot: int):
        idx = torch.randint(0, self.amp_obs.shape[0], (self.rows,), generator=self.gen, device=self.amp_obs.device)
        torch.index_select(self.amp_obs, 0, idx, out=self.shards[slot])

    def __call__(self) -> torch.Tensor:
        self._draw(0)
        return allgather_minibatch(self.shards[0], self.outs[0], self.group)

    def start(self) -> int:
        """Begin one gather in the next slot (waits first if that slot is still in flight); returns the slot."""
        slot = self._next
        self._next = (self._next + 1) % len(self.shards)
        if self.works[slot] is not None:
            self.works[slot].wait()
            self.works[slot] = None
        self._draw(slot)
        if self._async:
            self.works[slot] = dist.all_gather_into_tensor(self.outs[slot], self.shards[slot], group=self.group, async_op=True)
        else:
            allgather_minibatch(self.shards[slot], self.outs[slot], self.group)
        return slot

    def wait_all(self):
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None

    def minibatch_blocks(self, slot: int, i: int):
        """Minibatch ``i`` of a fused gather: ``world`` contiguous ``[rows, C]`` views, rank order (no copy)."""
        out = self.result(slot)
        world = out.shape[0] // self.rows
        full = out.view(world, self.minibatches, self.rows_per_minibatch, out.shape[1])
        return [full[r, i] for r in range(world)]

    def result(self, slot: int) -> torch.Tensor:
        if self.works[slot] is not None:
            self.works[slot].wait()
            self.works[slot] = None
        return self.outs[slot]
