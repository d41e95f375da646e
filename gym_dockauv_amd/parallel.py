"""
Multi-GPU layer: one process per GPU, envs sharded in contiguous ranges, ONE collective per step.

The reference is single-process; docking3d envs are independent, so the path shards with no
data-path exchange at all.  The only collective is the all-gather that concatenates every rank's packed
``[obs | reward | done]`` rows (float32 [n_local, n_obs + 2], written coalesced by the step kernel straight into
this rank's slice of the gather buffer) for a single learner: ``torch.distributed`` backend "nccl" = RCCL over xGMI.
Two gather buffers alternate, so the all-gather of step t overlaps the kernel of step t + 1 (the collective runs on
RCCL's stream; the kernel only waits for the gather that last used the buffer it is about to overwrite).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous env range [first, first + count) of `rank`; the first total % world ranks get one more env."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(int(total), world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


class ShardedStepper:
    """
    Steps this rank's shard and all-gathers the packed rows.

    :param n_local: envs on this rank (equal on every rank: all_gather_into_tensor needs equal shards)
    :param row_len: n_obs + 2
    :param step_fn: step_fn(actions_local, out_local) launches the env step of this rank ASYNCHRONOUSLY on the current
                    stream; out_local is this rank's [n_local, row_len] slice of a gather buffer
    :param group: torch.distributed process group (None = default); world size 1 without an initialised group works
    """

    def __init__(self, n_local: int, row_len: int, step_fn: Callable, device, world: int = 1, rank: int = 0,
                 group=None, overlap: bool = True, n_buffers: int = 2):
        import torch
        self.torch = torch
        self.n_local, self.row_len, self.world, self.rank = int(n_local), int(row_len), int(world), int(rank)
        self.step_fn = step_fn
        self.group = group
        self.overlap = bool(overlap) and world > 1
        self.bufs = [torch.zeros((world * n_local, row_len), device=device, dtype=torch.float32)
                     for _ in range(n_buffers if self.overlap else 1)]
        self.works: List[Optional[object]] = [None] * len(self.bufs)
        self.i = 0
        self.use_dist = world > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized())

    def local_slice(self, buf):
        return buf[self.rank * self.n_local:(self.rank + 1) * self.n_local]

    def step(self, actions_local):
        """Launch step + gather; returns the gather buffer that will hold all ranks' rows once `wait()` (or the
        returned work) completes."""
        import torch.distributed as dist
        k = self.i % len(self.bufs)
        buf = self.bufs[k]
        if self.works[k] is not None:
            self.works[k].wait()          # the kernel below overwrites this buffer: order it after its last gather
            self.works[k] = None
        out_local = self.local_slice(buf)
        self.step_fn(actions_local, out_local)
        if self.use_dist:
            w = dist.all_gather_into_tensor(buf, out_local, group=self.group, async_op=self.overlap)
            self.works[k] = w if self.overlap else None
        self.i += 1
        return buf

    def wait(self) -> None:
        for k, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[k] = None

    @staticmethod
    def split(buf, n_obs: int):
        """Views into a gathered buffer: obs [N, n_obs], reward [N], done [N] (bool)."""
        return buf[:, :n_obs], buf[:, n_obs], buf[:, n_obs + 1] > 0.5
