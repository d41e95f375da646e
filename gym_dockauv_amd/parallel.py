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
                 group=None, overlap: bool = True, n_buffers: int = 2, gather_dtype: str = "f32"):
        """gather_dtype "bf16": the rows the kernel writes (and the links carry) hold the observation columns as bfloat16
        pairs -- `row_len` is then the row length in 32-bit words, ceil(n_obs / 2) + 2 (BatchedDocking3d.packed_row_words,
        step_fn launches with packed="bf16"); reward and done stay float32.  Halves the bytes over xGMI; the learner gets
        observations rounded to nearest even bfloat16 (what a bf16 policy network would make of them anyway)."""
        import torch
        self.torch = torch
        self.n_local, self.row_len, self.world, self.rank = int(n_local), int(row_len), int(world), int(rank)
        if gather_dtype not in ("f32", "bf16"):
            raise ValueError("gather_dtype must be 'f32' or 'bf16'")
        self.gather_dtype = gather_dtype
        self.step_fn = step_fn
        self.group = group
        self.overlap = bool(overlap) and world > 1
        # (bf16 rows are addressed as 32-bit words: float32 storage, reinterpreted by split_bf16)
        self.bufs = [torch.zeros((world * n_local, row_len), device=device, dtype=torch.float32)
                     for _ in range(n_buffers if self.overlap else 1)]
        self.bytes_per_rank_per_step = self.n_local * self.row_len * 4
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

    @staticmethod
    def split_bf16(buf, n_obs: int):
        """The same for gather_dtype "bf16": obs [N, n_obs] as a bfloat16 view (no copy), reward [N] float32, done [N]."""
        import torch
        npair = (n_obs + 1) // 2
        obs = buf[:, :npair].view(torch.bfloat16)[:, :n_obs]
        return obs, buf[:, npair], buf[:, npair + 1] > 0.5


class _DevArray:
    """Minimal ``__cuda_array_interface__`` carrier so that torch can view memory the C ABI allocated."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class P2PGather:
    """
    Peer-to-peer gather of the packed rows (include/dockauv.h, "Multi-GPU" block): every rank pushes its rows into
    its slice of EVERY rank's gather buffer with one copy kernel over xGMI and raises a step stamp at its peers; a
    one-wave kernel waits (bounded) for the peers' stamps.  No collective-library call sits in the step loop.

    Buffers rotate: step t lands in buffer t % n_buffers.  ``lag = 0`` (closed loop: the learner reads step t before
    it emits the actions of step t + 1): the wait of step t is for stamp t, two buffers suffice -- a peer's stamp
    t - 1 proves it has consumed buffer (t - 2) % 2, provided the consumer runs on the step stream (it does: the
    actions depend on it).  ``lag = 1`` (open loop / pipelined): the wait of step t is for stamp t - 1, the fabric
    transfer overlaps the next step kernel, four buffers.

    The gather buffers are ordinary device memory (`uncached=False`): peers' rows arrive by system-scope write-through
    stores that are acknowledged before the stamp is raised, the stamp is read from fine-grained memory, and the consumer is a later
    kernel -- the same ordering RCCL relies on when peers write a receive buffer directly.  Fine-grained gather buffers
    (`uncached=True`) were measured at ~55 GB/s for the LOCAL copy alone (13 us for 4 096 envs) and are only a
    diagnostic.

    The bootstrap (exchange of the 64-byte IPC handles) uses the caller's torch.distributed group (gloo or nccl).
    """

    def __init__(self, n_local: int, row_len: int, device_index: int, world: int, rank: int, group=None, lag: int = 0,
                 n_buffers: Optional[int] = None, uncached: bool = False, max_spins: int = 8_000_000):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _capi
        self.C, self.torch, self._capi = C, torch, _capi
        self.lib = _capi.load_library()
        self.n_local, self.row_len, self.world, self.rank = int(n_local), int(row_len), int(world), int(rank)
        self.device_index = int(device_index)
        self.lag = int(lag)
        if self.lag not in (0, 1):
            raise ValueError("lag must be 0 or 1")
        self.nb = int(n_buffers) if n_buffers else (2 if self.lag == 0 else 4)
        if self.nb < (2 if self.lag == 0 else 4):
            raise ValueError("too few buffers for this lag (2 for lag 0, 4 for lag 1)")
        if world - 1 > _capi.P2P_MAX_PEERS:
            raise ValueError(f"at most {_capi.P2P_MAX_PEERS + 1} ranks")
        self.slice_bytes = self.n_local * self.row_len * 4
        if self.slice_bytes % 16:
            raise ValueError("n_local * row_len must be a multiple of 4 floats (16-byte slices)")
        self.max_spins = int(max_spins)
        self.t = 0
        self.closed = False
        self.group = group
        self._opened = []

        buf_bytes = self.nb * self.world * self.slice_bytes
        self._buf = C.c_void_p()
        self._flags = C.c_void_p()
        hbuf = C.create_string_buffer(_capi.P2P_HANDLE_BYTES)
        hflag = C.create_string_buffer(_capi.P2P_HANDLE_BYTES)
        err = None
        try:
            self._check(self.lib.dockauv_p2p_alloc(self.device_index, buf_bytes, 1 if uncached else 0,
                                                   C.byref(self._buf), hbuf), "dockauv_p2p_alloc(gather buffer)")
            # flags: uint32 [0, world) written by the peers; status = uint32 [32], [33] of the same allocation
            self._check(self.lib.dockauv_p2p_alloc(self.device_index, 256, 1, C.byref(self._flags), hflag),
                        "dockauv_p2p_alloc(flags)")
        except _capi.DockAUVError as e:
            err = e
        self._agree(err, "allocation")     # every rank raises or none does (a collective follows)
        self._status_ptr = self._flags.value + 128

        # exchange handles
        self.peer_buf = [None] * world    # base of rank r's gather buffer as mapped here (own: local pointer)
        self.peer_flags = [None] * world
        self.peer_buf[rank], self.peer_flags[rank] = self._buf.value, self._flags.value
        self._opened = []
        if world > 1:
            backend = dist.get_backend(group)
            dev = torch.device("cuda", self.device_index) if backend == "nccl" else torch.device("cpu")
            mine = torch.frombuffer(bytearray(hbuf.raw + hflag.raw), dtype=torch.uint8).to(dev)
            allh = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allh, mine, group=group)
            err = None
            try:
                for r in range(world):
                    if r == rank:
                        continue
                    raw = bytes(allh[r].cpu().numpy().tobytes())
                    for k, dst in ((0, self.peer_buf), (1, self.peer_flags)):
                        p = C.c_void_p()
                        self._check(self.lib.dockauv_p2p_open(self.device_index,
                                                              raw[k * _capi.P2P_HANDLE_BYTES:(k + 1) * _capi.P2P_HANDLE_BYTES],
                                                              C.byref(p)), f"dockauv_p2p_open(rank {r})")
                        dst[r] = p.value
                        self._opened.append(p.value)
            except _capi.DockAUVError as e:
                err = e
            self._agree(err, "mapping the peers' buffers")
        peers = [r for r in range(world) if r != rank]
        self.n_peers = len(peers)
        self._slots = (C.c_void_p * max(1, self.n_peers))(*[self.peer_flags[r] + 4 * rank for r in peers])
        # destination tables, one per buffer: this rank's slice in every rank's buffer k (own copy included)
        self._dsts = []
        for k in range(self.nb):
            off = (k * world + rank) * self.slice_bytes
            self._dsts.append((C.c_void_p * world)(*[self.peer_buf[r] + off for r in range(world)]))
        # plans of the one-kernel gather (dockauv_p2p_gather); counter = uint32 [48] of the flag allocation
        self._plans = (_capi.P2PPlan * self.nb)()
        for k in range(self.nb):
            pl = self._plans[k]
            for r in range(world):
                pl.dsts[r] = self._dsts[k][r]
            for i, r in enumerate(peers):
                pl.peer_slots[i] = self.peer_flags[r] + 4 * rank
            pl.my_flags, pl.status, pl.counter = self._flags.value, self._status_ptr, self._flags.value + 192
            pl.bytes, pl.max_spins = self.slice_bytes, self.max_spins
            pl.n_dsts, pl.n_peers, pl.world, pl.my_rank = world, self.n_peers, world, rank
        self._views = [torch.as_tensor(_DevArray(self._buf.value + k * world * self.slice_bytes,
                                                 (world * self.n_local, self.row_len), "<f4"),
                                       device=torch.device("cuda", self.device_index)) for k in range(self.nb)]
        self._status_view = torch.as_tensor(_DevArray(self._status_ptr, (2,), "<i4"),
                                            device=torch.device("cuda", self.device_index))

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            msg = self.lib.dockauv_last_error(None)
            raise self._capi.DockAUVError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

    def _agree(self, err, what: str) -> None:
        """All ranks learn whether any of them failed in this phase; then all raise (after releasing what they hold)."""
        import torch.distributed as dist
        n_bad = 1 if err is not None else 0
        if self.world > 1:
            backend = dist.get_backend(self.group)
            dev = self.torch.device("cuda", self.device_index) if backend == "nccl" else self.torch.device("cpu")
            flag = self.torch.tensor([n_bad], dtype=self.torch.int32, device=dev)
            dist.all_reduce(flag, group=self.group)
            n_bad = int(flag.item())
        if n_bad:
            self._release()
            raise self._capi.DockAUVError(f"P2PGather: {what} failed on {n_bad} rank(s)"
                                          + (f"; this rank: {err}" if err is not None else ""))

    def _release(self) -> None:
        for p in getattr(self, "_opened", []):
            self.lib.dockauv_p2p_close(p)
        self._opened = []
        for h in (self._buf, self._flags):
            if h.value:
                self.lib.dockauv_p2p_free(h)
                h.value = None

    def buffer(self, k: int):
        """torch view [world * n_local, row_len] of gather buffer k."""
        return self._views[k]

    def push(self, src_ptr: int, stream: int = 0):
        """Queue on `stream`: copy of this rank's rows (device pointer, n_local x row_len float32) into slot
        t % n_buffers of every rank, stamp t + 1 raised at the peers, wait for the peers' stamp t + 1 - lag.
        Returns the view of the buffer the step lands in."""
        k = self.t % self.nb
        self._check(self.lib.dockauv_p2p_push(src_ptr, self.slice_bytes, self._dsts[k], self.world, stream),
                    "dockauv_p2p_push")
        stamp = (self.t + 1) & 0xFFFFFFFF
        wait = (self.t + 1 - self.lag) & 0xFFFFFFFF if self.t + 1 - self.lag > 0 else 0
        if stamp == 0:
            stamp = 1     # 0 means "skip" in the ABI; after 2^32 steps one stamp repeats, harmless (>= compare)
        self._check(self.lib.dockauv_p2p_signal_wait(self._slots, self.n_peers, self._flags.value, self.world, self.rank,
                                                     stamp, wait, self.max_spins, self._status_ptr, stream),
                    "dockauv_p2p_signal_wait")
        self.t += 1
        return self._views[k]

    def gather(self, src_ptr: int, stream: int = 0):
        """`push` as ONE kernel (dockauv_p2p_gather): the block that finishes last raises and awaits the stamps."""
        k = self.t % self.nb
        stamp = ((self.t + 1) & 0xFFFFFFFF) or 1
        wait = (self.t + 1 - self.lag) & 0xFFFFFFFF if self.t + 1 - self.lag > 0 else 0
        self._check(self.lib.dockauv_p2p_gather(self.C.byref(self._plans[k]), src_ptr, stamp, wait, stream),
                    "dockauv_p2p_gather")
        self.t += 1
        return self._views[k]

    def wait(self, stream: int = 0) -> None:
        """Queue a wait for the stamps of the last pushed step (needed with lag = 1 before reading the last buffer)."""
        if self.t == 0 or self.lag == 0:
            return
        self._check(self.lib.dockauv_p2p_signal_wait(self._slots, self.n_peers, self._flags.value, self.world, self.rank,
                                                     0, self.t & 0xFFFFFFFF, self.max_spins, self._status_ptr, stream),
                    "dockauv_p2p_signal_wait")

    def timed_out(self) -> int:
        """Bit mask of ranks whose stamp did not arrive within max_spins (synchronises the device)."""
        self.torch.cuda.synchronize(self.device_index)
        return int(self._status_view[0].item()) & 0xFFFFFFFF

    def close(self) -> None:
        if self.closed:
            return
        import torch.distributed as dist
        self.closed = True
        self.torch.cuda.synchronize(self.device_index)
        if self.world > 1:
            dist.barrier(group=self.group)      # nobody unmaps while a peer may still write
        self._views, self._status_view = [], None
        for p in self._opened:
            self.lib.dockauv_p2p_close(p)
        self._opened = []
        if self.world > 1:
            dist.barrier(group=self.group)      # nobody frees while a peer still maps
        self._release()


class P2PShardedStepper:
    """ShardedStepper with the P2PGather transport.  `step` = step kernel -> rows -> one gather kernel on the current
    stream (closed loop; `lag=1`: the wait covers the previous step).  `run_sequence` = n open-loop steps queued by one
    host call (dockauv_step_gather_sequence); with `overlap` the gather of step t rides in the grid of step kernel
    t + 1."""

    def __init__(self, n_local: int, row_len: int, step_fn: Callable, device, world: int = 1, rank: int = 0,
                 group=None, overlap: bool = True, fused: bool = True, lag: int = 0, **kw):
        import torch
        self.torch = torch
        self.n_local, self.row_len, self.world, self.rank = int(n_local), int(row_len), int(world), int(rank)
        self.step_fn = step_fn
        self.device = torch.device(device)
        self.fused = bool(fused)
        self.overlap = bool(overlap)          # run_sequence: gathers ride in the next step kernel
        self.gather = P2PGather(n_local, row_len, self.device.index or 0, world, rank, group=group,
                                lag=lag, n_buffers=4, **kw)
        self.rows2 = [torch.zeros((n_local, row_len), device=self.device, dtype=torch.float32) for _ in range(2)]
        self.rows = self.rows2[0]             # rows of the last step
        self.bufs = [self.gather.buffer(k) for k in range(self.gather.nb)]
        self._gather_stream = None
        self._ride_ok = True

    def local_slice(self, buf):
        return buf[self.rank * self.n_local:(self.rank + 1) * self.n_local]

    def step(self, actions_local):
        self.rows = self.rows2[self.gather.t & 1]
        self.step_fn(actions_local, self.rows)
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        return (self.gather.gather if self.fused else self.gather.push)(self.rows.data_ptr(), s)

    def wait(self) -> None:
        self.gather.wait(self.torch.cuda.current_stream(self.device).cuda_stream)

    def make_sequence(self, env, action_ptrs):
        """StepIO array for `run_sequence`: step i reads action_ptrs[i]; valid for a start at an even OR odd global
        step (the row buffers alternate from whatever parity the sequence is run at -- fixed here at creation)."""
        from . import _capi
        t0 = self.gather.t
        ios = (_capi.StepIO * len(action_ptrs))()
        for i, a in enumerate(action_ptrs):
            ios[i].actions = a
            ios[i].obs = self.rows2[(t0 + i) & 1].data_ptr()
            ios[i].pack_reward_done = 1
        return (ios, len(action_ptrs), t0 & 1)

    def run_sequence(self, env, seq, two_streams: Optional[bool] = None, ride: Optional[bool] = None) -> None:
        """two_streams: gathers on a second stream beside the next step kernel.  Off by default: the two cross-stream
        dependencies per step cost more than they hide at every size measured on one GPU (4 096 envs: 17.4 vs 9.5 us
        per step; 32 768 envs with an 18 us local copy: 31 vs 24.5 us); it can only pay where the fabric transfer is
        several times the step kernel."""
        ios, n, parity = seq
        g = self.gather
        if (g.t & 1) != parity:
            raise ValueError("sequence was made for the other row-buffer parity")
        two_streams = bool(two_streams)
        cs = self.torch.cuda.current_stream(self.device).cuda_stream
        gs = cs
        if two_streams:
            if self._gather_stream is None:
                self._gather_stream = self.torch.cuda.Stream(device=self.device)
            gs = self._gather_stream.cuda_stream
        # ride (one stream): the gather of step t travels in the grid of step kernel t + 1, its transfer hidden behind
        # that step's arithmetic; every rank holds all rows of step t once ITS kernel t + 1 has ended (the last step
        # of the region: once the closing gather kernel has).  Without ride every gather is a kernel of its own that
        # awaits its step's stamps before the next step kernel starts.
        if ride is None:
            ride = self.overlap
        lag = 1 if (ride and not two_streams and self._ride_ok) else 0
        rc = g.lib.dockauv_step_gather_sequence(env._handle, ios, n, g._plans, g.nb, g.t, lag, cs, gs)
        if rc != 0 and lag == 1:
            msg = g.lib.dockauv_last_error(env._handle)
            if msg and b"lag 1 needs" in msg:       # float64 / general-expression kernels have no ride variant
                self._ride_ok = False
                rc = g.lib.dockauv_step_gather_sequence(env._handle, ios, n, g._plans, g.nb, g.t, 0, cs, gs)
        if rc != 0:
            msg = g.lib.dockauv_last_error(env._handle)
            raise g._capi.DockAUVError(f"dockauv_step_gather_sequence failed ({rc}): {msg.decode() if msg else '?'}")
        g.t += n
        if n:
            self.rows = self.rows2[(g.t - 1) & 1]

    split = ShardedStepper.split

    def close(self) -> None:
        self.gather.close()
