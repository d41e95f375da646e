"""
Device-resident env for a learner that lives on the same GPU: actions come in as a CUDA/HIP ``torch.Tensor`` and are
consumed in place, observations / rewards / dones come back as views of one device tensor -- no host copy, no sync
(SURVEY.md section 8f, rank 2).  torch is only the tensor container here: pointers go through the C ABI
(``dockauv_step`` with device pointers on torch's current stream).

The reference's caller is SB3's rollout loop (train.py:64-71): per step ``env.step(a)`` with NumPy arrays on the host.
Here the same loop is::

    env = TorchDocking3d(TRAIN_CONFIG, num_envs=65536, scenario="ObstaclesCurrentDocking3d")
    obs = env.reset()
    for _ in range(n_steps):
        actions = policy(obs)                    # torch, on device
        obs, reward, done = env.step(actions)    # views, valid until the next step()
"""
from __future__ import annotations

from typing import Optional

from ..config.env_config import BASE_CONFIG
from .batched import BatchedDocking3d


class TorchDocking3d:
    def __init__(self, env_config: dict = BASE_CONFIG, num_envs: int = 4096, scenario: str = "SimpleDocking3d",
                 device: int = 0, reset_mode: str = "device", device_seed: int = 0, vehicles=None,
                 double_buffer: bool = True, **kw):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("TorchDocking3d needs an MI355X: no HIP device visible (there is no CPU fallback)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.batch = BatchedDocking3d(env_config, num_envs=num_envs, scenario=scenario, device=device, precision="f32",
                                      reset_mode=reset_mode, device_seed=device_seed, rng="batched", vehicles=vehicles,
                                      **kw)
        self.num_envs, self.n_obs, self.n_u = self.batch.num_envs, self.batch.n_observations, self.batch.n_u
        self.observation_space, self.action_space = self.batch.observation_space, self.batch.action_space
        # the kernel writes packed rows [obs | reward | done]; two buffers so that the views handed out by step t
        # stay intact while step t + 1 is being written (a policy may still be reading them)
        n_buf = 2 if double_buffer else 1
        self._packed = [torch.zeros((self.num_envs, self.n_obs + 2), device=self.device, dtype=torch.float32)
                        for _ in range(n_buf)]
        self._terminal = None
        self._i = 0
        # mixed batches built with sort_vehicles=True: row j of every tensor handed in / out belongs to the caller's env
        # perm[j] (kind-sorted on the device, the caller's order within a kind); identity otherwise
        self.perm = torch.as_tensor(self.batch.perm, device=self.device)
        self.vehicles_by_row = None if vehicles is None else [list(vehicles)[int(i)] for i in self.batch.perm]

    def reset(self, seed: Optional[int] = None):
        """All envs: new episodes; returns the reference's reset observation (zeros, docking3d.py:269,322)."""
        self.batch.reset(seed=seed)
        self._packed[self._i % len(self._packed)].zero_()
        return self._packed[self._i % len(self._packed)][:, : self.n_obs]

    def step(self, actions, want_terminal_obs: bool = False):
        """actions: float32 [num_envs, n_u] on this device (contiguous).  Returns (obs, reward, done) views;
        with auto-reset the rows of finished envs already hold the reset observation and, if asked for,
        ``self.terminal_observation`` the last one of the finished episode."""
        torch = self.torch
        if actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous() \
                or tuple(actions.shape) != (self.num_envs, self.n_u):
            raise ValueError(f"actions must be a contiguous float32 [{self.num_envs}, {self.n_u}] tensor on {self.device}")
        self._i += 1
        out = self._packed[self._i % len(self._packed)]
        term_ptr = 0
        if want_terminal_obs:
            if self._terminal is None:
                self._terminal = torch.zeros((self.num_envs, self.n_obs), device=self.device, dtype=torch.float32)
            term_ptr = self._terminal.data_ptr()
        self.batch.step_device(actions.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream,
                               packed=True, terminal_obs_ptr=term_ptr)
        # the kernels' sticky status word (a tail role that gave up waiting: rows since are invalid) is host-coherent
        # memory: looked at every step without a synchronisation; what it shows belongs to steps that have already run
        self.batch.poll_status()
        return out[:, : self.n_obs], out[:, self.n_obs], out[:, self.n_obs + 1] > 0.5

    @property
    def terminal_observation(self):
        return self._terminal

    def close(self) -> None:
        """Raises DockAUVError (after releasing everything) if a step kernel of this env reported an internal time-out."""
        err = None
        try:
            if getattr(self.batch, "_handle", None) is not None and self.batch._handle.value:
                self.batch.synchronize()
        except Exception as e:
            err = e
        self.batch.close()
        if err is not None:
            raise err


class ShardedTorchDocking3d:
    """
    The same loop over several GPUs, one process per GPU, for a single learner: ``num_envs`` is
    the TOTAL over the ranks of the ``torch.distributed`` group, rank r owns the contiguous range
    ``shard_range(num_envs, world, r)``; ``step`` takes the learner's action batch (global ``[num_envs, n_u]`` -- the
    rank's rows are sliced out in place -- or just the local rows), steps the rank's shard and returns GLOBAL
    ``(obs, reward, done)``: every rank's packed rows, gathered by one RCCL all-gather (``transport="rccl"``, the
    default) or by the peer-to-peer transport (``"p2p"``, gym_dockauv_amd/parallel.py: P2PGather, closed loop: all rows
    of step t are there when the returned views are read on the current stream; accepted only after a bit-exact
    start-up check against RCCL, watched for late peers).  The views stay valid until the next step (rccl) / for two
    further steps (p2p).

        dist.init_process_group("nccl", ...)
        env = ShardedTorchDocking3d(TRAIN_CONFIG, num_envs=8 * 32768, scenario="ObstaclesDocking3d", device=local_rank)
        obs = env.reset()
        obs, reward, done = env.step(policy(obs))        # obs: [262144, n_obs] on every rank
    """

    def __init__(self, env_config: dict = BASE_CONFIG, num_envs: int = 4096, scenario: str = "SimpleDocking3d",
                 device: int = 0, transport: str = "rccl", group=None, device_seed: int = 0, host_seed: Optional[int] = None,
                 vehicles=None, verify_steps: int = 4, check_every: int = 8, p2p_max_spins: int = 8_000_000,
                 gather_dtype: str = "f32", **kw):
        """transport: "rccl" (default: one all_gather_into_tensor per step, what BASELINE.json names) or "p2p" (the
        peer-to-peer push of gym_dockauv_amd/parallel.py).  p2p is only kept if, on EVERY rank, `verify_steps` gathers of
        test rows equal an RCCL all-gather of the same rows bit for bit; otherwise the env falls back to RCCL
        (``self.transport`` says which one runs, ``self.transport_note`` why).  With p2p the time-out word of the
        transport is read every `check_every` steps and in close(): a peer whose step stamp did not arrive within the
        spin bound makes step() raise DockAUVError on every rank that waited for it (the rows it would have returned
        are stale) -- the job is then to be restarted as fresh processes.  The rows of up to `check_every` - 1 steps
        returned BEFORE the raise may already have been stale: a learner discards its last `check_every` steps on that
        error (default 8: one 8-byte read-back every eighth step).
        gather_dtype: "f32" (default: the gathered observations are the kernel's, bit for bit) or "bf16" (RCCL transport:
        the kernel writes the observation columns as bfloat16, round to nearest even, and the links carry half the
        bytes; step() then returns a bfloat16 observation view; reward / done stay float32)."""
        import numpy as np
        import torch
        import torch.distributed as dist
        from ..parallel import P2PShardedStepper, ShardedStepper, shard_range
        from .._capi import DockAUVError
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("ShardedTorchDocking3d needs an MI355X: no HIP device visible (there is no CPU fallback)")
        if transport not in ("p2p", "rccl"):
            raise ValueError("transport must be 'p2p' or 'rccl'")
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        if num_envs % self.world:
            raise ValueError("num_envs must be a multiple of the number of ranks (equal shards)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.num_envs = int(num_envs)
        self.first, self.n_local = shard_range(self.num_envs, self.world, self.rank)
        # Mixed batches with sort_vehicles=True (a keyword passed on to the shard's batch): every rank keeps ITS contiguous
        # range of the caller's envs and sorts it by vehicle kind on its device; global row j of the
        # gathered tensors -- and of a global action batch -- then belongs to the caller's env perm[j] (computed identically
        # on every rank from the full vehicle list; identity without the option)
        perm = np.arange(self.num_envs, dtype=np.int64)
        if vehicles is not None:
            vehicles = list(vehicles)
            if len(vehicles) != self.num_envs:
                raise ValueError("len(vehicles) must equal num_envs (the total over all ranks)")
            if kw.get("sort_vehicles"):
                kinds = np.array([0 if v == "BlueROV2" else 1 for v in vehicles])
                for r in range(self.world):
                    f, n = shard_range(self.num_envs, self.world, r)
                    perm[f:f + n] = f + np.argsort(kinds[f:f + n], kind="stable")
            self.vehicles_by_row = [vehicles[int(i)] for i in perm]
            vehicles = vehicles[self.first:self.first + self.n_local]
        else:
            self.vehicles_by_row = None
        self.perm = perm
        self.batch = BatchedDocking3d(env_config, num_envs=self.n_local, scenario=scenario, device=device, precision="f32",
                                      reset_mode="device", device_seed=device_seed + self.rank, rng="batched",
                                      vehicles=vehicles, **kw)
        if host_seed is not None:
            self.batch._gen = np.random.default_rng(host_seed + self.rank)
        self.n_obs, self.n_u = self.batch.n_observations, self.batch.n_u
        self.observation_space, self.action_space = self.batch.observation_space, self.batch.action_space
        self.transport, self.transport_note = transport, None
        if gather_dtype not in ("f32", "bf16"):
            raise ValueError("gather_dtype must be 'f32' or 'bf16'")
        if gather_dtype == "bf16" and transport != "rccl":
            raise ValueError("gather_dtype='bf16' is implemented for the RCCL transport")
        self.gather_dtype = gather_dtype
        packed = "bf16" if gather_dtype == "bf16" else True
        row_words = self.batch.packed_row_words(packed)
        self.check_every = max(1, int(check_every))
        self._steps = 0
        self._DockAUVError = DockAUVError

        def step_fn(actions_local, out_local):
            self.batch.step_device(actions_local.data_ptr(), out_local.data_ptr(),
                                   stream=torch.cuda.current_stream().cuda_stream, packed=packed)

        def rccl_stepper():
            return ShardedStepper(self.n_local, row_words, step_fn, self.device, world=self.world,
                                  rank=self.rank, group=group, overlap=False, gather_dtype=gather_dtype)

        if transport == "p2p":
            p2p, note = None, None
            try:
                p2p = P2PShardedStepper(self.n_local, self.n_obs + 2, step_fn, self.device, world=self.world,
                                        rank=self.rank, group=group, overlap=False, max_spins=p2p_max_spins)
            except (DockAUVError, ValueError) as e:      # (P2PGather agrees across ranks: all raise or none)
                note = f"p2p set-up failed ({e}): RCCL used"
            if p2p is not None and self.world > 1 and verify_steps > 0:
                # the transport itself, without stepping the envs: test rows through the gather vs. an RCCL all-gather
                ok = 1
                ref = torch.empty((self.world * self.n_local, self.n_obs + 2), device=self.device, dtype=torch.float32)
                stream = torch.cuda.current_stream().cuda_stream
                gen = torch.Generator(device=self.device)
                gen.manual_seed(1000 + self.rank)
                for i in range(int(verify_steps)):
                    rows = p2p.rows2[p2p.gather.t & 1]
                    rows.copy_(torch.rand(rows.shape, device=self.device, generator=gen))
                    buf = p2p.gather.gather(rows.data_ptr(), stream)
                    dist.all_gather_into_tensor(ref, rows, group=group)
                    ok &= int(torch.equal(buf.view(torch.int32), ref.view(torch.int32)))
                ok &= int(p2p.gather.timed_out() == 0)
                flag = torch.tensor([ok], device=self.device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if int(flag.item()) != 1:
                    note = "p2p rows differed from the RCCL all-gather (or a stamp timed out) in the start-up check: RCCL used"
                    p2p.close()
                    p2p = None
            if p2p is None:
                self.transport, self.transport_note = "rccl", note
                self.stepper = rccl_stepper()
            else:
                self.stepper = p2p
        else:
            self.stepper = rccl_stepper()
        self._zeros = None

    def _check_transport(self) -> None:
        """p2p: a peer's stamp that did not arrive within the spin bound is sticky in the status word (every later wait
        returns at once and the gather buffers would hand out rows of an earlier step)."""
        if self.transport != "p2p":
            return
        late = self.stepper.gather.timed_out()
        if late:
            raise self._DockAUVError(f"rank {self.rank}: the step stamps of rank(s) {[r for r in range(self.world) if late >> r & 1]} "
                                     f"did not arrive within the spin bound after {self._steps} steps: the gathered rows are "
                                     "stale; restart the job")

    def reset(self, seed: Optional[int] = None):
        """All envs of this rank's shard: new episodes; returns the reference's reset observation for ALL envs (zeros,
        docking3d.py:269,322)."""
        self.batch.reset(seed=seed)
        if self._zeros is None:
            self._zeros = self.torch.zeros((self.num_envs, self.n_obs), device=self.device, dtype=self.torch.float32)
        return self._zeros

    def step(self, actions):
        """actions: contiguous float32 on this device, [num_envs, n_u] (global; this rank's rows are used) or
        [n_local, n_u].  Returns global (obs [num_envs, n_obs], reward [num_envs], done [num_envs] bool) views."""
        torch = self.torch
        if actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            raise ValueError(f"actions must be a contiguous float32 tensor on {self.device}")
        if tuple(actions.shape) == (self.num_envs, self.n_u) and self.world > 1:
            actions = actions[self.first:self.first + self.n_local]
        elif tuple(actions.shape) != (self.n_local, self.n_u):
            raise ValueError(f"actions must be [{self.num_envs}, {self.n_u}] (global) or [{self.n_local}, {self.n_u}] (local)")
        buf = self.stepper.step(actions)
        self._steps += 1
        self.batch.poll_status()   # (the step kernels' own status word: host-coherent, no synchronisation)
        if self._steps % self.check_every == 0:
            self._check_transport()
        if self.gather_dtype == "bf16":
            from ..parallel import ShardedStepper
            return ShardedStepper.split_bf16(buf, self.n_obs)
        return buf[:, : self.n_obs], buf[:, self.n_obs], buf[:, self.n_obs + 1] > 0.5

    def close(self) -> None:
        err = None
        try:
            self._check_transport()
            if getattr(self.batch, "_handle", None) is not None and self.batch._handle.value:
                self.batch.synchronize()   # (reports the kernels' status word as well)
        except Exception as e:          # still release everything; re-raise afterwards
            err = e
        if hasattr(self.stepper, "close"):
            self.stepper.close()
        self.batch.close()
        if err is not None:
            raise err
