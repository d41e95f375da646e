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
        return out[:, : self.n_obs], out[:, self.n_obs], out[:, self.n_obs + 1] > 0.5

    @property
    def terminal_observation(self):
        return self._terminal

    def close(self) -> None:
        self.batch.close()
