"""
Single-environment classes with the reference's ``gym.Env`` surface, backed by the HIP step kernel.

Drop-in for ``gym_dockauv.envs.docking3d`` (envs/docking3d.py): same class names, same constructor
(``Env(env_config)``), ``reset(seed=None, return_info=False, options=None)``, ``step(action) -> (obs, reward, done,
info)`` with the gym 0.21 4-tuple, the same ``info`` keys (docking3d.py:388-400), ``action_space`` /
``observation_space`` (docking3d.py:116-125) and the attributes callers of the reference read: ``env.auv.u_bound``,
``env.auv.state / position / attitude`` (train.py:102,191; utils/datastorage.py:298-300), ``env.radar.n_rays / pos /
end_pos_n`` (docking3d.py:655-659), ``env.observation``, ``env.last_reward_arr``, ``env.cum_reward_arr``,
``env.meta_data_*`` (datastorage.py:54-61,254-259).

Each instance drives a one-env batch of ``BatchedDocking3d`` (all arithmetic of ``step`` runs in libdockauv.so; there
is no CPU path).  Episodes are generated on the host in the reference's draw order on a legacy ``RandomState``
stream, including the one ``normal`` draw ``Current.sim`` burns per step (objects/current.py:88), so
``reset(seed=s)`` reproduces the reference's initial conditions.  For throughput use ``BatchedDocking3d`` directly:
one launch steps 10^3-10^6 envs, one env per launch wastes the GPU.

``render`` is not on the accelerated path and raises.  The episode / full-run pickle storages
(utils/datastorage.py) are host-side logging: off by default (they write files), switched on with the config key
``"data_storage": True``; then they are fed and saved exactly where the reference does it
(``gym_dockauv_amd/utils/datastorage.py``).
"""
from __future__ import annotations

import copy
import time
from typing import Optional

import numpy as np

from .. import _capi
from ..config.env_config import BASE_CONFIG
from ..utils.datastorage import EpisodeDataStorage, FullDataStorage
from .batched import META_DATA_DONE, META_DATA_REWARD, BatchedDocking3d

try:                                  # gym is optional: subclass gym.Env when it is there (SB3 checks isinstance)
    import gym as _gym                # type: ignore
    _EnvBase = _gym.Env
except Exception:                     # pragma: no cover - gym is absent in the build image
    _EnvBase = object


class _AUVView:
    """``env.auv``: the attributes of the reference's AUVSim that callers touch, read from the device state."""

    def __init__(self, env: "BaseDocking3d"):
        self._env = env
        self._model = env._batch.auv

    def __getattr__(self, name):      # constants: u_bound, safety_radius, m, W, M_inv, ...
        return getattr(self._model, name)

    @property
    def state(self) -> np.ndarray:
        return self._env._batch.state[0]

    @property
    def position(self) -> np.ndarray:
        return self.state[0:3]

    @property
    def attitude(self) -> np.ndarray:
        return self.state[3:6]

    @property
    def relative_velocity(self) -> np.ndarray:
        return self.state[6:12]

    @property
    def u(self) -> np.ndarray:
        return self._env._batch.u[0]

    @property
    def state_dot(self) -> np.ndarray:
        """AUVSim._state_dot (objects/auvsim.py:62,108): zeros after reset, else the RHS at the new state."""
        return np.zeros(12) if self._env.t_steps == 0 else np.asarray(self._env._batch.state_dot[0], dtype=np.float64).copy()

    _state_dot = state_dot

    @property
    def euler_dot(self) -> np.ndarray:
        return self.state_dot[3:6]


class _RadarView:
    """``env.radar``: layout constants plus the last intersection distances / end points (objects/sensor.py:104-129)."""

    def __init__(self, env: "BaseDocking3d"):
        self._env = env
        self._layout = env._batch.radar

    def __getattr__(self, name):      # n_rays, max_dist, alpha, beta, rd_b, blocksize_reduce, ...
        return getattr(self._layout, name)

    @property
    def pos(self) -> np.ndarray:
        return self._env.auv.position

    @property
    def intersec_dist(self) -> np.ndarray:
        return np.asarray(self._env._batch.intersec_dist[0], dtype=np.float64)

    @property
    def rd_n(self) -> np.ndarray:
        phi, th, psi = self._env.auv.attitude
        cf, sf, ct, st_, cp, sp = np.cos(phi), np.sin(phi), np.cos(th), np.sin(th), np.cos(psi), np.sin(psi)
        R = np.array([[cp * ct, -sp * cf + cp * st_ * sf, sp * sf + cp * cf * st_],      # utils/geomutils.py:14-43
                      [sp * ct, cp * cf + sf * st_ * sp, -cp * sf + st_ * sp * cf],
                      [-st_, ct * sf, ct * cf]])
        d = self._layout.rd_b @ R.T
        return d / np.linalg.norm(d, axis=1)[:, None]

    @property
    def end_pos_n(self) -> np.ndarray:
        return self.pos[None, :] + self.rd_n * self.intersec_dist[:, None]

    @property
    def intersec_dist_reduced(self) -> np.ndarray:
        return self._env.observation[self._env.n_obs_without_radar:] * self._layout.max_dist


class BaseDocking3d(_EnvBase):
    """One docking3d environment (reference: ``BaseDocking3d``, envs/docking3d.py:31-704)."""

    scenario = "SimpleDocking3d"
    metadata = {"render.modes": []}

    def __init__(self, env_config: dict = BASE_CONFIG, device: int = 0, precision: str = "f32"):
        super().__init__()
        self.config = copy.deepcopy(env_config)
        self.title = self.config.get("title", "DEFAULT")
        self.verbose = self.config.get("verbose", 0)
        self._batch = BatchedDocking3d(self.config, num_envs=1, scenario=self.scenario, device=device,
                                       precision=precision, auto_reset=False, rng="per_env", reset_mode="none")
        b = self._batch
        self.auv = _AUVView(self)
        self.radar = _RadarView(self)
        self.action_space, self.observation_space = b.action_space, b.observation_space
        self.n_observations, self.n_obs_without_radar = b.n_observations, b.n_obs_without_radar
        self.n_rewards, self.n_cont_rewards = b.n_rewards, b.n_cont_rewards
        self.meta_data_reward, self.meta_data_done = META_DATA_REWARD, META_DATA_DONE
        self.meta_data_observation = (["delta_d", "delta_theta", "delta_psi", "u", "v", "w", "phi", "theta", "psi_sin",
                                       "psi_cos", "p", "q", "r", "u_c", "v_c", "w_c"]
                                      + [f"ray_{i}" for i in range(b.radar.n_rays_reduced)])   # docking3d.py:128-138
        self.max_timesteps = int(self.config["max_timesteps"])
        self.t_step_size = float(self.config["t_step_size"])
        self.reward_factors = self.config["reward_factors"]
        self.episode = 0
        self.t_total_steps = 0
        self._zero_episode()
        # storages (docking3d.py:207-212); opt-in here because they write pickles under save_path_folder
        self.save_path_folder = self.config.get("save_path_folder", "logs")
        self.interval_datastorage = int(self.config.get("interval_datastorage", 100))
        self.data_storage = bool(self.config.get("data_storage", False))
        self.episode_data_storage = None
        self.full_data_storage = None
        self.nu_c = np.zeros(6)
        if self.data_storage:
            self.full_data_storage = FullDataStorage()
            self.full_data_storage.set_up_full_storage(env=self, path_folder=self.save_path_folder, title=self.title)

    # ------------------------------------------------------------------------------------------ bookkeeping
    def _zero_episode(self) -> None:
        self.t_steps = 0
        self.observation = np.zeros(self.n_observations, dtype=np.float32)     # docking3d.py:269 (Q8)
        self.done = False
        self.goal_reached = False
        self.collision = False
        self.last_reward = 0.0
        self.last_reward_arr = np.zeros(self.n_rewards)
        self.cumulative_reward = 0.0
        self.cum_reward_arr = np.zeros(self.n_rewards)
        self.conditions = [False] * 5
        self.delta_d = self.delta_theta = self.delta_psi = self.delta_heading_goal = 0.0
        self.info = {}

    @property
    def goal_location(self) -> np.ndarray:
        return self._batch.get_field(_capi.F_GOAL)[0, 0:3]

    @property
    def heading_goal_reached(self) -> float:
        return float(self._batch.get_field(_capi.F_GOAL)[0, 3])

    @property
    def capsules(self) -> np.ndarray:
        """[n, 7] = bottom xyz, top xyz, radius of the capsules of the running episode."""
        if not self._batch.max_capsules:
            return np.zeros((0, 7))
        c = self._batch.get_field(_capi.F_CAPSULES)[0].reshape(-1, 7)
        return c[c[:, 6] > 0]

    # ------------------------------------------------------------------------------------------ gym.Env
    def reset(self, seed: Optional[int] = None, return_info: bool = False, options=None):
        """docking3d.py:222-322.  Returns the all-zero observation, like the reference (observe() is not called)."""
        return_info_dict = dict(self.info)
        if self.data_storage:
            # docking3d.py:252-259: close the running episode's storage, then the run-level one
            if self.episode_data_storage and (self.episode % self.interval_datastorage == 0 or self.episode == 1):
                self.episode_data_storage.update(self.nu_c)
                self.episode_data_storage.save()
            self.episode_data_storage = None
            if self.episode != 0:
                self.full_data_storage.update()
        self._batch.reset(seed=None if seed is None else [int(seed)])
        self.episode += 1
        self._zero_episode()
        if self.data_storage:
            self.nu_c = self._current_body(self.auv.attitude)
            if self.episode % self.interval_datastorage == 0 or self.episode == 1:      # docking3d.py:314-317
                self.init_episode_storage()
        if return_info:
            return self.observation, return_info_dict
        return self.observation

    # ------------------------------------------------------------------------------------------ logging (host side)
    def _current_body(self, attitude: np.ndarray) -> np.ndarray:
        """nu_c = [R(Theta)^T v_c^n, 0, 0, 0] (objects/current.py:33-76) from the device's current fields; for the logs."""
        vc, _, _, al, be = self._batch.get_field(_capi.F_CURRENT)[0]
        phi, th, psi = attitude
        cf, sf, ct, st_, cp, sp = np.cos(phi), np.sin(phi), np.cos(th), np.sin(th), np.cos(psi), np.sin(psi)
        R = np.array([[cp * ct, -sp * cf + cp * st_ * sf, sp * sf + cp * cf * st_],
                      [sp * ct, cp * cf + sf * st_ * sp, -cp * sf + st_ * sp * cf],
                      [-st_, ct * sf, ct * cf]])
        v_n = vc * np.array([np.cos(al) * np.cos(be), np.sin(be), np.sin(al) * np.cos(be)])
        return np.concatenate([R.T @ v_n, np.zeros(3)])

    def init_episode_storage(self) -> None:
        """docking3d.py:675-685."""
        shapes = [{"type": "Capsule", "position": 0.5 * (c[0:3] + c[3:6]), "vec_bot": c[0:3].copy(), "vec_top": c[3:6].copy(),
                   "radius": float(c[6])} for c in self.capsules]
        shapes.append({"type": "Sphere", "position": self.goal_location.copy(), "radius": 0.15})
        self.episode_data_storage = EpisodeDataStorage()
        self.episode_data_storage.set_up_episode_storage(path_folder=self.save_path_folder, env=self, nu_c_init=self.nu_c,
                                                         shapes=shapes, title=self.title, episode=self.episode)

    def step(self, action: np.ndarray):
        """docking3d.py:346-402."""
        t0 = time.perf_counter()
        a = np.asarray(action, dtype=np.float64).reshape(1, -1)
        att_pre = self.auv.attitude if self.episode_data_storage else None
        obs, rew, done, _ = self._batch.step(a, extras=True)
        b = self._batch
        self.observation = obs[0]
        self.last_reward = float(rew[0])
        self.last_reward_arr = np.asarray(b.last_reward_arr[0], dtype=np.float64).copy()
        self.cum_reward_arr = self.cum_reward_arr + self.last_reward_arr
        self.cumulative_reward += self.last_reward
        self.done = bool(done[0])
        self.conditions = [bool(c) for c in b.conditions[0]]
        cond_idx = [i for i, c in enumerate(self.conditions) if c]
        self.goal_reached = self.conditions[0]
        self.collision = self.conditions[4]
        nav = b.nav_errors[0]
        self.delta_d, self.delta_theta, self.delta_psi, self.delta_heading_goal = (float(x) for x in nav)
        self.t_total_steps += 1
        self.t_steps += 1
        self.info = {"episode_number": self.episode,
                     "t_step": self.t_steps,
                     "t_total_steps": self.t_total_steps,
                     "cumulative_reward": self.cumulative_reward,
                     "last_reward": self.last_reward,
                     "done": self.done,
                     "conditions_true": cond_idx,
                     "conditions_true_info": [self.meta_data_done[i] for i in cond_idx],
                     "collision": self.collision,
                     "goal_reached": self.goal_reached,
                     "simulation_time": time.perf_counter() - t0,
                     "delta_d": self.delta_d}
        if self.episode_data_storage:                       # docking3d.py:363-364
            self.nu_c = self._current_body(att_pre)         # current in the body frame at the PRE-step attitude (Q2)
            self.episode_data_storage.update(self.nu_c)
        return self.observation, self.last_reward, self.done, self.info

    def render(self, mode="human", real_time=False):
        raise NotImplementedError("rendering (utils/plotutils.py) is outside the accelerated path: SURVEY.md section 2")

    def save_full_data_storage(self):
        """docking3d.py:669-673; returns the file name (None when ``data_storage`` is off)."""
        return self.full_data_storage.save() if self.full_data_storage is not None else None

    def close(self) -> None:
        self._batch.close()

    def seed(self, seed=None):
        return self._batch.seed(None if seed is None else [int(seed)])


class SimpleDocking3d(BaseDocking3d):
    """docking3d.py:795-825: no current, no obstacles."""
    scenario = "SimpleDocking3d"


class SimpleCurrentDocking3d(BaseDocking3d):
    """docking3d.py:828-848: random constant current."""
    scenario = "SimpleCurrentDocking3d"


class CapsuleDocking3d(BaseDocking3d):
    """docking3d.py:851-886: dock onto a capsule."""
    scenario = "CapsuleDocking3d"


class CapsuleCurrentDocking3d(BaseDocking3d):
    """docking3d.py:889-906."""
    scenario = "CapsuleCurrentDocking3d"


class ObstaclesDocking3d(BaseDocking3d):
    """docking3d.py:909-946: capsule + four pillars."""
    scenario = "ObstaclesDocking3d"


class ObstaclesNoCapDocking3d(BaseDocking3d):
    """docking3d.py:949-965: pillars only."""
    scenario = "ObstaclesNoCapDocking3d"


class ObstaclesCurrentDocking3d(BaseDocking3d):
    """docking3d.py:968-988."""
    scenario = "ObstaclesCurrentDocking3d"


class SphereDocking3d(BaseDocking3d):
    """Build-defined: eight static sphere obstacles around the goal (the reference has the intersection routine,
    objects/shape.py:235-264, but no shipped env populates ``self.spheres``)."""
    scenario = "SphereDocking3d"
