"""
Episode / full-run storage with the reference's pickle schema (utils/datastorage.py:20-343), fed from the HIP env.

Host-side logging, off the accelerated path: the single-env classes call it exactly
where the reference does -- ``EpisodeDataStorage.update`` after every step (docking3d.py:363-364), save + re-arm in
``reset`` for episode 1 and every ``interval_datastorage``-th one (docking3d.py:252-259, 314-317),
``FullDataStorage.update`` at every episode end (docking3d.py:258-259), ``save_full_data_storage`` (docking3d.py:669-673).
The dictionaries have the reference's keys and array shapes, so its post-analysis of arrays (``plot_episode_states``,
``plot_u``, ``plot_observation``, ``plot_rewards``; utils/plotutils.py) reads them unchanged.  Two entries cannot be
the reference's Python objects and are plain data instead:

* ``storage["vehicle"]["object"]``: a dict of the vehicle's constants (name, u_bound, safety_radius), not an AUVSim;
* ``storage["shapes"]``: dicts ``{"type": "Capsule", "position", "vec_bot", "vec_top", "radius"}`` /
  ``{"type": "Sphere", "position", "radius"}`` instead of ``Shape`` instances (their plot meshes are rendering code);

``storage["vehicle"]["states_dot"]`` is AUVSim._state_dot (objects/auvsim.py:108), which the kernel evaluates in full
when a caller asks for it (dockauv_step_io.state_dot).

``BatchEpisodeStorage`` is the same storage for SELECTED ENVS OF A BATCH: the step kernel keeps their per-step arrays in
a device ring (include/dockauv.h: dockauv_trace_*), so that device-resident rollouts are logged without a host round
trip per step; every finished episode becomes one pickle of the schema above.
"""
from __future__ import annotations

import datetime
import os
import pickle
from typing import List, Optional

import numpy as np


class ArrayList:
    """Rows appended one at a time; the first row is the initial vector (what the reference's storages keep,
    utils/datastorage.py:120-161).  Rows are collected in a Python list of blocks and joined on demand."""

    _BLOCK = 256

    def __init__(self, init_array):
        first = np.array(init_array, dtype=float)
        self._shape = first.shape
        self._blocks = []
        self._cur = np.empty((self._BLOCK, *self._shape))
        self._n = 0
        self.add_row(first)

    @property
    def size(self) -> int:
        return len(self._blocks) * self._BLOCK + self._n

    def add_row(self, row) -> None:
        if self._n == self._BLOCK:
            self._blocks.append(self._cur)
            self._cur = np.empty((self._BLOCK, *self._shape))
            self._n = 0
        self._cur[self._n] = row
        self._n += 1

    def get_nparray(self) -> np.ndarray:
        return np.concatenate([*self._blocks, self._cur[:self._n]], axis=0)

    def __getitem__(self, index):
        return self.get_nparray()[index]


def _stamp() -> str:
    return datetime.datetime.utcnow().strftime('%Y_%m_%dT%H_%M_%S')


class FullDataStorage:
    """Per-episode cumulative rewards + info dicts of a whole run (reference: utils/datastorage.py:20-118)."""

    def __init__(self):
        self.file_save_name: Optional[str] = None
        self.env = None
        self.storage: Optional[dict] = None

    def set_up_full_storage(self, env, path_folder: str, title: str = "") -> None:
        self.env = env
        if len(path_folder) > 0:
            os.makedirs(path_folder, exist_ok=True)
        self.file_save_name = os.path.join(path_folder, f"{_stamp()}__{title}__FULL_DATA_STORAGE.pkl")
        self.storage = {
            "title": title,
            "cum_rewards": ArrayList(env.cum_reward_arr),
            "rewards": ArrayList(env.last_reward_arr),
            "meta_data_reward": env.meta_data_reward,
            "n_cont_rewards": env.n_cont_rewards,
            "infos": [],
        }

    def update(self) -> None:
        """At the end of each episode."""
        self.storage["cum_rewards"].add_row(self.env.cum_reward_arr)
        self.storage["rewards"].add_row(self.env.last_reward_arr)
        self.storage["infos"].append(self.env.info)

    def save(self) -> str:
        for k in ("cum_rewards", "rewards"):
            if isinstance(self.storage[k], ArrayList):
                self.storage[k] = self.storage[k].get_nparray()
        with open(self.file_save_name, 'wb') as outp:
            pickle.dump(self.storage, outp, pickle.HIGHEST_PROTOCOL)
        return self.file_save_name

    def load(self, file_name: str) -> dict:
        with open(file_name, 'rb') as inp:
            self.storage = pickle.load(inp)
        return self.storage


class EpisodeDataStorage:
    """Per-step vehicle / radar / reward / observation arrays of one episode (reference: utils/datastorage.py:164-343)."""

    def __init__(self):
        self.storage: Optional[dict] = None
        self.file_save_name: Optional[str] = None
        self.env = None

    def set_up_episode_storage(self, path_folder: str, env, nu_c_init: np.ndarray, shapes: Optional[List[dict]] = None,
                               title: str = "", episode: int = -1) -> None:
        if len(path_folder) > 0:
            os.makedirs(path_folder, exist_ok=True)
        self.file_save_name = os.path.join(path_folder, f"{_stamp()}__{title}__EPISODE_{episode}_DATA_STORAGE.pkl")
        self.env = env
        auv = env.auv
        state = np.asarray(auv.state, dtype=float)
        self.storage = {
            "vehicle": {
                "object": {"name": getattr(auv, "name", ""), "u_bound": np.asarray(auv.u_bound, dtype=float),
                           "safety_radius": float(auv.safety_radius)},
                "states": ArrayList(state),
                "states_dot": ArrayList(np.zeros(12)),            # AUVSim.reset: _state_dot = 0 (auvsim.py:62)
                "u": ArrayList(auv.u),
            },
            "radar": ArrayList(env.radar.end_pos_n),
            "nu_c": ArrayList(nu_c_init),
            "shapes": list(shapes or []),
            "title": title,
            "episode": episode,
            "step_size": env.t_step_size,
            "cum_rewards": ArrayList(env.cum_reward_arr),
            "rewards": ArrayList(env.last_reward_arr),
            "meta_data_reward": env.meta_data_reward,
            "n_cont_rewards": env.n_cont_rewards,
            "observation": ArrayList(env.observation),
            "meta_data_observation": env.meta_data_observation,
        }

    def update(self, nu_c: np.ndarray) -> None:
        """At the end of each simulation step."""
        auv = self.env.auv
        v = self.storage["vehicle"]
        v["states"].add_row(auv.state)
        v["states_dot"].add_row(getattr(auv, "state_dot", np.full(12, np.nan)))
        v["u"].add_row(auv.u)
        self.storage["nu_c"].add_row(nu_c)
        self.storage["cum_rewards"].add_row(self.env.cum_reward_arr)
        self.storage["rewards"].add_row(self.env.last_reward_arr)
        self.storage["observation"].add_row(self.env.observation)
        self.storage["radar"].add_row(self.env.radar.end_pos_n)

    def save(self) -> str:
        v = self.storage["vehicle"]
        for k in ("states", "states_dot", "u"):
            if isinstance(v[k], ArrayList):
                v[k] = v[k].get_nparray()
        for k in ("radar", "nu_c", "cum_rewards", "rewards", "observation"):
            if isinstance(self.storage[k], ArrayList):
                self.storage[k] = self.storage[k].get_nparray()
        with open(self.file_save_name, 'wb') as outp:
            pickle.dump(self.storage, outp, pickle.HIGHEST_PROTOCOL)
        return self.file_save_name

    def load(self, file_name: str) -> dict:
        with open(file_name, 'rb') as inp:
            self.storage = pickle.load(inp)
        return self.storage

    # the reference's convenience views (utils/datastorage.py:293-325)
    @property
    def states(self):
        return self.storage["vehicle"]["states"][:]

    @property
    def positions(self) -> np.ndarray:
        return self.storage["vehicle"]["states"][:, 0:3]

    @property
    def attitudes(self) -> np.ndarray:
        return self.storage["vehicle"]["states"][:, 3:6]

    @property
    def step_size(self) -> float:
        return self.storage["step_size"]

    @property
    def u(self) -> np.ndarray:
        return self.storage["vehicle"]["u"][:]

    @property
    def nu_c(self) -> np.ndarray:
        return self.storage["nu_c"][:]


class BatchEpisodeStorage:
    """EpisodeDataStorage for selected envs of a BatchedDocking3d, fed from the device trace ring.

    Row layout of a pickle (reference: utils/datastorage.py:262-287, 289-310): index 0 = the values at reset (state the
    episode started from, zeros for state_dot / u / rewards / observation -- Q8: reset returns zeros), index k = after
    step k.  ``nu_c`` rows are 6 wide (last three zero) as in the reference; row 0 repeats the first step's value (the
    reset-time current differs from it by one Gauss-Markov update of V_c)."""

    def __init__(self, env, path_folder: str, title: str, capacity: int):
        self.env = env
        self.path_folder = path_folder
        self.title = title
        self.capacity = int(capacity)
        self.ids = np.array(env._trace_ids)
        self._flushed = 0                                   # steps already looked at
        self._open = {int(i): [] for i in self.ids}         # env id -> list of per-step dicts of the running episode
        self._closed = {int(i): False for i in self.ids}    # env id -> its episode has ended and was written (reset_mode "none")
        self._episode_no = {int(i): 1 for i in self.ids}
        self.files = []
        if len(path_folder) > 0:
            os.makedirs(path_folder, exist_ok=True)

    def after_step(self, done: np.ndarray) -> None:
        """Host path: called by BatchedDocking3d.step; flushes when a selected env finished or the ring is half full."""
        if done[self.ids].any() or self.env.trace_steps() - self._flushed >= self.capacity // 2:
            self.flush()

    def flush(self):
        """Pull the steps recorded since the last flush from the device ring and write finished episodes."""
        n_now = self.env.trace_steps()
        n_new = n_now - self._flushed
        if n_new <= 0:
            return []
        if n_new > self.capacity:
            raise RuntimeError(f"trace ring overrun: {n_new} steps since the last flush, capacity {self.capacity}")
        tr = self.env.read_trace(self._flushed, n_new)
        self._flushed = n_now
        written = []
        for j, env_id in enumerate(self.ids.tolist()):
            for k in range(n_new):
                if self._closed[env_id]:
                    # reset_mode "none": a finished env that is stepped on reports its conditions on every step -- the
                    # episode closed on the 0 -> non-zero edge; what follows belongs to no episode until the env is reset
                    continue
                self._open[env_id].append({key: tr[key][k, j] for key in tr})
                if tr["conditions"][k, j] != 0:
                    written.append(self._save(env_id))
                    if not self.env.auto_reset:
                        self._closed[env_id] = True
        return written

    def on_reset(self, env_ids, save_partial: bool = True):
        """Host-side reset of `env_ids` (BatchedDocking3d.reset / reset_envs): the steps recorded so far are pulled from
        the ring first; a selected env's running (unfinished) episode is WRITTEN, as the reference saves its storage
        inside reset() (envs/docking3d.py:252-256: update + save before anything is reset) -- save_partial=False drops
        it instead -- so that its rows never join the next episode's."""
        written = self.flush()
        for env_id in np.intersect1d(np.asarray(env_ids, dtype=np.int64), self.ids).tolist():
            if self._open[env_id]:
                if save_partial:
                    written.append(self._save(env_id))
                else:
                    self._open[env_id] = []
            self._closed[env_id] = False
        return written

    def _save(self, env_id: int) -> str:
        steps = self._open[env_id]
        self._open[env_id] = []
        env = self.env
        n_u = int(env.vehicle_models[int(env.vehicle_id[env_id])].u_bound.shape[0])
        n_obs = env.n_observations
        T = len(steps)
        states = np.zeros((T + 1, 12))
        states_dot = np.zeros((T + 1, 12))
        u = np.zeros((T + 1, n_u))
        nu_c = np.zeros((T + 1, 6))
        rewards = np.zeros((T + 1, 13))
        obs = np.zeros((T + 1, n_obs))
        states[0] = steps[0]["state_pre"]
        for k, s in enumerate(steps):
            states[k + 1] = s["state"]
            states_dot[k + 1] = s["state_dot"]
            u[k + 1] = s["u"][:n_u]
            nu_c[k + 1, 0:3] = s["nu_c"]
            rewards[k + 1] = s["reward_terms"]
            obs[k + 1] = s["obs"]
        nu_c[0] = nu_c[1]
        episode = self._episode_no[env_id]
        self._episode_no[env_id] += 1
        model = env.vehicle_models[int(env.vehicle_id[env_id])]
        storage = {
            "vehicle": {"object": {"name": getattr(model, "name", ""), "u_bound": np.asarray(model.u_bound, dtype=float),
                                   "safety_radius": float(model.safety_radius)},
                        "states": states, "states_dot": states_dot, "u": u},
            "radar": None,            # ray end points are a rendering input (plotutils); the distances are in the observation
            "nu_c": nu_c,
            "shapes": [],
            "title": self.title,
            "episode": episode,
            "step_size": float(env.config["t_step_size"]),
            "cum_rewards": np.cumsum(rewards, axis=0),
            "rewards": rewards,
            "meta_data_reward": env.meta_data_reward,
            "n_cont_rewards": env.n_cont_rewards,
            "observation": obs,
            "meta_data_observation": None,
            "env_index": env_id,
            "conditions_last_step": int(steps[-1]["conditions"]),
        }
        name = os.path.join(self.path_folder, f"{_stamp()}__{self.title}__ENV_{env_id}__EPISODE_{episode}_DATA_STORAGE.pkl")
        with open(name, "wb") as f:
            pickle.dump(storage, f, pickle.HIGHEST_PROTOCOL)
        self.files.append(name)
        return name
