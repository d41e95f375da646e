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

and ``storage["vehicle"]["states_dot"]`` is NaN: the kernel evaluates of the post-step right-hand side only what the
reward needs (the Euler-angle rates).
"""
from __future__ import annotations

import datetime
import os
import pickle
from typing import List, Optional

import numpy as np


class ArrayList:
    """Growable n x c array (reference: utils/datastorage.py:120-161): first row = the initial vector."""

    def __init__(self, init_array):
        init_array = np.asarray(init_array, dtype=float)
        self.dim_col = init_array.shape
        self.capacity = 100
        self.array_grow_factor = 4
        self.data = np.zeros((self.capacity, *self.dim_col))
        self.size = 1
        self.data[0] = init_array

    def __getitem__(self, index):
        return self.data[:self.size][index]

    def add_row(self, row) -> None:
        if self.size == self.capacity:
            self.capacity *= self.array_grow_factor
            grown = np.zeros((self.capacity, *self.dim_col))
            grown[:self.size] = self.data
            self.data = grown
        self.data[self.size] = row
        self.size += 1

    def get_nparray(self) -> np.ndarray:
        return self.data[:self.size]


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
                "states_dot": ArrayList(np.full(12, np.nan)),
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
        v["states_dot"].add_row(np.full(12, np.nan))
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
