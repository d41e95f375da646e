"""
The drop-in boundary: what the reference's learner side touches (train.py:53-71,248-261 --
gym.make(id, env_config=...) -> MODEL('MlpPolicy', env) -> model.learn, which drives a VecEnv).

stable-baselines3 and gym are not installed in this image, so the contract is written down here from SB3 1.5's
``stable_baselines3/common/vec_env/base_vec_env.py`` (abstract methods of ``VecEnv``, what ``collect_rollouts`` of the
on-/off-policy algorithms reads per step) and checked on the real objects:

  * CPU: every VecEnv method / attribute exists on BatchedDocking3d with SB3's signature; the env ids of
    config/env_config.py resolve through ``gym.envs.registration.register`` / ``gym.make`` of a stub gym to the classes
    of this package (the reference's own entry points name ``gym_dockauv.envs:<Class>`` with an empty
    ``envs/__init__.py`` -- SURVEY.md section 2 row 17 -- so that string cannot be the contract; id -> class is).
  * GPU: a rollout loop written like SB3's ``collect_rollouts`` (clip actions to the action space, ``step``, read
    ``infos[i].get("terminal_observation")`` / ``"TimeLimit.truncated"``, store float32 rewards, bool dones) on
    BatchedDocking3d, and the single-env gym.Env protocol (4-tuple, info keys of docking3d.py:388-400, spaces).
"""
import inspect
import sys
import types

import numpy as np
import pytest

# SB3 1.5 VecEnv: @abstractmethod list and the signatures of the concrete helpers algorithms call
SB3_ABSTRACT = {
    "reset": [],
    "step_async": ["actions"],
    "step_wait": [],
    "close": [],
    "get_attr": ["attr_name", "indices"],
    "set_attr": ["attr_name", "value", "indices"],
    "env_method": ["method_name"],
    "env_is_wrapped": ["wrapper_class", "indices"],
}
SB3_CONCRETE = {"step": ["actions"], "seed": ["seed"], "render": ["mode"]}
SB3_ATTRS = ["num_envs", "observation_space", "action_space"]


def test_vecenv_surface_matches_sb3_signatures():
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    for name, params in {**SB3_ABSTRACT, **SB3_CONCRETE}.items():
        fn = getattr(BatchedDocking3d, name, None)
        assert callable(fn), f"VecEnv.{name} missing"
        sig = list(inspect.signature(fn).parameters)[1:]
        for p in params:
            assert p in sig, f"VecEnv.{name}: parameter {p!r} missing (has {sig})"
    # the attributes are set in __init__ (checked on an instance in the GPU test); statically: assigned there
    src = inspect.getsource(BatchedDocking3d.__init__)
    for a in SB3_ATTRS:
        assert f"self.{a} =" in src


def _stub_gym(monkeypatch):
    """gym 0.21's registration surface: register(id, entry_point, ...) + make(id, **kwargs) -> entry_point(**kwargs)."""
    import importlib
    registry = {}
    gym = types.ModuleType("gym")

    class Env:
        metadata = {}

    class Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low, self.high, self.dtype, self.shape = np.asarray(low, dtype), np.asarray(high, dtype), np.dtype(dtype), np.shape(low)

    def register(id, entry_point, **kw):
        if id in registry:
            raise RuntimeError(f"Cannot re-register id: {id}")
        registry[id] = entry_point

    def make(id, **kwargs):
        mod, cls = registry[id].split(":")
        return getattr(importlib.import_module(mod), cls)(**kwargs)

    gym.Env, gym.make = Env, make
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = register
    gym.spaces, gym.envs, envs.registration = spaces, envs, reg
    for n, m in (("gym", gym), ("gym.spaces", spaces), ("gym.envs", envs), ("gym.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, n, m)
    return gym, registry


def test_env_ids_register_and_resolve_through_gym(monkeypatch):
    """train.py:257: gym.make("ObstaclesDocking3d-v0", env_config=...) must reach the class of that name."""
    import importlib
    gym, registry = _stub_gym(monkeypatch)
    import gym_dockauv_amd
    from gym_dockauv_amd.config.env_config import REGISTRATION_DICT
    assert gym_dockauv_amd.register_envs() == len(REGISTRATION_DICT) == 7
    assert set(registry) == set(REGISTRATION_DICT)          # the reference's ids (config/env_config.py:9-17)
    for env_id, entry in registry.items():
        mod, cls = entry.split(":")
        klass = getattr(importlib.import_module(mod), cls)
        assert klass.__name__ == env_id.split("-")[0] and klass.__module__.startswith("gym_dockauv_amd.envs")
        sig = inspect.signature(klass.__init__)
        assert list(sig.parameters)[1] == "env_config"     # the keyword gym.make passes (train.py:257)
        for m, params in {"reset": ["seed", "return_info", "options"], "step": ["action"], "render": [],
                          "save_full_data_storage": []}.items():
            got = list(inspect.signature(getattr(klass, m)).parameters)
            for q in params:
                assert q in got, f"{cls}.{m}: parameter {q!r} missing"
    assert gym_dockauv_amd.register_envs() == 0            # a second import must not fail on re-registration


@pytest.mark.gpu
def test_rollout_loop_as_sb3_collects_it():
    from gym_dockauv_amd.envs.batched import BatchedDocking3d
    n_envs, n_steps = 48, 60
    cfg = None
    import copy
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    cfg = copy.deepcopy(BASE_CONFIG)
    cfg["max_timesteps"] = 25                       # several auto-resets inside the rollout
    env = BatchedDocking3d(cfg, num_envs=n_envs, scenario="ObstaclesDocking3d")
    try:
        for a in SB3_ATTRS:
            assert hasattr(env, a)
        assert env.num_envs == n_envs
        obs_space, act_space = env.observation_space, env.action_space
        assert obs_space.shape == (env.n_observations,) and act_space.shape == (6,)
        assert obs_space.dtype == np.float32 and act_space.dtype == np.float32
        assert obs_space.low[0] == 0 and obs_space.low[1] == -1 and obs_space.low[16] == 0 and (obs_space.high == 1).all()
        assert env.seed(7) == list(range(7, 7 + n_envs))
        last_obs = env.reset()
        assert last_obs.shape == (n_envs,) + obs_space.shape and last_obs.dtype == np.float32 and not last_obs.any()   # Q8
        rewards_buf = np.zeros((n_steps, n_envs), dtype=np.float32)
        dones_buf = np.zeros((n_steps, n_envs), dtype=np.float32)
        rs = np.random.RandomState(0)
        n_term = 0
        for t in range(n_steps):
            actions = rs.normal(scale=1.2, size=(n_envs,) + act_space.shape).astype(np.float32)
            clipped = np.clip(actions, act_space.low, act_space.high)       # on_policy_algorithm.collect_rollouts
            env.step_async(clipped)
            new_obs, rewards, dones, infos = env.step_wait()
            assert new_obs.shape == last_obs.shape and new_obs.dtype == np.float32
            assert rewards.shape == (n_envs,) and dones.shape == (n_envs,) and dones.dtype == bool
            assert isinstance(infos, (list, tuple)) and len(infos) == n_envs
            rewards_buf[t] = rewards                                         # RolloutBuffer.add casts to float32
            dones_buf[t] = dones
            for i, done in enumerate(dones):
                info = infos[i]
                assert not info.get("TimeLimit.truncated", False)           # bootstrap branch of collect_rollouts
                term = info.get("terminal_observation")
                if done:
                    n_term += 1
                    assert term is not None and term.shape == obs_space.shape and term.dtype == np.float32
                    assert not new_obs[i].any()                              # the row of a finished env is the reset observation
                    assert (term >= obs_space.low - 1e-6).all() and (term <= obs_space.high + 1e-6).all()
                    info["episode"] = {"r": 0.0, "l": 1}                     # what VecMonitor writes into a finished env's info
                else:
                    assert term is None
            assert np.isfinite(new_obs).all() and np.isfinite(rewards).all()
            last_obs = new_obs
        assert n_term >= n_envs, "every env must have finished at least once (max_timesteps = 25)"
        # VecEnv helpers algorithms / callbacks use
        assert env.env_is_wrapped(object) == [False] * n_envs
        assert len(env.get_attr("n_observations")) == n_envs and env.get_attr("n_observations", [0, 1]) == [env.n_observations] * 2
        env.set_attr("t_total_steps", 0)
        assert env.t_total_steps == 0
        with pytest.raises(NotImplementedError):
            env.render()
    finally:
        env.close()
    env.close()      # idempotent, as SB3 closes envs twice on some paths


@pytest.mark.gpu
def test_single_env_gym_protocol(monkeypatch, tmp_path):
    """gym.make -> reset -> step loop of train.py:108-117 / predict(): 4-tuple, info keys, spaces, auv.u_bound."""
    gym, registry = _stub_gym(monkeypatch)
    import copy
    import gym_dockauv_amd
    from gym_dockauv_amd.config.env_config import BASE_CONFIG
    gym_dockauv_amd.register_envs()
    cfg = copy.deepcopy(BASE_CONFIG)
    cfg["save_path_folder"] = str(tmp_path)
    env = gym.make("CapsuleDocking3d-v0", env_config=cfg)
    try:
        obs = env.reset(seed=3)
        assert obs.shape == env.observation_space.shape and obs.dtype == np.float32 and not obs.any()
        obs2, info0 = env.reset(seed=3, return_info=True)
        assert isinstance(info0, dict)
        assert env.auv.u_bound.shape == (6, 2) and np.allclose(env.action_space.low, env.auv.u_bound[:, 0])
        for t in range(5):
            ob, rew, done, info = env.step(env.action_space.sample() if hasattr(env.action_space, "sample") else np.zeros(6))
            assert ob.shape == env.observation_space.shape and ob.dtype == np.float32
            assert isinstance(rew, float) and isinstance(done, bool) and isinstance(info, dict)
            assert set(info) >= {"episode_number", "t_step", "t_total_steps", "cumulative_reward", "last_reward", "done",
                                 "conditions_true", "conditions_true_info", "collision", "goal_reached",
                                 "simulation_time", "delta_d"}                # docking3d.py:388-400
        assert info["t_step"] == 5 and info["episode_number"] == 2
    finally:
        env.close()
