"""
gym_dockauv_amd -- the docking3d step() of Erikx3/gym_dockauv as one fused HIP kernel for MI355X (gfx950), behind the
reference's env API.  ``gym_dockauv_amd.envs`` has the batched env (``BatchedDocking3d``) and the single-env classes
with the reference's names; when ``gym`` is importable the reference's env ids are registered on import, like
``gym_dockauv/__init__.py:4-8`` does.
"""
from .config.env_config import REGISTRATION_DICT


def register_envs() -> int:
    """Register the env ids of config/env_config.py with gym; returns how many were registered (0 without gym)."""
    try:
        from gym.envs.registration import register  # type: ignore
    except Exception:
        return 0
    n = 0
    for env_id, entry_point in REGISTRATION_DICT.items():
        try:
            register(id=env_id, entry_point=entry_point)
            n += 1
        except Exception:      # already registered
            pass
    return n


register_envs()
