"""CPU: the product's host-side scenario generators (gym_dockauv_amd/scenarios.py) fed with a legacy RandomState
stream reproduce the reference's reset draws stored in the golden trajectories (G8), incl. later episodes after the
per-step normal burn."""
import numpy as np
import pytest

from gym_dockauv_amd import scenarios
from tests import helpers as H


@pytest.mark.parametrize("name", [n for n in H.TRAJ if str(H.load(n)["meta_env"]) in scenarios.SCENARIOS
                                  and str(H.load(n)["meta_env"]) != "SphereDocking3d"])
def test_reset_draws_match_reference(name):
    g = H.load(name)
    scenario = str(g["meta_env"])
    rs = np.random.RandomState(int(g["meta_seed"]))
    max_caps = max(int(g["ep_n_capsules"].max()), scenarios.N_CAPSULES[scenario])
    ep_start = g["ep_start"].tolist() + [int(g["meta_T"])]
    for e in range(len(g["ep_start"])):
        if e > 0:
            rs.normal(size=ep_start[e] - ep_start[e - 1])        # one normal per elapsed step (current.py:88)
        U = rs.random_sample(scenarios.N_DRAWS[scenario])[None, :]
        ep = scenarios.episodes_from_uniforms(scenario, U, np.pi / 3, 20.0, max_caps, 0)
        ref = H.episode_arrays(g, [e], max_caps, 0)
        pose_ref = g["ep_pose_drawn"][e][None, :] if "ep_pose_drawn" in g.files else ref["pose"]   # ("near" fixtures)
        np.testing.assert_allclose(ep["pose"], pose_ref, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(ep["goal"], ref["goal"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(ep["current"], ref["current"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(ep["capsules"], ref["capsules"], rtol=1e-12, atol=1e-12)


def test_unknown_scenario_and_capacity():
    with pytest.raises(KeyError):
        scenarios.episodes_from_uniforms("Nope", np.zeros((1, 12)), 1.0, 20.0, 0, 0)
    with pytest.raises(ValueError):
        scenarios.episodes_from_uniforms("ObstaclesDocking3d", np.random.rand(2, 12), 1.0, 20.0, 2, 0)
