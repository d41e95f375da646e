"""Ray-distance error statistics of the f32 HIP path against the golden trajectories (teacher-forced)."""
import os, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers as H
from gym_dockauv_amd import _capi
names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(os.path.dirname(H.__file__), "golden", "traj_*.npz")))
for name in names:
    g = H.load(name)
    if "ray_dist" not in g or g["ray_dist"].min() >= g["ray_dist"].max():
        continue
    T = int(g["meta_T"]); n_u = int(g["meta_n_u"])
    env, max_caps, max_sph = H.make_batched(g, T, "f32", auto_reset=False)
    state, u, vc, tsteps, w = H.prestep_inputs(g)
    ep = H.episode_arrays(g, g["ep_index"], max_caps, max_sph)
    env.load_episodes(np.arange(T), ep)
    ep["current"][:, 0] = vc
    env.set_field(_capi.F_CURRENT, ep["current"]); env.set_field(_capi.F_STATE, state); env.set_field(_capi.F_U, u)
    env.set_field(_capi.F_TSTEPS, tsteps[:, None].astype(float))
    actions = np.zeros((T, env.n_u)); actions[:, :n_u] = g["action"]
    env.step(actions, noise=w, extras=True)
    d = env.intersec_dist; r = g["ray_dist"]; md = env.radar.max_dist
    err = np.abs(d - r)
    hit = (r < md) | (d < md)
    flips = ((r < md) != (d < md))
    both = (r < md) & (d < md)
    print(f"{name[5:]:55s} rays {err.size:6d} hit {hit.sum():6d} flips {flips.sum():3d} | both-hit err: max {err[both].max() if both.any() else 0:.2e} "
          f">5e-5 {(err[both] > 5e-5).sum():4d} >2e-4 {(err[both] > 2e-4).sum():3d} >1e-3 {(err[both] > 1e-3).sum():3d} p99 {np.percentile(err[both], 99) if both.any() else 0:.1e}")
    env.close()
