"""Ad-hoc report (GPU): per-trajectory worst errors of the f32 HIP path vs the golden float64 reference, free-running."""
import numpy as np
from tests import helpers as H


def main():
    print(f"{'trajectory':58s} {'obs':>9s} {'obs(noray)':>10s} {'rew_rel':>9s} {'r6_abs':>9s} {'state':>9s} {'pos':>9s} {'att':>9s} {'vel':>9s} {'angvel':>9s} {'badrays':>7s}")
    for name in H.TRAJ:
        g = H.load(name)
        T, n_u = int(g["meta_T"]), int(g["meta_n_u"])
        env, mc, ms = H.make_batched(g, 1, "f32", auto_reset=False)
        _, _, _, _, w = H.prestep_inputs(g)
        ep_start = g["ep_start"].tolist()
        e = -1
        wo = wo2 = wr = w6 = ws = 0.0
        wp = wa = wv = wq = 0.0
        bad = 0
        for t in range(T):
            if t in ep_start:
                e += 1
                env.reset_envs([0], H.episode_arrays(g, [e], mc, ms))
            a = np.zeros((1, env.n_u)); a[0, :n_u] = g["action"][t]
            obs, rew, done, _ = env.step(a, noise=w[t:t + 1], extras=True)
            rb = np.abs(env.intersec_dist[0] - g["ray_dist"][t]) > 1e-3
            bad += int(rb.sum())
            wrap = abs(abs(g["nav"][t, 2]) - np.pi) < 1e-3
            if wrap or rb.any():
                continue
            wo = max(wo, float(np.abs(obs[0] - g["obs"][t]).max()))
            wo2 = max(wo2, float(np.abs(obs[0, :16] - g["obs"][t, :16]).max()))
            wr = max(wr, abs(float(rew[0]) - g["reward"][t]) / max(1.0, abs(g["reward"][t])))
            w6 = max(w6, abs(float(env.last_reward_arr[0, 6]) - g["reward_arr"][t, 6]))
            st = env.state[0]
            ws = max(ws, float(np.abs(st[[0, 1, 2, 6, 7, 8, 9, 10, 11]] - g["state"][t][[0, 1, 2, 6, 7, 8, 9, 10, 11]]).max()))
            wp = max(wp, float(np.abs(st[0:3] - g["state"][t][0:3]).max()))
            wa = max(wa, float(H.angle_diff(st[3:6], g["state"][t][3:6]).max()))
            wv = max(wv, float(np.abs(st[6:9] - g["state"][t][6:9]).max()))
            wq = max(wq, float(np.abs(st[9:12] - g["state"][t][9:12]).max()))
        env.close()
        print(f"{name:58s} {wo:9.2e} {wo2:10.2e} {wr:9.2e} {w6:9.2e} {ws:9.2e} {wp:9.2e} {wa:9.2e} {wv:9.2e} {wq:9.2e} {bad:7d}")


if __name__ == "__main__":
    main()
