"""
oracle/philox_ref.py -- NumPy restatement of the Philox4x32-10 counter RNG (Salmon, Moraes, Dror, Shaw: "Parallel
random numbers: as easy as 1, 2, 3", SC'11; constants of the Random123 reference implementation) and of the
counter/key convention of the in-kernel episode generator (gym_dockauv_amd/csrc/dockauv_step.hip.inc).

TEST INFRASTRUCTURE ONLY.  Pinned by the Random123 known-answer vectors in tests/test_philox.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """counter [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32."""
    c = np.array(counter, dtype=np.uint64)
    k = np.broadcast_to(np.array(key, dtype=np.uint64), c.shape[:-1] + (2,)).copy()
    for _ in range(10):
        p0 = M0 * c[..., 0]
        p1 = M1 * c[..., 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        n0 = hi1 ^ c[..., 1] ^ k[..., 0]
        n2 = hi0 ^ c[..., 3] ^ k[..., 1]
        c = np.stack([n0, lo1, n2, lo0], axis=-1)
        k[..., 0] = (k[..., 0] + W0) & MASK
        k[..., 1] = (k[..., 1] + W1) & MASK
    return c.astype(np.uint32)


def episode_uniforms(seed: int, env: np.ndarray, episode: np.ndarray) -> np.ndarray:
    """The 12 uniforms the kernel draws for (env, episode): counter = (env, episode, block, 0), key = seed,
    U = (x >> 8) * 2^-24.  Returns [n, 12] float64."""
    env = np.asarray(env, dtype=np.uint64)
    episode = np.asarray(episode, dtype=np.uint64)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    out = []
    for blk in range(3):
        ctr = np.stack([env, episode, np.full_like(env, blk), np.zeros_like(env)], axis=-1)
        x = philox4x32_10(ctr, key)
        out.append((x >> np.uint32(8)).astype(np.float64) / 16777216.0)
    return np.concatenate(out, axis=-1)


def philox_normal(seed: int, env: np.ndarray, episode: np.ndarray, t_steps: np.ndarray) -> np.ndarray:
    """The standard normal the kernel draws for the current's white noise of (env, episode, step) when
    dockauv_config.device_noise is set: counter = (env, episode, t_steps, 1), key = seed, Box-Muller on the first two
    words with u1 = ((x0 >> 8) + 0.5) 2^-24 in (0, 1), u2 = (x1 >> 8) 2^-24:  z = sqrt(-2 ln u1) cos(2 pi u2)."""
    env = np.asarray(env, dtype=np.uint64)
    episode = np.broadcast_to(np.asarray(episode, dtype=np.uint64), env.shape)
    t_steps = np.broadcast_to(np.asarray(t_steps, dtype=np.uint64), env.shape)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    ctr = np.stack([env, episode, t_steps, np.ones_like(env)], axis=-1)
    x = philox4x32_10(ctr, key)
    u1 = ((x[..., 0] >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    u2 = (x[..., 1] >> np.uint32(8)).astype(np.float64) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
