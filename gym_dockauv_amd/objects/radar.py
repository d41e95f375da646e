"""
Ray fan layout ("Radar") on the host: which rays exist, their body-frame directions and the obstacle-avoidance
weights.  Mirrors ``Radar.__init__`` of the reference (objects/sensor.py:43-87); the per-step work (rotate, intersect,
clamp, block-max reduce) runs inside the HIP step kernel.
"""
from __future__ import annotations

import numpy as np


class RadarLayout:
    def __init__(self, freq: float = 1, alpha: float = 2 * np.pi, beta: float = 2 * np.pi,
                 ray_per_deg: float = 5.0 * np.pi / 180, max_dist: float = 25, blocksize_reduce: int = 2):
        self.freq = freq                        # unused by the reference as well ("TODO" in env_config.py:84)
        self.max_dist = float(max_dist)
        tol = 10e-8
        if (alpha + tol) % ray_per_deg > 0.001 or (beta + tol) % ray_per_deg > 0.001:   # sensor.py:51-52
            raise KeyError("Initialize the radar with valid ray_per_deg for alpha and beta.")
        self.alpha_max = alpha / 2
        self.beta_max = beta / 2
        a = np.arange(-alpha / 2, alpha / 2 + tol, ray_per_deg)
        b = np.arange(-beta / 2, beta / 2 + tol, ray_per_deg)
        self.n_vertical, self.n_horizontal = a.shape[0], b.shape[0]
        self.alpha = np.repeat(a, self.n_horizontal)      # ray index = iv * n_horizontal + ih
        self.beta = np.tile(b, self.n_vertical)
        self.n_rays = self.alpha.shape[0]
        d = np.stack([np.ones(self.n_rays), np.sin(self.beta), np.sin(self.alpha)], axis=1)
        self.rd_b = d / np.linalg.norm(d, axis=1)[:, None]
        self.blocksize_reduce = int(blocksize_reduce)
        bs = self.blocksize_reduce
        self.n_rays_reduced = (-(-self.n_vertical // bs)) * (-(-self.n_horizontal // bs))

    def beta_oa(self, epsilon_oa: float = 0.01) -> np.ndarray:
        """Reward.beta_oa, envs/docking3d.py:786-788."""
        return (1 - np.abs(self.alpha) / self.alpha_max) * (1 - np.abs(self.beta) / self.beta_max) + epsilon_oa

    def ray_table(self) -> np.ndarray:
        """[n_rays][4] float64: unit body direction, obstacle-avoidance weight (dockauv_config.ray_table)."""
        return np.ascontiguousarray(np.concatenate([self.rd_b, self.beta_oa()[:, None]], axis=1), dtype=np.float64)
