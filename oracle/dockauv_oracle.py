"""
oracle/dockauv_oracle.py -- CPU restatement (NumPy, float64, one env at a time) of the
reference's docking3d hot path.

*** TEST INFRASTRUCTURE ONLY. ***  Nothing under ``gym_dockauv_amd/`` may import this file.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker / the timed CPU baseline -- never as the product path.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference (in the build
container only) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
file against those vectors and against the known-answer values of the reference's own unit
tests.

Every function cites the reference lines it restates (paths relative to the reference root,
``gym_dockauv/...``).  The code is written from the maths, not transcribed: 6x6 matrices are
never assembled with hstack/vstack, Coriolis terms are cross products, etc.  Quirks Q1..Q15 of
SURVEY.md section 8a are reproduced and marked ``# Qn``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

PI = math.pi
TWO_PI = 2.0 * math.pi

# --------------------------------------------------------------------------------------
# Geometry helpers  (utils/geomutils.py)
# --------------------------------------------------------------------------------------


def ssa(angle):
    """Wrap to [-pi, pi) with Python floor-mod semantics.  utils/geomutils.py:4-11."""
    return (angle + PI) % TWO_PI - PI


def rot_zyx(phi: float, theta: float, psi: float) -> np.ndarray:
    """R_b^n(Theta), body -> NED.  utils/geomutils.py:14-43."""
    cf, sf = math.cos(phi), math.sin(phi)
    ct, st = math.cos(theta), math.sin(theta)
    cp, sp = math.cos(psi), math.sin(psi)
    return np.array([
        [cp * ct, -sp * cf + cp * st * sf, sp * sf + cp * cf * st],
        [sp * ct, cp * cf + sf * st * sp, -cp * sf + st * sp * cf],
        [-st, ct * sf, ct * cf],
    ])


def t_zyx(phi: float, theta: float) -> np.ndarray:
    """Euler-rate transformation T_Theta.  utils/geomutils.py:46-75 (tan, 1/cos: singular at +-pi/2)."""
    sf, cf = math.sin(phi), math.cos(phi)
    tt, ct = math.tan(theta), math.cos(theta)
    return np.array([
        [1.0, sf * tt, cf * tt],
        [0.0, cf, -sf],
        [0.0, sf / ct, cf / ct],
    ])


def skew(a: Sequence[float]) -> np.ndarray:
    """S(a), S(a) b = a x b.  utils/geomutils.py:106-128."""
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


# --------------------------------------------------------------------------------------
# Vehicle parameter tables (values of objects/vehicles/BlueROV2.xml:11-42, LAUV.xml:10-57 and
# of tests/objects/test_BlueROV2.xml:24-26 for the older added-mass set).
# --------------------------------------------------------------------------------------

_BASE_KEYS = (
    "m BY I_x I_y I_z I_xy I_xz I_yz x_G y_G z_G x_B y_B z_B "
    "X_udot Y_vdot Z_wdot K_pdot M_qdot N_rdot X_u Y_v Z_w K_p M_q N_r X_uu Y_vv Z_ww K_pp M_qq N_rr"
).split()
_LAUV_KEYS = (
    "N_urf N_uvf N_uvb M_uqf M_uwf M_uwb Z_uqf Z_uwf Z_uwb Y_urf Y_uvf Y_uvb "
    "N_vv M_ww Z_qq Y_rr N_v M_w Z_q Y_r N_uudr M_uuds Z_uuds Y_uudr"
).split()

BLUEROV2_PARAMS: Dict[str, float] = dict(
    m=11.5, BY=114.8, I_x=0.21, I_y=0.245, I_z=0.245, z_G=0.02,
    X_udot=-7.57, Y_vdot=-7.57, Z_wdot=-7.57, K_pdot=-0.12, M_qdot=-0.12, N_rdot=-0.12,
    X_u=-4.03, Y_v=-6.22, Z_w=-5.18, K_p=-0.07, M_q=-0.07, N_r=-0.07,
    X_uu=-18.18, Y_vv=-21.66, Z_ww=-36.99, K_pp=-1.55, M_qq=-1.55, N_rr=-1.55,
)
# the reference's *test* fixture uses the pre-18.05.2022 added mass values
BLUEROV2_TEST_PARAMS: Dict[str, float] = dict(BLUEROV2_PARAMS, X_udot=-5.5, Y_vdot=-12.7, Z_wdot=-14.57)

LAUV_PARAMS: Dict[str, float] = dict(
    m=18.0, BY=177.58, I_x=0.0405, I_y=1.07, I_z=1.07, z_G=0.01,
    X_udot=-1.0291, Y_vdot=-16.153, Z_wdot=-16.153, K_pdot=0.0, M_qdot=0.758, N_rdot=0.758,
    X_u=-2.4, Y_v=-23.0, Z_w=-23.0, K_p=-0.3, M_q=-9.7, N_r=-9.7,
    X_uu=-2.4, Y_vv=-80.0, Z_ww=-80.0, K_pp=-0.0006, M_qq=-9.1, N_rr=-9.1,
    N_urf=-3.072, N_uvf=7.68, N_uvb=3.3088, M_uqf=-3.072, M_uwf=-7.68, M_uwb=-3.3088,
    Z_uqf=-7.68, Z_uwf=-19.2, Z_uwb=-10.956, Y_urf=7.68, Y_uvf=-19.2, Y_uvb=-10.956,
    N_vv=-1.5, M_ww=1.5, Z_qq=-0.3, Y_rr=0.3, N_v=-3.1, M_w=3.1, Z_q=-11.5, Y_r=11.5,
    N_uudr=-7.68, M_uuds=-7.68, Z_uuds=-19.2, Y_uudr=19.2,
)

G_ACC = 9.81  # objects/statespace.py:62


class VehicleModel:
    """
    Fossen 6-DOF model of one vehicle type: constant matrices plus the velocity / attitude
    dependent terms.  Restates objects/statespace.py:86-397 and the two vehicle classes
    (objects/vehicles/BlueROV2.py:27-88, objects/vehicles/LAUV.py:29-110).

    kind: "bluerov2" (joystick, 6 inputs), "bluerov2_direct" (8 thrusters), "lauv" (3 inputs).
    """

    def __init__(self, kind: str = "bluerov2", params: Optional[Dict[str, float]] = None):
        self.kind = kind
        if params is None:
            params = LAUV_PARAMS if kind == "lauv" else BLUEROV2_PARAMS
        p = {k: 0.0 for k in _BASE_KEYS + _LAUV_KEYS}
        p.update(params)
        self.p = p
        self.m = p["m"]
        self.BY = p["BY"]
        self.W = self.m * G_ACC                                           # statespace.py:86-88
        self.r_G = np.array([p["x_G"], p["y_G"], p["z_G"]])
        self.r_B = np.array([p["x_B"], p["y_B"], p["z_B"]])
        # Q6: element [2,0] carries +I_xz in the reference (statespace.py:96-101); inert while I_xz == 0
        self.I_g = np.array([
            [p["I_x"], -p["I_xy"], -p["I_xz"]],
            [-p["I_xy"], p["I_y"], -p["I_yz"]],
            [p["I_xz"], -p["I_yz"], p["I_z"]],
        ])
        S = skew(self.r_G)
        self.I_b = self.I_g + self.m * S @ S.T                            # statespace.py:105-117
        # M_RB = H^T diag(m I3, I_g) H with H = [[I, S(r_G)^T],[0, I]]     statespace.py:138-161, geomutils.py:131-157
        H = np.eye(6)
        H[0:3, 3:6] = S.T
        M_cg = np.zeros((6, 6))
        M_cg[0:3, 0:3] = self.m * np.eye(3)
        M_cg[3:6, 3:6] = self.I_g
        self.M_RB = H.T @ M_cg @ H
        self.ma_diag = -np.array([p["X_udot"], p["Y_vdot"], p["Z_wdot"], p["K_pdot"], p["M_qdot"], p["N_rdot"]])
        self.M_A = np.diag(self.ma_diag)                                  # statespace.py:164-187
        self.M_inv = np.linalg.inv(self.M_RB + self.M_A)                  # statespace.py:190-197
        self.d_lin = np.array([p["X_u"], p["Y_v"], p["Z_w"], p["K_p"], p["M_q"], p["N_r"]])
        self.d_quad = np.array([p["X_uu"], p["Y_vv"], p["Z_ww"], p["K_pp"], p["M_qq"], p["N_rr"]])

        if kind == "bluerov2":                                            # BlueROV2.py:34-51
            self.B_const = np.diag([2.83, 2.83, 4.0, 0.436, 0.24, 0.378]) * 20.0
            self.u_bound = np.array([[-1.0, 1.0]] * 6)
        elif kind == "bluerov2_direct":                                   # BlueROV2.py:53-72
            T = np.array([
                [0.707, 0.707, -0.707, -0.707, 0, 0, 0, 0],
                [-0.707, 0.707, -0.707, 0.707, 0, 0, 0, 0],
                [0, 0, 0, 0, -1, -1, -1, -1],
                [0.06, -0.06, 0.06, -0.06, -0.218, -0.218, 0.218, 0.218],
                [0.06, 0.06, -0.06, -0.06, 0.120, -0.120, 0.120, -0.120],
                [-0.189, 0.189, 0.189, -0.189, 0, 0, 0, 0],
            ])
            self.B_const = T @ np.diag([40.0] * 8)
            self.u_bound = np.array([[-1.0, 1.0]] * 8)
        elif kind == "lauv":                                              # LAUV.py:103-110
            self.B_const = None
            d30 = 30.0 * PI / 180.0
            self.u_bound = np.array([[0.0, 14.0], [-d30, d30], [-d30, d30]])
        else:
            raise KeyError(f"unknown vehicle kind {kind!r}")
        self.n_u = self.u_bound.shape[0]

    # -- velocity dependent terms -------------------------------------------------------
    def coriolis_force(self, nu: np.ndarray) -> np.ndarray:
        """(C_RB(nu) + C_A(nu)) nu as cross products.  statespace.py:199-286."""
        v1, v2 = nu[0:3], nu[3:6]
        m, rg = self.m, self.r_G
        # C_RB = [[m S(v2), -m S(v2) S(rG)], [m S(rG) S(v2), -S(I_b v2)]]
        top = m * np.cross(v2, v1) - m * np.cross(v2, np.cross(rg, v2))
        bot = m * np.cross(rg, np.cross(v2, v1)) - np.cross(self.I_b @ v2, v2)
        # C_A = [[0, -S(a1)], [-S(a1), -S(a2)]] with a = M_A nu (diagonal M_A)
        a1 = self.ma_diag[0:3] * v1
        a2 = self.ma_diag[3:6] * v2
        top = top - np.cross(a1, v2)
        bot = bot - np.cross(a1, v1) - np.cross(a2, v2)
        return np.concatenate([top, bot])

    def damping_matrix(self, nu: np.ndarray) -> np.ndarray:
        """D(nu).  statespace.py:288-351; LAUV override LAUV.py:69-101 (cross terms, lift * abs(u))."""
        a = np.abs(nu)
        D = -np.diag(self.d_lin + self.d_quad * a)
        if self.kind == "lauv":
            p = self.p
            au = a[0]
            D[1, 5] += -(p["Y_r"] + p["Y_rr"] * a[5] + p["Y_urf"] * au)
            D[2, 4] += -(p["Z_q"] + p["Z_qq"] * a[4] + p["Z_uqf"] * au)
            D[4, 2] += -(p["M_w"] + p["M_ww"] * a[2] + (p["M_uwb"] + p["M_uwf"]) * au)
            D[5, 1] += -(p["N_v"] + p["N_vv"] * a[1] + (p["N_uvb"] + p["N_uvf"]) * au)
            D[1, 1] += -(p["Y_uvb"] + p["Y_uvf"]) * au
            D[2, 2] += -(p["Z_uwb"] + p["Z_uwf"]) * au
            D[4, 4] += -p["M_uqf"] * au
            D[5, 5] += -p["N_urf"] * au
        return D

    def restoring(self, phi: float, theta: float) -> np.ndarray:
        """g(eta).  statespace.py:353-397."""
        W, B = self.W, self.BY
        xg, yg, zg = self.r_G
        xb, yb, zb = self.r_B
        sf, cf, st, ct = math.sin(phi), math.cos(phi), math.sin(theta), math.cos(theta)
        return np.array([
            (W - B) * st,
            -(W - B) * ct * sf,
            -(W - B) * ct * cf,
            -(yg * W - yb * B) * ct * cf + (zg * W - zb * B) * ct * sf,
            (zg * W - zb * B) * st + (xg * W - xb * B) * ct * cf,
            -(xg * W - xb * B) * ct * sf - (yg * W - yb * B) * st,
        ])

    def input_matrix(self, nu: np.ndarray) -> np.ndarray:
        """B(nu).  BlueROV2.py:76-77 (constant); LAUV.py:59-67 (fins scale with signed u**2)."""
        if self.kind != "lauv":
            return self.B_const
        p = self.p
        uu = nu[0] ** 2
        return np.array([
            [1.0, 0.0, 0.0],
            [0.0, p["Y_uudr"] * uu, 0.0],
            [0.0, 0.0, p["Z_uuds"] * uu],
            [0.0, 0.0, 0.0],
            [0.0, 0.0, p["M_uuds"] * uu],
            [0.0, p["N_uudr"] * uu, 0.0],
        ])

    def unnormalize(self, action: np.ndarray) -> np.ndarray:
        """clip to [-1,1] then affine map onto u_bound.  objects/auvsim.py:67-75."""
        a = np.clip(action, -1.0, 1.0)
        lo, hi = self.u_bound[:, 0], self.u_bound[:, 1]
        return lo + (hi - lo) * (a + 1.0) / 2.0

    def state_dot(self, state: np.ndarray, u: np.ndarray, nu_c: np.ndarray) -> np.ndarray:
        """RHS of the 12 ODEs.  objects/auvsim.py:110-160."""
        phi, theta, psi = state[3], state[4], state[5]
        nu_r = state[6:12]
        nu = nu_r + nu_c
        out = np.empty(12)
        out[0:3] = rot_zyx(phi, theta, psi) @ nu[0:3]
        out[3:6] = t_zyx(phi, theta) @ nu[3:6]
        tau = self.input_matrix(nu_r) @ u
        rhs = tau - self.damping_matrix(nu_r) @ nu_r - self.coriolis_force(nu_r) - self.restoring(phi, theta)
        out[6:12] = self.M_inv @ rhs
        return out


def rkf45_step4(f, y: np.ndarray, h: float) -> np.ndarray:
    """
    One fixed step of the Fehlberg 4(5) tableau, returning the 4th-order solution only.
    utils/odesolver45.py:18-28.  Q1: the reference computes stage 6 and the 5th-order ``q`` and
    throws both away (auvsim.py:98); they do not influence any output and are skipped here.
    """
    s1 = f(y)
    s2 = f(y + h * s1 / 4.0)
    s3 = f(y + 3.0 * h * s1 / 32.0 + 9.0 * h * s2 / 32.0)
    s4 = f(y + 1932.0 * h * s1 / 2197.0 - 7200.0 * h * s2 / 2197.0 + 7296.0 * h * s3 / 2197.0)
    s5 = f(y + 439.0 * h * s1 / 216.0 - 8.0 * h * s2 + 3680.0 * h * s3 / 513.0 - 845.0 * h * s4 / 4104.0)
    return y + h * (25.0 * s1 / 216.0 + 1408.0 * s3 / 2565.0 + 2197.0 * s4 / 4104.0 - s5 / 5.0)


LOWPASS_T1 = 0.2  # objects/auvsim.py:40


def lowpass_alpha(h: float) -> float:
    """alpha = h / (h + T1).  utils/lowpassfilter.py:13-27 (Q5: frozen at first use = the config step size)."""
    return h / (h + LOWPASS_T1)


def auv_step(model: VehicleModel, state: np.ndarray, u_prev: np.ndarray, action: np.ndarray,
             nu_c: np.ndarray, h: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """
    AUVSim.step: un-normalise, low-pass, RKF45(4) step, wrap angles, post-step RHS.
    objects/auvsim.py:77-108.  Returns (state', u', state_dot').
    """
    a = lowpass_alpha(h)
    u = a * model.unnormalize(action) + (1.0 - a) * u_prev               # lowpassfilter.py:41
    new = rkf45_step4(lambda y: model.state_dot(y, u, nu_c), state, h)   # Q2: nu_c frozen over the step
    new[3:6] = ssa(new[3:6])                                             # Q3: wrap only after the step
    sdot = model.state_dot(new, u, nu_c)                                 # Q4: post-wrap RHS with the new u
    return new, u, sdot


# --------------------------------------------------------------------------------------
# Ocean current  (objects/current.py)
# --------------------------------------------------------------------------------------


@dataclass
class CurrentState:
    mu: float = 0.005
    V_min: float = 0.0
    V_max: float = 0.0
    V_c: float = 0.0
    alpha: float = 0.0
    beta: float = 0.0
    sigma: float = 0.0

    def ned(self) -> np.ndarray:
        """objects/current.py:55-76."""
        return np.array([
            self.V_c * math.cos(self.alpha) * math.cos(self.beta),
            self.V_c * math.sin(self.beta),
            self.V_c * math.sin(self.alpha) * math.cos(self.beta),
        ])

    def body(self, att: np.ndarray) -> np.ndarray:
        """nu_c = [R(Theta)^T v_c^n, 0, 0, 0].  objects/current.py:33-53."""
        out = np.zeros(6)
        out[0:3] = rot_zyx(att[0], att[1], att[2]).T @ self.ned()
        return out

    def sim(self, h: float, w: float) -> None:
        """Euler step of the Gauss-Markov speed + clip.  objects/current.py:78-96 (``w`` drawn by the caller)."""
        self.V_c += (-self.mu * self.V_c + w) * h
        self.V_c = float(np.clip(self.V_c, self.V_min, self.V_max))


# --------------------------------------------------------------------------------------
# Ray fan ("Radar")  (objects/sensor.py)
# --------------------------------------------------------------------------------------


class RayFan:
    """Ray layout and the per-step clamp / reduce.  objects/sensor.py:43-144."""

    def __init__(self, alpha: float, beta: float, ray_per_deg: float, max_dist: float, blocksize_reduce: int = 2,
                 freq: float = 1.0):
        tol = 10e-8
        if (alpha + tol) % ray_per_deg > 0.001 or (beta + tol) % ray_per_deg > 0.001:      # sensor.py:51-52
            raise KeyError("Initialize the radar with valid ray_per_deg for alpha and beta.")
        self.max_dist = float(max_dist)
        self.alpha_max = alpha / 2.0
        self.beta_max = beta / 2.0
        av = np.arange(-alpha / 2.0, alpha / 2.0 + tol, ray_per_deg)                       # sensor.py:56-61
        bh = np.arange(-beta / 2.0, beta / 2.0 + tol, ray_per_deg)
        self.n_v, self.n_h = av.shape[0], bh.shape[0]
        self.alpha = np.repeat(av, self.n_h)             # ray index = iv * n_h + ih
        self.beta = np.tile(bh, self.n_v)
        self.n_rays = self.n_v * self.n_h
        d = np.stack([np.ones(self.n_rays), np.sin(self.beta), np.sin(self.alpha)], axis=1)  # sensor.py:66-71
        self.rd_b = d / np.linalg.norm(d, axis=1)[:, None]
        self.block = int(blocksize_reduce)
        self.n_vr = -(-self.n_v // self.block)
        self.n_hr = -(-self.n_h // self.block)
        self.n_rays_reduced = self.n_vr * self.n_hr

    def directions_ned(self, att: np.ndarray) -> np.ndarray:
        """rd_n = normalise(R(Theta) rd_b).  sensor.py:97-102 (Q7: the R^T of __init__ is overwritten)."""
        rd = (rot_zyx(att[0], att[1], att[2]) @ self.rd_b.T).T
        return rd / np.linalg.norm(rd, axis=1)[:, None]

    def clamp(self, dist: Optional[np.ndarray]) -> np.ndarray:
        """sensor.py:104-120: None -> all max_dist; d < 0 or d > max -> max (NaN survives, Q13)."""
        if dist is None:
            return np.full(self.n_rays, self.max_dist)
        d = np.array(dist, dtype=float)
        d[(d < 0) | (d > self.max_dist)] = self.max_dist
        return d

    def reduce(self, dist: np.ndarray) -> np.ndarray:
        """
        block x block max over the [n_v, n_h] image, zero padded.  sensor.py:131-137 calls
        skimage.measure.block_reduce(func=np.max) (scikit-image ~0.19.3, requirements.txt:35; not vendored,
        not installed): documented behaviour = pad with cval=0 up to a multiple of the block, then func
        over each block.  np.max propagates NaN.
        """
        b = self.block
        img = np.zeros((self.n_vr * b, self.n_hr * b))
        img[: self.n_v, : self.n_h] = dist.reshape(self.n_v, self.n_h)
        return img.reshape(self.n_vr, b, self.n_hr, b).max(axis=(1, 3)).reshape(-1)


# --------------------------------------------------------------------------------------
# Ray / obstacle geometry  (objects/shape.py)
# --------------------------------------------------------------------------------------

NEG_INF = -math.inf


def ray_capsule(origin: np.ndarray, rd: np.ndarray, cap1: np.ndarray, cap2: np.ndarray, rad: float) -> float:
    """
    One ray against one capsule, with the exact case structure of the *vectorised* reference routine
    (objects/shape.py:327-390; that is the one the env calls, docking3d.py:424-429).  ``rd`` need not be unit.
    Can return a negative distance (capsule behind the origin).  Q13: y<=0 / y>=0 overlap at 0, a == 0 divides
    by zero (IEEE inf/NaN propagate exactly as in NumPy).
    """
    with np.errstate(all="ignore"):
        ba = cap2 - cap1
        oa = origin - cap1
        rd = rd / np.linalg.norm(rd)
        baba = float(ba @ ba)
        bard = float(rd @ ba)
        baoa = float(oa @ ba)
        rdoa = float(rd @ oa)
        oaoa = float(oa @ oa)
        a = np.float64(baba - bard * bard)
        b = np.float64(baba * rdoa - baoa * bard)
        c = np.float64(baba * oaoa - baoa * baoa - rad * rad * baba)
        h = b * b - a * c
        if not (h >= 0):
            t = np.float64(NEG_INF)
        else:
            t = (-b - np.sqrt(h)) / a
        y = baoa + t * bard
        res = np.float64(0.0)
        body = bool(h >= 0) and bool(y > 0) and bool(y < baba)
        if body:
            res = t
        # cap selection: y <= 0 -> bottom sphere, then y >= 0 overrides with top sphere; NaN -> zero vector
        oc = np.zeros(3)
        if y <= 0.0:
            oc = oa
        if y >= 0.0:
            oc = origin - cap2
        b2 = np.float64(rd @ oc)
        c2 = np.float64(oc @ oc - rad * rad)
        h2 = b2 * b2 - c2
        if bool(h >= 0) and bool(h2 > 0.0) and not body:
            res = -b2 - np.sqrt(h2)
        if bool(h <= 0) or bool(res == 0):
            res = np.float64(NEG_INF)
        return float(res)


def ray_spheres(origin: np.ndarray, rd: np.ndarray, centers: np.ndarray, radii: np.ndarray) -> float:
    """
    One ray against all spheres: nearest *positive* entry distance, else the first sphere's value.
    objects/shape.py:235-264 (origin inside a sphere counts as a miss, :262).
    """
    with np.errstate(all="ignore"):
        rd = rd / np.linalg.norm(rd)
        oc = origin[None, :] - centers
        b = oc @ rd
        c = np.linalg.norm(oc, axis=1) ** 2 - radii ** 2
        h = b * b - c
        h = np.where(h < 0.0, NEG_INF, h)
        h = np.where(h >= 0.0, np.sqrt(np.where(h >= 0.0, h, 0.0)), h)
        res = np.minimum(-b + h, -b - h)
        return float(res[np.where(res > 0, res, np.inf).argmin()])


def seg_point_distance(po: np.ndarray, l1: np.ndarray, l2: np.ndarray) -> float:
    """Distance point <-> segment.  objects/shape.py:393-417."""
    d = (l2 - l1) / np.linalg.norm(l2 - l1)
    s = float((l1 - po) @ d)
    t = float((po - l2) @ d)
    hh = max(s, t, 0.0)
    c = np.cross(po - l1, d)
    return float(np.hypot(hh, np.linalg.norm(c)))


def vec_line_point(po: np.ndarray, l1: np.ndarray, l2: np.ndarray) -> np.ndarray:
    """Vector from the point to its projection on the (infinite) line.  objects/shape.py:420-433."""
    d = (l2 - l1) / np.linalg.norm(l2 - l1)
    t = float((po - l1) @ d)
    return l1 + t * d - po


def capsule_from_center(position: np.ndarray, radius: float, vec_top: np.ndarray) -> Tuple[np.ndarray, np.ndarray, float]:
    """Capsule ctor: vec_bot = 2 * position - vec_top.  objects/shape.py:98-108.  Returns (bot, top, r)."""
    position = np.asarray(position, dtype=float)
    vec_top = np.asarray(vec_top, dtype=float)
    return position - (vec_top - position), vec_top, float(radius)


# --------------------------------------------------------------------------------------
# Reward pieces  (envs/docking3d.py:706-792)
# --------------------------------------------------------------------------------------


def log_precision(x: float, x_goal: float, x_max: float) -> float:
    """docking3d.py:711-723."""
    eps = 0.001
    with np.errstate(all="ignore"):
        return float(1.0 - np.clip(np.log(max(x, eps) / x_max) / np.log(max(x_goal, eps) / x_max), 0.0, 1.0))


def cont_goal_constraints(x, delta_d, x_des, delta_d_des, x_max, delta_d_max, x_exp=1.0, delta_d_exp=1.0,
                          x_rev=False, delta_d_rev=False) -> float:
    """docking3d.py:742-764."""
    r_x = abs(float(x_rev) - log_precision(x, x_des, x_max)) ** x_exp
    r_d = abs(float(delta_d_rev) - log_precision(delta_d, delta_d_des, delta_d_max)) ** delta_d_exp
    return r_x * r_d


def obstacle_avoidance(theta_r, psi_r, d_r, theta_max, psi_max, d_max, gamma_c=1.0, epsilon_c=0.001,
                       epsilon_oa=0.01) -> float:
    """docking3d.py:766-792: sum(beta) / sum(max((gamma (1-c))^2, eps_c) beta) - 1 over ALL rays (unreduced)."""
    beta = (1.0 - np.abs(theta_r) / theta_max) * (1.0 - np.abs(psi_r) / psi_max) + epsilon_oa
    c = np.clip(1.0 - d_r / d_max, 0.0, 1.0)
    return float(np.sum(beta) / (np.maximum((gamma_c * (1.0 - c)) ** 2, epsilon_c) @ beta) - 1.0)


# --------------------------------------------------------------------------------------
# Default config (same key schema as config/env_config.py:20-91; values are the reference defaults)
# --------------------------------------------------------------------------------------


def default_config() -> dict:
    return {
        "max_timesteps": 1000,
        "t_step_size": 0.10,
        "max_dist_from_goal": 20,
        "max_attitude": 60 / 180 * PI,
        "dist_goal_reached_tol": 0.5,
        "vehicle": "BlueROV2",
        "u_max": 2.0, "v_max": 1.5, "w_max": 1.5,
        "p_max": 90 * PI / 180, "q_max": 90 * PI / 180, "r_max": 120 * PI / 180,
        "reward_set": 1,
        "reward_factors": {
            "w_d": 1.1, "w_delta_psi": 0.5, "w_delta_theta": 0.3, "w_phi": 0.3, "w_theta": 0.3,
            "w_Thetadot": 0.2, "w_t": 0.05, "w_oa": 0.20, "w_goal": 400.0, "w_deltad_max": -200.0,
            "w_Theta_max": -200.0, "w_t_max": -100.0, "w_col": -300.0,
        },
        "action_reward_factors": 6.0,
        "radar": {"freq": 1, "alpha": 60 * PI / 180, "beta": 80 * PI / 180, "ray_per_deg": 10 * PI / 180,
                  "max_dist": 10, "blocksize_reduce": 2},
    }


VEHICLE_KINDS = {"BlueROV2": "bluerov2", "BlueROV2_direct": "bluerov2_direct", "LAUV": "lauv"}

SCENARIOS = ("SimpleDocking3d", "SimpleCurrentDocking3d", "CapsuleDocking3d", "CapsuleCurrentDocking3d",
             "ObstaclesDocking3d", "ObstaclesNoCapDocking3d", "ObstaclesCurrentDocking3d", "SphereDocking3d")
SPHERE_SEED = 10_000   # SphereDocking3d (build-defined, BASELINE config 3): obstacle field of episode k

SAFETY_RADIUS = 1.0  # Q9: hard-wired, objects/auvsim.py:43 (config "radius" is ignored)


@dataclass
class Episode:
    """What a reset produces (inputs of the step path)."""
    position: np.ndarray
    attitude: np.ndarray
    goal: np.ndarray
    heading_goal: float
    current: CurrentState
    capsules: List[Tuple[np.ndarray, np.ndarray, float]] = field(default_factory=list)
    sphere_centers: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    sphere_radii: np.ndarray = field(default_factory=lambda: np.zeros(0))


def sphere_field(rs: np.random.RandomState, goal: np.ndarray, n_spheres: int = 8, r_min: float = 3.0,
                 r_max: float = 12.0) -> Tuple[np.ndarray, np.ndarray]:
    """Obstacle field of the build-defined SphereDocking3d scenario (the reference
    ships the sphere routines, objects/shape.py:235-264, but no env that uses them): centres uniform in direction,
    distance U(r_min, r_max) from the goal, radii U(0.5, 1.5).  Same draws as SphereDocking3d.generate_environment
    of oracle/gen_golden.py, which produced the traj_SphereDocking3d_* fixtures on the reference."""
    d = rs.normal(size=(n_spheres, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    centers = np.asarray(goal, dtype=float)[None, :3] + d * rs.uniform(r_min, r_max, n_spheres)[:, None]
    radii = rs.uniform(0.5, 1.5, n_spheres)
    return centers, radii


def generate_episode(scenario: str, rng: np.random.RandomState, max_attitude: float, max_dist_from_goal: float,
                     sphere_rng: Optional[np.random.RandomState] = None) -> Episode:
    """
    Scenario generators with the reference's draw order on a legacy MT19937 stream.
    envs/docking3d.py:687-703 (random pos / att) and :795-988 (the seven scenarios).
    SphereDocking3d = SimpleDocking3d + sphere_field(sphere_rng) (its own stream, as in oracle/gen_golden.py).
    """
    # --- SimpleDocking3d.generate_environment, docking3d.py:803-825
    goal = np.zeros(3)
    heading = (rng.random_sample() - 0.5) * PI
    r = rng.random_sample(3) - 0.5                                           # generate_random_pos :694-696
    r[2] = abs(r[0] + r[1]) / 3.0 * np.sign(r[2])
    position = goal + r * (15.0 / np.linalg.norm(r))                         # Q10: relative to goal (0,0,0)
    a = (rng.random_sample(3) - 0.5) * 2.0                                   # generate_random_att :699-703
    attitude = a * np.array([max_attitude * 0.7, max_attitude * 0.7, PI])
    cur = CurrentState(mu=0.005, V_min=0.0, V_max=0.0, V_c=0.0, alpha=0.0, beta=0.0, sigma=0.0)
    ep = Episode(position=position, attitude=attitude, goal=goal, heading_goal=heading, current=cur)

    if scenario == "SimpleDocking3d":
        return ep
    if scenario == "SphereDocking3d":
        if sphere_rng is None:
            raise ValueError("SphereDocking3d needs sphere_rng")
        ep.sphere_centers, ep.sphere_radii = sphere_field(sphere_rng, ep.goal)
        return ep
    if scenario == "SimpleCurrentDocking3d":                                # docking3d.py:844-849
        ang = (rng.random_sample(2) - 0.5) * 2.0 * np.array([PI / 2, PI])
        speed = rng.random_sample() * 1.0
        ep.current = CurrentState(0.005, speed, speed, 0.5, ang[0], ang[1], 0.0)
        return ep

    # --- CapsuleDocking3d, docking3d.py:860-886
    cap_r, cap_h = 1.0, 4.0
    theta = rng.rand() * TWO_PI
    rad = cap_r + SAFETY_RADIUS
    gz = (rng.rand() - 0.5) * cap_h
    ep.goal = np.array([math.cos(theta) * rad, math.sin(theta) * rad, gz])
    cap = capsule_from_center(np.zeros(3), cap_r, np.array([0.0, 0.0, -cap_h / 2.0]))
    ep.capsules = [cap]
    v = vec_line_point(ep.goal, cap[1], cap[0])
    ep.heading_goal = float(ssa(math.atan2(v[1], v[0])))

    if scenario.startswith("Obstacles"):                                    # docking3d.py:919-946
        height = 2.0 * max_dist_from_goal
        th = rng.rand() * TWO_PI
        for _ in range(4):
            x, y = math.cos(th) * 6.0, math.sin(th) * 6.0
            th += TWO_PI / 4
            ep.capsules.append(capsule_from_center(np.array([x, y, 0.0]), 1.0, np.array([x, y, -height / 2.0])))
        if scenario == "ObstaclesNoCapDocking3d":                           # docking3d.py:957-965
            ep.capsules.pop(0)

    if scenario in ("CapsuleCurrentDocking3d", "ObstaclesCurrentDocking3d"):  # docking3d.py:904-908, 984-988
        ang = (rng.random_sample(2) - 0.5) * 2.0 * np.array([PI / 2, PI])
        ep.current = CurrentState(0.005, 0.5, 0.5, 0.5, ang[0], ang[1], 0.0)
    elif scenario not in SCENARIOS:
        raise KeyError(scenario)
    return ep


class OracleEnv:
    """
    One docking3d environment, float64, reference step order (envs/docking3d.py:346-402).

    ``rng`` stands for the reference's global legacy ``np.random`` stream: ``reset(seed)`` re-seeds it
    (docking3d.py:296-298), every ``step`` burns one ``normal`` draw (Q11, current.py:88), and scenario
    generation draws from it in the reference order.
    """

    n_rewards = 13

    def __init__(self, scenario: str = "SimpleDocking3d", config: Optional[dict] = None,
                 vehicle_params: Optional[Dict[str, float]] = None):
        cfg = default_config()
        if config:
            for k, v in config.items():
                if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                    cfg[k] = dict(cfg[k], **v)
                else:
                    cfg[k] = v
        self.cfg = cfg
        self.scenario = scenario
        self.model = VehicleModel(VEHICLE_KINDS[cfg["vehicle"]], vehicle_params)
        self.h = float(cfg["t_step_size"])
        rc = cfg["radar"]
        self.fan = RayFan(rc["alpha"], rc["beta"], rc["ray_per_deg"], rc["max_dist"], rc.get("blocksize_reduce", 2))
        self.n_obs = 16 + self.fan.n_rays_reduced
        rf = cfg["reward_factors"]
        self.w_done = np.array([rf["w_goal"], rf["w_deltad_max"], rf["w_Theta_max"], rf["w_t_max"], rf["w_col"]])
        self.rng = np.random.RandomState()
        self.episode = 0
        self.t_total_steps = 0
        self._clear()

    def _clear(self):
        self.state = np.zeros(12)
        self.state_dot = np.zeros(12)
        self.u = np.zeros(self.model.n_u)
        self.t_steps = 0
        self.collision = False
        self.goal_reached = False
        self.done = False
        self.conditions = [False] * 5
        self.last_reward = 0.0
        self.cumulative_reward = 0.0
        self.last_reward_arr = np.zeros(self.n_rewards)
        self.cum_reward_arr = np.zeros(self.n_rewards)
        self.observation = np.zeros(self.n_obs, dtype=np.float32)
        self.delta_d = self.delta_theta = self.delta_psi = self.delta_heading_goal = 0.0
        self.intersec_dist = np.full(self.fan.n_rays, self.fan.max_dist)

    # -- reset ---------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, episode: Optional[Episode] = None) -> np.ndarray:
        """docking3d.py:222-322.  Q8: returns the all-zero observation."""
        self._clear()
        if seed is not None:
            self.rng = np.random.RandomState(seed)
        self.episode += 1
        if episode is None:
            srng = np.random.RandomState(SPHERE_SEED + self.episode) if self.scenario == "SphereDocking3d" else None
            episode = generate_episode(self.scenario, self.rng, self.cfg["max_attitude"], self.cfg["max_dist_from_goal"],
                                       sphere_rng=srng)
        self.load_episode(episode)
        return self.observation

    def load_episode(self, ep: Episode) -> None:
        self.state[0:3] = ep.position
        self.state[3:6] = ep.attitude
        self.goal = np.array(ep.goal, dtype=float)
        self.heading_goal = float(ep.heading_goal)
        self.current = ep.current
        self.capsules = list(ep.capsules)
        self.sphere_centers = np.array(ep.sphere_centers, dtype=float).reshape(-1, 3)
        self.sphere_radii = np.array(ep.sphere_radii, dtype=float).reshape(-1)
        self.nu_c = self.current.body(self.state[3:6])

    # -- step ----------------------------------------------------------------------------
    def step(self, action: np.ndarray, noise: Optional[float] = None):
        action = np.asarray(action, dtype=float)
        # 1. current speed (Q11: one normal draw per step even with sigma = 0)
        w = self.rng.normal(0.0, self.current.sigma) if noise is None else noise
        self.current.sim(self.h, w)
        # 2. current in body frame at the PRE-step attitude (Q2); also what observe() reports
        self.nu_c = self.current.body(self.state[3:6])
        # 3. vehicle
        self.state, self.u, self.state_dot = auv_step(self.model, self.state, self.u, action, self.nu_c, self.h)
        # 4-6. rays
        self.intersec_dist = self.fan.clamp(self._ray_distances())
        # 7. body collision
        self.collision = self._body_collision()
        # 9. navigation errors (docking3d.py:404-413)
        diff = self.goal - self.state[0:3]
        self.delta_d = float(np.linalg.norm(diff))
        self.delta_theta = float(self.state[4] + ssa(math.atan2(diff[2], float(np.linalg.norm(diff[0:2])))))
        self.delta_psi = float(ssa(math.atan2(diff[1], diff[0]) - self.state[5]))
        self.delta_heading_goal = float(ssa(self.heading_goal - self.state[5]))
        # 10-12
        self.observation = self.observe()
        self.done = self._is_done()
        self.last_reward = self._reward(action)
        self.cumulative_reward += self.last_reward
        self.t_total_steps += 1
        self.t_steps += 1
        return self.observation, self.last_reward, self.done, {"conditions": list(self.conditions)}

    def _ray_distances(self) -> Optional[np.ndarray]:
        """docking3d.py:415-442: per ray the smallest positive over obstacle groups, else group 0's value."""
        groups = len(self.capsules) + (1 if self.sphere_radii.shape[0] > 0 else 0)
        if groups == 0:
            return None
        rd = self.fan.directions_ned(self.state[3:6])
        pos = self.state[0:3]
        out = np.empty(self.fan.n_rays)
        for k in range(self.fan.n_rays):
            vals = [ray_capsule(pos, rd[k], c[0], c[1], c[2]) for c in self.capsules]
            if self.sphere_radii.shape[0] > 0:
                vals.append(ray_spheres(pos, rd[k], self.sphere_centers, self.sphere_radii))
            v = np.array(vals)
            with np.errstate(all="ignore"):
                out[k] = v[np.where(v > 0, v, np.inf).argmin()]
        return out

    def _body_collision(self) -> bool:
        """docking3d.py:444-460, shape.py:182-210."""
        pos = self.state[0:3]
        hit = False
        if self.sphere_radii.shape[0] > 0:
            hit |= bool(np.any(np.linalg.norm(self.sphere_centers - pos[None, :], axis=1) <= SAFETY_RADIUS + self.sphere_radii))
        for bot, top, r in self.capsules:
            hit |= seg_point_distance(pos, bot, top) <= r + SAFETY_RADIUS
        return bool(hit)

    def observe(self) -> np.ndarray:
        """docking3d.py:462-488."""
        c = self.cfg
        s = self.state
        obs = np.zeros(self.n_obs, dtype=np.float32)
        with np.errstate(all="ignore"):
            obs[0] = np.clip(1 - (np.log(self.delta_d / c["max_dist_from_goal"])
                                  / np.log(c["dist_goal_reached_tol"] / c["max_dist_from_goal"])), 0, 1)
        obs[1] = np.clip(self.delta_theta / (PI / 2), -1, 1)
        obs[2] = np.clip(self.delta_psi / PI, -1, 1)
        obs[3] = np.clip(s[6] / c["u_max"], -1, 1)
        obs[4] = np.clip(s[7] / c["v_max"], -1, 1)
        obs[5] = np.clip(s[8] / c["w_max"], -1, 1)
        obs[6] = np.clip(s[3] / c["max_attitude"], -1, 1)
        obs[7] = np.clip(s[4] / c["max_attitude"], -1, 1)
        obs[8] = np.clip(math.sin(s[5]), -1, 1)
        obs[9] = np.clip(math.cos(s[5]), -1, 1)
        obs[10] = np.clip(s[9] / c["p_max"], -1, 1)
        obs[11] = np.clip(s[10] / c["q_max"], -1, 1)
        obs[12] = np.clip(s[11] / c["r_max"], -1, 1)
        obs[13:16] = np.clip(self.nu_c[0:3] / 2, -1, 1)
        obs[16:] = np.clip(self.fan.reduce(self.intersec_dist) / self.fan.max_dist, 0, 1)
        return obs

    def _is_done(self) -> bool:
        """docking3d.py:597-631 (t_steps is the pre-increment counter)."""
        c = self.cfg
        self.conditions = [
            bool(self.delta_d < c["dist_goal_reached_tol"]),
            bool(self.delta_d > c["max_dist_from_goal"]),
            bool(np.any(np.abs(self.state[3:5]) > c["max_attitude"])),
            bool(self.t_steps >= c["max_timesteps"]),
            bool(self.collision),
        ]
        if self.conditions[0]:
            self.goal_reached = True
        return bool(any(self.conditions))

    def _reward(self, action: np.ndarray) -> float:
        """docking3d.py:490-595."""
        c = self.cfg
        rf = c["reward_factors"]
        r = self.last_reward_arr
        dtol, dmax = c["dist_goal_reached_tol"], c["max_dist_from_goal"]
        r[0] = -rf["w_d"] * log_precision(self.delta_d, dtol, dmax)
        oa = obstacle_avoidance(self.fan.alpha, self.fan.beta, self.intersec_dist, self.fan.alpha_max,
                                self.fan.beta_max, self.fan.max_dist, 1.0, 0.001, 0.01)
        if c["reward_set"] == 1:
            r[1] = -rf["w_delta_theta"] * (self.delta_theta / (PI / 2)) ** 2
            r[2] = -rf["w_delta_psi"] * (self.delta_psi / PI) ** 2
            r[6] = -rf["w_oa"] * oa
        elif c["reward_set"] == 2:
            r[1] = -rf["w_delta_theta"] * cont_goal_constraints(abs(self.delta_theta), self.delta_d, 0.0, dtol,
                                                                  PI / 2, dmax, 4.0, 4.0)
            r[2] = -rf["w_delta_psi"] * cont_goal_constraints(abs(self.delta_psi), self.delta_d, 0.0, dtol,
                                                                PI, dmax, 4.0, 4.0)
            r[6] = -rf["w_oa"] * cont_goal_constraints(abs(oa), self.delta_d, 0.0, dtol, 1.0, dmax, 4.0, 4.0)
        r[3] = -rf["w_phi"] * (self.state[3] / (PI / 2)) ** 2
        r[4] = -rf["w_theta"] * (self.state[4] / (PI / 2)) ** 2
        r[5] = -rf["w_Thetadot"] * (float(np.linalg.norm(self.state_dot[3:6])) / c["p_max"]) ** 2
        r[7] = -float(np.sum((np.abs(action) / self.model.n_u) ** 2 * np.asarray(c["action_reward_factors"])))
        r[8:13] = np.array(self.conditions, dtype=float) * self.w_done          # Q12: several may fire, they add
        self.cum_reward_arr = self.cum_reward_arr + r
        return float(np.sum(r))
