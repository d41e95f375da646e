"""
Host-side vehicle models: flat-XML parameter loader and the constant matrices handed to libdockauv.

Mirrors the reference's vehicle interface for this path (same names, same XML schema, same error behaviour):
``StateSpace.read_phys_para_from_xml`` (objects/statespace.py:428-448), the cached constants ``W, I_g, I_b, M_RB,
M_A, M_inv`` (statespace.py:86-197), ``BlueROV2(xml_path, control_mode)`` (objects/vehicles/BlueROV2.py:27-88) and
``LAUV(xml_path)`` (objects/vehicles/LAUV.py:29-110).  The velocity dependent terms (C, D, g, B(nu)) are evaluated
on the GPU; here they only exist as the numbers the kernels need.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET
from typing import Dict, Optional

import numpy as np

from .. import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
G_ACC = 9.81  # statespace.py:62

_STATESPACE_FLOATS = (
    "m BY I_x I_y I_z I_xy I_xz I_yz x_G y_G z_G x_B y_B z_B X_udot Y_vdot Z_wdot K_pdot M_qdot N_rdot "
    "X_u Y_v Z_w K_p M_q N_r X_uu Y_vv Z_ww K_pp M_qq N_rr"
).split()
_LAUV_FLOATS = (
    "N_urf N_uvf N_uvb M_uqf M_uwf M_uwb Z_uqf Z_uwf Z_uwb Y_urf Y_uvf Y_uvb N_vv M_ww Z_qq Y_rr N_v M_w Z_q Y_r "
    "N_uudr M_uuds Z_uuds Y_uudr"
).split()


class VehicleModel:
    """Constants of one vehicle type.  Attribute names follow the reference's StateSpace."""

    name = "AUV_name_here"
    version = 0.0
    kind = _capi.VEH_CONSTB
    _extra_floats: tuple = ()

    def __init__(self):
        self.g = G_ACC
        for k in _STATESPACE_FLOATS + list(self._extra_floats):
            setattr(self, k, 0.0)
        self.safety_radius = 1  # objects/auvsim.py:43 (hard-wired in the reference)
        self._B = None
        self._u_bound = None

    # -- XML -----------------------------------------------------------------------------------------
    def read_phys_para_from_xml(self, xml_path: str) -> None:
        """Flat XML -> attributes; the type is inferred from the pre-initialised attribute and unknown tags are
        rejected, as in the reference (objects/statespace.py:439-448)."""
        root = ET.parse(xml_path).getroot()
        for child in root:
            if hasattr(self, child.tag):
                setattr(self, child.tag, type(getattr(self, child.tag))(child.text))
            else:
                raise AttributeError("Bad and not allowed practice: Trying to parse xml data tag without it being "
                                     "initialized in init")

    # -- constants (statespace.py:86-197) -------------------------------------------------------------
    @property
    def W(self) -> float:
        return self.m * self.g

    @property
    def r_G(self) -> np.ndarray:
        return np.array([self.x_G, self.y_G, self.z_G], dtype=float)

    @property
    def r_B(self) -> np.ndarray:
        return np.array([self.x_B, self.y_B, self.z_B], dtype=float)

    @staticmethod
    def _skew(a) -> np.ndarray:
        return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])

    @property
    def I_g(self) -> np.ndarray:
        # the reference puts +I_xz at [2,0] (statespace.py:100); only inert while I_xz == 0, so insist on that
        if self.I_xz != 0.0:
            raise ValueError("I_xz != 0 is not supported: the reference's I_g is not symmetric in that case")
        return np.array([[self.I_x, -self.I_xy, -self.I_xz],
                         [-self.I_xy, self.I_y, -self.I_yz],
                         [self.I_xz, -self.I_yz, self.I_z]], dtype=float)

    @property
    def I_b(self) -> np.ndarray:
        S = self._skew(self.r_G)
        return self.I_g + self.m * S @ S.T

    @property
    def M_RB(self) -> np.ndarray:
        S = self._skew(self.r_G)
        H = np.eye(6)
        H[0:3, 3:6] = S.T
        M = np.zeros((6, 6))
        M[0:3, 0:3] = self.m * np.eye(3)
        M[3:6, 3:6] = self.I_g
        return H.T @ M @ H

    @property
    def M_A(self) -> np.ndarray:
        return -np.diag([self.X_udot, self.Y_vdot, self.Z_wdot, self.K_pdot, self.M_qdot, self.N_rdot])

    @property
    def M_inv(self) -> np.ndarray:
        return np.linalg.inv(self.M_RB + self.M_A)

    @property
    def u_bound(self) -> np.ndarray:
        return self._u_bound

    def B(self, nu=None) -> Optional[np.ndarray]:
        return self._B

    # -- hand-over to the C ABI ------------------------------------------------------------------------
    def to_capi(self) -> "_capi.Vehicle":
        v = _capi.Vehicle()
        v.kind = self.kind
        ub = np.asarray(self.u_bound, dtype=float)
        n_u = ub.shape[0]
        if n_u > _capi.MAX_U:
            raise ValueError(f"at most {_capi.MAX_U} inputs are supported")
        v.n_u = n_u
        v.m, v.W, v.BY = float(self.m), float(self.W), float(self.BY)
        v.r_G[:] = self.r_G.tolist()
        v.r_B[:] = self.r_B.tolist()
        v.I_b[:] = self.I_b.reshape(-1).tolist()
        v.ma_diag[:] = np.diag(self.M_A).tolist()
        v.d_lin[:] = [self.X_u, self.Y_v, self.Z_w, self.K_p, self.M_q, self.N_r]
        v.d_quad[:] = [self.X_uu, self.Y_vv, self.Z_ww, self.K_pp, self.M_qq, self.N_rr]
        v.M_inv[:] = self.M_inv.reshape(-1).tolist()
        Bm = np.zeros((6, _capi.MAX_U))
        if self.kind == _capi.VEH_CONSTB:
            Bm[:, :n_u] = np.asarray(self._B, dtype=float)
        v.B[:] = Bm.reshape(-1).tolist()
        lo, hi = np.zeros(_capi.MAX_U), np.zeros(_capi.MAX_U)
        lo[:n_u], hi[:n_u] = ub[:, 0], ub[:, 1]
        v.u_lo[:] = lo.tolist()
        v.u_hi[:] = hi.tolist()
        v.lauv[:] = self._lauv_block()
        return v

    def _lauv_block(self):
        return [0.0] * 20


class BlueROV2(VehicleModel):
    """BlueROV2 heavy; ``control_mode`` "joystick" (6 inputs, B = diag * 20) or "direct" (8 thrusters).
    Reference: objects/vehicles/BlueROV2.py:27-88."""

    kind = _capi.VEH_CONSTB

    def __init__(self, xml_path: str = os.path.join(_HERE, "vehicles", "BlueROV2.xml"), control_mode: str = "joystick"):
        super().__init__()
        self.read_phys_para_from_xml(xml_path)
        if control_mode == "joystick":
            self.K_thrust = 20
            self._B = np.diag([2.83, 2.83, 4.0, 0.436, 0.24, 0.378]) * self.K_thrust
            self._u_bound = np.array([[-1.0, 1.0]] * 6)
        elif control_mode == "direct":
            self.K_thrust = np.diag([40.0] * 8)
            self.T_thrust = np.array([
                [0.707, 0.707, -0.707, -0.707, 0, 0, 0, 0],
                [-0.707, 0.707, -0.707, 0.707, 0, 0, 0, 0],
                [0, 0, 0, 0, -1, -1, -1, -1],
                [0.06, -0.06, 0.06, -0.06, -0.218, -0.218, 0.218, 0.218],
                [0.06, 0.06, -0.06, -0.06, 0.120, -0.120, 0.120, -0.120],
                [-0.189, 0.189, 0.189, -0.189, 0, 0, 0, 0]])
            self._B = self.T_thrust @ self.K_thrust
            self._u_bound = np.array([[-1.0, 1.0]] * 8)
        else:
            raise KeyError("Invalid control mode for BlueROV2 initialization.")

    # the reference exposes these for its tests (BlueROV2.py:84-88)
    def set_B(self, value):
        self._B = np.asarray(value, dtype=float)

    def set_u_bound(self, value):
        self._u_bound = np.asarray(value, dtype=float)


class LAUV(VehicleModel):
    """LAUV: 3 inputs (thrust 0..14 N, rudder / stern +-30 deg).  Reference: objects/vehicles/LAUV.py:29-110."""

    kind = _capi.VEH_LAUV
    _extra_floats = tuple(_LAUV_FLOATS)

    def __init__(self, xml_path: str = os.path.join(_HERE, "vehicles", "LAUV.xml")):
        super().__init__()
        self.read_phys_para_from_xml(xml_path)
        d30 = 30 * np.pi / 180
        self._u_bound = np.array([[0, 14], [-d30, d30], [-d30, d30]], dtype=float)

    def B(self, nu=None):
        u = 0.0 if nu is None else nu[0]
        uu = u ** 2
        return np.array([[1, 0, 0], [0, self.Y_uudr * uu, 0], [0, 0, self.Z_uuds * uu], [0, 0, 0],
                         [0, 0, self.M_uuds * uu], [0, self.N_uudr * uu, 0]], dtype=float)

    def _lauv_block(self):
        return [self.Y_r, self.Y_rr, self.Y_urf, self.Z_q, self.Z_qq, self.Z_uqf,
                self.M_w, self.M_ww, self.M_uwb + self.M_uwf, self.N_v, self.N_vv, self.N_uvb + self.N_uvf,
                self.Y_uvb + self.Y_uvf, self.Z_uwb + self.Z_uwf, self.M_uqf, self.N_urf,
                self.Y_uudr, self.Z_uuds, self.M_uuds, self.N_uudr]


VEHICLES: Dict[str, type] = {"BlueROV2": BlueROV2, "LAUV": LAUV}


def make_vehicle(name: str) -> VehicleModel:
    """The reference loads ``gym_dockauv.objects.vehicles.<name>.<name>`` dynamically (envs/docking3d.py:76-78)."""
    if name not in VEHICLES:
        raise ModuleNotFoundError(f"No vehicle named {name!r}; available: {sorted(VEHICLES)}")
    return VEHICLES[name]()
