"""Warm start from the reference trajectory (warm_start_reference.py:12-96 upstream): states are the
reference states (first one replaced by the measured state), controls are the inverse dynamics of
the reference -- `pin.rnea` upstream, the batched RNEA kernel behind the C ABI here."""

from __future__ import annotations

import itertools

import numpy as np

from .trajectory import TrajectoryPoint
from .warm_start_base import WarmStartBase


class WarmStartReference(WarmStartBase):
    def __init__(self) -> None:
        super().__init__()
        self._rnea = None
        self._nx = 0
        self._nv = 0

    def setup(self, rmodel) -> None:
        """`rmodel`: anything with `.nq`, `.nv` and a batched `rnea(q, v, a)`; an OCP of this package
        (its device problem is used) or a `backend.HipOcp` qualify."""
        hip = getattr(rmodel, "_hip", rmodel)
        assert hasattr(hip, "rnea"), "WarmStartReference.setup needs an object exposing rnea(q, v, a)"
        self._rnea = hip.rnea
        self._nv = hip.nv
        self._nx = 2 * hip.nv

    def generate(self, initial_state: TrajectoryPoint, reference_trajectory: list[TrajectoryPoint]):
        assert self._rnea is not None, "Robot model is missing in warmstart. please use warmstart.setup(rmodel)"
        n_states = len(reference_trajectory)
        x0 = np.concatenate([initial_state.robot_configuration, initial_state.robot_velocity])
        assert x0.shape[0] == self._nx, f"Expected x0 shape {self._nx},from provided reference got {x0.shape}"
        head = list(itertools.chain([initial_state], reference_trajectory[1:]))
        x_init = [np.hstack([p.robot_configuration, p.robot_velocity]) for p in head]
        assert np.array(x_init).shape == (n_states, self._nx)
        ctrl_pts = head[:-1]  # one control fewer than states
        q = np.array([p.robot_configuration for p in ctrl_pts])
        v = np.array([p.robot_velocity for p in ctrl_pts])
        a = np.array([p.robot_acceleration for p in ctrl_pts])
        u_init = list(self._rnea(q, v, a)) if ctrl_pts else []
        assert np.array(u_init).shape == (n_states - 1, self._nv)
        return x0, x_init, u_init
