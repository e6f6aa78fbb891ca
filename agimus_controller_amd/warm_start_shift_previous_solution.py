"""Warm start by shifting the previous solution by the first time step
(warm_start_shift_previous_solution.py:23-109 upstream).  Nodes whose step equals the first one
take their successor's state/control; coarser nodes are advanced by one fine Euler step with
their own control kept.  The Euler step is the HIP `integrate` kernel (no costs involved)."""

from __future__ import annotations

import numpy as np

from .trajectory import TrajectoryPoint
from .warm_start_base import WarmStartBase
from .warm_start_reference import WarmStartReference  # noqa: F401  (kept importable from here like upstream)


class WarmStartShiftPreviousSolution(WarmStartBase):
    def __init__(self) -> None:
        super().__init__()
        self._integrate = None

    def setup(self, robot_models, ocp_params, ocp=None) -> None:
        """`ocp` (optional): an OCP of this package whose device problem provides the Euler step;
        without it a bare device problem with no costs is created from `robot_models`."""
        self._timesteps = ocp_params.timesteps
        self._dt = self._timesteps[0]
        assert ocp_params.dt == self._timesteps[0]
        assert all(dt >= self._dt for dt in self._timesteps)
        if ocp is not None:
            self._integrate = ocp.integrate
        else:
            from . import _abi, backend

            po = _abi.PackedOcp(robot_models.table.nv, [self._dt], [], [])
            self._hip = backend.HipOcp(robot_models.table, po, 1)
            self._integrate = lambda x, u: self._hip.integrate(x, u)[0]

    def generate(self, initial_state: TrajectoryPoint, reference_trajectory: list[TrajectoryPoint]):
        assert self._previous_solution is not None, (
            "WarmStartBase.update_previous_solution should have been called before generate can work."
        )
        self.shift()
        x0 = np.concatenate([initial_state.robot_configuration, initial_state.robot_velocity])
        return x0, self._previous_solution.states.copy(), self._previous_solution.feed_forward_terms.copy()

    def shift(self):
        xs = self._previous_solution.states
        us = self._previous_solution.feed_forward_terms
        n = len(self._timesteps)
        assert len(xs) == n + 1 and len(us) == n
        for i, dt in enumerate(self._timesteps):
            if dt == self._dt:
                xs[i] = xs[i + 1]
                if i < n - 1:  # the last control has no successor: it is duplicated
                    us[i] = us[i + 1]
            else:
                assert dt > self._dt
                xs[i] = np.array(self._integrate(xs[i], us[i]))
