"""Receding-horizon orchestration with the reference's `MPC` surface
(agimus_controller/agimus_controller/mpc.py:13-95): setup / run / integrate / append_*."""

from __future__ import annotations

import time

import numpy.typing as npt

from .mpc_data import MPCDebugData, OCPResults
from .ocp_base import OCPBase
from .trajectory import TrajectoryBuffer, TrajectoryPoint, WeightedTrajectoryPoint
from .warm_start_base import WarmStartBase


class MPC:
    def __init__(self) -> None:
        self._ocp: OCPBase = None
        self._warm_start: WarmStartBase = None
        self._buffer: TrajectoryBuffer = None
        self._mpc_debug_data: MPCDebugData = None

    def setup(self, ocp: OCPBase, warm_start: WarmStartBase, buffer: TrajectoryBuffer) -> None:
        self._ocp, self._warm_start, self._buffer = ocp, warm_start, buffer
        self._mpc_debug_data = MPCDebugData(ocp=ocp.debug_data)

    def run(self, initial_state: TrajectoryPoint, current_time_ns: int) -> OCPResults:
        assert self._ocp is not None and self._warm_start is not None
        t_begin = time.perf_counter_ns()
        if len(self._buffer) < self._ocp.n_controls + 1:
            return None
        horizon = self._extract_horizon_from_buffer()
        self._ocp.set_reference_weighted_trajectory(horizon)
        t_refs = time.perf_counter_ns()
        points = [wp.point for wp in horizon]
        x0, x_init, u_init = self._warm_start.generate(initial_state, points)
        assert len(x_init) == self._ocp.n_controls + 1 and len(u_init) == self._ocp.n_controls
        t_ws = time.perf_counter_ns()
        self._ocp.solve(x0, x_init, u_init)
        self._warm_start.update_previous_solution(self._ocp.ocp_results)
        self._buffer.clear_past()
        t_end = time.perf_counter_ns()
        dbg = self._mpc_debug_data
        dbg.ocp = self._ocp.debug_data
        dbg.reference_id = points[0].id
        dbg.duration_iteration_ns = t_end - t_begin
        dbg.duration_horizon_update_ns = t_refs - t_begin
        dbg.duration_generate_warm_start_ns = t_ws - t_refs
        dbg.duration_ocp_solve_ns = t_end - t_ws
        return self._ocp.ocp_results

    def integrate(self, state: TrajectoryPoint, control: npt.NDArray) -> TrajectoryPoint:
        """One OCP time step forward from `state` under `control` (modifies and returns `state`)."""
        x = self._ocp.integrate(state.robot_state, control)
        nq = len(state.robot_configuration)
        state.time_ns += int(self._ocp.dt * 1e-9)  # as upstream (mpc.py:77); SURVEY App. C notes the factor
        state.robot_configuration = x[:nq]
        state.robot_velocity = x[nq:]
        return state

    @property
    def mpc_debug_data(self) -> MPCDebugData:
        return self._mpc_debug_data

    def append_trajectory_point(self, trajectory_point: WeightedTrajectoryPoint):
        self._buffer.append(trajectory_point)

    def append_trajectory_points(self, trajectory_points: list[WeightedTrajectoryPoint]):
        self._buffer.extend(trajectory_points)

    def _extract_horizon_from_buffer(self):
        return self._buffer.horizon
