"""HIP-backed counterpart of the reference's OCPBaseCroco
(agimus_controller/agimus_controller/ocp_base_croco.py:16-215): same constructor arguments,
properties and `solve` / `integrate` behaviour, with the Crocoddyl ShootingProblem +
mim_solvers.SolverCSQP pair replaced by one device-resident problem behind the C ABI.

Subclasses describe their costs by returning row tables (instead of Crocoddyl action models)
from `create_running_model_list` / `create_terminal_model`.
"""

from __future__ import annotations

import abc

import numpy as np
import numpy.typing as npt

from . import _abi, backend
from .factory.robot_model import RobotModels
from .mpc_data import OCPDebugData, OCPResults
from .ocp_base import OCPBase
from .ocp_param_base import OCPParamsBaseCroco


class OCPBaseCroco(OCPBase):
    def __init__(self, robot_models: RobotModels, ocp_params: OCPParamsBaseCroco, use_colmpc_state: bool = False, device: int = 0) -> None:
        self._robot_models = robot_models
        self._collision_model = robot_models.collision_model
        self._armature = robot_models.armature
        self._use_colmpc_state = use_colmpc_state
        self._ocp_params = ocp_params
        self._ocp_results: OCPResults = None
        self._debug_data: OCPDebugData = OCPDebugData()
        self._table = robot_models.table
        nv = self._table.nv
        self._nv, self._nx = nv, 2 * nv
        # one row table per node type; every running node shares the table, its time step is per node
        self._running_rows = self.create_running_model_list()
        self._terminal_rows = self.create_terminal_model()
        running_cons, terminal_cons = self.create_constraint_lists()
        self._packed = _abi.PackedOcp(
            nv,
            ocp_params.timesteps,
            self._running_rows,
            self._terminal_rows,
            running_constraints=running_cons,
            terminal_constraints=terminal_cons,
            termination_tolerance=ocp_params.termination_tolerance,
            max_qp_iters=ocp_params.qp_iters,
            eps_abs=ocp_params.eps_abs,
            eps_rel=ocp_params.eps_rel,
            use_filter_line_search=bool(ocp_params.use_filter_line_search),
        )
        self._hip = backend.HipOcp(self._table, self._packed, batch=1, device=device)
        self._ref_tile = self._packed.new_ref_tile(1)
        self._frames = self._packed.default_frames(1)
        self._last_status = None

    # -- the reference's properties ------------------------------------------
    @property
    def n_controls(self) -> int:
        return self._ocp_params.n_controls

    @property
    def dt(self) -> float:
        return self._ocp_params.dt

    @property
    def problem(self):
        """The reference exposes its crocoddyl.ShootingProblem here; the HIP path has no such
        object.  The packed problem description is returned for introspection."""
        return self._packed

    @abc.abstractmethod
    def create_running_model_list(self) -> list[_abi.RowSpec]: ...

    @abc.abstractmethod
    def create_terminal_model(self) -> list[_abi.RowSpec]: ...

    def create_constraint_lists(self):
        """(running, terminal) lists of _abi.ConstraintSpec; none by default."""
        return [], []

    def set_reference_weighted_trajectory(self, reference_weighted_trajectory=None):
        pass

    def update_geometry_placement(self, geometry_name: str, placement, geometry_type=None):
        """Updates placement of the obstacles (reference ocp_base_croco.py:110-132): `placement` is an
        SE3 (or 12 doubles R|p) in the geometry's parent joint frame, world for environment objects."""
        cmodel = self._robot_models.collision_model
        if cmodel is None or not cmodel.existGeometryName(geometry_name):
            raise RuntimeError(f"Unknown geometry name '{geometry_name}' in collision model!")
        from .se3 import as_se3_12

        self._hip.set_geom_placement(cmodel.getGeometryId(geometry_name), as_se3_12(placement))

    def fill_debug_data(self, res: bool, ocp_results: OCPResults) -> None:
        st = self._last_status
        self._debug_data.problem_solved = bool(res)
        self._debug_data.result = ocp_results
        self._debug_data.kkt_norm = float(st["kkt"][0])
        self._debug_data.nb_iter = int(st["iter"][0])
        self._debug_data.nb_qp_iter = int(st["qp_iters"][0])

    def solve(self, x0: npt.NDArray[np.float64], x_warmstart: list, u_warmstart: list,
              use_iteration_limits_and_timeout: bool = True) -> None:  # fmt: skip
        T = self.n_controls
        assert len(x_warmstart) == T + 1 and len(u_warmstart) == T
        max_iters = self._ocp_params.solver_iters if use_iteration_limits_and_timeout else 1000
        max_time = 0.0
        if self._ocp_params.max_solve_time is not None and use_iteration_limits_and_timeout:
            max_time = float(self._ocp_params.max_solve_time)
        xs_ws = np.asarray(x_warmstart, dtype=np.float64).reshape(1, T + 1, self._nx)
        us_ws = np.asarray(u_warmstart, dtype=np.float64).reshape(1, T, self._nv)
        xs, us, K, st = self._hip.solve(np.asarray(x0, dtype=np.float64).reshape(1, self._nx), xs_ws, us_ws, max_iters, max_time)
        self._last_status = st
        # lists of writable per-node arrays: the warm start mutates them in place
        # (warm_start_shift_previous_solution.py:95-104)
        ocp_results = OCPResults(
            states=[xs[0, t].copy() for t in range(T + 1)],
            ricatti_gains=[K[0, t].copy() for t in range(T)],
            feed_forward_terms=[us[0, t].copy() for t in range(T)],
        )
        if self._ocp_params.use_debug_data:
            self.fill_debug_data(res=bool(st["solved"][0]), ocp_results=ocp_results)
        self._ocp_results = ocp_results

    def integrate(self, state, control):
        return self._hip.integrate(np.asarray(state, dtype=np.float64), np.asarray(control, dtype=np.float64))[0]

    @property
    def ocp_results(self) -> OCPResults:
        return self._ocp_results

    @ocp_results.setter
    def ocp_results(self, value: OCPResults) -> None:
        self._ocp_results = value

    @property
    def debug_data(self) -> OCPDebugData:
        return self._debug_data

    @debug_data.setter
    def debug_data(self, value: OCPDebugData) -> None:
        self._debug_data = value
