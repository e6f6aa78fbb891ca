"""Result / debug containers with the reference's field names
(agimus_controller/agimus_controller/mpc_data.py:7-42)."""

from __future__ import annotations

import dataclasses
import typing as T

import numpy as np
import numpy.typing as npt

Array = npt.NDArray[np.float64]


@dataclasses.dataclass
class OCPResults:
    states: list[Array] = dataclasses.field(default_factory=list)
    ricatti_gains: list[Array] = dataclasses.field(default_factory=list)
    feed_forward_terms: list[Array] = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class OCPDebugData:
    result: OCPResults = dataclasses.field(default_factory=OCPResults)
    references: list[T.Tuple[str, Array]] = dataclasses.field(default_factory=list)
    residuals: list[T.Tuple[str, T.List[Array]]] = dataclasses.field(default_factory=list)
    kkt_norm: np.float64 = 0.0
    nb_iter: np.int64 = 0
    nb_qp_iter: np.int64 = 0
    problem_solved: bool = False


@dataclasses.dataclass
class MPCDebugData:
    ocp: OCPDebugData = dataclasses.field(default_factory=OCPDebugData)
    reference_id: int = -1
    duration_iteration_ns: int = 0
    duration_horizon_update_ns: int = 0
    duration_generate_warm_start_ns: int = 0
    duration_ocp_solve_ns: int = 0
