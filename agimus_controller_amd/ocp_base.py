"""Abstract OCP interface, the drop-in boundary of the reference
(agimus_controller/agimus_controller/ocp_base.py:11-107)."""

from __future__ import annotations

import abc
import warnings

import numpy as np
import numpy.typing as npt

from .mpc_data import OCPDebugData, OCPResults
from .trajectory import WeightedTrajectoryPoint


class OCPBase(abc.ABC):
    def __init__(self) -> None:
        pass

    @abc.abstractmethod
    def set_reference_weighted_trajectory(self, reference_weighted_trajectory: list[WeightedTrajectoryPoint]) -> None:
        """References and cost weights of every node (n_controls + 1 points)."""

    @property
    def horizon_size(self) -> int:
        warnings.warn("Use n_controls instead", DeprecationWarning)
        return self.n_controls

    @property
    @abc.abstractmethod
    def n_controls(self) -> int: ...

    @property
    @abc.abstractmethod
    def dt(self) -> float: ...

    @abc.abstractmethod
    def solve(self, x0: npt.NDArray[np.float64], x_warmstart: list, u_warmstart: list,
              use_iteration_limits_and_timeout: bool = True) -> None: ...  # fmt: skip

    @abc.abstractmethod
    def integrate(self, state: npt.NDArray[np.float64], control: npt.NDArray) -> npt.NDArray[np.float64]: ...

    @property
    @abc.abstractmethod
    def ocp_results(self) -> OCPResults: ...

    @ocp_results.setter
    def ocp_results(self, value: OCPResults) -> None: ...

    @property
    @abc.abstractmethod
    def debug_data(self) -> OCPDebugData: ...
