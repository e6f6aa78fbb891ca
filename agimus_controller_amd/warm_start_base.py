"""Warm-start interface of the reference (agimus_controller/agimus_controller/warm_start_base.py:24-92)."""

from __future__ import annotations

import abc
import typing as T

import numpy as np
import numpy.typing as npt

from .mpc_data import OCPResults
from .trajectory import TrajectoryPoint

WarmStart = tuple[npt.NDArray[np.float64], list[npt.NDArray[np.float64]], list[npt.NDArray[np.float64]]]


class WarmStartBase(abc.ABC):
    """Produces (x0, xs_init, us_init) for the next solve."""

    def __init__(self) -> None:
        super().__init__()
        self._previous_solution: T.Optional[OCPResults] = None

    @abc.abstractmethod
    def generate(self, initial_state: TrajectoryPoint, reference_trajectory: list[TrajectoryPoint]) -> WarmStart: ...

    @abc.abstractmethod
    def setup(self, *args, **kwargs) -> None: ...

    def update_previous_solution(self, previous_solution: OCPResults) -> None:
        self._previous_solution = previous_solution
