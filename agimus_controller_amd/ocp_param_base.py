"""OCP parameters, field-for-field compatible with the reference's
agimus_controller/agimus_controller/ocp_param_base.py:6-85 (same names, defaults and derived
quantities) so that callers construct them unchanged."""

from __future__ import annotations

import dataclasses
import itertools
import typing as T


@dataclasses.dataclass
class DTFactorsNSeq:
    """Piecewise-constant time-step pattern: `n_steps[i]` nodes of length `factors[i] * dt`."""

    factors: list[int]
    n_steps: list[int]


@dataclasses.dataclass
class OCPParamsBaseCroco:
    dt: float
    solver_iters: int
    dt_factor_n_seq: DTFactorsNSeq
    _n_controls: int = dataclasses.field(init=False)
    horizon_size: int
    timesteps: tuple[float] = dataclasses.field(init=False)
    total_time: float = dataclasses.field(init=False)
    qp_iters: int = 200
    termination_tolerance: float = 1e-3
    max_solve_time: T.Optional[float] = None
    eps_abs: float = 1e-6
    eps_rel: float = 0.0
    callbacks: bool = False
    use_debug_data: bool = True
    use_filter_line_search = False
    n_threads: int = 1

    def __post_init__(self):
        seq = self.dt_factor_n_seq
        self._n_controls = int(sum(seq.n_steps))
        self.timesteps = tuple(
            itertools.chain.from_iterable(itertools.repeat(self.dt * f, n) for f, n in zip(seq.factors, seq.n_steps))
        )
        self.total_time = sum(self.timesteps)
        assert self.horizon_size == self._n_controls, (
            f"The horizon size {self.horizon_size} must be equal to the sum of the time steps {self._n_controls}."
        )

    @property
    def n_controls(self) -> int:
        return self._n_controls
