"""Trajectory points, weights and the horizon buffer -- same public names and semantics as
agimus_controller/agimus_controller/trajectory.py:10-279 (field names, `robot_state`,
`TrajectoryBuffer.horizon` with non-uniform indexes, `interpolate_weights`)."""

from __future__ import annotations

import copy
import dataclasses

import numpy as np
import numpy.typing as npt

from .ocp_param_base import DTFactorsNSeq

Array = npt.NDArray[np.float64]


def _same(a, b) -> bool:
    """Equality that understands None, numpy arrays and dicts of either."""
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, dict) or isinstance(b, dict):
        if not (isinstance(a, dict) and isinstance(b, dict)) or a.keys() != b.keys():
            return False
        return all(_same(a[k], b[k]) for k in a)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.array_equal(a, b)
    return bool(a == b)


@dataclasses.dataclass(eq=False)
class TrajectoryPoint:
    """One sample of the reference the MPC tracks."""

    id: int | None = None
    time_ns: int | None = None
    robot_configuration: Array | None = None
    robot_velocity: Array | None = None
    robot_acceleration: Array | None = None
    robot_effort: Array | None = None
    forces: dict | None = None
    end_effector_poses: dict | None = None
    end_effector_velocities: dict | None = None

    @property
    def robot_state(self) -> Array:
        return np.concatenate((self.robot_configuration, self.robot_velocity))

    _COMPARED = ("time_ns", "robot_configuration", "robot_velocity", "robot_acceleration", "robot_effort", "forces",
                 "end_effector_poses", "end_effector_velocities")  # fmt: skip

    def __eq__(self, other):
        return isinstance(other, TrajectoryPoint) and all(_same(getattr(self, f), getattr(other, f)) for f in self._COMPARED)


@dataclasses.dataclass(eq=False)
class TrajectoryPointWeights:
    """Weights of the cost terms attached to one trajectory point."""

    w_robot_configuration: Array | None = None
    w_robot_velocity: Array | None = None
    w_robot_acceleration: Array | None = None
    w_robot_effort: Array | None = None
    w_forces: dict | None = None
    w_end_effector_poses: dict | None = None
    w_end_effector_velocities: dict | None = None
    w_collision_avoidance: np.float64 | None = None

    @property
    def w_robot_state(self) -> Array:
        return np.concatenate((self.w_robot_configuration, self.w_robot_velocity))

    def __eq__(self, other):
        return isinstance(other, TrajectoryPointWeights) and all(
            _same(getattr(self, f.name), getattr(other, f.name)) for f in dataclasses.fields(self)
        )


@dataclasses.dataclass(eq=False)
class WeightedTrajectoryPoint:
    point: TrajectoryPoint
    weights: TrajectoryPointWeights

    def __eq__(self, other):
        return isinstance(other, WeightedTrajectoryPoint) and self.point == other.point and self.weights == other.weights


class TrajectoryBuffer:
    """Growing list of weighted points; `horizon` picks the nodes of the OCP out of it."""

    def __init__(self, dt_factor_n_seq: DTFactorsNSeq):
        self._buffer: list = []
        self.dt_factor_n_seq = copy.deepcopy(dt_factor_n_seq)
        self.horizon_indexes = self.compute_horizon_indexes()

    def compute_horizon_indexes(self) -> list[int]:
        seq = self.dt_factor_n_seq
        steps = [f for f, n in zip(seq.factors, seq.n_steps) for _ in range(n)]
        indexes = [0] + [int(v) for v in np.cumsum(steps)]
        assert len(indexes) == sum(seq.n_steps) + 1
        assert all(a <= b for a, b in zip(indexes, indexes[1:])), "Time steps must be increasing"
        return indexes

    @property
    def horizon(self) -> list:
        assert self.horizon_indexes[-1] < len(self._buffer), "Size of buffer must be at least horizon_indexes[-1]."
        return [self._buffer[i] for i in self.horizon_indexes]

    def append(self, item):
        self._buffer.append(item)

    def extend(self, items):
        self._buffer.extend(items)

    def pop(self, index=-1):
        return self._buffer.pop(index)

    def clear_past(self):
        if self._buffer:
            del self._buffer[0]

    def __len__(self):
        return len(self._buffer)

    def __getitem__(self, index):
        return self._buffer[index]

    def __setitem__(self, index, value):
        self._buffer[index] = value


def interpolate_weights(p1: TrajectoryPointWeights, p2: TrajectoryPointWeights, alpha: float) -> TrajectoryPointWeights:
    """(1 - alpha) p1 + alpha p2 on every weight, alpha clipped to [0, 1]; a frame missing on
    one side is blended against zeros."""
    a = float(np.clip(alpha, 0.0, 1.0))

    def mix(u, v):
        return (1.0 - a) * u + a * v

    def mix_dicts(d1: dict, d2: dict) -> dict:
        out = {}
        for key in set(d1) | set(d2):
            u = d1.get(key)
            v = d2.get(key)
            if u is None:
                u = np.zeros_like(v)
            if v is None:
                v = np.zeros_like(u)
            out[key] = mix(u, v)
        return out

    values = {}
    for f in dataclasses.fields(TrajectoryPointWeights):
        u, v = getattr(p1, f.name), getattr(p2, f.name)
        if u is None and v is None:
            values[f.name] = None  # unset on both sides stays unset
        elif isinstance(u, dict) or isinstance(v, dict):
            values[f.name] = mix_dicts(u or {}, v or {})
        else:
            values[f.name] = mix(u, v)
    return TrajectoryPointWeights(**values)
