"""Base of the reference generators (trajectories/trajectory_base.py:9-56 upstream).  Forward
kinematics and inverse dynamics come from a device problem (`backend.HipOcp`-like object with
`frame_placement` and `rnea`) instead of a pinocchio model/data pair."""

from __future__ import annotations

import abc

import numpy as np

from ..se3 import SE3, SE3ToXYZQUAT
from ..trajectory import WeightedTrajectoryPoint


class TrajectoryBase(abc.ABC):
    def __init__(self, ee_frame_name) -> None:
        self.ee_frame_name = ee_frame_name
        self.trajectory_is_done = False
        self.ee_frame_id = None
        self.pin_model = None
        self.q0 = self.q = self.dq = self.ddq = None
        self.is_initialized = False
        self._dyn = None

    def initialize(self, pin_model, q0, dynamics=None) -> None:
        """`pin_model`: the TableModel of RobotModels; `dynamics`: object with rnea / frame_placement
        (an OCP of this package or a backend.HipOcp)."""
        self.pin_model = pin_model
        assert pin_model.existFrame(self.ee_frame_name), "Frame does not exist."
        self.ee_frame_id = pin_model.getFrameId(self.ee_frame_name)
        self._dyn = getattr(dynamics, "_hip", dynamics)
        assert self._dyn is not None, "TrajectoryBase.initialize needs a dynamics provider (OCP or HipOcp)"
        self.q0 = np.asarray(q0, dtype=float)
        self.q = self.q0.copy()
        self.dq = np.zeros(pin_model.nv)
        self.ddq = np.zeros(pin_model.nv)
        self.is_initialized = True

    def get_end_effector_pose_from_q_as_se3(self, q) -> SE3:
        m = self._dyn.frame_placement(self.ee_frame_id, q)[0]
        return SE3(m[:9].reshape(3, 3), m[9:])

    def get_end_effector_pose_from_q(self, q):
        return SE3ToXYZQUAT(self.get_end_effector_pose_from_q_as_se3(q))

    @abc.abstractmethod
    def get_traj_point_at_t(self, t) -> WeightedTrajectoryPoint: ...
