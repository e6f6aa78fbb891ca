"""Trajectory fed by the user as q / dq / ddq arrays (trajectories/generic_trajectory.py:13-87
upstream): feed-forward effort by RNEA and end-effector pose by forward kinematics for every
sample -- evaluated on the device in one batch instead of one pinocchio call per sample."""

from __future__ import annotations

import numpy as np

from ..se3 import SE3
from ..trajectory import TrajectoryPoint, TrajectoryPointWeights, WeightedTrajectoryPoint
from .trajectory_base import TrajectoryBase


class GenericTrajectory(TrajectoryBase):
    def __init__(self, ee_frame_name, w_q, w_qdot, w_qddot, w_robot_effort, w_pose, w_collision_avoidance):
        super().__init__(ee_frame_name)
        self.trajectory = None
        self.traj_idx = 0
        self.w_q, self.w_qdot, self.w_qddot = w_q, w_qdot, w_qddot
        self.w_robot_effort, self.w_pose = w_robot_effort, w_pose
        self.robot_frame = self.ee_frame_name
        self.w_collision_avoidance = w_collision_avoidance

    def build_trajectory_from_q_dq_ddq_arrays(self, q_array, dq_array, ddq_array) -> list[TrajectoryPoint]:
        """Builds list of Trajectory points based on given trajectory of q, dq and ddq."""
        assert len(q_array) == len(dq_array) and len(q_array) == len(ddq_array)
        q = np.asarray(q_array, dtype=float)
        dq = np.asarray(dq_array, dtype=float)
        ddq = np.asarray(ddq_array, dtype=float)
        effort = self._dyn.rnea(q, dq, ddq)  # [n, nv]
        poses = self._dyn.frame_placement(self.ee_frame_id, q)  # [n, 12]
        trajectory = []
        for idx in range(q.shape[0]):
            ee_pose = SE3(poses[idx, :9].reshape(3, 3).copy(), poses[idx, 9:].copy())
            trajectory.append(
                TrajectoryPoint(robot_configuration=q[idx].copy(), robot_velocity=dq[idx].copy(), robot_acceleration=ddq[idx].copy(),
                                robot_effort=effort[idx].copy(), end_effector_poses={self.robot_frame: ee_pose}))  # fmt: skip
        return trajectory

    def add_trajectory(self, trajectory: list[TrajectoryPoint]) -> None:
        """Initialize the trajectory if it wasn't, otherwise extend the trajectory."""
        self.trajectory_is_done = False
        if self.trajectory is None:
            self.trajectory = list(trajectory)
        else:
            self.trajectory.extend(list(trajectory))

    def get_traj_point_at_t(self, t) -> WeightedTrajectoryPoint:
        traj_point = self.trajectory[self.traj_idx]
        self.trajectory_is_done = self.traj_idx == len(self.trajectory) - 1
        self.traj_idx = min(self.traj_idx + 1, len(self.trajectory) - 1)
        traj_weights = TrajectoryPointWeights(
            w_robot_configuration=self.w_q, w_robot_velocity=self.w_qdot, w_robot_acceleration=self.w_qddot,
            w_robot_effort=self.w_robot_effort, w_end_effector_poses={self.robot_frame: self.w_pose},
            w_collision_avoidance=self.w_collision_avoidance)  # fmt: skip
        return WeightedTrajectoryPoint(point=traj_point, weights=traj_weights)
