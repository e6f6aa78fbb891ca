"""Sine wave of the end effector in cartesian space (trajectories/sine_wave_cartesian_space.py:15-152
upstream): the desired pose is the initial pose translated by A s(t) sin(w t), joint positions come
from an iterative inverse kinematics (damped by nothing: J^T (J J^T)^-1, LOCAL frame error), joint
velocities from the LOCAL_WORLD_ALIGNED Jacobian, feed-forward effort from RNEA with zero
acceleration.  Forward kinematics, Jacobians and RNEA are evaluated on the device."""

from __future__ import annotations

from copy import deepcopy

import numpy as np

from ..se3 import SE3, SE3ToXYZQUAT, log6
from ..trajectory import TrajectoryPoint, TrajectoryPointWeights, WeightedTrajectoryPoint
from .quintic_trajectory import QuinticTrajectory
from .sine_wave_params import SinWaveParams
from .trajectory_base import TrajectoryBase


class SinusWaveCartesianSpace(TrajectoryBase):
    """Define the trajectory of a sine-wave in cartesian space."""

    def __init__(self, sine_wave_params: SinWaveParams, ee_frame_name, w_q, w_qdot, w_qddot, w_robot_effort, w_pose,
                 mask=(True, True, True, True, True, True)):  # fmt: skip
        super().__init__(ee_frame_name)
        self.quint_traj = QuinticTrajectory(scale_duration=sine_wave_params.scale_duration)
        self.amp = np.array(sine_wave_params.amplitude)
        self.w = np.array(sine_wave_params.pulsation)
        self.w_q, self.w_qdot, self.w_qddot = w_q, w_qdot, w_qddot
        self.w_robot_effort, self.w_pose = w_robot_effort, w_pose
        self.mask = np.asarray(mask, dtype=bool)  # inverse kinematics DoF mask [x, y, z, roll, pitch, yaw]
        self.ik_q = None

    def initialize(self, pin_model, q0, dynamics=None) -> None:
        super().initialize(pin_model, q0, dynamics)
        self.ik_q = self.q0.copy()
        self.ee_init_pos = self.get_end_effector_pose_from_q_as_se3(self.q0)

    def inverse_kinematics(self, ee_des_pos: SE3, ee_des_vel: np.ndarray, precision=1e-5, it_max=10000):
        """Compute the inverse kinematics of the robot to reach the desired end effector pose."""
        i = 0
        success = False
        while True:
            self.ik_ee_pose = self.get_end_effector_pose_from_q_as_se3(self.ik_q)
            dMi = ee_des_pos.actInv(self.ik_ee_pose)
            error = log6(dMi).vector[self.mask]
            if np.linalg.norm(error) < precision:
                success = True
                break
            if i > it_max:
                break
            Jee = self._dyn.frame_jacobian(self.ee_frame_id, self.ik_q, local=True)[0][self.mask, :]
            dq = -Jee.T @ np.linalg.solve(Jee @ Jee.T, error)
            self.ik_q[:] = self.ik_q + dq  # revolute joints: pin.integrate is the plain sum
            i += 1
        if not success:
            raise RuntimeError(f"Inverse kinematics 6D failed to converge with error: {error}. Number of iteration: {i}")
        Jee = self._dyn.frame_jacobian(self.ee_frame_id, self.ik_q, local=False)[0][self.mask, :]
        dq = Jee.T @ np.linalg.solve(Jee @ Jee.T, ee_des_vel[self.mask])
        return self.ik_q.copy(), dq.copy()

    def get_traj_point_at_t(self, t) -> WeightedTrajectoryPoint:
        quint, dquint, _ = self.quint_traj.get_value_at_t(t)
        sin_wt, cos_wt = np.sin(self.w * t), np.cos(self.w * t)
        ee_des_pos = self.ee_init_pos.copy()
        ee_des_vel = np.zeros(6)
        ee_des_pos.translation = ee_des_pos.translation + self.amp * quint * sin_wt
        ee_des_vel[:3] = self.amp * (dquint * sin_wt + quint * self.w * cos_wt)
        q, dq = self.inverse_kinematics(ee_des_pos, ee_des_vel)
        ddq = np.zeros(self.pin_model.nv)
        u = self._dyn.rnea(q, dq, ddq)[0]
        traj_point = TrajectoryPoint(time_ns=t, robot_configuration=q, robot_velocity=dq, robot_acceleration=ddq, robot_effort=u,
                                     end_effector_poses={self.ee_frame_name: SE3ToXYZQUAT(ee_des_pos)})  # fmt: skip
        traj_weights = TrajectoryPointWeights(w_robot_configuration=self.w_q, w_robot_velocity=self.w_qdot,
                                              w_robot_acceleration=self.w_qddot, w_robot_effort=self.w_robot_effort,
                                              w_end_effector_poses={self.ee_frame_name: self.w_pose})  # fmt: skip
        return WeightedTrajectoryPoint(point=deepcopy(traj_point), weights=deepcopy(traj_weights))
