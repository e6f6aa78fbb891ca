"""Sine wave in configuration space: q(t) = q0 + A s(t) sin(w t) with a quintic ramp s
(trajectories/sine_wave_configuration_space.py:15-72 upstream); feed-forward effort by RNEA and
end-effector pose by forward kinematics, both evaluated on the device."""

from __future__ import annotations

import numpy as np

from ..trajectory import TrajectoryPoint, TrajectoryPointWeights, WeightedTrajectoryPoint
from .quintic_trajectory import QuinticTrajectory
from .sine_wave_params import SinWaveParams
from .trajectory_base import TrajectoryBase


class SinusWaveConfigurationSpace(TrajectoryBase):
    def __init__(self, sine_wave_params: SinWaveParams, ee_frame_name: str, w_q, w_qdot, w_qddot, w_robot_effort, w_pose):
        super().__init__(ee_frame_name)
        self.quint_traj = QuinticTrajectory(scale_duration=sine_wave_params.scale_duration)
        self.amp = np.array(sine_wave_params.amplitude)
        self.w = np.array(sine_wave_params.pulsation)
        self.w_q, self.w_qdot, self.w_qddot = w_q, w_qdot, w_qddot
        self.w_robot_effort, self.w_pose = w_robot_effort, w_pose

    def get_traj_point_at_t(self, t) -> WeightedTrajectoryPoint:
        ramp, dramp, ddramp = self.quint_traj.get_value_at_t(t)
        w = self.w
        s, c = np.sin(w * t), np.cos(w * t)
        self.q = self.q0 + self.amp * ramp * s
        self.dq = self.amp * (dramp * s + ramp * w * c)
        self.ddq = self.amp * (ddramp * s + 2 * dramp * w * c - ramp * w * w * s)
        ee_pose = self.get_end_effector_pose_from_q(self.q)
        u = self._dyn.rnea(self.q, self.dq, self.ddq)[0]
        point = TrajectoryPoint(time_ns=t, robot_configuration=self.q, robot_velocity=self.dq, robot_acceleration=self.ddq,
                                robot_effort=u, end_effector_poses={self.ee_frame_name: ee_pose})  # fmt: skip
        weights = TrajectoryPointWeights(w_robot_configuration=self.w_q, w_robot_velocity=self.w_qdot,
                                         w_robot_acceleration=self.w_qddot, w_robot_effort=self.w_robot_effort,
                                         w_end_effector_poses={self.ee_frame_name: self.w_pose})  # fmt: skip
        return WeightedTrajectoryPoint(point=point, weights=weights)
