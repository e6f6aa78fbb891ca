"""Quintic 0 -> 1 ramp with zero velocity/acceleration at both ends
(agimus_controller/agimus_controller/trajectories/quintic_trajectory.py:6-42)."""

from __future__ import annotations

import numpy as np


class QuinticTrajectory:
    def __init__(self, scale_duration):
        self.scale_duration = np.asarray(scale_duration, dtype=float).reshape(-1)
        n = self.scale_duration.size
        self.p, self.v, self.a = np.zeros(n), np.zeros(n), np.zeros(n)

    def get_value_at_t(self, t: float):
        d = self.scale_duration
        s = np.clip(t / d, 0.0, 1.0)
        inside = (t > 0) & (t < d)
        self.p[:] = 10 * s**3 - 15 * s**4 + 6 * s**5
        self.v[:] = np.where(inside, (30 * s**2 - 60 * s**3 + 30 * s**4) / d, 0.0)
        self.a[:] = np.where(inside, (60 * s - 180 * s**2 + 120 * s**3) / d**2, 0.0)
        return self.p, self.v, self.a
