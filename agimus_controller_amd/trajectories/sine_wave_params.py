"""Amplitude / period / ramp duration of a sine-wave trajectory
(agimus_controller/agimus_controller/trajectories/sine_wave_params.py:4-41)."""

from __future__ import annotations

import numpy as np


class SinWaveParams:
    def __init__(self, amplitude, period, scale_duration):
        self.amplitude = amplitude
        self.period = period
        self.scale_duration = scale_duration

    @property
    def frequency(self):
        p = np.asarray(self.period, dtype=float)
        with np.errstate(divide="ignore"):
            f = np.where(np.abs(p) < 1e-6, 0.0, 1.0 / np.where(np.abs(p) < 1e-6, 1.0, p))
        return f.tolist()

    @property
    def pulsation(self):
        return (2.0 * np.pi * np.asarray(self.frequency)).tolist()
