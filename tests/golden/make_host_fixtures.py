"""Golden vectors from the numpy-only modules of the reference that import here
(SURVEY.md 8c): ocp_param_base (timesteps / total_time / n_controls), quintic_trajectory,
sine_wave_params.  Run in the build container: python tests/golden/make_host_fixtures.py
Outputs tests/golden/host_fixtures.json (data only)."""
import json
import pathlib
import sys

import numpy as np

sys.path.insert(0, "/root/reference/agimus_controller")
from agimus_controller.ocp_param_base import DTFactorsNSeq, OCPParamsBaseCroco  # noqa: E402
from agimus_controller.trajectories.quintic_trajectory import QuinticTrajectory  # noqa: E402
from agimus_controller.trajectories.sine_wave_params import SinWaveParams  # noqa: E402

out = {"params": [], "quintic": [], "sine": []}
for dt, factors, n_steps in [(0.01, [1], [100]), (0.01, [1, 2, 4], [30, 20, 10]), (0.1, [1, 2], [2, 1]), (0.05, [2, 1], [2, 2])]:
    p = OCPParamsBaseCroco(dt=dt, solver_iters=10, horizon_size=sum(n_steps), dt_factor_n_seq=DTFactorsNSeq(factors=factors, n_steps=n_steps))
    out["params"].append({"dt": dt, "factors": factors, "n_steps": n_steps, "timesteps": list(p.timesteps),
                          "total_time": p.total_time, "n_controls": p.n_controls, "qp_iters": p.qp_iters,
                          "termination_tolerance": p.termination_tolerance, "eps_abs": p.eps_abs, "eps_rel": p.eps_rel,
                          "n_threads": p.n_threads, "use_filter_line_search": p.use_filter_line_search})
q = QuinticTrajectory([0.2, 0.5, 1.0])
for t in [-0.1, 0.0, 0.05, 0.1, 0.2, 0.35, 0.7, 1.0, 1.5]:
    p, v, a = q.get_value_at_t(t)
    out["quintic"].append({"t": t, "p": p.tolist(), "v": v.tolist(), "a": a.tolist()})
s = SinWaveParams(amplitude=[0.1, 0.2], period=[4.0, 0.0], scale_duration=[0.2, 0.2])
out["sine"].append({"period": [4.0, 0.0], "frequency": s.frequency, "pulsation": s.pulsation})
dst = pathlib.Path(__file__).resolve().parent / "host_fixtures.json"
dst.write_text(json.dumps(out, indent=1))
print("wrote", dst)
