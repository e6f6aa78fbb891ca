"""Development: per-kernel device times of the large-model path (30-DoF humanoid, B=512, T=50)."""
import sys, pathlib, os
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from agimus_controller_amd import backend, workloads
from agimus_controller_amd.factory import robot_tables as rt
B = int(os.environ.get("B", 512)); T = int(os.environ.get("T", 50))
tab = rt.humanoid30_table()
po, ref, x0, xs, us = workloads.random_goal_problem(tab, T, 0.01, 4, 5, frame=len(tab.frame_names) - 1)
reps = B // 4
hb = backend.HipOcp(tab, po, B)
hb.set_refs(np.tile(ref, (reps, 1, 1))); hb.upload_x0(np.tile(x0, (reps, 1))); hb.upload_warmstart(np.tile(xs, (reps, 1, 1)), np.tile(us, (reps, 1, 1)))
out = []
for which, name in ((3, "calc_qp"), (1, "riccati"), (5, "ric_bwd"), (6, "gains"), (2, "step")):
    ms = min(hb.time_kernel(which, 3) for _ in range(2))
    out.append(f"{name} {ms:.2f}ms")
print("B", B, "T", T, " | ".join(out))
