"""Development: per-kernel device times at bench scale."""
import sys, pathlib, os
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from agimus_controller_amd import backend, workloads
from agimus_controller_amd.factory import robot_tables as rt
B = int(os.environ.get("B", 1024)); T = 100
tab = rt.panda_table()
po, ref, x0, xs, us = workloads.random_goal_problem(tab, T, 0.01, 4, 5, frame=tab.frame_id("panda_hand_tcp"))
reps = B // 4
hb = backend.HipOcp(tab, po, B)
hb.set_refs(np.tile(ref, (reps, 1, 1))); hb.upload_x0(np.tile(x0, (reps, 1))); hb.upload_warmstart(np.tile(xs, (reps, 1, 1)), np.tile(us, (reps, 1, 1)))
out = []
for which, name in ((3, "calc_qp"), (0, "calc_qp+term"), (1, "riccati"), (5, "ric_bwd"), (6, "gains"), (7, "pair"), (2, "step")):
    ms = min(hb.time_kernel(which, 10) for _ in range(3))
    out.append(f"{name} {ms*1e3:.1f}us")
print(os.environ.get("AGX_LIB", "default"), "B", B, " | ".join(out))
