"""Development: status words of the instances of the cartesian workload that do not converge within max_iter."""
import pathlib, sys
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from agimus_controller_amd import backend, workloads  # noqa: E402

B, T, dt = 256, 200, 0.01
table, tcp, po = bench.make_problem(T, "cartesian")
hip = backend.HipOcp(table, po, B)
w = dict(workloads.SINE_WEIGHTS)
cq0, camp, cpuls = workloads.cartesian_sine_batch_params(B, seed0=1234, lower=table.lower_position_limit, upper=table.upper_position_limit)
hip.cartesian_sine_trajectory(30 + T + 2, dt, cq0, camp, cpuls, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
np.set_printoptions(linewidth=200, precision=4)
for k in range(25):
    hip.mpc_step(k, 10, first=(k == 0))
    st = hip.download_first(copy=True)[3]
    bad = np.nonzero(st["iter"] >= 3)[0]
    print("step", k, "iters>=3:", [(int(b), int(st["iter"][b]), int(st["solved"][b]), int(st["flags"][b]), int(st["qp_iters"][b]), float(st["kkt"][b])) for b in bad][:8],
          "names", [n for n in st.dtype.names] if k == 0 else "")
