"""Development: in-situ time of the derivative kernel and the direction sweep of the default bench workload (B = 1024, T = 100)."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from agimus_controller_amd import _abi, backend, workloads
from agimus_controller_amd.factory import robot_tables as rt
T, B, dt = 100, 1024, 0.01
table = rt.panda_table(0.1)
tcp = table.frame_id("panda_hand_tcp")
running, terminal = workloads.goal_reaching_rows(tcp)
po = _abi.PackedOcp(7, [dt] * T, running, terminal, termination_tolerance=1e-3, max_qp_iters=100)
h = backend.HipOcp(table, po, B)
p = workloads.sine_batch_params(B, lower=table.lower_position_limit, upper=table.upper_position_limit)
w = workloads.SINE_WEIGHTS
h.sine_trajectory(T + 20, dt, *p, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
for k in range(3):
    h.mpc_step(k, 10, first=(k == 0))
print("K1 (running + terminal) %.4f ms   direction sweep %.4f ms" % (min(h.time_kernel(3, 5) for _ in range(3)), min(h.time_kernel(1, 5) for _ in range(3))))
