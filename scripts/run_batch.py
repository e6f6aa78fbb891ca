"""Development driver: the default bench workload (sine references, goal-reaching costs) at a given batch, a few MPC
steps, per-step host wall time.  For rocprofv3 timelines of small batches:  rocprofv3 --kernel-trace -d out -- python3
scripts/run_batch.py --batch 1 --steps 30"""
import argparse, pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from agimus_controller_amd import _abi, backend, workloads  # noqa: E402
from agimus_controller_amd.factory import robot_tables as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--horizon", type=int, default=100)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--max-iter", type=int, default=10)
a = ap.parse_args()
T, B, dt = a.horizon, a.batch, 0.01
table = rt.panda_table(0.1)
tcp = table.frame_id("panda_hand_tcp")
running, terminal = workloads.goal_reaching_rows(tcp)
po = _abi.PackedOcp(7, [dt] * T, running, terminal, termination_tolerance=1e-3, max_qp_iters=100)
h = backend.HipOcp(table, po, B)
p = workloads.sine_batch_params(B, lower=table.lower_position_limit, upper=table.upper_position_limit)
w = workloads.SINE_WEIGHTS
h.sine_trajectory(a.warmup + a.steps + T + 2, dt, *p, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
for k in range(a.warmup):
    h.mpc_step(k, a.max_iter, first=(k == 0))
    h.download_first(copy=False)
h.sync()
lat, its = [], []
for k in range(a.warmup, a.warmup + a.steps):
    t1 = time.perf_counter()
    h.mpc_step(k, a.max_iter, first=False)
    st = h.download_first(copy=False)[3]
    lat.append((time.perf_counter() - t1) * 1e3)
    its.append(float(st["iter"].mean()))
lat = np.sort(np.array(lat))
print(f"batch {B} T {T}: median {np.median(lat):.4f} ms  mean {lat.mean():.4f}  p99 {lat[int(0.99 * (len(lat) - 1))]:.4f}  "
      f"steps/s {B * 1e3 / lat.mean():.1f}  mean iters {np.mean(its):.2f}")
h.close()
