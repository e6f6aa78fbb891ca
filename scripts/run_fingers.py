"""Development: MPC step time of a Panda with its two finger joints unlocked (nv = 9: runs at the 16-joint capacity) on the
default bench workload shape (sine references, goal-reaching costs)."""
import argparse, pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from agimus_controller_amd import _abi, backend, workloads  # noqa: E402
from test_model_sizes import _model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--horizon", type=int, default=50)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
table = _model(9, "panda_fingers")
tcp = table.frame_id("panda_hand_tcp")
B, T, dt = a.batch, a.horizon, 0.01
running, terminal = workloads.goal_reaching_rows(tcp)
po = _abi.PackedOcp(9, [dt] * T, running, terminal)
h = backend.HipOcp(table, po, B)
q0, amp, puls, scale, t0 = workloads.sine_batch_params(B, nv=9, seed0=3, q0=np.zeros(9), lower=table.lower_position_limit, upper=table.upper_position_limit)
w = workloads.SINE_WEIGHTS
h.sine_trajectory(a.steps + 5 + T + 2, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
for k in range(5):
    h.mpc_step(k, 10, first=(k == 0))
    h.download_first(copy=False)
h.sync()
lat = []
for k in range(5, 5 + a.steps):
    t1 = time.perf_counter()
    h.mpc_step(k, 10, first=False)
    st = h.download_first(copy=False)[3]
    lat.append((time.perf_counter() - t1) * 1e3)
print(f"nv 9 batch {B} T {T}: median {np.median(lat):.3f} ms per step, {B * 1e3 / np.mean(lat):.0f} steps/s, mean iters {st['iter'].mean():.2f}")
h.close()
