"""Compare the direction of the MFMA-layout sweep with the lane-grid sweep on the same tiles (GPU)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agimus_controller_amd import backend, workloads  # noqa: E402
from agimus_controller_amd.factory import robot_tables  # noqa: E402

T, B = int(sys.argv[1]), int(sys.argv[2])
panda = robot_tables.panda_table()
tcp = panda.frame_id("panda_hand_tcp")
po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=31, frame=tcp)
out = {}
for mx in ("0", "1"):
    os.environ["AGX_RICCATI_MX"] = mx
    h = backend.HipOcp(panda, po, B)
    h.set_refs(ref)
    h.upload_x0(x0)
    h.upload_warmstart(xs, us)
    out[mx] = h.direction()
    h.close()
for name, a, b in zip(("K", "k", "dx", "du", "kkt"), out["0"], out["1"]):
    d = np.abs(a - b)
    print(name, "max abs diff", d.max(), "scale", np.abs(a).max())
    if a.ndim >= 2 and d.max() > 1e-9 * max(1, np.abs(a).max()):
        per_node = d.reshape(d.shape[0], d.shape[1], -1).max(axis=2)[0]
        bad = np.nonzero(per_node > 1e-9 * np.abs(a).max())[0]
        print("   nodes off (instance 0):", bad[:10], "...", bad[-5:], "count", len(bad))
