"""Development: phase stamps of k_riccati_mfma (library built with -DAGX_WG_PROFILE and the RSTAMP patch)."""
import ctypes as C, os, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from agimus_controller_amd import backend, workloads
from agimus_controller_amd.factory import robot_tables as rt
B, T = int(os.environ.get("B", 512)), int(os.environ.get("T", 50))
tab = rt.humanoid30_table()
po, ref, x0, xs, us = workloads.random_goal_problem(tab, T, 0.01, 4, 5, frame=len(tab.frame_names) - 1)
reps = B // 4
hb = backend.HipOcp(tab, po, B)
hb.set_refs(np.tile(ref, (reps, 1, 1))); hb.upload_x0(np.tile(x0, (reps, 1))); hb.upload_warmstart(np.tile(xs, (reps, 1, 1)), np.tile(us, (reps, 1, 1)))
lib = backend.lib()
ts = (C.c_longlong * 64)(); n = C.c_int(0)
hb.time_kernel(3, 1)
lib.agx_dev_wg_stamps(ts, C.byref(n))
print("riccati bwd %.3f ms" % hb.time_kernel(5, 1))
lib.agx_dev_wg_stamps(ts, C.byref(n))
t = np.array(ts[: n.value], dtype=np.int64)
print("n", n.value, "deltas (10 ns ticks):", np.diff(t).tolist())
