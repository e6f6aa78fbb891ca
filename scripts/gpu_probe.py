"""Development probe: HIP path vs oracle on a few problems (run through gpurun)."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from agimus_controller_amd import _abi, backend, workloads
from agimus_controller_amd.factory import robot_tables as rt
from oracle.oracle import Oracle

def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))

print("devices", backend.device_count())
# 1. rigid body pieces
tab = rt.panda_table()
po, ref, x0, xs, us = workloads.random_goal_problem(tab, 12, 0.01, 5, 3, frame=tab.frame_id("panda_hand_tcp"))
h = backend.HipOcp(tab, po, 5)
o = Oracle(tab, po, 5)
rng = np.random.default_rng(0)
q, v, a = rng.uniform(-1, 1, (3, 16, 7))
print("rnea", rel(h.rnea(q, v, a), o.rnea(q, v, a).reshape(16, 7)))
print("frame", rel(h.frame_placement(tab.frame_id("panda_hand_tcp"), q), o.frame_placement(tab.frame_id("panda_hand_tcp"), q)))
xx = np.concatenate([q, v], 1)
print("integrate", rel(h.integrate(xx, a * 5), o.integrate(xx, a * 5).reshape(16, 14)))
# 2. derivative tiles
h.set_refs(ref); h.upload_x0(x0); h.upload_warmstart(xs, us)
t_h = h.calc_diff()
t_o = o.calc_diff(ref, None, xs, us)
sl = _abi.tile_slices(7)
for name, s in sl.items():
    print("tile", name, rel(t_h[..., s], t_o[..., s]), float(np.abs(t_o[..., s]).max()))
# 3. direction
xs[:, 0] = x0
h.upload_warmstart(xs, us)
K, k, dx, du, kkt = h.direction()
Ko, ko, dxo, duo, kkto = o.direction(o.calc_diff(ref, None, xs, us))
print("dir dx", rel(dx, dxo), "du", rel(du, duo), "K", rel(K, Ko), "kkt", kkt, kkto)
# 4. full solve
xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 20)
xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 20)
print("solve xs", rel(xs_h, xs_o), "us", rel(us_h, us_o), "K", rel(K_h, K_o))
print(st_h); print(st_o)
# 5. golden
table, po, ref, x0, xs0, us0 = workloads.golden_problem()
g = np.load(pathlib.Path(__file__).resolve().parents[1] / "tests/golden/simple_ocp_croco_results.npz")
hg = backend.HipOcp(table, po, 1)
hg.set_refs(ref)
xs_h, us_h, K_h, st_h = hg.solve(x0, xs0, us0, 100)
print("golden xs", np.abs(xs_h[0] - g["states"]).max(), "us", np.abs(us_h[0] - g["feed_forward_terms"]).max(), "K", np.abs(K_h[0] - g["ricatti_gains"]).max(), st_h)
# 6. timing at scale
B, T = 1024, 100
po, ref, x0, xs, us = workloads.random_goal_problem(tab, T, 0.01, 4, 5, frame=tab.frame_id("panda_hand_tcp"))
reps = B // 4
hb = backend.HipOcp(tab, po, B)
hb.set_refs(np.tile(ref, (reps, 1, 1))); hb.upload_x0(np.tile(x0, (reps, 1))); hb.upload_warmstart(np.tile(xs, (reps, 1, 1)), np.tile(us, (reps, 1, 1)))
for which, name in ((0, "calc_diff"), (1, "direction"), (2, "linesearch")):
    ms = hb.time_kernel(which, 5)
    print(name, "ms", ms, "us/node", ms * 1e3 / (B * (T + 1)))
t0 = time.time(); hb.solve_resident(10); hb.sync(); t1 = time.time()
xs_b, us_b, K_b, st_b = hb.download()
print("solve B=1024 T=100 wall", t1 - t0, "iters", np.bincount(st_b["iter"]), "solved", st_b["solved"].sum())
