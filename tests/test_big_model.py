"""30-DoF tree (BASELINE.json configs[4], synthetic humanoid of factory/robot_tables.py): the large-model
kernels (agx_big.hpp: LDS Riccati sweep, scratch-array derivative pass) against the CPU checker."""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="module")
def humanoid():
    t = rt.humanoid30_table()
    assert t.nv == 30 and not np.array_equal(t.parent, np.arange(30) - 1)  # a real tree
    return t


def test_primitives_30dof(hip_backend, humanoid):
    frame = len(humanoid.frame_names) - 1
    po, *_ = workloads.random_goal_problem(humanoid, 3, 0.01, 2, seed=1, frame=frame)
    h, o = hip_backend.HipOcp(humanoid, po, 2), Oracle(humanoid, po, 2)
    rng = np.random.default_rng(0)
    q, v, a = rng.uniform(-1.0, 1.0, (3, 9, 30))
    assert rel(h.rnea(q, v, a), o.rnea(q, v, a).reshape(9, 30)) < 1e-12
    assert rel(h.frame_placement(frame, q), o.frame_placement(frame, q)) < 1e-12
    x = np.concatenate([q, v], axis=1)
    assert rel(h.integrate(x, 3 * a), o.integrate(x, 3 * a).reshape(9, 60)) < 1e-10
    h.close()


def test_derivative_tiles_30dof(hip_backend, humanoid):
    frame = len(humanoid.frame_names) - 1
    B, T = 2, 3
    po, ref, x0, xs, us = workloads.random_goal_problem(humanoid, T, 0.01, B, seed=3, frame=frame)
    h, o = hip_backend.HipOcp(humanoid, po, B), Oracle(humanoid, po, B)
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    got, want = h.calc_diff(), o.calc_diff(ref, None, xs, us)
    for field, s in _abi.tile_slices(30).items():
        scale = max(np.abs(want[..., s]).max(), 1e-300)
        assert np.abs(got[..., s] - want[..., s]).max() <= 1e-10 * scale + 1e-13, field
    h.close()


def test_direction_30dof(hip_backend, humanoid):
    """QP tiles (scratch-array K1) + LDS Riccati sweep + KKT shares + exit gains at a fixed point."""
    frame = len(humanoid.frame_names) - 1
    B, T = 3, 6
    po, ref, x0, xs, us = workloads.random_goal_problem(humanoid, T, 0.01, B, seed=6, frame=frame)
    h, o = hip_backend.HipOcp(humanoid, po, B), Oracle(humanoid, po, B)
    xs[:, 0] = x0
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    K, k, dx, du, kkt = h.direction()
    Ko, ko, dxo, duo, kkto = o.direction(o.calc_diff(ref, None, xs, us))
    assert rel(dx, dxo) < 1e-8 and rel(du, duo) < 1e-8
    assert rel(K, Ko) < 1e-7
    np.testing.assert_allclose(kkt, kkto, rtol=1e-6)
    h.close()


def test_full_solve_30dof(hip_backend, humanoid):
    frame = len(humanoid.frame_names) - 1
    B, T = 3, 8
    po, ref, x0, xs, us = workloads.random_goal_problem(humanoid, T, 0.01, B, seed=9, frame=frame)
    h, o = hip_backend.HipOcp(humanoid, po, B), Oracle(humanoid, po, B)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 8)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 8, nthreads=4)
    np.testing.assert_array_equal(st_h["iter"], st_o["iter"])
    np.testing.assert_array_equal(st_h["solved"], st_o["solved"])
    assert rel(xs_h, xs_o) < 1e-8 and rel(us_h, us_o) < 1e-7
    assert rel(K_h, K_o) < 1e-6
    np.testing.assert_allclose(st_h["kkt"], st_o["kkt"], rtol=1e-5, atol=1e-12)
    h.close()


def test_gains_gemm_on_matrix_cores_equals_scalar_path(hip_backend, humanoid, monkeypatch):
    """K = M Kw - taux through v_mfma_f64_16x16x4_f64 (default for nv > 16) against the scalar kernel."""
    frame = len(humanoid.frame_names) - 1
    B, T = 2, 5
    po, ref, x0, xs, us = workloads.random_goal_problem(humanoid, T, 0.01, B, seed=21, frame=frame)
    xs[:, 0] = x0
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("AGX_GAINS_MFMA", flag)
        h = hip_backend.HipOcp(humanoid, po, B)
        h.set_refs(ref)
        h.upload_warmstart(xs, us)
        out[flag] = h.direction()[0]
        h.close()
    assert np.abs(out["1"]).max() > 1e-3
    assert rel(out["1"], out["0"]) < 1e-13
