"""The oracle against the reference's own golden vectors and known answers (CPU).

Golden file: agimus_controller/tests/resources/simple_ocp_croco_results.pkl, checked upstream
by agimus_controller/tests/test_ocp_croco_base.py:175-204 to 6 decimals.  It was produced by
Crocoddyl + mim_solvers.SolverCSQP on example-robot-data's Panda; the Panda table of this
repository reproduces it, which pins the oracle to the real reference binaries."""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt
from oracle.oracle import Oracle


@pytest.mark.parametrize("analytic", [False, True])
def test_golden_cold_start_reproduced(golden, analytic):
    """analytic = True: the same with the kernels' analytical derivatives on the CPU (oracle/agx_analytic.cpp) -- the
    reference's golden file pins that derivation too, without a GPU."""
    table, po, ref, x0, xs0, us0 = workloads.golden_problem()
    o = Oracle(table, po, 1)
    if analytic:
        assert o.set_analytic(True)
    xs, us, K, st = o.solve(ref, None, x0, xs0, us0, 100)
    assert st["solved"][0] == 1 and st["kkt"][0] <= 1e-3
    # upstream asserts 6 decimals; the restatement is good to ~1e-10
    np.testing.assert_allclose(xs[0], golden["states"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(us[0], golden["feed_forward_terms"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(K[0], golden["ricatti_gains"], rtol=0, atol=1e-8)


def test_golden_point_is_stationary_and_gains_match(golden):
    table, po, ref, x0, _, _ = workloads.golden_problem()
    o = Oracle(table, po, 1)
    xs, us, K, st = o.solve(ref, None, x0, golden["states"][None], golden["feed_forward_terms"][None], 5)
    assert st["iter"][0] == 0 and st["solved"][0] == 1  # already a KKT point
    np.testing.assert_array_equal(xs[0], golden["states"])
    np.testing.assert_allclose(K[0], golden["ricatti_gains"], rtol=0, atol=1e-8)


def test_golden_dynamics_consistency(golden):
    """Model sanity probe of SURVEY App. B: (M + armature) dv/h + nle = u along the golden trajectory."""
    table, po, *_ = workloads.golden_problem()
    o = Oracle(table, po, 1)
    xs, us = golden["states"], golden["feed_forward_terms"]
    h = 1e-3
    for t in range(9):
        q, v = xs[t, :7], xs[t, 7:]
        a = (xs[t + 1, 7:] - v) / h
        np.testing.assert_allclose(xs[t + 1, :7], q + h * xs[t + 1, 7:], atol=1e-12)  # semi-implicit Euler
        tau = o.rnea(q, v, a) + 0.1 * a
        np.testing.assert_allclose(tau, us[t], atol=1e-6)


def test_state_and_control_residual_known_answers():
    """tests/test_ocp_croco_generic.py:48-72 and :93-113: r = x - xref, cost = 1/2 sum w (x - xref)^2."""
    table = rt.chain_table(6, seed=1)
    running = [_abi.RowSpec(_abi.RES_STATE), _abi.RowSpec(_abi.RES_CONTROL)]
    po = _abi.PackedOcp(6, [1.0], running, [_abi.RowSpec(_abi.RES_STATE)])
    o = Oracle(table, po, 1)
    rng = np.random.default_rng(0)
    x, u = rng.random(12), rng.random(6)
    ref = po.new_ref_tile(1)
    # default references: zeros, unit weights
    wi, r, aw = po.row_view(ref, False, 0)
    r[...] = 0.0
    wi2, r2, aw2 = po.row_view(ref, False, 1)
    r2[...] = 0.0
    _, cost, res = o.node_calc(False, 1.0, x, u, ref[0, 0])
    np.testing.assert_array_equal(res[:12], x)
    np.testing.assert_array_equal(res[12:18], u)
    assert cost == pytest.approx(np.sum(0.5 * x**2) + np.sum(0.5 * u**2), rel=1e-14)
    # after update(): new references and weights (0.5 on q, 10 on v; 0.5 on u)
    xref, uref = rng.random(12), rng.random(6)
    r[...] = xref
    aw[..., :6] = 0.5
    aw[..., 6:] = 10.0
    r2[...] = uref
    aw2[...] = 0.5
    _, cost, res = o.node_calc(False, 1.0, x, u, ref[0, 0])
    np.testing.assert_array_equal(res[:12], x - xref)
    np.testing.assert_array_equal(res[12:18], u - uref)
    w = np.concatenate([0.5 * np.ones(6), 10 * np.ones(6)])
    assert cost == pytest.approx(np.sum(0.5 * w * (x - xref) ** 2) + np.sum(0.25 * (u - uref) ** 2), rel=1e-14)


def test_pendulum_closed_form():
    table = rt.pendulum_table(length=0.7, mass=2.0)
    po = _abi.PackedOcp(1, [0.01], [_abi.RowSpec(_abi.RES_STATE)], [_abi.RowSpec(_abi.RES_STATE)])
    o = Oracle(table, po, 1)
    q, v, a = np.array([0.3]), np.array([1.2]), np.array([-0.7])
    np.testing.assert_allclose(o.rnea(q, v, a), 2 * 0.49 * a + 2 * 9.81 * 0.7 * np.sin(q), rtol=1e-14)
    np.testing.assert_allclose(o.mass_matrix(q), [[0.98]], rtol=1e-14)
    # semi-implicit Euler: v+ = v + h a, q+ = q + h v+
    u = np.array([0.4])
    acc = (u - 2 * 9.81 * 0.7 * np.sin(q)) / 0.98
    xn = o.integrate(np.concatenate([q, v]), u)
    np.testing.assert_allclose(xn, [q[0] + 0.01 * (v[0] + 0.01 * acc[0]), v[0] + 0.01 * acc[0]], rtol=1e-14)


@pytest.mark.parametrize("table", [rt.panda_table(), rt.chain_table(5, 3), rt.humanoid30_table()], ids=["panda", "chain5", "humanoid30"])
def test_dynamics_identities(table):
    nv = table.nv
    po = _abi.PackedOcp(nv, [0.01], [_abi.RowSpec(_abi.RES_STATE)], [_abi.RowSpec(_abi.RES_STATE)])
    o = Oracle(table, po, 1)
    rng = np.random.default_rng(5)
    q, v, a = rng.uniform(-1, 1, (3, nv))
    M = o.mass_matrix(q)
    np.testing.assert_allclose(M, M.T, atol=1e-12)
    assert np.linalg.eigvalsh(M).min() > 0
    nle = o.rnea(q, v, np.zeros(nv))
    np.testing.assert_allclose(o.rnea(q, v, a), (M - np.diag(table.armature)) @ a + nle, rtol=1e-10, atol=1e-10)
    # forward dynamics inverts (M + armature) a + nle = u
    u = M @ a + nle
    np.testing.assert_allclose(o.forward_dynamics(q, v, u), a, rtol=1e-9, atol=1e-9)
    # power balance: d/dt (1/2 v'(M-arm)v) = v'(tau - g) along a short explicit trajectory
    g = o.rnea(q, np.zeros(nv), np.zeros(nv))
    eps = 1e-6
    Mq = M - np.diag(table.armature)
    M2 = o.mass_matrix(q + eps * v) - np.diag(table.armature)
    dK = (0.5 * (v + eps * a) @ M2 @ (v + eps * a) - 0.5 * v @ Mq @ v) / eps
    np.testing.assert_allclose(dK, v @ (o.rnea(q, v, a) - g), rtol=2e-4, atol=1e-4)


def test_tiles_match_finite_differences(panda):
    tcp = panda.frame_id("panda_hand_tcp")
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, 3, 0.01, 1, seed=2, frame=tcp)
    o = Oracle(panda, po, 1)
    sl = _abi.tile_slices(7)
    x, u = xs[0, 1], us[0, 1]
    tile, xnext, _ = o.node_calc_diff(False, 0.01, x, u, ref[0, 1])
    Fx, Fu = tile[sl["Fx"]].reshape(14, 14), tile[sl["Fu"]].reshape(14, 7)
    Lx, Lu = tile[sl["Lx"]], tile[sl["Lu"]]
    eps = 1e-6
    for i in range(14):
        d = np.zeros(14)
        d[i] = eps
        xp, cp, _ = o.node_calc(False, 0.01, x + d, u, ref[0, 1])
        xm, cm, _ = o.node_calc(False, 0.01, x - d, u, ref[0, 1])
        np.testing.assert_allclose(Fx[:, i], (xp - xm) / (2 * eps), rtol=1e-5, atol=1e-7)
        assert Lx[i] == pytest.approx((cp - cm) / (2 * eps), rel=1e-5, abs=1e-7)
    for i in range(7):
        d = np.zeros(7)
        d[i] = eps
        xp, cp, _ = o.node_calc(False, 0.01, x, u + d, ref[0, 1])
        xm, cm, _ = o.node_calc(False, 0.01, x, u - d, ref[0, 1])
        np.testing.assert_allclose(Fu[:, i], (xp - xm) / (2 * eps), rtol=1e-5, atol=1e-9)
        assert Lu[i] == pytest.approx((cp - cm) / (2 * eps), rel=1e-5, abs=1e-9)


def test_direction_solves_the_qp(panda):
    """dx, du of the oracle satisfy the linearised dynamics and make the QP stationary."""
    tcp = panda.frame_id("panda_hand_tcp")
    T = 8
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.02, 1, seed=4, frame=tcp)
    o = Oracle(panda, po, 1)
    xs[:, 0] = x0
    tiles = o.calc_diff(ref, None, xs, us)
    K, k, dx, du, kkt = o.direction(tiles, preg=0.0, dreg=0.0)
    sl = _abi.tile_slices(7)
    lam_next = None
    for t in range(T, -1, -1):
        n = tiles[0, t]
        Lx, Lxx = n[sl["Lx"]], n[sl["Lxx"]].reshape(14, 14)
        if t == T:
            lam = Lx + Lxx @ dx[0, T]
        else:
            Fx, Fu, f = n[sl["Fx"]].reshape(14, 14), n[sl["Fu"]].reshape(14, 7), n[sl["f"]]
            Lu, Luu, Lxu = n[sl["Lu"]], n[sl["Luu"]].reshape(7, 7), n[sl["Lxu"]].reshape(14, 7)
            np.testing.assert_allclose(dx[0, t + 1], Fx @ dx[0, t] + Fu @ du[0, t] + f, atol=1e-10)
            np.testing.assert_allclose(Lu + Luu @ du[0, t] + Lxu.T @ dx[0, t] + Fu.T @ lam_next, 0, atol=1e-7)
            lam = Lx + Lxx @ dx[0, t] + Lxu @ du[0, t] + Fx.T @ lam_next
        lam_next = lam
    assert np.all(dx[0, 0] == 0)


def test_warm_start_shift_semantics():
    """tests/test_warm_start_shift_previous_reference.py:107-117 with timesteps (0.1, 0.1, 0.2):
    u_init[i-1] == controls[i] and x_init[i] == Euler(states[i], controls[i], dt0) for i = 1, 2."""
    table = rt.chain_table(3, seed=9)
    rows = [_abi.RowSpec(_abi.RES_STATE)]
    po = _abi.PackedOcp(3, [0.1, 0.1, 0.2], rows, rows)
    o = Oracle(table, po, 1)
    rng = np.random.default_rng(1)
    controls = rng.random((3, 3))
    states = np.zeros((4, 6))
    for i, h in enumerate([0.1, 0.1, 0.2]):
        pstep = _abi.PackedOcp(3, [h], rows, rows)
        states[i + 1] = Oracle(table, pstep, 1).integrate(states[i], controls[i])
    xs, us = o.shift_warmstart(states[None], controls[None])
    np.testing.assert_array_equal(xs[0, 0], states[1])
    np.testing.assert_array_equal(us[0, 0], controls[1])
    np.testing.assert_array_equal(xs[0, 1], states[2])
    np.testing.assert_array_equal(us[0, 1], controls[2])
    np.testing.assert_array_equal(xs[0, 2], Oracle(table, _abi.PackedOcp(3, [0.1], rows, rows), 1).integrate(states[2], controls[2]))
    np.testing.assert_array_equal(us[0, 2], controls[2])
    np.testing.assert_array_equal(xs[0, 3], states[3])


def test_analytic_cpu_leg_agrees_with_the_automatic_differentiation_checker():
    """oracle/agx_analytic.cpp (bench.py's "port-analytic" CPU baseline: the kernels' analytical RNEA / CRBA derivatives
    compiled for the host) against the dual-number checker: same SQP path, same iterates -- a third derivation of the
    tiles that runs without a GPU.  Serial chain (Panda) and tree (30 DoF), dt factors, frame and collision rows."""
    from agimus_controller_amd import _abi, workloads
    from agimus_controller_amd.factory import robot_tables as rt
    from oracle.oracle import Oracle

    cases = [(rt.panda_table(0.1), "goal", 12, 3, None), (rt.panda_collision_table(0.1, obstacle_xyz=(0.45, 0.1, 0.45), obstacle_radius=0.08,
              obstacle_length=0.3), "collision", 10, 2, None), (rt.humanoid30_table(), "goal", 6, 2, [0.01, 0.01, 0.02, 0.02, 0.04, 0.04])]
    for table, rows, T, B, ts in cases:
        frame = table.frame_id("panda_hand_tcp") if table.nv == 7 else len(table.frame_names) - 1
        po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=4, frame=frame, rows=rows, timesteps=ts)
        o = Oracle(table, po, B)
        r_ad = o.solve(ref, None, x0, xs, us, 12)
        assert o.set_analytic(True)
        r_an = o.solve(ref, None, x0, xs, us, 12)
        assert np.array_equal(r_ad[3]["iter"], r_an[3]["iter"]) and np.array_equal(r_ad[3]["solved"], r_an[3]["solved"])
        np.testing.assert_allclose(r_an[0], r_ad[0], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r_an[1], r_ad[1], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(r_an[2], r_ad[2], rtol=1e-7, atol=1e-7)
    # not covered: general rows / constraints keep the automatic-differentiation path
    table = rt.panda_table(0.1)
    rows = [_abi.RowSpec(_abi.RES_STATE), _abi.RowSpec(_abi.RES_CONTROL_GRAV)]
    assert not Oracle(table, _abi.PackedOcp(7, [0.01] * 3, rows, rows[:1]), 1).set_analytic(True)
