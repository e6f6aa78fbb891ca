"""Constraints through ADMM (SolverCSQP.computeDirection; section 8 a-1 / a-6 / f-2).

No fixture of the reference exercises constraints and mim_solvers is absent: parity with the reference
binaries is UNPINNED for this path.  These tests pin the CPU checker's constrained solve through
properties (feasibility, KKT, agreement with the unconstrained solve when no bound is active) and the
HIP path against the CPU checker.
"""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt


def _oracle(table, po, B=1):
    from oracle.oracle import Oracle

    return Oracle(table, po, B)


def _control_limit_problem(limit, T=16, B=2, seed=3, max_qp=200):
    table = rt.panda_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    running, terminal = workloads.goal_reaching_rows(tcp)
    cons = [] if limit is None else [_abi.ConstraintSpec(_abi.RES_CONTROL, lower=-np.asarray(limit), upper=np.asarray(limit), name="ulim")]
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=max_qp, running_constraints=cons)
    _, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed, frame=tcp)
    return table, po, ref, x0, xs, us


def test_inactive_control_limits_reproduce_the_unconstrained_solve():
    table, po0, ref, x0, xs, us = _control_limit_problem(None)
    r0 = _oracle(table, po0, 2).solve(ref, None, x0, xs, us, 30)
    _, po1, *_ = _control_limit_problem(np.full(7, 1e3))
    r1 = _oracle(table, po1, 2).solve(ref, None, x0, xs, us, 30)
    assert np.array_equal(r0[3]["iter"], r1[3]["iter"])
    np.testing.assert_allclose(r1[0], r0[0], atol=1e-6)
    np.testing.assert_allclose(r1[1], r0[1], atol=1e-4)
    assert np.all(r1[3]["qp_iters"] < 200)


def test_active_control_limits_are_respected_and_converge():
    lim = np.full(7, 15.0)
    table, po, ref, x0, xs, us = _control_limit_problem(lim, max_qp=400)
    xs_c, us_c, K, st = _oracle(table, po, 2).solve(ref, None, x0, xs, us, 40)
    assert np.all(st["solved"] == 1)
    assert np.abs(us_c).max() <= 15.0 + 1e-4
    assert np.all(st["kkt"] <= 1e-3)
    # the unconstrained optimum needs more torque than allowed: the bound really is active
    _, po0, *_ = _control_limit_problem(None)
    us_u = _oracle(table, po0, 2).solve(ref, None, x0, xs, us, 40)[1]
    assert np.abs(us_u).max() > 30.0


def test_duals_persist_across_solves_like_the_solver_object():
    """reset_y = reset_rho = false: a second solve from the same warm start starts from the
    multipliers of the first one and needs no more QP iterations than the first."""
    lim = np.full(7, 15.0)
    table, po, ref, x0, xs, us = _control_limit_problem(lim, B=1, max_qp=400)
    o = _oracle(table, po, 1)
    r1 = o.solve(ref[:1], None, x0[:1], xs[:1], us[:1], 40)
    r2 = o.solve(ref[:1], None, x0[:1], r1[0], r1[1], 40)
    assert r2[3]["iter"][0] <= 1  # warm start at the solution: converged immediately or after one step
    o.reset_duals()
    r3 = o.solve(ref[:1], None, x0[:1], xs[:1], us[:1], 40)
    np.testing.assert_allclose(r3[0], r1[0], atol=1e-9)


def test_collision_constraint_keeps_the_pair_apart():
    """ocp_traj_tracking_collision_avoidance.yaml:48-56: distance >= lower on the collision pair."""
    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.45, 0.1, 0.45), obstacle_radius=0.08, obstacle_length=0.3)
    tcp = table.frame_id("panda_hand_tcp")
    T, B = 12, 2
    running, terminal = workloads.goal_reaching_rows(tcp)
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    con = [_abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.05, upper=np.inf, frame=fa, frame_b=fb, name="collision")]
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=400, running_constraints=con, terminal_constraints=con)
    _, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, 23, frame=tcp)
    o = _oracle(table, po, B)
    xs_c, us_c, K, st = o.solve(ref, None, x0, xs, us, 60)
    for b in range(B):
        for t in range(1, T + 1):
            g, Gx, Gu = o.node_constraints(t == T, xs_c[b, t], None if t == T else us_c[b, min(t, T - 1)])
            assert g[0] >= 0.05 - 2e-3, (b, t, g[0])
    assert np.all(np.isfinite(K))


def _translation_box_problem(T=12, B=2, seed=23, max_qp=400, with_collision=False):
    """goal reaching with the end effector confined to a box around a point (ConstraintModelResidual on
    ResidualModelFrameTranslation, ocp_croco_generic.py:252-275, 594-620): lower <= p(q) - pref <= upper."""
    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.45, 0.1, 0.45), obstacle_radius=0.08, obstacle_length=0.3)
    tcp = table.frame_id("panda_hand_tcp")
    running, terminal = workloads.goal_reaching_rows(tcp)
    _, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed, frame=tcp)
    return table, tcp, running, terminal, ref, x0, xs, us


def test_frame_translation_constraint_confines_the_end_effector():
    table, tcp, running, terminal, ref, x0, xs, us = _translation_box_problem()
    T, B = 12, 2
    o0 = _oracle(table, _abi.PackedOcp(7, [0.01] * T, running, terminal), B)
    xs_u = o0.solve(ref, None, x0, xs, us, 60)[0]
    p0 = o0.frame_placement(tcp, x0[:, :7])[:, 9:]
    pu = np.stack([o0.frame_placement(tcp, xs_u[:, t, :7])[:, 9:] for t in range(T + 1)], 1)
    travel = np.abs(pu - p0[:, None, :]).max()
    assert travel > 0.02  # the unconstrained solution moves the end effector by more than the box allows
    half = 0.4 * travel
    for b in range(B):
        con = [_abi.ConstraintSpec(_abi.RES_FRAME_TRANSLATION, lower=-half, upper=half, ref=p0[b], frame=tcp, name="ee_box")]
        po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=400, running_constraints=con, terminal_constraints=con)
        o = _oracle(table, po, 1)
        xs_c, us_c, K, st = o.solve(ref[b:b + 1], None, x0[b:b + 1], xs[b:b + 1], us[b:b + 1], 60)
        pc = np.stack([o.frame_placement(tcp, xs_c[:, t, :7])[0, 9:] for t in range(T + 1)])
        assert np.abs(pc[1:] - p0[b]).max() <= half + 2e-3
        assert np.all(np.isfinite(K))
        # Jacobian rows of the constraint = finite differences of the frame translation
        g, Gx, Gu = o.node_constraints(False, xs_c[0, 3], us_c[0, 3])
        h = 1e-6
        for j in range(7):
            e = np.zeros(14)
            e[j] = h
            gp = o.node_constraints(False, xs_c[0, 3] + e, us_c[0, 3])[0]
            gm = o.node_constraints(False, xs_c[0, 3] - e, us_c[0, 3])[0]
            np.testing.assert_allclose(Gx[:, j], (gp - gm) / (2 * h), atol=1e-7)
        assert not np.any(Gx[:, 7:]) and not np.any(Gu)


def test_yaml_constraints_lower_to_rows():
    from agimus_controller_amd.ocp import ocp_croco_generic as g
    from agimus_controller_amd.factory.robot_model import RobotModelParameters, RobotModels

    table = rt.panda_collision_table(0.1)
    rm = RobotModels(RobotModelParameters(table=table, armature=table.armature, collision_pairs=[("panda_link7_capsule_0", "obstacle")]))
    diff = g.create_croco_dataclasses({
        "class": "DifferentialActionModelFreeFwdDynamics",
        "costs": [{"name": "state_reg", "cost": {"class": "CostModelResidual", "residual": {"class": "ResidualModelState"}}}],
        "constraints": [
            {"name": "collision", "constraint": {"class": "ConstraintModelResidual", "lower": 0.01, "upper": "inf",
                                                   "residual": {"class": "ResidualDistanceCollision", "collision_pair_id": 0}}},
            {"name": "torque", "constraint": {"class": "ConstraintModelControlLimit"}},
            {"name": "ee_box", "constraint": {"class": "ConstraintModelResidual", "lower": [-0.1, -0.1, 0.0], "upper": [0.1, 0.1, 0.3],
                                                "residual": {"class": "ResidualModelFrameTranslation", "id": "panda_hand_tcp",
                                                             "pref": [0.4, 0.0, 0.4]}}},
        ],
    })
    data = g.BuildData(rm.robot_model, 7, rm.collision_model)
    run = diff.lower_constraints(data, False)
    assert [c.kind for c in run] == [_abi.RES_COLLISION, _abi.RES_CONTROL, _abi.RES_FRAME_TRANSLATION]
    assert run[2].frame == table.frame_id("panda_hand_tcp")
    np.testing.assert_allclose(run[2].ref, [0.4, 0.0, 0.4])
    np.testing.assert_allclose(run[2].upper, [0.1, 0.1, 0.3])
    assert float(np.asarray(run[0].lower).reshape(-1)[0]) == pytest.approx(0.01) and np.isinf(np.asarray(run[0].upper)).all()
    np.testing.assert_allclose(run[1].upper, table.effort_limit)
    np.testing.assert_allclose(run[1].lower, -table.effort_limit)
    term = diff.lower_constraints(data, True)
    assert term[0].active and not term[1].active  # no control at the terminal node
    po = _abi.PackedOcp(7, [0.01] * 4, diff.lower(data), diff.lower(data), running_constraints=run, terminal_constraints=term)
    assert po.desc.n_running_constraints == 3 and po.desc.n_terminal_constraints == 3


@pytest.mark.gpu
def test_hip_constraints_with_the_filter_line_search():
    """use_filter_line_search (ocp_param_base.py:64) together with constraints: the filter compares cost,
    gaps and constraint violation of the trial point (k_step<FILTER, CON>)."""
    from agimus_controller_amd import backend

    lim = np.full(7, 15.0)
    table, po0, ref, x0, xs, us = _control_limit_problem(lim, T=12, B=3, max_qp=100)
    po = _abi.PackedOcp(7, [0.01] * 12, po0.running, po0.terminal, max_qp_iters=100, running_constraints=po0.running_constraints,
                        use_filter_line_search=True)
    o = _oracle(table, po, 3)
    hb = backend.HipOcp(table, po, 3)
    hb.set_refs(ref)
    r_o = o.solve(ref, None, x0, xs, us, 12)
    r_h = hb.solve(x0, xs, us, 12)
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["solved"], r_o[3]["solved"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
    hb.close()


@pytest.mark.gpu
def test_hip_control_limits_match_the_checker():
    from agimus_controller_amd import backend

    lim = np.full(7, 15.0)
    table, po, ref, x0, xs, us = _control_limit_problem(lim, T=12, B=4, max_qp=100)
    o = _oracle(table, po, 4)
    hb = backend.HipOcp(table, po, 4)
    hb.set_refs(ref)
    # one SQP iteration first: identical ADMM iteration counts, tight agreement
    r_o = o.solve(ref, None, x0, xs, us, 1)
    r_h = hb.solve(x0, xs, us, 1)
    assert np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(r_h[2], r_o[2], rtol=1e-6, atol=1e-6)
    # full solve from fresh multipliers
    o.reset_duals()
    hb.reset_duals()
    r_o = o.solve(ref, None, x0, xs, us, 30)
    r_h = hb.solve(x0, xs, us, 30)
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"])
    assert np.array_equal(r_h[3]["solved"], r_o[3]["solved"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
    assert np.abs(r_h[1]).max() <= 15.0 + 1e-4
    hb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("box", [None, (0.1, 0.15, 0.08)])
def test_hip_collision_constraint_matches_the_checker(box):
    from agimus_controller_amd import backend

    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.45, 0.1, 0.45), obstacle_radius=0.08, obstacle_length=0.3, obstacle_box=box)
    tcp = table.frame_id("panda_hand_tcp")
    T, B = 10, 3
    running, terminal = workloads.collision_avoidance_rows(table, tcp, alpha=0.05)
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    con = [_abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.05, upper=np.inf, frame=fa, frame_b=fb, name="collision"),
           _abi.ConstraintSpec(_abi.RES_STATE, lower=-5.0, upper=5.0, name="box")]
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=con)
    _, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, 23, frame=tcp, rows="collision")
    o = _oracle(table, po, B)
    hb = backend.HipOcp(table, po, B)
    hb.set_refs(ref)
    r_o = o.solve(ref, None, x0, xs, us, 2)
    r_h = hb.solve(x0, xs, us, 2)
    assert np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[2], r_o[2], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-5, atol=1e-8)
    hb.close()


@pytest.mark.gpu
def test_hip_frame_translation_constraint_matches_the_checker():
    """end effector confined to a box (three components with dense Jacobian rows) next to a collision
    constraint (the fourth Jacobian slot): HIP ADMM loop against the CPU checker."""
    from agimus_controller_amd import backend

    table, tcp, running, terminal, ref, x0, xs, us = _translation_box_problem(T=10, B=3)
    T, B = 10, 3
    o0 = _oracle(table, _abi.PackedOcp(7, [0.01] * T, running, terminal), B)
    p0 = o0.frame_placement(tcp, x0[:, :7])[:, 9:]
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    con = [_abi.ConstraintSpec(_abi.RES_FRAME_TRANSLATION, lower=[-0.02, -0.03, -0.01], upper=[0.02, 0.01, 0.03], ref=p0.mean(0), frame=tcp, name="ee_box"),
           _abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.05, upper=np.inf, frame=fa, frame_b=fb, name="collision")]
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=con)
    o = _oracle(table, po, B)
    hb = backend.HipOcp(table, po, B)
    hb.set_refs(ref)
    r_o = o.solve(ref, None, x0, xs, us, 2)
    r_h = hb.solve(x0, xs, us, 2)
    assert np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[2], r_o[2], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-5, atol=1e-8)
    hb.close()
    # a tenth dense-Jacobian component does not fit (placement 6 + translation 3 + collision 1)
    P0 = o0.frame_placement(tcp, x0[:1, :7])[0]
    con2 = con + [_abi.ConstraintSpec(_abi.RES_FRAME_PLACEMENT, lower=-1.0, upper=1.0, ref=P0, frame=tcp, name="pose")]
    po2 = _abi.PackedOcp(7, [0.01] * T, running, terminal, running_constraints=con2)
    with pytest.raises(backend.HipError, match="dense Jacobian"):
        backend.HipOcp(table, po2, 1)


@pytest.mark.gpu
def test_hip_lqr_pass_next_to_the_factorisation_is_the_same_solve(monkeypatch):
    """The constrained SQP iteration launches the plain LQR pass and the factorisation of the first ADMM iteration in
    one kernel (k_riccati_lqr_prefactor); AGX_ADMM_PREFACTOR=0 runs them one after the other, the factorisation
    inside the first ADMM iteration.  Same iterations, same result up to round-off."""
    from agimus_controller_amd import backend

    table, tcp, running, terminal, ref, x0, xs, us = _translation_box_problem(T=10, B=3)
    T, B = 10, 3
    p0 = _oracle(table, _abi.PackedOcp(7, [0.01] * T, running, terminal), B).frame_placement(tcp, x0[:, :7])[:, 9:]
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    con = [_abi.ConstraintSpec(_abi.RES_FRAME_TRANSLATION, lower=[-0.02, -0.03, -0.01], upper=[0.02, 0.01, 0.03], ref=p0.mean(0), frame=tcp, name="ee_box"),
           _abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.05, upper=np.inf, frame=fa, frame_b=fb, name="collision")]
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=con)
    out = {}
    for pre in ("0", "1"):
        monkeypatch.setenv("AGX_ADMM_PREFACTOR", pre)
        hb = backend.HipOcp(table, po, B)
        hb.set_refs(ref)
        out[pre] = hb.solve(x0, xs, us, 4)
        hb.close()
    a, b = out["0"], out["1"]
    for key in ("iter", "qp_iters", "solved", "flags"):
        assert np.array_equal(a[3][key], b[3][key]), key
    np.testing.assert_allclose(b[0], a[0], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(b[2], a[2], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(b[3]["kkt"], a[3]["kkt"], rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_hip_admm_iterations_in_one_launch_are_the_same_solve(monkeypatch):
    """k_admm_loop runs the gradient-only ADMM iterations between two rho checks of an instance in one launch (sweep, node update,
    norms and convergence test on the instance's own workgroup); the solver uses it for the tail of a step, when few instances
    are left.  AGX_ADMM_LOOP=2 forces it for every instance, 0 launches every iteration as sweep / update / reduce: same ADMM
    iteration counts and results, against each other and against the checker -- also across a rho update (max_qp_iters 100,
    loose problem: more than 25 iterations) and with a batch quorum below 1 (chunks of the host's polling schedule)."""
    from agimus_controller_amd import backend

    lim = np.full(7, 15.0)
    table, po, ref, x0, xs, us = _control_limit_problem(lim, T=12, B=4, max_qp=100)
    r_o = _oracle(table, po, 4).solve(ref, None, x0, xs, us, 6)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("AGX_ADMM_LOOP", mode)
        hb = backend.HipOcp(table, po, 4)
        hb.set_refs(ref)
        out[mode] = hb.solve(x0, xs, us, 6)
        hb.set_quorum(1.0, 0.75)
        hb.reset_duals()
        out[mode + "q"] = hb.solve(x0, xs, us, 6)
        hb.close()
    for a, b in ((out["0"], out["2"]), (out["0q"], out["2q"])):
        for key in ("iter", "qp_iters", "solved", "flags"):
            assert np.array_equal(a[3][key], b[3][key]), key
        np.testing.assert_allclose(b[0], a[0], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(b[1], a[1], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(b[2], a[2], rtol=1e-9, atol=1e-9)
    r_h = out["2"]
    assert r_h[3]["qp_iters"].max() > 25  # a rho check was crossed
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["rotation+translation", "placement"])
def test_hip_frame_rotation_and_placement_constraints_match_the_checker(which):
    """ConstraintModelResidual on ResidualModelFrameRotation (log3, 3 components) and
    ResidualModelFramePlacement (log6, 6 components) next to a collision constraint."""
    from agimus_controller_amd import backend

    table, tcp, running, terminal, ref, x0, xs, us = _translation_box_problem(T=10, B=3)
    T, B = 10, 3
    o0 = _oracle(table, _abi.PackedOcp(7, [0.01] * T, running, terminal), B)
    P0 = o0.frame_placement(tcp, x0[:1, :7])[0]
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    coll = _abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.05, upper=np.inf, frame=fa, frame_b=fb, name="collision")
    if which == "placement":
        con = [_abi.ConstraintSpec(_abi.RES_FRAME_PLACEMENT, lower=[-0.03, -0.03, -0.03, -0.05, -0.05, -0.05],
                                   upper=[0.03, 0.03, 0.03, 0.05, 0.05, 0.05], ref=P0, frame=tcp, name="pose"), coll]
    else:
        con = [_abi.ConstraintSpec(_abi.RES_FRAME_ROTATION, lower=-0.04, upper=0.04, ref=P0[:9], frame=tcp, name="rot"),
               _abi.ConstraintSpec(_abi.RES_FRAME_TRANSLATION, lower=-0.03, upper=0.03, ref=P0[9:], frame=tcp, name="box"), coll]
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=con)
    o = _oracle(table, po, B)
    hb = backend.HipOcp(table, po, B)
    hb.set_refs(ref)
    r_o = o.solve(ref, None, x0, xs, us, 2)
    r_h = hb.solve(x0, xs, us, 2)
    assert np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[2], r_o[2], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-5, atol=1e-8)
    # the constraint really bites: without it the end effector leaves the bounds
    r_u = o0.solve(ref, None, x0, xs, us, 2)
    assert np.abs(r_u[0] - r_o[0]).max() > 1e-3
    hb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["frame_velocity", "control_grav"])
def test_hip_velocity_and_gravity_torque_constraints_match_the_checker(which):
    """Residuals that depend on v and u as constraints (rows with dense Jacobians [Gq | Gv | Gu]):
    ResidualModelFrameVelocity (LOCAL_WORLD_ALIGNED, bounded end-effector speed) and
    ResidualModelControlGrav (bounded torque beyond gravity compensation)."""
    from agimus_controller_amd import backend

    table, tcp, running, terminal, ref, x0, xs, us = _translation_box_problem(T=10, B=3)
    T, B = 10, 3
    if which == "frame_velocity":
        con = [_abi.ConstraintSpec(_abi.RES_FRAME_VELOCITY, lower=[-0.05, -0.05, -0.05, -0.5, -0.5, -0.5], upper=[0.05, 0.05, 0.05, 0.5, 0.5, 0.5],
                                   ref=np.zeros(6), frame=tcp, frame_b=2, name="ee_speed")]
        term = con
    else:
        con = [_abi.ConstraintSpec(_abi.RES_CONTROL_GRAV, lower=-4.0, upper=4.0, name="tau_minus_g")]
        term = []
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=term)
    o = _oracle(table, po, B)
    hb = backend.HipOcp(table, po, B)
    hb.set_refs(ref)
    r_o = o.solve(ref, None, x0, xs, us, 2)
    r_h = hb.solve(x0, xs, us, 2)
    assert np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[2], r_o[2], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-5, atol=1e-8)
    # the constraint changes the solution
    o0 = _oracle(table, _abi.PackedOcp(7, [0.01] * T, running, terminal), B)
    r_u = o0.solve(ref, None, x0, xs, us, 2)
    assert np.abs(r_u[1] - r_o[1]).max() > 1e-2
    hb.close()


def _obstacle_entering_problem():
    """Two instances of the collision workload of bench.py (T = 200): the reference of the first one ends
    inside the obstacle (instance 14 of the benchmark batch), the second one is a regular instance."""
    import bench

    T, dt = 200, 0.01
    table, tcp, po = bench.make_problem(T, "collision")
    sel = [14, 0]  # instance 14 of the benchmark batch is such a case, instance 0 is a regular one
    q0, amp, puls, scale, t0 = workloads.sine_batch_params(16, nv=7, seed0=1234, lower=table.lower_position_limit, upper=table.upper_position_limit)
    q0, amp, puls, scale, t0 = q0[sel], amp[sel], puls[sel], scale[sel], t0[sel]
    B = len(sel)
    o = _oracle(table, po, B)
    w = workloads.SINE_WEIGHTS
    ref = po.new_ref_tile(B)
    xs, us = np.empty((B, T + 1, 14)), np.empty((B, T, 7))
    for t in range(T + 1):
        tt = t0 + t * dt
        s_ = np.clip(tt[:, None] / scale, 0.0, 1.0)
        ramp = 10 * s_**3 - 15 * s_**4 + 6 * s_**5
        dramp = np.where((s_ > 0) & (s_ < 1), (30 * s_**2 - 60 * s_**3 + 30 * s_**4) / scale, 0.0)
        ddramp = np.where((s_ > 0) & (s_ < 1), (60 * s_ - 180 * s_**2 + 120 * s_**3) / scale**2, 0.0)
        sw, cw = np.sin(puls * tt[:, None]), np.cos(puls * tt[:, None])
        q = q0 + amp * ramp * sw
        dq = amp * (dramp * sw + ramp * puls * cw)
        ddq = amp * (ddramp * sw + 2 * dramp * puls * cw - ramp * puls**2 * sw)
        u, pose = o.rnea(q, dq, ddq).reshape(B, 7), o.frame_placement(tcp, q)
        xs[:, t] = np.concatenate([q, dq], 1)
        if t < T:
            us[:, t] = u
        rows, offs = (po.terminal, po.terminal_offsets) if t == T else (po.running, po.running_offsets)
        for r, off in zip(rows, offs):
            seg = ref[:, t, off:]
            seg[:, 0] = 1.0
            if r.kind == _abi.RES_STATE:
                seg[:, 1:15], seg[:, 15:22], seg[:, 22:29] = xs[:, t], w["w_q"], w["w_qdot"]
            elif r.kind == _abi.RES_CONTROL:
                seg[:, 1:8], seg[:, 8:15] = u, w["w_effort"]
            elif r.kind == _abi.RES_FRAME_PLACEMENT:
                seg[:, 1:13], seg[:, 13:19] = pose, w["w_pose"]
            else:
                seg[:, 1] = 0.0
    x0 = xs[:, 0].copy()
    dist = np.array([o.node_constraints(False, xs[0, t], us[0, t])[0][0] for t in range(T)])
    assert dist.min() < 0.0 < dist[0]  # the reference of the first instance really enters the obstacle
    return table, po, ref, x0, xs, us, B


def test_checker_discards_the_direction_when_quu_is_not_positive_definite():
    """the CPU checker on the case above: LLT failure -> direction discarded (flags bit 0), step rejected
    (bit 1; bit 2: step lengths were rejected), regularisation x10 per iteration; the multipliers stay finite and the solve gets going
    again once the regularisation is large enough."""
    table, po, ref, x0, xs, us, B = _obstacle_entering_problem()
    o = _oracle(table, po, B)
    xs2, us2, K2, st2 = o.solve(ref, None, x0, xs, us, 2)
    assert st2["flags"][0] == 7 and st2["qp_iters"][0] == 1 and np.array_equal(xs2[0], xs[0])
    assert np.isnan(st2["kkt"][0])  # no KKT residual for a discarded direction
    assert st2["flags"][1] == 0 and st2["solved"][1] == 1
    xs10, us10, K10, st10 = _oracle(table, po, B).solve(ref, None, x0, xs, us, 10)
    assert st10["qp_iters"][0] > 1 and not np.array_equal(xs10[0], xs[0])
    assert np.all(np.isfinite(xs10)) and np.all(np.isfinite(us10))


@pytest.mark.gpu
def test_hip_breakdown_of_the_factorisation_is_handled_like_the_checker():
    """The QuadExp cost has negative curvature near the obstacle: Quu loses positive definiteness, the
    direction is discarded (flags bit 0), the step rejected (bit 1), the regularisation raised -- and
    after enough iterations the solve proceeds.  HIP (sign of the reciprocal pivots) and checker (LLT
    failure) must walk the same path."""
    from agimus_controller_amd import backend

    table, po, ref, x0, xs, us, B = _obstacle_entering_problem()
    merit_h = {}
    for iters in (2, 7, 8, 10):
        r_o = _oracle(table, po, B).solve(ref, None, x0, xs, us, iters)
        hb = backend.HipOcp(table, po, B)
        hb.set_refs(ref)
        r_h = hb.solve(x0, xs, us, iters)
        hb.close()
        # the discrete path is the same in every phase: iterations, QP iterations, discarded-direction / rejected-step flags
        assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
        assert np.array_equal(r_h[3]["flags"], r_o[3]["flags"])
        merit_h[iters] = r_h[3]["merit"].copy()
        if iters <= 7:
            # deterministic phase (the checker's trace: directions discarded and steps rejected while the regularisation climbs
            # 1e-9 -> 1e-2; the first accepted step comes in iteration 8): exact
            np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-5, atol=1e-8, equal_nan=True)
            np.testing.assert_allclose(r_h[3]["merit"], r_o[3]["merit"], rtol=1e-9)
            np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
            assert r_o[3]["flags"][0] == 7 and np.array_equal(r_o[0][0], xs[0]) and np.array_equal(r_h[0][0], xs[0])
        else:
            # The recovery (100 ADMM iterations on a nearly singular QP) amplifies round-off: the checker built with AVX2 and
            # with AVX-512 vectorisation differs from itself by 2e-2 relative in the KKT value and 8e-3 in xs after the 10
            # iterations (measured, same source).  Values are therefore compared loosely here and the path is pinned by
            # invariants instead: same discrete path (above), x0 pinned, finite iterates, and a merit that never increases
            # along the accepted steps and agrees with the checker's.
            assert r_o[3]["qp_iters"][0] > 1 and not np.array_equal(r_o[0][0], xs[0]) and not np.array_equal(r_h[0][0], xs[0])
            np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=5e-2, atol=1e-8, equal_nan=True)
            np.testing.assert_allclose(r_h[3]["merit"], r_o[3]["merit"], rtol=5e-2)
            np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=3e-2)
            np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-1)
            np.testing.assert_array_equal(r_h[0][:, 0], x0)
            assert np.all(np.isfinite(r_h[0])) and np.all(np.isfinite(r_h[1]))
    # merit of the point each run stopped at (evaluated at its last SQP iteration): non-increasing once steps are accepted
    assert np.all(merit_h[10] <= merit_h[8] * (1 + 1e-12)) and np.all(merit_h[8] <= merit_h[7] * (1 + 1e-12))
    assert np.all(np.isfinite(r_h[2]))


@pytest.mark.gpu
def test_hip_config3_full_size_properties():
    """BASELINE.json configs[2] shape (horizon 200, batch 256, collision-avoidance costs + distance
    constraint) on the resident sine-wave workload: after MPC steps every solved instance keeps the
    pair apart and satisfies the KKT tolerance; a few instances are compared with the CPU checker."""
    from agimus_controller_amd import backend

    T, B, dt, lower = 200, 256, 0.01, 0.05
    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.27, 0.22, 0.70), obstacle_radius=0.06, obstacle_length=0.0)
    tcp = table.frame_id("panda_hand_tcp")
    running, terminal = workloads.collision_avoidance_rows(table, tcp, alpha=1e-4)
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    con = [_abi.ConstraintSpec(_abi.RES_COLLISION, lower=lower, upper=np.inf, frame=fa, frame_b=fb, name="collision")]
    po = _abi.PackedOcp(7, [dt] * T, running, terminal, max_qp_iters=50, running_constraints=con)
    hb = backend.HipOcp(table, po, B)
    q0, amp, puls, scale, t0 = workloads.sine_batch_params(B, lower=table.lower_position_limit, upper=table.upper_position_limit)
    w = workloads.SINE_WEIGHTS
    hb.sine_trajectory(T + 8, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
    for k in range(2):
        hb.mpc_step(k, 6, first=(k == 0))
    xs, us, K, st = hb.download()
    assert np.all(np.isfinite(xs)) and np.all(np.isfinite(K))
    d = hb.residuals(3)[..., 0]  # distance row of every running node at the returned trajectory
    solved = st["solved"] == 1
    assert solved.mean() > 0.5
    assert np.all(st["kkt"][solved] <= 1e-3)
    assert d[solved][:, 1:].min() >= lower - 1e-3
    # the constraint is really active for part of the batch
    assert (d[:, 1:].min(axis=1) < lower + 5e-3).sum() >= 3
    # checker on three instances (the ones closest to the obstacle), same two MPC steps
    idx = np.argsort(d[:, 1:].min(axis=1))[:3]
    o = _oracle(table, po, 3)
    pts = [hb.traj_point(k) for k in range(T + 2)]
    def window(k0):
        ref = po.new_ref_tile(3)
        for t in range(T + 1):
            q, dq, _, u, pose = (a[idx] for a in pts[k0 + t])
            rows, offs = (terminal, po.terminal_offsets) if t == T else (running, po.running_offsets)
            for r, off in zip(rows, offs):
                seg = ref[:, t, off:]
                seg[:, 0] = 1.0
                if r.kind == _abi.RES_STATE:
                    seg[:, 1:15] = np.concatenate([q, dq], 1)
                    seg[:, 15:22], seg[:, 22:29] = w["w_q"], w["w_qdot"]
                elif r.kind == _abi.RES_CONTROL:
                    seg[:, 1:8], seg[:, 8:15] = u, w["w_effort"]
                elif r.kind == _abi.RES_FRAME_PLACEMENT:
                    seg[:, 1:13], seg[:, 13:19] = pose, w["w_pose"]
        return ref
    xs_c = np.stack([np.concatenate([p[0][idx], p[1][idx]], 1) for p in pts[: T + 1]], 1)
    us_c = np.stack([p[3][idx] for p in pts[:T]], 1)
    x0 = xs_c[:, 0].copy()
    r = o.solve(window(0), None, x0, xs_c, us_c, 6)
    xs_s, us_s = o.shift_warmstart(r[0], r[1])
    r = o.solve(window(1), None, r[0][:, 1].copy(), xs_s, us_s, 6)
    assert np.array_equal(r[3]["iter"], st["iter"][idx])
    np.testing.assert_allclose(xs[idx], r[0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(us[idx], r[1], rtol=1e-4, atol=1e-4)
    hb.close()
