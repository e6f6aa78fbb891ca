"""Model sizes at run time: the kernels are compiled for a few capacities (7 joints: eight lanes per node; 30 and 32:
one workgroup per node) and a model with any number of joints up to 32 runs at the smallest capacity that holds it,
padded with massless joints that couple to nothing (agx_model_create, csrc/agimus_hip.hip).  The reference takes any
URDF and any set of locked joints (factory/robot_model.py:231-257): a Panda with its two finger joints unlocked has
nv = 9, a mobile manipulator more.  Every case goes through the C ABI in the caller's own layout (nv joints) and is
compared with the CPU checker, which is generic in nv."""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _model(nv, kind):
    if kind == "panda_fingers":
        # the Panda with two more joints at the hand (the finger joints robot_model.py locks by default): chain of 9
        p = rt.panda_table(0.1)
        f = rt.chain_table(2, seed=5, armature=0.1)
        import dataclasses

        def cat(a, b):
            return np.concatenate([np.asarray(a), np.asarray(b)])

        frame_parent = np.asarray(p.frame_parent).copy()
        return dataclasses.replace(
            p, name="panda_fingers", joint_names=list(p.joint_names) + ["finger_joint1", "finger_joint2"],
            parent=np.arange(-1, 8, dtype=np.int32), placement=cat(p.placement, f.placement), axis=cat(p.axis, f.axis),
            mass=cat(p.mass, 0.05 * f.mass), com=cat(p.com, 0.2 * f.com), inertia=cat(p.inertia, 1e-3 * f.inertia),
            armature=cat(p.armature, f.armature), effort_limit=cat(p.effort_limit, [20.0, 20.0]),
            lower_position_limit=cat(p.lower_position_limit, [-0.5, -0.5]), upper_position_limit=cat(p.upper_position_limit, [0.5, 0.5]),
            velocity_limit=cat(p.velocity_limit, [1.0, 1.0]), frame_parent=frame_parent)
    if kind == "chain":
        return rt.chain_table(nv, seed=20 + nv)
    return rt.tree_table(nv, seed=40 + nv)


CASES = [(5, "chain"), (5, "tree"), (8, "chain"), (9, "panda_fingers"), (12, "tree"), (16, "tree"), (24, "tree"), (31, "chain"), (32, "tree")]


@pytest.mark.parametrize("nv,kind", CASES)
def test_primitives_and_derivative_tiles(hip_backend, nv, kind):
    """RNEA, frame placement / Jacobian, integrate and the canonical derivative tiles (1e-10) in the caller's layout."""
    table = _model(nv, kind)
    assert table.nv == nv
    frame = len(table.frame_names) - 1
    B, T = 3, 4
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=nv, frame=frame, timesteps=[0.01, 0.01, 0.02, 0.02])
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    rng = np.random.default_rng(nv)
    q, v, a = rng.uniform(-1.0, 1.0, (3, 5, nv))
    assert rel(h.rnea(q, v, a), o.rnea(q, v, a).reshape(5, nv)) < 1e-11
    assert rel(h.frame_placement(frame, q), o.frame_placement(frame, q)) < 1e-12
    x = np.concatenate([q, v], axis=1)
    assert rel(h.integrate(x, 3 * a), o.integrate(x, 3 * a).reshape(5, 2 * nv)) < 1e-10
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    got, want = h.calc_diff(), o.calc_diff(ref, None, xs, us)
    assert got.shape == want.shape == (B, T + 1, _abi.tile_doubles(nv))
    for field, s in _abi.tile_slices(nv).items():
        scale = max(np.abs(want[..., s]).max(), 1e-300)
        assert np.abs(got[..., s] - want[..., s]).max() <= 1e-10 * scale + 1e-13, field
    h.close()


@pytest.mark.parametrize("nv,kind", CASES)
def test_full_solve_and_shift(hip_backend, nv, kind):
    """SQP solve (same iterations, xs / us to 1e-8, gains to 1e-7 relative), first-node download, warm-start shift with mixed dt."""
    table = _model(nv, kind)
    frame = len(table.frame_names) - 1
    B, T = 2, 6
    ts = [0.01] * 4 + [0.02] * 2
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=100 + nv, frame=frame, timesteps=ts)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 8)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 8)
    assert xs_h.shape == (B, T + 1, 2 * nv) and us_h.shape == (B, T, nv) and K_h.shape == (B, T, nv, 2 * nv)
    assert np.array_equal(st_h["iter"], st_o["iter"]) and np.array_equal(st_h["solved"], st_o["solved"])
    np.testing.assert_allclose(xs_h, xs_o, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(us_h, us_o, rtol=1e-8, atol=1e-8)
    assert rel(K_h, K_o) < 1e-7
    np.testing.assert_allclose(st_h["kkt"], st_o["kkt"], rtol=1e-5, atol=1e-10)
    us0, K0, x1, st1 = h.download_first()
    np.testing.assert_array_equal(us0, us_h[:, 0])
    np.testing.assert_array_equal(K0, K_h[:, 0])
    np.testing.assert_array_equal(x1, xs_h[:, 1])
    h.shift_warmstart()
    xs_s, us_s, _, _ = h.download(want_K=False)
    xs_so, us_so = o.shift_warmstart(xs_h, us_h)
    np.testing.assert_allclose(xs_s, xs_so, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(us_s, us_so, rtol=1e-10, atol=1e-12)
    h.close()


def test_resident_sine_trajectory_nine_joints(hip_backend):
    """The device-side reference generator and the resident MPC step for a padded model (nv = 9 at capacity 30)."""
    table = _model(9, "panda_fingers")
    tcp = table.frame_id("panda_hand_tcp")
    B, T, dt = 3, 10, 0.01
    running, terminal = workloads.goal_reaching_rows(tcp)
    po = _abi.PackedOcp(9, [dt] * T, running, terminal)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    q0, amp, puls, scale, t0 = workloads.sine_batch_params(B, nv=9, seed0=3, q0=np.zeros(9), lower=table.lower_position_limit,
                                                           upper=table.upper_position_limit)
    w = workloads.SINE_WEIGHTS
    h.sine_trajectory(T + 8, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
    q, v, a, u, pose = h.traj_point(3)
    assert q.shape == (B, 9) and u.shape == (B, 9)
    np.testing.assert_allclose(u, o.rnea(q, v, a).reshape(B, 9), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(pose, o.frame_placement(tcp, q), rtol=1e-11, atol=1e-12)
    h.mpc_step(0, 10, first=True)
    us0, K0, x1, st = h.download_first()
    assert us0.shape == (B, 9) and K0.shape == (B, 9, 18) and np.all(st["solved"] == 1)
    h.mpc_step(1, 10, first=False)
    assert np.all(h.download_first()[3]["solved"] == 1)
    h.close()


def test_constraints_on_a_five_joint_chain(hip_backend):
    """ConstraintModelControlLimit / state bounds through the ADMM loop for a padded model: the pad components are unbounded."""
    table = rt.chain_table(5, seed=77)
    B, T = 2, 8
    po0, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.02, B, seed=8)
    lim = np.full(5, 6.0)
    con = [_abi.ConstraintSpec(_abi.RES_CONTROL, lower=-lim, upper=lim, name="ctrl_limit")]
    po = _abi.PackedOcp(5, [0.02] * T, po0.running, po0.terminal, max_qp_iters=100, running_constraints=con)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 6)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 6)
    assert np.array_equal(st_h["iter"], st_o["iter"]) and np.array_equal(st_h["qp_iters"], st_o["qp_iters"])
    np.testing.assert_allclose(xs_h, xs_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(us_h, us_o, rtol=1e-5, atol=1e-5)
    assert np.abs(us_h).max() <= 6.0 + 1e-3
    h.close()


@pytest.mark.parametrize("nv,kind", [(9, "panda_fingers"), (24, "tree")])
def test_filter_line_search_for_large_models(hip_backend, nv, kind):
    """use_filter_line_search = True (ocp_param_base.py:64) on the workgroup-per-node path: the accept kernel is the same for
    every model size."""
    table = _model(nv, kind)
    frame = len(table.frame_names) - 1
    B, T = 2, 6
    po0, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=300 + nv, frame=frame)
    po = _abi.PackedOcp(nv, [0.01] * T, po0.running, po0.terminal, use_filter_line_search=True)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 6)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 6)
    assert np.array_equal(st_h["iter"], st_o["iter"]) and np.array_equal(st_h["flags"], st_o["flags"])
    np.testing.assert_allclose(xs_h, xs_o, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(us_h, us_o, rtol=1e-8, atol=1e-8)
    h.close()


@pytest.mark.parametrize("nv,kind,limit", [(9, "panda_fingers", 6.0), (16, "tree", 5.0), (30, "humanoid", 40.0), (31, "chain", 8.0)])
def test_control_limits_on_large_models(hip_backend, nv, kind, limit):
    """ConstraintModelControlLimit (ocp_croco_generic.py:624-640) for models above 7 joints: bounds on u through the ADMM loop on
    the workgroup-per-node path (k_admm_tile_big / k_riccati_blk on the augmented tile / k_admm_update_big, csrc/agx_big.hpp).
    Same SQP and ADMM iteration counts, iterate and bounded controls as the checker; other constraint kinds are refused."""
    table = rt.humanoid30_table() if kind == "humanoid" else _model(nv, kind)
    assert table.nv == nv
    frame = len(table.frame_names) - 1
    B, T = 2, 8
    po0, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.02, B, seed=500 + nv, frame=frame)
    lim = np.full(nv, limit)
    con = [_abi.ConstraintSpec(_abi.RES_CONTROL, lower=-lim, upper=lim, name="ctrl_limit")]
    po = _abi.PackedOcp(nv, [0.02] * T, po0.running, po0.terminal, max_qp_iters=100, running_constraints=con)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 6)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 6)
    assert np.abs(us_o).max() > 0.98 * limit  # the bound is active
    assert np.array_equal(st_h["iter"], st_o["iter"]) and np.array_equal(st_h["qp_iters"], st_o["qp_iters"])
    np.testing.assert_allclose(xs_h, xs_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(us_h, us_o, rtol=1e-5, atol=1e-5)
    assert np.abs(us_h).max() <= limit + 1e-3
    h.close()
    # a FrameVelocity constraint on a large model is refused with a message
    bad = [_abi.ConstraintSpec(_abi.RES_FRAME_VELOCITY, lower=-np.ones(6), upper=np.ones(6), frame=frame, name="ee_twist")]
    with pytest.raises(Exception, match="at most 7 joints"):
        hip_backend.HipOcp(table, _abi.PackedOcp(nv, [0.02] * T, po0.running, po0.terminal, running_constraints=bad), B)


def _panda_collision_with_fingers():
    """The collision-avoidance Panda of tests/test_constraints.py (arm capsules, one obstacle) with its two finger joints unlocked:
    nv = 9, runs at the 16-joint capacity."""
    import dataclasses

    p = rt.panda_collision_table(0.1, obstacle_xyz=(0.45, 0.1, 0.45), obstacle_radius=0.08, obstacle_length=0.3)
    f = rt.chain_table(2, seed=5, armature=0.1)

    def cat(a, b):
        return np.concatenate([np.asarray(a), np.asarray(b)])

    return dataclasses.replace(
        p, name="panda_collision_fingers", joint_names=list(p.joint_names) + ["finger_joint1", "finger_joint2"],
        parent=np.arange(-1, 8, dtype=np.int32), placement=cat(p.placement, f.placement), axis=cat(p.axis, f.axis),
        mass=cat(p.mass, 0.05 * f.mass), com=cat(p.com, 0.2 * f.com), inertia=cat(p.inertia, 1e-3 * f.inertia),
        armature=cat(p.armature, f.armature), effort_limit=cat(p.effort_limit, [20.0, 20.0]),
        lower_position_limit=cat(p.lower_position_limit, [-0.5, -0.5]), upper_position_limit=cat(p.upper_position_limit, [0.5, 0.5]),
        velocity_limit=cat(p.velocity_limit, [1.0, 1.0]))


def test_collision_constraint_and_torque_limits_on_nine_joints(hip_backend):
    """The constraint of ocp_traj_tracking_collision_avoidance.yaml (distance of a capsule pair >= bound) together with torque limits
    for the Panda with unlocked fingers: constraint values and Jacobian rows from k_con_eval_wg, ADMM on the workgroup path."""
    table = _panda_collision_with_fingers()
    nv = table.nv
    assert nv == 9
    tcp = table.frame_id("panda_hand_tcp")
    T, B = 10, 3
    running, terminal = workloads.collision_avoidance_rows(table, tcp, alpha=0.05)
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    lim = np.full(nv, 30.0)
    con = [_abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.05, upper=np.inf, frame=fa, frame_b=fb, name="collision"),
           _abi.ConstraintSpec(_abi.RES_CONTROL, lower=-lim, upper=lim, name="ctrl_limit")]
    po = _abi.PackedOcp(nv, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con,
                        terminal_constraints=[con[0]])
    _, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, 23, frame=tcp, rows="collision")
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    r_o = o.solve(ref, None, x0, xs, us, 3)
    r_h = h.solve(x0, xs, us, 3)
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-4, atol=1e-8)
    h.close()


@pytest.mark.parametrize("which", ["translation+collision", "rotation", "placement"])
def test_frame_constraints_on_nine_joints(hip_backend, which):
    """ConstraintModelResidual on the frame residuals (ocp_croco_generic.py:198-356, 594-620) for a model above 7 joints: the end
    effector of the Panda with unlocked fingers confined to a box / its orientation / its pose bounded, next to the collision
    constraint -- the nr Jacobian rows on q of each row come from k_con_eval_wg.  Same ADMM iteration counts as the checker."""
    table = _panda_collision_with_fingers()
    nv = table.nv
    tcp = table.frame_id("panda_hand_tcp")
    T, B = 10, 3
    running, terminal = workloads.goal_reaching_rows(tcp)
    _, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, 23, frame=tcp)
    o0 = Oracle(table, _abi.PackedOcp(nv, [0.01] * T, running, terminal), B)
    P0 = o0.frame_placement(tcp, x0[:, :nv])
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    if which == "translation+collision":
        con = [_abi.ConstraintSpec(_abi.RES_FRAME_TRANSLATION, lower=[-0.02, -0.03, -0.01], upper=[0.02, 0.01, 0.03], ref=P0[:, 9:].mean(0), frame=tcp, name="ee_box"),
               _abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.05, upper=np.inf, frame=fa, frame_b=fb, name="collision")]
    elif which == "rotation":
        con = [_abi.ConstraintSpec(_abi.RES_FRAME_ROTATION, lower=-0.05, upper=0.05, ref=P0[0, :9], frame=tcp, name="ee_rot")]
    else:
        con = [_abi.ConstraintSpec(_abi.RES_FRAME_PLACEMENT, lower=-0.04, upper=0.04, ref=P0[0], frame=tcp, name="ee_pose")]
    po = _abi.PackedOcp(nv, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=con)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    r_o = o.solve(ref, None, x0, xs, us, 2)
    r_h = h.solve(x0, xs, us, 2)
    assert np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-4, atol=1e-8)
    h.close()


@pytest.mark.parametrize("nv,kind", [(9, "panda_fingers"), (30, "humanoid")])
def test_state_bounds_and_torque_limits_on_large_models(hip_backend, nv, kind):
    """Bounds on the state (here: joint-velocity limits) together with torque limits for models above 7 joints: 2 nv + nv
    constraint components per node (the workgroup path takes up to 104)."""
    table = rt.humanoid30_table() if kind == "humanoid" else _model(nv, kind)
    frame = len(table.frame_names) - 1
    B, T = 2, 8
    po0, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.02, B, seed=600 + nv, frame=frame)
    lim = np.full(nv, 40.0 if kind == "humanoid" else 6.0)
    xb = np.full(2 * nv, np.inf)
    xb[nv:] = 0.6
    con = [_abi.ConstraintSpec(_abi.RES_STATE, lower=-xb, upper=xb, name="velocity_limit"),
           _abi.ConstraintSpec(_abi.RES_CONTROL, lower=-lim, upper=lim, name="ctrl_limit")]
    po = _abi.PackedOcp(nv, [0.02] * T, po0.running, po0.terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=[con[0]])
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    r_h = h.solve(x0, xs, us, 4)
    r_o = o.solve(ref, None, x0, xs, us, 4)
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-4, atol=1e-8)
    h.close()
