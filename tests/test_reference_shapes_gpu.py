"""Configurations exactly as the reference ships them, through the HIP path and against the CPU checker:

* the pick-and-place OCP of agimus_controller_examples/main/panda_pick_and_place/config/
  (agimus_controller_params.yaml:6-21, ocp_definition_file.yaml, trajectory_weigths_params.yaml:4-9);
* an arm / arm capsule pair (fer_link7_sc_capsule_0 / fer_link3_sc_capsule_0, agimus_controller_params.yaml:17-21):
  both geometries move with the arm, both halves of d'(q) = n' (J1(p1) - J2(p2)) are non-zero;
* ResidualModelVisualServoing with an input transform (ocp_croco_generic.py:436-495);
* max_solve_time (ocp_base_croco.py:70-71,166-171);
* BASELINE configs[2] as written: sine_wave_cartesian_space references + collision-avoidance costs + the
  distance >= 1 cm constraint, T = 200, B = 256, max_qp_iters 100.

Parity of the colmpc distance residual / activations and of the ADMM loop is UNPINNED (recalled forms, DESIGN.md
section 2): those tests compare the HIP path with this repository's own checker."""
import io

import numpy as np
import pytest
import yaml

from agimus_controller_amd import _abi, se3, workloads
from agimus_controller_amd.factory import robot_tables as rt
from agimus_controller_amd.factory.robot_model import RobotModelParameters, RobotModels, panda_robot_models
from agimus_controller_amd.mpc import MPC
from agimus_controller_amd.ocp.ocp_croco_generic import OCPCrocoGeneric
from agimus_controller_amd.ocp_param_base import DTFactorsNSeq, OCPParamsBaseCroco
from agimus_controller_amd.trajectory import TrajectoryBuffer, TrajectoryPoint, TrajectoryPointWeights, WeightedTrajectoryPoint
from agimus_controller_amd.warm_start_reference import WarmStartReference
from agimus_controller_amd.warm_start_shift_previous_solution import WarmStartShiftPreviousSolution
from agimus_controller_amd.workloads import PANDA_Q0
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _pick_and_place_points(n, dt, seed=3):
    """Joint trajectory as tests/test_generic_trajectory.py:147-160 builds it (smooth random accelerations integrated
    twice), with the weights of trajectory_weigths_params.yaml:4-9."""
    rng = np.random.default_rng(seed)
    ddq = np.cumsum(rng.normal(0.0, 0.4, (n, 7)), axis=0) * dt
    ddq -= ddq.mean(0)
    dq = np.cumsum(ddq, axis=0) * dt
    q = PANDA_Q0 + np.cumsum(dq, axis=0) * dt
    w = TrajectoryPointWeights(w_robot_configuration=3.0 * np.ones(7), w_robot_velocity=0.12 * np.ones(7), w_robot_acceleration=1e-6 * np.ones(7),
                               w_robot_effort=8e-4 * np.ones(7), w_end_effector_poses={"panda_hand_tcp": np.zeros(6)}, w_collision_avoidance=0.0)
    return q, dq, ddq, w


def test_pick_and_place_configuration_as_shipped(hip_backend):
    """T = 60 with n_steps [30, 20, 10] x factors [1, 2, 4], max_iter 3, max_qp_iter 200, control_reg + state_reg (terminal
    weight 0): three MPC.run steps (reference warm start, then shifts with integration of the coarse nodes) against the checker
    driven with the same inputs."""
    T, dt = 60, 0.01
    rm = panda_robot_models(0.1)
    seq = DTFactorsNSeq(factors=[1, 2, 4], n_steps=[30, 20, 10])
    params = OCPParamsBaseCroco(dt=dt, horizon_size=T, dt_factor_n_seq=seq, solver_iters=3, qp_iters=200, callbacks=False)
    assert params.n_controls == T and params.timesteps[0] == dt and params.timesteps[30] == 2 * dt and params.timesteps[59] == 4 * dt
    ocp = OCPCrocoGeneric(rm, params, OCPCrocoGeneric.get_default_yaml_file("ocp_regulation.yaml"))
    po = ocp.problem
    assert [r.kind for r in po.running] == [_abi.RES_CONTROL, _abi.RES_STATE] and [r.kind for r in po.terminal] == [_abi.RES_STATE]
    buffer = TrajectoryBuffer(seq)
    idx = buffer.horizon_indexes
    assert len(idx) == T + 1 and idx[30] == 30 and idx[31] == 32 and idx[50] == 70 and idx[51] == 74 and idx[60] == 110
    n_pts = idx[-1] + 8
    q, dq, ddq, w = _pick_and_place_points(n_pts, dt)
    o = Oracle(rm.table, po, 1)
    u_ff = o.rnea(q, dq, ddq).reshape(n_pts, 7)
    pts = []
    for k in range(n_pts):
        pose = o.frame_placement(rm.robot_model.getFrameId("panda_hand_tcp"), q[k:k + 1])[0]
        pts.append(WeightedTrajectoryPoint(
            TrajectoryPoint(id=k, time_ns=k, robot_configuration=q[k].copy(), robot_velocity=dq[k].copy(), robot_acceleration=ddq[k].copy(),
                            robot_effort=u_ff[k].copy(), end_effector_poses={"panda_hand_tcp": se3.SE3(pose[:9].reshape(3, 3), pose[9:])}), w))
    mpc = MPC()
    ws_ref = WarmStartReference()
    ws_ref.setup(ocp)
    mpc.setup(ocp, ws_ref, buffer)
    mpc.append_trajectory_points(pts)
    state = TrajectoryPoint(robot_configuration=q[0].copy(), robot_velocity=dq[0].copy(), robot_acceleration=ddq[0].copy(), time_ns=0)
    ws_shift = WarmStartShiftPreviousSolution()
    ws_shift.setup(rm, params, ocp)
    xs_prev = us_prev = None
    for step in range(3):
        res = mpc.run(state, step)
        # the same step on the checker: same reference tile, same x0, same warm start
        x0 = state.robot_state[None]
        if step == 0:
            xs_ws = np.stack([state.robot_state] + [np.concatenate([q[idx[t]], dq[idx[t]]]) for t in range(1, T + 1)])[None]
            us_ws = np.stack([o.rnea(state.robot_configuration, state.robot_velocity, state.robot_acceleration).reshape(7)] +
                             [u_ff[idx[t]] for t in range(1, T)])[None]
        else:
            xs_ws, us_ws = o.shift_warmstart(xs_prev, us_prev)
        xs_o, us_o, K_o, st_o = o.solve(ocp._ref_tile, ocp._frames, x0, xs_ws, us_ws, 3)
        assert mpc.mpc_debug_data.ocp.nb_iter == st_o["iter"][0] and mpc.mpc_debug_data.ocp.problem_solved == bool(st_o["solved"][0])
        np.testing.assert_allclose(np.array(res.states), xs_o[0], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(np.array(res.feed_forward_terms), us_o[0], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(np.array(res.ricatti_gains), K_o[0], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(mpc.mpc_debug_data.ocp.kkt_norm, st_o["kkt"][0], rtol=1e-6, atol=1e-12)
        xs_prev, us_prev = xs_o, us_o
        if step == 0:
            ws_shift.update_previous_solution(res)
            mpc.setup(ocp, ws_shift, buffer)
        state = TrajectoryPoint(robot_configuration=res.states[1][:7].copy(), robot_velocity=res.states[1][7:].copy(), time_ns=step + 1)


def _arm_arm_problem(T, B, as_constraint, seed=31):
    table = rt.panda_collision_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    pair = ("panda_link7_capsule_0", "panda_link3_capsule_0")
    fa, fb = table.frame_id(pair[0]), table.frame_id(pair[1])
    assert table.frame_parent[fa] == 6 and table.frame_parent[fb] == 2  # both on the arm
    running, terminal = workloads.collision_avoidance_rows(table, tcp, pair=pair, alpha=0.05)
    con = [_abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.08, upper=np.inf, frame=fa, frame_b=fb, name="self_collision")] if as_constraint else []
    po = _abi.PackedOcp(7, [0.01] * T, running, terminal, max_qp_iters=100, running_constraints=con, terminal_constraints=con)
    _, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed, frame=tcp, rows="collision")
    # fold the arm so that the pair is close (within the bell of the activation): elbow and wrist bent
    xs[..., 3] = -2.6 + 0.2 * np.sin(np.arange(T + 1))[None, :]
    xs[..., 5] = 3.2
    x0[:, 3], x0[:, 5] = xs[:, 0, 3], 3.2
    return table, po, ref, x0, xs, us, (fa, fb)


def test_arm_arm_capsule_pair_cost_tiles_and_solve(hip_backend):
    """Self-collision pair link7 / link3 as a cost row: derivative tiles against the checker's automatic differentiation
    (both Jacobian halves of the distance are non-zero) and the full solve."""
    B, T = 4, 10
    table, po, ref, x0, xs, us, (fa, fb) = _arm_arm_problem(T, B, as_constraint=False)
    o, h = Oracle(table, po, B), hip_backend.HipOcp(table, po, B)
    h.set_refs(ref)
    h.upload_x0(x0)
    h.upload_warmstart(xs, us)
    d = h.residuals(3)[..., 0]
    assert d.min() < 0.25, "the pair must be close enough for the cost to matter"
    # the joints between link 3 and link 7 change the distance, the ones below link 3 move both bodies together
    sl = _abi.tile_slices(7)
    got, want = h.calc_diff(), o.calc_diff(ref, None, xs, us)
    for name in ("Lx", "Lxx", "cost", "Fx", "Fu"):
        np.testing.assert_allclose(got[..., sl[name]], want[..., sl[name]], rtol=1e-9, atol=1e-10, err_msg=name)
    # isolate the distance row: its gradient in q must vanish for joints 0..2 (rigid motion of the pair) and not for 3..6
    rows = [_abi.RowSpec(_abi.RES_COLLISION, activation=_abi.ACT_QUAD_EXP, alpha=0.05, frame=fa, frame_b=fb)]
    po1 = _abi.PackedOcp(7, [0.01] * T, rows, rows)
    h1, o1 = hip_backend.HipOcp(table, po1, B), Oracle(table, po1, B)
    r1 = po1.new_ref_tile(B)
    h1.set_refs(r1)
    h1.upload_warmstart(xs, us)
    g1, w1 = h1.calc_diff()[..., sl["Lx"]], o1.calc_diff(r1, None, xs, us)[..., sl["Lx"]]
    np.testing.assert_allclose(g1, w1, rtol=1e-9, atol=1e-12)
    assert np.abs(g1[..., :3]).max() < 1e-12 * max(np.abs(g1).max(), 1.0) + 1e-13 and np.abs(g1[..., 3:7]).max() > 1e-6
    h1.close()
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 15)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 15)
    assert np.array_equal(st_h["iter"], st_o["iter"])
    np.testing.assert_allclose(xs_h, xs_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(us_h, us_o, rtol=1e-6, atol=1e-6)
    h.close()


def test_arm_arm_capsule_pair_as_constraint(hip_backend):
    """The same pair as ConstraintModelResidual (distance >= 8 cm) through the ADMM loop."""
    B, T = 3, 10
    table, po, ref, x0, xs, us, _ = _arm_arm_problem(T, B, as_constraint=True)
    o, h = Oracle(table, po, B), hip_backend.HipOcp(table, po, B)
    h.set_refs(ref)
    r_o, r_h = o.solve(ref, None, x0, xs, us, 3), h.solve(x0, xs, us, 3)
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"])
    assert r_o[3]["qp_iters"].max() > 1, "the constraint has to be active"
    np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r_h[2], r_o[2], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r_h[3]["kkt"], r_o[3]["kkt"], rtol=1e-5, atol=1e-8)
    h.close()


def _goal_doc(residual):
    doc = yaml.safe_load(open(OCPCrocoGeneric.get_default_yaml_file("ocp_goal_reaching.yaml")))
    for part in ("running_model", "terminal_model"):
        for item in doc[part]["differential"]["costs"]:
            if item["name"] == "goal_tracking":
                item["cost"]["residual"] = dict(residual)
    return io.StringIO(yaml.safe_dump(doc))


def test_visual_servoing_row_equals_the_composed_frame_placement(hip_backend):
    """ResidualModelVisualServoing through YAML: with wMo in input_transforms and oMf under `<robot_frame>_vs` the row is
    the FramePlacement row with reference wMo * oMf (ocp_croco_generic.py:470-476); without a transform the target is taken
    as it is; non-zero weights without a transform are refused."""
    T, dt = 8, 0.02
    rm = panda_robot_models(0.1)
    params = OCPParamsBaseCroco(dt=dt, horizon_size=T, dt_factor_n_seq=DTFactorsNSeq([1], [T]), solver_iters=20, callbacks=False)
    vs = OCPCrocoGeneric(rm, params, _goal_doc({"class": "ResidualModelVisualServoing", "world_frame": "universe", "object_frame": "box",
                                                "robot_frame": "panda_hand_tcp"}))
    fp = OCPCrocoGeneric(rm, params, OCPCrocoGeneric.get_default_yaml_file("ocp_goal_reaching.yaml"))
    assert vs.input_transforms == {("universe", "box"): None}
    rng = np.random.default_rng(8)
    wMo = se3.SE3(rt.rpy(0.4, -0.3, 0.9), np.array([0.35, -0.1, 0.2]))
    oMf = se3.SE3(rt.rpy(-0.2, 0.5, 0.1), np.array([0.1, 0.25, 0.3]))
    wMf = wMo * oMf
    weights = dict(w_robot_configuration=0.05 * np.ones(7), w_robot_velocity=0.1 * np.ones(7), w_robot_effort=1e-4 * np.ones(7))

    def point(key, pose, w_pose):
        return WeightedTrajectoryPoint(
            TrajectoryPoint(robot_configuration=PANDA_Q0, robot_velocity=np.zeros(7), robot_effort=np.zeros(7), end_effector_poses={key: pose}),
            TrajectoryPointWeights(w_end_effector_poses={key: w_pose}, **weights))

    w_pose = rng.uniform(5.0, 20.0, 6)
    with pytest.raises(AssertionError, match="no transform"):
        vs.set_reference_weighted_trajectory([point("panda_hand_tcp_vs", oMf, w_pose)] * (T + 1))
    vs.set_reference_weighted_trajectory([point("panda_hand_tcp_vs", oMf, np.zeros(6))] * (T + 1))  # inactive: allowed
    with pytest.raises(AssertionError, match="should contains key"):
        vs.set_reference_weighted_trajectory([point("panda_hand_tcp", oMf, w_pose)] * (T + 1))
    vs.input_transforms[("universe", "box")] = wMo
    vs.set_reference_weighted_trajectory([point("panda_hand_tcp_vs", oMf, w_pose)] * (T + 1))
    fp.set_reference_weighted_trajectory([point("panda_hand_tcp", wMf, w_pose)] * (T + 1))
    np.testing.assert_allclose(vs._ref_tile, fp._ref_tile, rtol=0, atol=1e-15)
    tcp_id = rm.robot_model.getFrameId("panda_hand_tcp")
    assert set(np.unique(vs._frames)) <= {-1, tcp_id} and vs.problem.running[2].frame == tcp_id  # the row's own frame: robot_frame
    x0 = np.concatenate([PANDA_Q0, np.zeros(7)])
    for ocp in (vs, fp):
        ocp.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    np.testing.assert_allclose(np.array(vs.ocp_results.states), np.array(fp.ocp_results.states), rtol=0, atol=1e-12)
    # and it is the checker's solution for the composed target
    o = Oracle(rm.table, vs.problem, 1)
    xs_o, us_o, K_o, st_o = o.solve(vs._ref_tile, vs._frames, x0[None], np.tile(x0, (1, T + 1, 1)), np.zeros((1, T, 7)), 20)
    assert vs.debug_data.nb_iter == st_o["iter"][0]
    np.testing.assert_allclose(np.array(vs.ocp_results.states), xs_o[0], rtol=1e-8, atol=1e-9)
    # a new object pose moves the target without touching the trajectory point
    vs.input_transforms[("universe", "box")] = se3.SE3(np.eye(3), np.array([0.0, 0.0, 0.1])) * wMo
    before = vs._ref_tile.copy()
    vs.set_reference_weighted_trajectory([point("panda_hand_tcp_vs", oMf, w_pose)] * (T + 1))
    assert not np.array_equal(before, vs._ref_tile)


def test_max_solve_time_caps_the_sqp_loop(hip_backend):
    """OCPParamsBaseCroco.max_solve_time (ocp_base_croco.py:70-71, 166-171; ROS default 0.1 s): a solve that would need its 100
    iterations stops at the wall-clock cap with problem_solved = False, fewer iterations than the cap, finite gains; with
    use_iteration_limits_and_timeout = False neither limit applies."""
    T, dt = 200, 0.05
    rm = panda_robot_models(0.1)
    seq = DTFactorsNSeq(factors=[1], n_steps=[T])

    def make(max_time):
        params = OCPParamsBaseCroco(dt=dt, horizon_size=T, dt_factor_n_seq=seq, solver_iters=100, callbacks=False, max_solve_time=max_time,
                                    termination_tolerance=1e-9)
        ocp = OCPCrocoGeneric(rm, params, OCPCrocoGeneric.get_default_yaml_file("ocp_goal_reaching.yaml"))
        pt = WeightedTrajectoryPoint(
            TrajectoryPoint(robot_configuration=np.zeros(7), robot_velocity=np.zeros(7), robot_effort=np.zeros(7),
                            end_effector_poses={"panda_hand_tcp": se3.SE3(np.eye(3), np.array([0.5, 0.2, 0.5]))}),
            TrajectoryPointWeights(w_robot_configuration=0.01 * np.ones(7), w_robot_velocity=0.01 * np.ones(7), w_robot_effort=1e-4 * np.ones(7),
                                   w_end_effector_poses={"panda_hand_tcp": 1e3 * np.ones(6)}))
        ocp.set_reference_weighted_trajectory([pt] * (T + 1))
        return ocp

    x0 = np.zeros(14)
    free = make(None)
    free.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    n_free = free.debug_data.nb_iter
    assert n_free >= 30, "the uncapped solve must be long enough for the cap to bite"
    import time
    t0 = time.perf_counter()
    free.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    t_free = time.perf_counter() - t0
    capped = make(t_free / 6.0)
    t0 = time.perf_counter()
    capped.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    t_cap = time.perf_counter() - t0
    dd = capped.debug_data
    assert not dd.problem_solved and 1 <= dd.nb_iter < n_free, (dd.nb_iter, n_free)
    assert t_cap < 0.7 * t_free
    K = np.array(capped.ocp_results.ricatti_gains)
    assert np.isfinite(K).all() and np.abs(K).max() > 0 and np.isfinite(np.array(capped.ocp_results.states)).all()
    # the iterate it stopped at is the checker's iterate after the same number of iterations
    o = Oracle(rm.table, capped.problem, 1)
    xs_o, us_o, K_o, st_o = o.solve(capped._ref_tile, capped._frames, x0[None], np.tile(x0, (1, T + 1, 1)), np.zeros((1, T, 7)), dd.nb_iter)
    np.testing.assert_allclose(np.array(capped.ocp_results.states), xs_o[0], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(K, K_o[0], rtol=1e-4, atol=1e-5)
    # without limits the cap is ignored
    capped.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T, use_iteration_limits_and_timeout=False)
    assert capped.debug_data.nb_iter >= n_free


def test_config3_as_written_cartesian_references_with_collision_constraint(hip_backend):
    """BASELINE.json configs[2]: sine_wave_cartesian_space references (IK of the batch), collision-avoidance costs,
    distance >= 1 cm (lower 0.01) constraint, max_qp_iters 100, T = 200, B = 256: two resident MPC steps; three instances
    against the checker, the whole batch through properties (finite, constraint respected where the solver converged)."""
    B, T, dt, steps = 256, 200, 0.01, 2
    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.49, 0.222, 0.487), obstacle_radius=0.06, obstacle_length=0.0)
    tcp = table.frame_id("panda_hand_tcp")
    running, terminal = workloads.collision_avoidance_rows(table, tcp, alpha=1e-4)
    fa, fb = table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle")
    con = [_abi.ConstraintSpec(_abi.RES_COLLISION, lower=0.01, upper=np.inf, frame=fa, frame_b=fb, name="collision")]
    po = _abi.PackedOcp(7, [dt] * T, running, terminal, termination_tolerance=1e-3, max_qp_iters=100, running_constraints=con)
    hip = hip_backend.HipOcp(table, po, B)
    n_points = T + steps + 2
    q0, amp, puls = workloads.cartesian_sine_batch_params(B, lower=table.lower_position_limit, upper=table.upper_position_limit)
    gq, gdq, gddq = workloads.cartesian_sine_batch_arrays(hip, tcp, n_points, dt, q0, amp, puls)
    w = workloads.SINE_WEIGHTS
    hip.generic_trajectory(gq, gdq, gddq, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
    # the checker follows three instances: the closest approach of the batch and two others
    o_all = Oracle(table, po, 1)
    chosen = [0, B // 2, B - 1]
    oc = Oracle(table, po, len(chosen))
    refs = []
    xs_prev = us_prev = None
    for k in range(steps):
        hip.mpc_step(k, 10, first=(k == 0))
        xs_h, us_h, K_h, st_h = hip.download()
        assert np.isfinite(xs_h).all() and np.isfinite(us_h).all() and np.isfinite(K_h).all()
        # reference tile of the window, rebuilt on the host for the chosen instances
        ref = po.new_ref_tile(len(chosen))
        for t in range(T + 1):
            qk, dqk, ddqk = gq[chosen, k + t], gdq[chosen, k + t], gddq[chosen, k + t]
            uk = oc.rnea(qk, dqk, ddqk).reshape(len(chosen), 7)
            pose = oc.frame_placement(tcp, qk)
            rows, offs = (po.terminal, po.terminal_offsets) if t == T else (po.running, po.running_offsets)
            for r, off in zip(rows, offs):
                seg = ref[:, t, off:]
                seg[:, 0] = r.weight
                if r.kind == _abi.RES_STATE:
                    seg[:, 1:15] = np.concatenate([qk, dqk], 1)
                    seg[:, 15:22], seg[:, 22:29] = w["w_q"], w["w_qdot"]
                elif r.kind == _abi.RES_CONTROL:
                    seg[:, 1:8], seg[:, 8:15] = uk, w["w_effort"]
                elif r.kind == _abi.RES_FRAME_PLACEMENT:
                    seg[:, 1:13], seg[:, 13:19] = pose, w["w_pose"]
                else:
                    seg[:, 1:] = 0.0
        if k == 0:
            xs_ws = np.stack([np.concatenate([gq[chosen, t], gdq[chosen, t]], 1) for t in range(T + 1)], 1)
            us_ws = np.stack([oc.rnea(gq[chosen, t], gdq[chosen, t], gddq[chosen, t]).reshape(len(chosen), 7) for t in range(T)], 1)
            x0 = xs_ws[:, 0].copy()
        else:
            x0 = xs_prev[:, 1].copy()
            xs_ws, us_ws = oc.shift_warmstart(xs_prev, us_prev)
        xs_o, us_o, K_o, st_o = oc.solve(ref, None, x0, xs_ws, us_ws, 10)
        np.testing.assert_array_equal(st_h["iter"][chosen], st_o["iter"])
        np.testing.assert_array_equal(st_h["qp_iters"][chosen], st_o["qp_iters"])
        np.testing.assert_allclose(xs_h[chosen], xs_o, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(us_h[chosen], us_o, rtol=1e-5, atol=1e-5)
        xs_prev, us_prev = xs_o, us_o
        # the distance constraint along the horizon of every instance the solver reports as solved
        d = np.stack([o_all.node_constraints(False, xs_h[b, t], us_h[b, t])[0][0] for b in np.flatnonzero(st_h["solved"])[:16] for t in range(1, T, 10)])
        assert d.min() >= 0.01 - 2e-3
        assert st_h["solved"].mean() > 0.75  # the first steps start from the reference; bench.py reports ~ 0.99 in steady state
    hip.close()
