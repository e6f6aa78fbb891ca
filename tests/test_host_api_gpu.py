"""The reference's Python surface on the HIP path (needs an MI355X): OCPCrocoGeneric from YAML,
WarmStartReference / WarmStartShiftPreviousSolution, MPC.run -- written like the reference's own
tests (tests/test_ocp_croco_generic.py, test_mpc_unicycle.py, test_warm_start_*.py)."""
import numpy as np
import pytest

from agimus_controller_amd import _abi, se3
from agimus_controller_amd.factory.robot_model import panda_robot_models
from agimus_controller_amd.mpc import MPC
from agimus_controller_amd.ocp.ocp_croco_generic import OCPCrocoGeneric
from agimus_controller_amd.ocp_param_base import DTFactorsNSeq, OCPParamsBaseCroco
from agimus_controller_amd.trajectories.sine_wave_configuration_space import SinusWaveConfigurationSpace
from agimus_controller_amd.trajectories.sine_wave_params import SinWaveParams
from agimus_controller_amd.trajectory import TrajectoryBuffer, TrajectoryPoint, TrajectoryPointWeights, WeightedTrajectoryPoint
from agimus_controller_amd.warm_start_reference import WarmStartReference
from agimus_controller_amd.warm_start_shift_previous_solution import WarmStartShiftPreviousSolution
from agimus_controller_amd.workloads import PANDA_Q0
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def make_ocp(T, dt, iters=100, factors=None, n_steps=None, yaml_name="ocp_goal_reaching.yaml", **kw):
    rm = panda_robot_models(0.1)
    seq = DTFactorsNSeq(factors=factors or [1], n_steps=n_steps or [T])
    params = OCPParamsBaseCroco(dt=dt, horizon_size=T, dt_factor_n_seq=seq, solver_iters=iters, callbacks=False, **kw)
    return rm, params, OCPCrocoGeneric(rm, params, OCPCrocoGeneric.get_default_yaml_file(yaml_name))


def test_ocp_solution_reaches_the_goal(hip_backend):
    """tests/test_ocp_croco_generic.py:161-221: Panda, T = 200, dt = 0.05, 100 iterations,
    end-effector within 0.05 m (places=1) of (0.5, 0.2, 0.5)."""
    T = 200
    rm, params, ocp = make_ocp(T, 0.05)
    q0 = np.zeros(7)
    ee_pose = se3.SE3(np.eye(3), np.array([0.5, 0.2, 0.5]))
    pt = WeightedTrajectoryPoint(
        TrajectoryPoint(robot_configuration=q0, robot_velocity=np.zeros(7), robot_effort=np.zeros(7),
                        end_effector_poses={"panda_hand_tcp": ee_pose}),
        TrajectoryPointWeights(w_robot_configuration=0.01 * np.ones(7), w_robot_velocity=0.01 * np.ones(7),
                               w_robot_effort=0.0001 * np.ones(7), w_end_effector_poses={"panda_hand_tcp": 1e3 * np.ones(6)}))
    ocp.set_reference_weighted_trajectory([pt] * (T + 1))
    x0 = np.concatenate([q0, np.zeros(7)])
    ocp.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    res = ocp.ocp_results
    assert len(res.states) == T + 1 and len(res.feed_forward_terms) == T and res.ricatti_gains[0].shape == (7, 14)
    tcp = rm.robot_model.getFrameId("panda_hand_tcp")
    final = ocp._hip.frame_placement(tcp, res.states[-1][:7])[0]
    assert np.linalg.norm(final[9:] - ee_pose.translation) < 0.05
    # and it is the oracle's answer
    o = Oracle(rm.table, ocp.problem, 1)
    xs_o, us_o, K_o, st_o = o.solve(ocp._ref_tile, ocp._frames, x0[None], np.tile(x0, (1, T + 1, 1)), np.zeros((1, T, 7)), 100)
    assert ocp.debug_data.nb_iter == st_o["iter"][0] and ocp.debug_data.problem_solved == bool(st_o["solved"][0])
    # a long solve from the singular zero posture with pose weight 1e3 amplifies round-off: 1e-6 here
    np.testing.assert_allclose(np.array(res.states), xs_o[0], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ocp.debug_data.kkt_norm, st_o["kkt"][0], rtol=1e-5)


def test_update_semantics_references_weights_and_frames(hip_backend):
    """update() chain (ocp_croco_generic.py:158-160,174-176,203-210,571-574): refs, weights and
    frame id reach the kernels; default unit weights before the first update."""
    T = 5
    rm, params, ocp = make_ocp(T, 0.01)
    po = ocp.problem
    wi, ref, aw = po.row_view(ocp._ref_tile, False, 1)
    assert np.all(aw == 1.0) and np.all(ref == 0.0) and np.all(wi == 1.0)
    rng = np.random.default_rng(0)
    pts = []
    for t in range(T + 1):
        name = "panda_hand_tcp" if t % 2 == 0 else "panda_link5"
        pts.append(WeightedTrajectoryPoint(
            TrajectoryPoint(robot_configuration=rng.random(7), robot_velocity=rng.random(7), robot_effort=rng.random(7),
                            end_effector_poses={name: se3.SE3ToXYZQUAT(se3.SE3.Random(rng))}),
            TrajectoryPointWeights(w_robot_configuration=np.array([0.5]), w_robot_velocity=10 * np.ones(7),
                                   w_robot_effort=0.5 * np.ones(7), w_end_effector_poses={name: rng.random(6)})))
    ocp.set_reference_weighted_trajectory(pts)
    np.testing.assert_array_equal(ref[0, 2], pts[2].point.robot_state)
    np.testing.assert_array_equal(aw[0, 3], np.concatenate([0.5 * np.ones(7), 10 * np.ones(7)]))
    assert ocp._frames[0, 1, 2] == rm.robot_model.getFrameId("panda_link5")
    assert ocp._frames[0, T, 1] == rm.robot_model.getFrameId("panda_link5" if T % 2 else "panda_hand_tcp")
    with pytest.raises(AssertionError):
        ocp.set_reference_weighted_trajectory(pts[:-1])
    # the tiles the device computes with these references equal the oracle's
    xs = rng.normal(0, 0.3, (1, T + 1, 14))
    us = rng.normal(0, 1.0, (1, T, 7))
    ocp._hip.upload_warmstart(xs, us)
    got = ocp._hip.calc_diff()
    want = Oracle(rm.table, po, 1).calc_diff(ocp._ref_tile, ocp._frames, xs, us)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)


def test_integrate_matches_first_node(hip_backend):
    rm, params, ocp = make_ocp(4, 0.02)
    rng = np.random.default_rng(1)
    x, u = rng.normal(0, 0.5, 14), rng.normal(0, 2.0, 7)
    xn = ocp.integrate(x, u)
    np.testing.assert_allclose(xn, Oracle(rm.table, ocp.problem, 1).integrate(x, u), rtol=1e-12)
    # semi-implicit Euler
    np.testing.assert_allclose(xn[:7], x[:7] + 0.02 * xn[7:], rtol=1e-13)


def test_warm_start_reference_layout(hip_backend):
    """tests/test_warm_start_reference.py:48-75: x_init = [x_meas, ref[1:]], u_init = RNEA of [x_meas, ref[1:-1]]."""
    rm, params, ocp = make_ocp(6, 0.01)
    ws = WarmStartReference()
    with pytest.raises(AssertionError):
        ws.generate(TrajectoryPoint(robot_configuration=np.zeros(7), robot_velocity=np.zeros(7)), [])
    ws.setup(ocp)
    rng = np.random.default_rng(2)
    mk = lambda: TrajectoryPoint(robot_configuration=rng.random(7), robot_velocity=rng.random(7), robot_acceleration=rng.random(7))  # noqa: E731
    init, refs = mk(), [mk() for _ in range(7)]
    x0, x_init, u_init = ws.generate(init, refs)
    np.testing.assert_array_equal(x0, init.robot_state)
    assert len(x_init) == 7 and len(u_init) == 6
    np.testing.assert_array_equal(x_init[0], init.robot_state)
    np.testing.assert_array_equal(x_init[3], refs[3].robot_state)
    o = Oracle(rm.table, ocp.problem, 1)
    np.testing.assert_allclose(u_init[0], o.rnea(init.robot_configuration, init.robot_velocity, init.robot_acceleration), rtol=1e-12)
    np.testing.assert_allclose(u_init[5], o.rnea(refs[5].robot_configuration, refs[5].robot_velocity, refs[5].robot_acceleration), rtol=1e-12)


def test_warm_start_shift_previous_solution(hip_backend):
    """tests/test_warm_start_shift_previous_reference.py:55-117 with timesteps (0.1, 0.1, 0.2)."""
    rm, params, ocp = make_ocp(3, 0.1, factors=[1, 2], n_steps=[2, 1])
    assert params.timesteps == (0.1, 0.1, 0.2)
    ws = WarmStartShiftPreviousSolution()
    with pytest.raises(AssertionError):
        ws.generate(None, None)
    ws.setup(rm, params)
    rng = np.random.default_rng(3)
    controls = [rng.random(7) for _ in range(3)]
    from agimus_controller_amd.mpc_data import OCPResults
    states = [np.zeros(14)]
    for i, h in enumerate(params.timesteps):
        po = _abi.PackedOcp(7, [h], [], [])
        states.append(hip_backend.HipOcp(rm.table, po, 1).integrate(states[i], controls[i])[0])
    ws.update_previous_solution(OCPResults(states=[s.copy() for s in states], ricatti_gains=[], feed_forward_terms=[c.copy() for c in controls]))
    init = TrajectoryPoint(robot_configuration=-10 * np.ones(7), robot_velocity=100 * np.ones(7))
    x0, x_init, u_init = ws.generate(init, [])
    assert len(x_init) == 4 and len(u_init) == 3
    np.testing.assert_array_equal(x0[:7], init.robot_configuration)
    np.testing.assert_array_equal(x0[7:], init.robot_velocity)
    for i in range(1, 3):
        np.testing.assert_array_equal(u_init[i - 1], controls[i])
        np.testing.assert_allclose(x_init[i], ws._integrate(states[i], controls[i]), rtol=0, atol=1e-15)


def test_mpc_run_sine_wave_closed_loop(hip_backend):
    """MPC.run (mpc.py:32-66) on the sine-wave reference: buffer / horizon alignment as in
    tests/test_mpc_unicycle.py:197-257 (xs[0] == x0, buffer shrinks by one per step), debug timers
    filled, tracking error stays small; warm start by reference first, then shift."""
    T, dt = 20, 0.01
    rm, params, ocp = make_ocp(T, dt, iters=10)
    traj = SinusWaveConfigurationSpace(SinWaveParams(amplitude=[0.1] * 7, period=[4.0] * 7, scale_duration=[0.2] * 7),
                                       "panda_hand_tcp", np.array([1.0]), np.array([0.1]), np.array([1e-6]),
                                       np.array([3e-4]), np.array([0.1]))
    traj.initialize(rm.robot_model, PANDA_Q0, ocp)
    buffer = TrajectoryBuffer(params.dt_factor_n_seq)
    mpc = MPC()
    ws_ref = WarmStartReference()
    ws_ref.setup(ocp)
    mpc.setup(ocp, ws_ref, buffer)
    n_pts = T + 8
    pts = [traj.get_traj_point_at_t(k * dt) for k in range(n_pts)]
    for k, p in enumerate(pts):
        p.point.id = k
    assert mpc.run(pts[0].point, 0) is None  # not enough points buffered yet
    mpc.append_trajectory_points(pts)
    state = TrajectoryPoint(robot_configuration=pts[0].point.robot_configuration.copy(), robot_velocity=pts[0].point.robot_velocity.copy(),
                            robot_acceleration=pts[0].point.robot_acceleration.copy(), time_ns=0)
    res = mpc.run(state, 0)
    assert len(buffer) == n_pts - 1 and mpc.mpc_debug_data.reference_id == 0
    np.testing.assert_array_equal(res.states[0], state.robot_state)
    assert mpc.mpc_debug_data.duration_ocp_solve_ns > 0 and mpc.mpc_debug_data.ocp.problem_solved
    ws_shift = WarmStartShiftPreviousSolution()
    ws_shift.setup(rm, params, ocp)
    ws_shift.update_previous_solution(res)
    mpc.setup(ocp, ws_shift, buffer)
    for step in range(1, 6):
        state = TrajectoryPoint(robot_configuration=res.states[1][:7].copy(), robot_velocity=res.states[1][7:].copy(), time_ns=step)
        res = mpc.run(state, step)
        assert mpc.mpc_debug_data.reference_id == step and len(buffer) == n_pts - 1 - step
        ref_next = pts[step + 1].point.robot_state
        assert np.abs(res.states[1][:7] - ref_next[:7]).max() < 5e-3
        assert mpc.mpc_debug_data.ocp.nb_iter <= 10
    K0 = res.ricatti_gains[0]
    assert K0.shape == (7, 14) and np.isfinite(K0).all()
    # MPC.integrate: one Euler step of the node-0 model
    before = state.robot_state.copy()
    out = mpc.integrate(state, res.feed_forward_terms[0])
    np.testing.assert_allclose(out.robot_state, ocp.integrate(before, res.feed_forward_terms[0]), rtol=1e-13)


def test_rolling_buffer_mode_equals_full_update(hip_backend):
    """expect_rolling_buffer (ocp_croco_generic.py:865-881): circularAppend + last-node update must
    give the same references as a full update when the horizon slides by one point."""
    T = 6
    rm = panda_robot_models(0.1)
    params = OCPParamsBaseCroco(dt=0.01, horizon_size=T, dt_factor_n_seq=DTFactorsNSeq([1], [T]), solver_iters=5)
    y = OCPCrocoGeneric.get_default_yaml_file("ocp_goal_reaching.yaml")
    full, roll = OCPCrocoGeneric(rm, params, y), OCPCrocoGeneric(rm, params, y, expect_rolling_buffer=True)
    rng = np.random.default_rng(5)
    mk = lambda: WeightedTrajectoryPoint(  # noqa: E731
        TrajectoryPoint(robot_configuration=rng.random(7), robot_velocity=rng.random(7), robot_effort=rng.random(7),
                        end_effector_poses={"panda_hand_tcp": se3.SE3.Random(rng)}),
        TrajectoryPointWeights(w_robot_configuration=rng.random(7), w_robot_velocity=rng.random(7), w_robot_effort=rng.random(7),
                               w_end_effector_poses={"panda_hand_tcp": rng.random(6)}))
    pts = [mk() for _ in range(T + 4)]
    for k in range(3):
        window = pts[k:k + T + 1]
        full.set_reference_weighted_trajectory(window)
        roll.set_reference_weighted_trajectory(window)
        np.testing.assert_array_equal(full._ref_tile, roll._ref_tile)


def test_debug_references_and_residuals(hip_backend):
    import io
    import yaml
    T = 4
    rm = panda_robot_models(0.1)
    params = OCPParamsBaseCroco(dt=0.01, horizon_size=T, dt_factor_n_seq=DTFactorsNSeq([1], [T]), solver_iters=3)
    doc = yaml.safe_load(open(OCPCrocoGeneric.get_default_yaml_file("ocp_goal_reaching.yaml")))
    doc["running_model"]["differential"]["costs"][2]["publish_residual"] = True
    ocp = OCPCrocoGeneric(rm, params, io.StringIO(yaml.safe_dump(doc)))
    assert [n for n, _ in ocp.debug_data.references] == ["control_reg", "state_reg", "goal_tracking"]
    assert [n for n, _ in ocp.debug_data.residuals] == ["goal_tracking"]
    pose = se3.SE3(np.eye(3), np.array([0.4, 0.1, 0.5]))
    pt = WeightedTrajectoryPoint(
        TrajectoryPoint(robot_configuration=PANDA_Q0, robot_velocity=np.zeros(7), robot_effort=np.zeros(7), end_effector_poses={"panda_hand_tcp": pose}),
        TrajectoryPointWeights(w_robot_configuration=np.ones(7), w_robot_velocity=np.ones(7), w_robot_effort=1e-3 * np.ones(7),
                               w_end_effector_poses={"panda_hand_tcp": np.ones(6)}))
    ocp.set_reference_weighted_trajectory([pt] * (T + 1))
    x0 = np.concatenate([PANDA_Q0, np.zeros(7)])
    ocp.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    refs = dict(ocp.debug_data.references)
    np.testing.assert_allclose(refs["goal_tracking"], se3.SE3ToXYZQUAT(pose), atol=1e-15)
    np.testing.assert_array_equal(refs["state_reg"], x0)
    name, r = ocp.debug_data.residuals[0]
    assert r.shape == (T, 6)
    o = Oracle(rm.table, ocp.problem, 1)
    res = ocp.ocp_results
    _, _, rr = o.node_calc(False, 0.01, res.states[2], res.feed_forward_terms[2], ocp._ref_tile[0, 2], ocp._frames[0, 2])
    np.testing.assert_allclose(r[2], rr[21:27], rtol=1e-10, atol=1e-13)
    assert ocp.n_controls == T and ocp.dt == 0.01 and ocp.input_transforms == {}
    with pytest.warns(DeprecationWarning):
        assert ocp.horizon_size == T
    with pytest.raises(RuntimeError, match="Unknown geometry"):
        ocp.update_geometry_placement("obstacle", pose)


def test_collision_avoidance_yaml_end_to_end(hip_backend):
    """ocp_traj_tracking_collision_avoidance.yaml through the class surface: QuadExp distance cost,
    distance >= 1 cm constraint (ADMM), w_collision_avoidance as the item weight
    (ocp_croco_generic.py:713-719) and update_geometry_placement for a moving obstacle."""
    from agimus_controller_amd.factory import robot_tables as rt
    from agimus_controller_amd.factory.robot_model import RobotModelParameters, RobotModels

    T, dt = 20, 0.02
    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.42, 0.05, 0.55), obstacle_radius=0.06, obstacle_length=0.2)
    rm = RobotModels(RobotModelParameters(table=table, armature=table.armature, q0=PANDA_Q0,
                                          collision_pairs=[("panda_link7_capsule_0", "obstacle")]))
    seq = DTFactorsNSeq(factors=[1], n_steps=[T])
    params = OCPParamsBaseCroco(dt=dt, horizon_size=T, dt_factor_n_seq=seq, solver_iters=30, qp_iters=100, callbacks=False)
    ocp = OCPCrocoGeneric(rm, params, OCPCrocoGeneric.get_default_yaml_file("ocp_traj_tracking_collision_avoidance.yaml"))
    assert ocp.problem.desc.n_running_constraints == 1 and ocp.problem.desc.n_terminal_constraints == 0
    tcp = rm.robot_model.getFrameId("panda_hand_tcp")
    start = ocp._hip.frame_placement(tcp, PANDA_Q0)[0]
    goal = se3.SE3(start[:9].reshape(3, 3), start[9:] + np.array([0.12, 0.05, 0.05]))
    pt = WeightedTrajectoryPoint(
        TrajectoryPoint(robot_configuration=PANDA_Q0, robot_velocity=np.zeros(7), robot_effort=np.zeros(7),
                        end_effector_poses={"panda_hand_tcp": goal}),
        TrajectoryPointWeights(w_robot_configuration=0.01 * np.ones(7), w_robot_velocity=0.1 * np.ones(7),
                               w_robot_effort=1e-4 * np.ones(7), w_end_effector_poses={"panda_hand_tcp": 50.0 * np.ones(6)},
                               w_collision_avoidance=2.0))
    ocp.set_reference_weighted_trajectory([pt] * (T + 1))
    x0 = np.concatenate([PANDA_Q0, np.zeros(7)])
    ocp.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    res = ocp.ocp_results
    xs = np.array(res.states)
    # the checker agrees on the whole constrained solve
    o = Oracle(rm.table, ocp.problem, 1)
    xs_o, us_o, K_o, st_o = o.solve(ocp._ref_tile, ocp._frames, x0[None], np.tile(x0, (1, T + 1, 1)), np.zeros((1, T, 7)), 30)
    assert ocp.debug_data.nb_iter == st_o["iter"][0]
    np.testing.assert_allclose(xs, xs_o[0], rtol=1e-5, atol=1e-6)
    assert ocp.debug_data.nb_qp_iter == st_o["qp_iters"][0]
    # the pair stays apart along the running nodes
    for t in range(1, T):
        g, _, _ = o.node_constraints(False, xs[t], res.feed_forward_terms[t])
        assert g[0] >= 0.01 - 2e-3
    # moving the obstacle away: unknown names raise, known ones change the distances
    with pytest.raises(RuntimeError, match="Unknown geometry name"):
        ocp.update_geometry_placement("no_such_geometry", se3.SE3(np.eye(3), np.zeros(3)))
    d0 = ocp._hip.residuals(3)[0, 1, 0]
    ocp.update_geometry_placement("obstacle", se3.SE3(rt._ry(np.pi / 2), np.array([1.535, 0.0, 0.43])))
    assert ocp._hip.residuals(3)[0, 1, 0] > d0 + 0.3


def test_generic_trajectory_host_and_resident(hip_backend):
    """GenericTrajectory (tests/test_generic_trajectory.py:140-163 upstream: smooth random accelerations
    integrated to dq, q): the host class and the device-resident generator give the same samples, and
    an MPC step on the resident window equals one on the host-built reference list."""
    from agimus_controller_amd import backend, workloads
    from agimus_controller_amd.trajectories.generic_trajectory import GenericTrajectory

    T, dt, N = 12, 0.01, 40
    rm, params, ocp = make_ocp(T, dt, iters=10, yaml_name="ocp_regulation.yaml")
    rng = np.random.default_rng(0)
    ddq = 0.5 * (rng.random((N, 7)) - 0.5)
    dq = np.zeros((N, 7))
    q = np.tile(PANDA_Q0, (N, 1))
    for i in range(N - 1):
        dq[i + 1] = dq[i] + ddq[i] * dt
        q[i + 1] = q[i] + dq[i + 1] * dt
    w = dict(w_q=3.0, w_qdot=0.12, w_effort=8e-4, w_pose=0.0)  # pick-and-place trajectory_weigths_params.yaml:4-9
    gen = GenericTrajectory("panda_hand_tcp", w["w_q"] * np.ones(7), w["w_qdot"] * np.ones(7), np.zeros(7), w["w_effort"] * np.ones(7),
                            w["w_pose"] * np.ones(6), 0.0)
    gen.initialize(rm.robot_model, PANDA_Q0, ocp)
    gen.add_trajectory(gen.build_trajectory_from_q_dq_ddq_arrays(list(q), list(dq), list(ddq)))
    pts = [gen.get_traj_point_at_t(k * dt) for k in range(N)]
    assert gen.trajectory_is_done and len(pts) == N
    # resident generator on a second handle with the same row tables
    hb = backend.HipOcp(rm.table, ocp.problem, 1)
    tcp = rm.robot_model.getFrameId("panda_hand_tcp")
    hb.generic_trajectory(q[None], dq[None], ddq[None], w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
    for k in (0, 7, N - 1):
        qk, vk, ak, uk, pose = hb.traj_point(k)
        np.testing.assert_allclose(qk[0], pts[k].point.robot_configuration, atol=1e-15)
        np.testing.assert_allclose(uk[0], pts[k].point.robot_effort, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(pose[0, 9:], pts[k].point.end_effector_poses["panda_hand_tcp"].translation, atol=1e-13)
    # one MPC step both ways
    x0 = np.concatenate([q[0], dq[0]])
    ocp.set_reference_weighted_trajectory(pts[: T + 1])
    xs_ws = [np.concatenate([p.point.robot_configuration, p.point.robot_velocity]) for p in pts[: T + 1]]
    us_ws = [p.point.robot_effort for p in pts[:T]]
    ocp.solve(x0, xs_ws, us_ws)
    hb.mpc_step(0, 10, first=True)
    xs_r, us_r, K_r, st_r = hb.download()
    np.testing.assert_allclose(xs_r[0], np.array(ocp.ocp_results.states), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(us_r[0], np.array(ocp.ocp_results.feed_forward_terms), rtol=1e-8, atol=1e-9)
    hb.close()


def test_resident_window_with_nonuniform_horizon_indexes(hip_backend):
    """dt factors (ocp_param_base.py:67-81) + TrajectoryBuffer.horizon_indexes (trajectory.py:195-216):
    the resident window gathered at [0,1,2,4,6,9,...] equals the host-built horizon, for the first
    solve and after a shift."""
    from agimus_controller_amd import backend, workloads

    factors, n_steps = [1, 2, 3], [2, 2, 3]
    T, dt = sum(n_steps), 0.01
    rm, params, ocp = make_ocp(T, dt, iters=20, factors=factors, n_steps=n_steps)
    buf = TrajectoryBuffer(DTFactorsNSeq(factors=factors, n_steps=n_steps))
    idx = buf.compute_horizon_indexes()
    assert idx == [0, 1, 2, 4, 6, 9, 12, 15]
    sp = SinWaveParams(amplitude=np.full(7, 0.1), period=np.full(7, 4.0), scale_duration=np.full(7, 0.2))
    w = workloads.SINE_WEIGHTS
    gen = SinusWaveConfigurationSpace(sp, "panda_hand_tcp", w["w_q"] * np.ones(7), w["w_qdot"] * np.ones(7), 1e-6 * np.ones(7),
                                      w["w_effort"] * np.ones(7), w["w_pose"] * np.ones(6))
    gen.initialize(rm.robot_model, PANDA_Q0, ocp)
    n_points = idx[-1] + 4
    pts = [gen.get_traj_point_at_t(k * dt) for k in range(n_points)]
    hb = backend.HipOcp(rm.table, ocp.problem, 1)
    tcp = rm.robot_model.getFrameId("panda_hand_tcp")
    hb.sine_trajectory(n_points, dt, PANDA_Q0, sp.amplitude, sp.pulsation, sp.scale_duration, 0.0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
    hb.set_horizon_indexes(idx)
    for k0 in (0, 1):
        horizon = [pts[k0 + i] for i in idx]
        ocp.set_reference_weighted_trajectory(horizon)
        xs_ws = [np.concatenate([p.point.robot_configuration, p.point.robot_velocity]) for p in horizon]
        us_ws = [p.point.robot_effort for p in horizon[:-1]]
        ocp.solve(xs_ws[0], xs_ws, us_ws)
        hb.set_window(k0)
        hb.warmstart_from_reference()
        hb.solve_resident(20)
        xs_r, us_r, _, st_r = hb.download()
        assert st_r["iter"][0] == ocp.debug_data.nb_iter
        np.testing.assert_allclose(xs_r[0], np.array(ocp.ocp_results.states), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(us_r[0], np.array(ocp.ocp_results.feed_forward_terms), rtol=1e-8, atol=1e-9)
    with pytest.raises(backend.HipError, match="increasing"):
        hb.set_horizon_indexes([0, 2, 1] + list(range(3, T + 1)))
    hb.close()


def test_control_grav_and_frame_velocity_rows_from_yaml(hip_backend):
    """ResidualModelControlGrav + ResidualModelFrameVelocity through the YAML schema and the update chain
    (ocp_croco_generic.py:186-194, 360-395): the effort weights feed the gravity-compensated control
    row, the end-effector twist and its weights feed the velocity row; result equals the checker's."""
    import io

    yaml_text = """
running_model:
  class: IntegratedActionModelEuler
  differential:
    class: DifferentialActionModelFreeFwdDynamics
    costs:
      - {name: ctrl_grav, update: true, weight: 1.0, cost: {class: CostModelResidual, residual: {class: ResidualModelControlGrav},
         activation: {class: ActivationModelWeightedQuad, weights: 1.0}}}
      - {name: state_reg, update: true, weight: 1.0, cost: {class: CostModelResidual, residual: {class: ResidualModelState},
         activation: {class: ActivationModelWeightedQuad, weights: 1.0}}}
      - {name: ee_vel, update: true, weight: 1.0, cost: {class: CostModelResidual,
         residual: {class: ResidualModelFrameVelocity, id: panda_hand_tcp, reference_frame: LOCAL_WORLD_ALIGNED},
         activation: {class: ActivationModelWeightedQuad, weights: 1.0}}}
terminal_model:
  class: IntegratedActionModelEuler
  differential:
    class: DifferentialActionModelFreeFwdDynamics
    costs:
      - {name: state_reg, update: true, weight: 1.0, cost: {class: CostModelResidual, residual: {class: ResidualModelState},
         activation: {class: ActivationModelWeightedQuad, weights: 1.0}}}
"""
    T, dt = 12, 0.01
    rm = panda_robot_models(0.1)
    params = OCPParamsBaseCroco(dt=dt, horizon_size=T, dt_factor_n_seq=DTFactorsNSeq(factors=[1], n_steps=[T]), solver_iters=30, callbacks=False)
    ocp = OCPCrocoGeneric(rm, params, io.StringIO(yaml_text))
    rows = ocp.problem.running
    assert [r.kind for r in rows] == [_abi.RES_CONTROL_GRAV, _abi.RES_STATE, _abi.RES_FRAME_VELOCITY] and rows[2].frame_b == 2
    twist = np.array([0.05, -0.02, 0.03, 0.0, 0.0, 0.1])
    pt = WeightedTrajectoryPoint(
        TrajectoryPoint(robot_configuration=PANDA_Q0, robot_velocity=np.zeros(7), robot_effort=np.zeros(7),
                        end_effector_velocities={"panda_hand_tcp": twist}),
        TrajectoryPointWeights(w_robot_configuration=1.0 * np.ones(7), w_robot_velocity=0.1 * np.ones(7),
                               w_robot_effort=1e-3 * np.ones(7), w_end_effector_velocities={"panda_hand_tcp": 5.0 * np.ones(6)}))
    ocp.set_reference_weighted_trajectory([pt] * (T + 1))
    x0 = np.concatenate([PANDA_Q0, np.zeros(7)])
    ocp.solve(x0, [x0] * (T + 1), [np.zeros(7)] * T)
    o = Oracle(rm.table, ocp.problem, 1)
    xs_o, us_o, K_o, st_o = o.solve(ocp._ref_tile, ocp._frames, x0[None], np.tile(x0, (1, T + 1, 1)), np.zeros((1, T, 7)), 30)
    assert ocp.debug_data.nb_iter == st_o["iter"][0]
    np.testing.assert_allclose(np.array(ocp.ocp_results.states), xs_o[0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.array(ocp.ocp_results.feed_forward_terms), us_o[0], rtol=1e-6, atol=1e-8)
    # gravity compensation: the solution's torques stay near g(q) and the end effector picks up the twist
    g = ocp._hip.rnea(PANDA_Q0, np.zeros(7), np.zeros(7))[0]
    assert np.abs(np.array(ocp.ocp_results.feed_forward_terms)[0] - g).max() < 0.5 * np.abs(g).max()
