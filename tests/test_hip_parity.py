"""Parity of the HIP path against the oracle and the golden fixture (needs an MI355X).

Everything goes through the C ABI of libagimus_hip.so.  Tolerances: fp64, relative 1e-9 on
kernel outputs and on xs/us after a full multi-iteration solve, 1e-7 on the Riccati gains K
(they amplify round-off through Quu^-1), identical iteration counts."""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


MODELS = {
    "panda": lambda: rt.panda_table(),
    "chain4": lambda: rt.chain_table(4, seed=7),
    "chain6": lambda: rt.chain_table(6, seed=8),
    "pendulum": lambda: rt.pendulum_table(),
}


@pytest.mark.parametrize("name", list(MODELS))
def test_rigid_body_primitives(hip_backend, name):
    table = MODELS[name]()
    nv = table.nv
    frame = len(table.frame_names) - 1
    po, ref, x0, xs, us = workloads.random_goal_problem(table, 4, 0.01, 2, seed=1, frame=frame)
    h, o = hip_backend.HipOcp(table, po, 2), Oracle(table, po, 2)
    rng = np.random.default_rng(0)
    q, v, a = rng.uniform(-1.5, 1.5, (3, 33, nv))
    assert rel(h.rnea(q, v, a), o.rnea(q, v, a).reshape(33, nv)) < 1e-12
    assert rel(h.frame_placement(frame, q), o.frame_placement(frame, q)) < 1e-12
    x = np.concatenate([q, v], axis=1)
    assert rel(h.integrate(x, 3 * a), o.integrate(x, 3 * a).reshape(33, 2 * nv)) < 1e-11


@pytest.mark.parametrize("name,rows", [("panda", "goal"), ("panda", "reg"), ("chain4", "goal"), ("chain6", "goal"), ("pendulum", "goal")])
def test_derivative_tiles(hip_backend, name, rows):
    table = MODELS[name]()
    frame = len(table.frame_names) - 1
    B, T = 6, 11
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=3, frame=frame, rows=rows,
                                                        timesteps=[0.01] * 6 + [0.02] * 3 + [0.04] * 2)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    got, want = h.calc_diff(), o.calc_diff(ref, None, xs, us)
    for field, s in _abi.tile_slices(table.nv).items():
        scale = max(np.abs(want[..., s]).max(), 1e-300)
        assert np.abs(got[..., s] - want[..., s]).max() <= 1e-11 * scale + 1e-14, field


def test_frame_id_override_per_node(hip_backend, panda):
    """obj.id = get_frame_id(...) at every update (ocp_croco_generic.py:208): per-node frame ids."""
    tcp, l5 = panda.frame_id("panda_hand_tcp"), panda.frame_id("panda_link5")
    B, T = 2, 5
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=5, frame=tcp)
    frames = po.default_frames(B)
    frames[:, ::2, 2] = l5  # running row 2 = placement
    frames[:, T, 1] = l5  # terminal row 1 = placement
    h, o = hip_backend.HipOcp(panda, po, B), Oracle(panda, po, B)
    h.set_refs(ref, frames)
    h.upload_warmstart(xs, us)
    got, want = h.calc_diff(), o.calc_diff(ref, frames, xs, us)
    assert rel(got, want) < 1e-11
    h.set_refs(ref)
    assert rel(h.calc_diff(), want) > 1e-3  # and it matters


def test_direction_kernels_against_oracle(hip_backend, panda):
    """Production K1 (QP tiles) + K2 (Riccati/forward) + K4 prologue (du, KKT) + exit path (gains)
    at a fixed point, against the oracle's direction on the oracle's own tiles."""
    tcp = panda.frame_id("panda_hand_tcp")
    B, T = 5, 30
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=6, frame=tcp)
    h, o = hip_backend.HipOcp(panda, po, B), Oracle(panda, po, B)
    xs[:, 0] = x0
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    K, k, dx, du, kkt = h.direction()
    Ko, ko, dxo, duo, kkto = o.direction(o.calc_diff(ref, None, xs, us))
    assert rel(dx, dxo) < 1e-9 and rel(du, duo) < 1e-9
    assert rel(K, Ko) < 1e-8
    np.testing.assert_allclose(kkt, kkto, rtol=1e-7)


@pytest.mark.parametrize("name,T,timesteps", [("panda", 100, None), ("panda", 61, [0.01] * 31 + [0.02] * 20 + [0.04] * 10),
                                              ("panda", 511, None), ("chain4", 37, None), ("chain6", 18, None), ("pendulum", 9, None)])
def test_mfma_layout_sweep_agrees_with_the_lane_grid_sweep(hip_backend, monkeypatch, name, T, timesteps):
    """K2 has two implementations for nv <= 7: the 8 x 8 lane grid (k_riccati, AGX_RICCATI_MX=0) and the MFMA operand
    layout (k_riccati_mx, default).  Same tiles in, same direction / gains out to round-off -- over a long horizon
    too: the MFMA sweep symmetrises the value function at every node, without which round-off asymmetry grows
    ~1.5 x per node (visible beyond ~50 nodes)."""
    table = MODELS[name]()
    B = 3
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=31, frame=len(table.frame_names) - 1, timesteps=timesteps)
    xs[:, 0] = x0
    out = {}
    for mx in ("0", "1"):
        monkeypatch.setenv("AGX_RICCATI_MX", mx)
        h = hip_backend.HipOcp(table, po, B)
        h.set_refs(ref)
        h.upload_warmstart(xs, us)
        out[mx] = h.direction()
        h.close()
    for a, b, tol in zip(out["0"], out["1"], (1e-9, 1e-9, 1e-10, 1e-10, 1e-10)):  # K, k, dx, du, kkt
        assert np.isfinite(b).all()
        assert rel(b, a) < tol


def test_qp_tile_cost_line_tail_is_zero(hip_backend, panda):
    """The MFMA-layout Riccati sweep loads the element behind `cost` of a node's QP tile for its pad lanes: it must be
    zero after derivative passes, warm-start shifts and solves (nobody writes it; buffers are cleared on allocation)."""
    import ctypes as C

    T, B = 12, 3
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=5, frame=panda.frame_id("panda_hand_tcp"))
    h = hip_backend.HipOcp(panda, po, B)
    h.set_refs(ref)
    h.solve(x0, xs, us, 5)
    lib = hip_backend.lib()
    qs, as_ = C.c_int(0), C.c_int(0)
    assert lib.agx_ocp_qp_tiles(h._h, None, None, C.byref(qs), C.byref(as_)) == 0
    qt = np.empty((B, T + 1, qs.value))
    aux = np.empty((B, T + 1, as_.value))
    assert lib.agx_ocp_qp_tiles(h._h, qt.ctypes.data_as(C.c_void_p), aux.ctypes.data_as(C.c_void_p), C.byref(qs), C.byref(as_)) == 0
    cost = qs.value - 8
    assert np.all(qt[..., cost + 1:] == 0.0) and np.any(qt[..., cost] != 0.0)
    h.close()


def test_node_shares_inside_the_forward_pass_agree_with_the_node_kernel(hip_backend, monkeypatch, panda):
    """AGX_FUSED_KKT=1 computes du and the KKT / cost / gap totals inside the forward pass of k_riccati_mx instead of in
    k_node_kkt: same solve (iterations, xs, us, K, status) up to the summation order of the totals."""
    T, B = 37, 6  # not a multiple of the prefetch depth
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=41, frame=panda.frame_id("panda_hand_tcp"))
    out = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("AGX_FUSED_KKT", fused)
        h = hip_backend.HipOcp(panda, po, B)
        h.set_refs(ref)
        out[fused] = h.solve(x0, xs, us, 10)
        h.close()
    (xs0, us0, K0, st0), (xs1, us1, K1, st1) = out["0"], out["1"]
    np.testing.assert_array_equal(st0["iter"], st1["iter"])
    np.testing.assert_array_equal(st0["solved"], st1["solved"])
    assert rel(xs1, xs0) < 1e-11 and rel(us1, us0) < 1e-10 and rel(K1, K0) < 1e-9
    np.testing.assert_allclose(st1["kkt"], st0["kkt"], rtol=1e-9)
    np.testing.assert_allclose(st1["cost"], st0["cost"], rtol=1e-12)


@pytest.mark.parametrize("name,rows,T,seed", [("panda", "goal", 25, 10), ("panda", "reg", 40, 11), ("chain4", "goal", 12, 12), ("chain6", "goal", 15, 13)])
def test_full_solve_matches_oracle(hip_backend, name, rows, T, seed):
    table = MODELS[name]()
    frame = len(table.frame_names) - 1
    B = 7
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=seed, frame=frame, rows=rows)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 12)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 12, nthreads=8)
    np.testing.assert_array_equal(st_h["iter"], st_o["iter"])
    np.testing.assert_array_equal(st_h["solved"], st_o["solved"])
    assert rel(xs_h, xs_o) < 1e-9 and rel(us_h, us_o) < 1e-9
    assert rel(K_h, K_o) < 1e-7
    np.testing.assert_allclose(st_h["kkt"], st_o["kkt"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(st_h["cost"], st_o["cost"], rtol=1e-10)
    np.testing.assert_array_equal(xs_h[:, 0], x0)  # the solver pins xs[0] = x0


def test_iteration_cap_and_unsolved_status(hip_backend, panda):
    tcp = panda.frame_id("panda_hand_tcp")
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, 20, 0.01, 3, seed=14, frame=tcp)
    h, o = hip_backend.HipOcp(panda, po, 3), Oracle(panda, po, 3)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 1)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 1)
    assert list(st_h["iter"]) == [1, 1, 1] and not st_h["solved"].any()
    assert rel(xs_h, xs_o) < 1e-10 and rel(K_h, K_o) < 1e-8


def test_golden_fixture_through_the_c_abi(hip_backend, golden):
    """The reference's golden case (tests/test_ocp_croco_base.py:175-204, 6 decimals upstream)."""
    table, po, ref, x0, xs0, us0 = workloads.golden_problem()
    h = hip_backend.HipOcp(table, po, 1)
    h.set_refs(ref)
    xs, us, K, st = h.solve(x0, xs0, us0, 100)
    assert st["solved"][0] == 1
    np.testing.assert_allclose(xs[0], golden["states"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(us[0], golden["feed_forward_terms"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(K[0], golden["ricatti_gains"], rtol=0, atol=1e-7)


def test_warm_start_shift_on_device(hip_backend):
    table = rt.chain_table(3, seed=9)
    rows = [_abi.RowSpec(_abi.RES_STATE)]
    ts = [0.1, 0.1, 0.2, 0.2, 0.4]
    po = _abi.PackedOcp(3, ts, rows, rows)
    B = 4
    rng = np.random.default_rng(2)
    xs, us = rng.normal(size=(B, 6, 6)), rng.normal(size=(B, 5, 3))
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.upload_warmstart(xs, us)
    h.shift_warmstart()
    xs_h, us_h, _, _ = h.download(want_K=False)
    xs_o, us_o = o.shift_warmstart(xs, us)
    np.testing.assert_array_equal(xs_h[:, :2], xs_o[:, :2])  # pure copies are bit exact
    np.testing.assert_array_equal(us_h, us_o)
    assert rel(xs_h, xs_o) < 1e-13
    np.testing.assert_array_equal(xs_h[:, -1], xs[:, -1])


def test_residual_readback(hip_backend, panda):
    tcp = panda.frame_id("panda_hand_tcp")
    B, T = 2, 6
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=15, frame=tcp)
    h, o = hip_backend.HipOcp(panda, po, B), Oracle(panda, po, B)
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    offs = np.cumsum([0] + [_abi.row_nr(r.kind, 7) for r in po.running])
    for row in range(3):
        got = h.residuals(row)
        for b in range(B):
            for t in range(T):
                _, _, res = o.node_calc(False, 0.01, xs[b, t], us[b, t], ref[b, t])
                np.testing.assert_allclose(got[b, t], res[offs[row]:offs[row + 1]], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("factors", [None, [1, 1, 1, 2, 2, 2, 4, 4]])
def test_resident_sine_trajectory_and_mpc_steps(hip_backend, panda, factors):
    """Device-resident reference generator + receding-horizon steps vs a host restatement built on the
    oracle; with dt factors (OCPParamsBaseCroco.timesteps, TrajectoryBuffer.compute_horizon_indexes) the
    warm-start shift integrates the nodes whose dt differs from the first one
    (warm_start_shift_previous_solution.py:95-104)."""
    tcp = panda.frame_id("panda_hand_tcp")
    B, dt = 3, 0.01
    T = 16 if factors is None else len(factors)
    fac = np.ones(T, dtype=int) if factors is None else np.asarray(factors)
    hidx = np.concatenate([[0], np.cumsum(fac)]).astype(np.int32)  # trajectory point of node t, relative to the window start
    NP = int(hidx[-1]) + 8
    running, terminal = workloads.goal_reaching_rows(tcp)
    po = _abi.PackedOcp(7, list(dt * fac), running, terminal)
    h, o = hip_backend.HipOcp(panda, po, B), Oracle(panda, po, B)
    q0, amp, puls, scale, t0 = workloads.sine_batch_params(B)
    w = workloads.SINE_WEIGHTS
    h.sine_trajectory(NP, dt, q0, amp, puls, scale, t0, w["w_q"], w["w_qdot"], w["w_effort"], w["w_pose"], tcp)
    if factors is not None:
        h.set_horizon_indexes(hidx)
    # host restatement of sine_wave_configuration_space.py:41-72 with oracle RNEA / FK
    def sample(k):
        t = t0 + k * dt
        s = np.clip(t[:, None] / scale, 0.0, 1.0)
        ramp = 10 * s**3 - 15 * s**4 + 6 * s**5
        dramp = np.where((s > 0) & (s < 1), (30 * s**2 - 60 * s**3 + 30 * s**4) / scale, 0.0)
        ddramp = np.where((s > 0) & (s < 1), (60 * s - 180 * s**2 + 120 * s**3) / scale**2, 0.0)
        sw, cw = np.sin(puls * t[:, None]), np.cos(puls * t[:, None])
        q = q0 + amp * ramp * sw
        dq = amp * (dramp * sw + ramp * puls * cw)
        ddq = amp * (ddramp * sw + 2 * dramp * puls * cw - ramp * puls**2 * sw)
        return q, dq, ddq, o.rnea(q, dq, ddq).reshape(B, 7), o.frame_placement(tcp, q)
    for k in (0, 5, NP - 1):
        got = h.traj_point(k)
        for g, wnt in zip(got, sample(k)):
            np.testing.assert_allclose(g, wnt, rtol=1e-11, atol=1e-12)
    def window_tile(k0):
        ref = po.new_ref_tile(B)
        for t in range(T + 1):
            q, dq, ddq, u, pose = sample(k0 + int(hidx[t]))
            term = t == T
            rows = terminal if term else running
            offs = po.terminal_offsets if term else po.running_offsets
            for r, off in zip(rows, offs):
                nref, nr = _abi.row_nref(r.kind, 7), _abi.row_nr(r.kind, 7)
                seg = ref[:, t, off:off + 1 + nref + nr]
                seg[:, 0] = 1.0
                if r.kind == _abi.RES_STATE:
                    seg[:, 1:15] = np.concatenate([q, dq], 1)
                    seg[:, 15:22], seg[:, 22:29] = w["w_q"], w["w_qdot"]
                elif r.kind == _abi.RES_CONTROL:
                    seg[:, 1:8], seg[:, 8:15] = u, w["w_effort"]
                else:
                    seg[:, 1:13], seg[:, 13:19] = pose, w["w_pose"]
        return ref
    # three receding-horizon steps, closed on the own prediction
    xs_o = us_o = None
    for step in range(3):
        h.mpc_step(step, 10, first=(step == 0))
        xs_h, us_h, K_h, st_h = h.download()
        ref = window_tile(step)
        if step == 0:
            pts = [sample(int(hidx[t])) for t in range(T + 1)]
            xs_ws = np.stack([np.concatenate([p[0], p[1]], 1) for p in pts], 1)
            us_ws = np.stack([p[3] for p in pts[:T]], 1)
            x0 = xs_ws[:, 0].copy()
        else:
            x0 = xs_o[:, 1].copy()
            xs_ws, us_ws = o.shift_warmstart(xs_o, us_o)
        xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs_ws, us_ws, 10)
        np.testing.assert_array_equal(st_h["iter"], st_o["iter"])
        assert rel(xs_h, xs_o) < 1e-9 and rel(us_h, us_o) < 1e-8 and rel(K_h, K_o) < 1e-7
    us0, K0, x1, st = h.download_first()
    np.testing.assert_array_equal(us0, us_h[:, 0])
    np.testing.assert_array_equal(K0, K_h[:, 0])
    np.testing.assert_array_equal(x1, xs_h[:, 1])


def test_full_size_properties(hip_backend, panda):
    """BASELINE.json sizes (T = 100, B = 1024): size-independent properties instead of an oracle run:
    replicated instances give bit-identical results (no cross-instance coupling, no races), the
    result is a KKT point of its own linearisation (direction ~ 0 after convergence), xs[0] = x0."""
    tcp = panda.frame_id("panda_hand_tcp")
    B, T, R = 1024, 100, 4
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, R, seed=20, frame=tcp)
    rep = B // R
    h = hip_backend.HipOcp(panda, po, B)
    h.set_refs(np.tile(ref, (rep, 1, 1)))
    xs_h, us_h, K_h, st = h.solve(np.tile(x0, (rep, 1)), np.tile(xs, (rep, 1, 1)), np.tile(us, (rep, 1, 1)), 30)
    assert st["solved"].all() and np.isfinite(K_h).all()
    for r in range(R):
        np.testing.assert_array_equal(xs_h[r::R], np.broadcast_to(xs_h[r], (rep,) + xs_h[r].shape))
        np.testing.assert_array_equal(K_h[r::R], np.broadcast_to(K_h[r], (rep,) + K_h[r].shape))
        np.testing.assert_array_equal(st["iter"][r::R], st["iter"][r])
    np.testing.assert_array_equal(xs_h[:, 0], np.tile(x0, (rep, 1)))
    # the first R instances agree with the oracle
    xs_o, us_o, K_o, st_o = Oracle(panda, po, R).solve(ref, None, x0, xs, us, 30, nthreads=4)
    np.testing.assert_array_equal(st["iter"][:R], st_o["iter"])
    assert rel(xs_h[:R], xs_o) < 1e-9 and rel(K_h[:R], K_o) < 1e-7
    # re-solving from the solution returns immediately with the same gains
    xs2, us2, K2, st2 = h.solve(np.tile(x0, (rep, 1)), xs_h, us_h, 30)
    assert (st2["iter"] == 0).all() and st2["solved"].all()
    np.testing.assert_array_equal(xs2, xs_h)
    assert rel(K2, K_h) < 1e-12


@pytest.mark.gpu
def test_feedback_rollout_is_the_riccati_feedback_law_on_the_model(hip_backend):
    """u = us[0] + K[0] (x0 - x) (agimus_controller.py:418-426 feeds exactly these to the
    linear feedback controller), integrated with the model's own semi-implicit Euler at the control rate."""
    table = rt.panda_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    B, T = 3, 10
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, 41, frame=tcp)
    hb = hip_backend.HipOcp(table, po, B)
    hb.set_refs(ref)
    xs_s, us_s, K_s, st = hb.solve(x0, xs, us, 20)
    dist = np.random.default_rng(0).normal(0, 0.5, (B, 7))
    n_sub, dt_sub = 10, 1e-3
    hb.feedback_rollout(n_sub, dt_sub, dist)
    got = hb.download_x0()
    o = Oracle(table, po, B)
    x = x0.copy()
    for _ in range(n_sub):
        u = us_s[:, 0] + dist + np.einsum("bij,bj->bi", K_s[:, 0], x0 - x)
        a = o.forward_dynamics(x[:, :7], x[:, 7:], u).reshape(B, 7)
        v = x[:, 7:] + dt_sub * a
        x = np.concatenate([x[:, :7] + dt_sub * v, v], 1)
    np.testing.assert_allclose(got, x, rtol=1e-10, atol=1e-12)
    # one sub-step of the node's own dt without disturbance is the plan itself: x0 -> xs[1] up to the
    # dynamics gap the solver tolerates (KKT <= 1e-3)
    hb.upload_x0(x0)
    hb.feedback_rollout(1, 0.01, None)
    assert np.abs(hb.download_x0() - xs_s[:, 1]).max() < 2e-3
    hb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T,B", [(1, 1), (2, 3), (511, 2)])
def test_extreme_horizons(hip_backend, panda, T, B):
    """Shortest horizon (one control), tiny batches, and the longest horizon the step kernel takes
    (T + 1 = 512 nodes); T + 1 > 512 is refused with a message."""
    tcp = panda.frame_id("panda_hand_tcp")
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=31, frame=tcp)
    h, o = hip_backend.HipOcp(panda, po, B), Oracle(panda, po, B)
    h.set_refs(ref)
    iters = 4 if T > 100 else 10
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, iters)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, iters, nthreads=2)
    np.testing.assert_array_equal(st_h["iter"], st_o["iter"])
    assert rel(xs_h, xs_o) < 1e-8 and rel(us_h, us_o) < 1e-7 and rel(K_h, K_o) < 1e-6
    h.close()


@pytest.mark.gpu
def test_horizon_beyond_the_step_kernel_is_refused(hip_backend, panda):
    tcp = panda.frame_id("panda_hand_tcp")
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, 512, 0.01, 1, seed=32, frame=tcp)
    h = hip_backend.HipOcp(panda, po, 1)
    h.set_refs(ref)
    with pytest.raises(hip_backend.HipError, match="511"):
        h.solve(x0, xs, us, 2)
    h.close()


@pytest.mark.gpu
def test_inactive_and_zero_weight_rows(hip_backend, panda):
    """CostModelSumItem.active = False and weight 0 rows contribute nothing (ocp_croco_generic.py:578-585, 691)."""
    tcp = panda.frame_id("panda_hand_tcp")
    B, T = 3, 8
    po, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=33, frame=tcp)
    # same problem with the placement rows switched off two ways
    run_off = [_abi.RowSpec(r.kind, active=(r.kind != _abi.RES_FRAME_PLACEMENT), frame=r.frame, name=r.name) for r in po.running]
    term_off = [_abi.RowSpec(r.kind, active=(r.kind != _abi.RES_FRAME_PLACEMENT), frame=r.frame, name=r.name) for r in po.terminal]
    po_inactive = _abi.PackedOcp(7, [0.01] * T, run_off, term_off)
    ref_zero = ref.copy()
    for term, rows in ((False, po.running), (True, po.terminal)):
        for i, r in enumerate(rows):
            if r.kind == _abi.RES_FRAME_PLACEMENT:
                po.row_view(ref_zero, term, i)[0][...] = 0.0
    h1, h2, o = hip_backend.HipOcp(panda, po_inactive, B), hip_backend.HipOcp(panda, po, B), Oracle(panda, po_inactive, B)
    h1.set_refs(ref)
    h2.set_refs(ref_zero)
    r1 = h1.solve(x0, xs, us, 10)
    r2 = h2.solve(x0, xs, us, 10)
    ro = o.solve(ref, None, x0, xs, us, 10)
    assert rel(r1[0], ro[0]) < 1e-9 and rel(r2[0], ro[0]) < 1e-9
    np.testing.assert_array_equal(r1[3]["iter"], ro[3]["iter"])
    h1.close(); h2.close()


@pytest.mark.gpu
def test_filter_line_search_matches_the_checker(hip_backend, panda):
    """use_filter_line_search = True (ocp_param_base.py:64, SolverCSQP filter of size 1): accept a step unless
    it is no better than the current point in cost, gap norm and constraint norm at once."""
    tcp = panda.frame_id("panda_hand_tcp")
    B, T = 5, 20
    po0, ref, x0, xs, us = workloads.random_goal_problem(panda, T, 0.01, B, seed=77, frame=tcp)
    po = _abi.PackedOcp(7, [0.01] * T, po0.running, po0.terminal, use_filter_line_search=True)
    h, o = hip_backend.HipOcp(panda, po, B), Oracle(panda, po, B)
    h.set_refs(ref)
    r_h = h.solve(x0, xs, us, 15)
    r_o = o.solve(ref, None, x0, xs, us, 15, nthreads=4)
    np.testing.assert_array_equal(r_h[3]["iter"], r_o[3]["iter"])
    assert rel(r_h[0], r_o[0]) < 1e-8 and rel(r_h[2], r_o[2]) < 1e-6
    # and it is a different algorithm from the merit search on this problem or at least converges
    assert np.all(r_h[3]["kkt"] < 1e-3) or np.all(r_h[3]["iter"] == 15)
    h.close()
