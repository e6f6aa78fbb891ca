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


def _expected_qp_blocks(tiles, dts, preg=1e-9):
    """Acceleration-input QP blocks from the canonical tile of the checker:
    M = dt (Fu_v)^-1, taux = -M a_x, H = [taux M]' D [taux M] + Lxx, D = Luu + preg."""
    nv = 30
    sl = _abi.tile_slices(nv)
    B, T1 = tiles.shape[:2]
    out = {k: np.zeros((B, T1, nv, nv)) for k in ("Hqq", "Hqv", "Hvv", "Hqw", "Hvw", "Hww", "M", "tq", "tv", "Lqq")}
    out.update(gx=np.zeros((B, T1, 2 * nv)), gw=np.zeros((B, T1, nv)), f=np.zeros((B, T1, 2 * nv)), cost=np.zeros((B, T1)))
    for b in range(B):
        for t in range(T1):
            z = tiles[b, t]
            Lxx = z[sl["Lxx"]].reshape(2 * nv, 2 * nv)
            Lx, Lu = z[sl["Lx"]], z[sl["Lu"]]
            o = {k: v[b, t] for k, v in out.items()}
            o["Lqq"][:] = Lxx[:nv, :nv]
            out["cost"][b, t] = z[sl["cost"]][0]
            out["f"][b, t] = z[sl["f"]]
            if t == T1 - 1:
                o["Hqq"][:], o["Hqv"][:], o["Hvv"][:] = Lxx[:nv, :nv], Lxx[:nv, nv:], Lxx[nv:, nv:]
                out["gx"][b, t] = Lx
                continue
            dt = dts[t]
            Fx, Fu = z[sl["Fx"]].reshape(2 * nv, 2 * nv), z[sl["Fu"]].reshape(2 * nv, nv)
            M = np.linalg.inv(Fu[nv:] / dt)
            tq, tv = -M @ (Fx[nv:, :nv] / dt), -M @ ((Fx[nv:, nv:] - np.eye(nv)) / dt)
            D = np.diag(np.diag(z[sl["Luu"]].reshape(nv, nv)) + preg)
            o["M"][:], o["tq"][:], o["tv"][:] = M, tq, tv
            o["Hww"][:], o["Hqw"][:], o["Hvw"][:] = M.T @ D @ M, tq.T @ D @ M, tv.T @ D @ M
            o["Hqq"][:] = Lxx[:nv, :nv] + tq.T @ D @ tq
            o["Hqv"][:] = Lxx[:nv, nv:] + tq.T @ D @ tv
            o["Hvv"][:] = Lxx[nv:, nv:] + tv.T @ D @ tv
            out["gw"][b, t] = M @ Lu
            out["gx"][b, t] = Lx + np.concatenate([tq.T @ Lu, tv.T @ Lu])
    return out


def _dense_rows_problem(table, T, B, seed):
    """Goal-reaching rows plus a FrameTranslation, a FrameRotation (other hand) and a capsule / sphere collision cost."""
    import dataclasses

    lh, rh = table.frame_id("l_hand"), table.frame_id("r_hand")
    nf = len(table.frame_names)
    t2 = dataclasses.replace(
        table,
        frame_names=list(table.frame_names) + ["l_forearm_capsule", "ball"],
        frame_parent=np.concatenate([table.frame_parent, [table.joint_names.index("l_elbow"), -1]]).astype(np.int32),
        frame_placement=np.concatenate([table.frame_placement, [rt.se3(None, [0.0, 0.0, -0.1]), rt.se3(None, [0.35, 0.3, 1.1])]]),
        frame_radius=np.concatenate([np.zeros(nf), [0.05, 0.12]]),
        frame_halflen=np.concatenate([np.zeros(nf), [0.08, 0.0]]),
        frame_box=np.zeros((nf + 2, 3)),
    )
    running = [_abi.RowSpec(_abi.RES_STATE), _abi.RowSpec(_abi.RES_CONTROL), _abi.RowSpec(_abi.RES_FRAME_PLACEMENT, frame=lh),
               _abi.RowSpec(_abi.RES_FRAME_TRANSLATION, frame=rh), _abi.RowSpec(_abi.RES_FRAME_ROTATION, frame=rh),
               _abi.RowSpec(_abi.RES_COLLISION, activation=_abi.ACT_QUAD_EXP, alpha=0.05, frame=nf, frame_b=nf + 1)]
    terminal = [_abi.RowSpec(_abi.RES_STATE), _abi.RowSpec(_abi.RES_FRAME_PLACEMENT, frame=lh),
                _abi.RowSpec(_abi.RES_COLLISION, activation=_abi.ACT_EXP, alpha=0.3, frame=nf, frame_b=nf + 1)]
    po = _abi.PackedOcp(30, [0.01, 0.02, 0.01, 0.03][:T] + [0.01] * max(T - 4, 0), running, terminal)
    rng = np.random.default_rng(seed)
    ref = po.new_ref_tile(B)
    ref[...] = rng.uniform(0.2, 1.5, ref.shape)
    for term, rows in ((False, running), (True, terminal)):
        for r, row in enumerate(rows):
            w, rr, aw = po.row_view(ref, term, r)
            if row.kind in (_abi.RES_FRAME_PLACEMENT, _abi.RES_FRAME_ROTATION):
                for b in range(B):
                    for t in range(rr.shape[1]):
                        rr[b, t, :9] = rt.rpy(*rng.uniform(-1.0, 1.0, 3)).reshape(9)
    xs = rng.uniform(-0.8, 0.8, (B, T + 1, 60))
    us = rng.uniform(-20, 20, (B, T, 30))
    return t2, po, ref, xs, us


def test_qp_tiles_30dof_against_the_checker(hip_backend, humanoid):
    """The workgroup-per-node derivative pass (agx_big_k1.hpp: LDS + fp64 MFMA): every block of the QP / aux tiles against
    the blocks rebuilt in numpy from the checker's canonical tile; all dense-row kinds, non-uniform dt."""
    B, T = 2, 4
    table, po, ref, xs, us = _dense_rows_problem(humanoid, T, B, seed=11)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    q, a = h.qp_tiles()
    want = _expected_qp_blocks(o.calc_diff(ref, None, xs, us), po.dt)
    got = dict(q, **{k: a[k] for k in ("M", "tq", "tv", "Lqq")})
    for name, w in want.items():
        scale = max(np.abs(w).max(), 1e-300)
        assert np.abs(got[name] - w).max() <= 1e-9 * scale + 1e-12, name
    h.close()


def test_integrate_and_shift_30dof_with_dt_factors(hip_backend, humanoid):
    """Warm-start shift of a large model with dt factors: copies for dt_i == dt_0, one integration workgroup per other node."""
    frame = len(humanoid.frame_names) - 1
    B, T = 2, 6
    running, terminal = workloads.goal_reaching_rows(frame)
    po = _abi.PackedOcp(30, [0.01, 0.01, 0.02, 0.02, 0.04, 0.04], running, terminal)
    h, o = hip_backend.HipOcp(humanoid, po, B), Oracle(humanoid, po, B)
    rng = np.random.default_rng(5)
    xs, us = rng.uniform(-0.5, 0.5, (B, T + 1, 60)), rng.uniform(-10, 10, (B, T, 30))
    h.upload_warmstart(xs, us)
    h.shift_warmstart()
    xs_h, us_h, _, _ = h.download(want_K=False)
    xs_o, us_o = o.shift_warmstart(xs, us)
    assert rel(xs_h, xs_o) < 1e-10 and rel(us_h, us_o) < 1e-14
    np.testing.assert_array_equal(xs_h[:, :2], xs[:, 1:3])  # plain copies are bit exact
    h.close()


def test_full_size_humanoid_replicas_are_bit_identical(hip_backend, humanoid):
    """BASELINE configs[4] at full size (B = 512, T = 50) in one process with the small models of the other tests:
    512 copies of three instances must come out bit-identical (no cross-instance state, no scratch aliasing), and the
    three agree with the checker."""
    frame = len(humanoid.frame_names) - 1
    B, T, n = 512, 50, 3
    po, ref, x0, xs, us = workloads.random_goal_problem(humanoid, T, 0.01, n, seed=17, frame=frame)
    rep = lambda a: np.ascontiguousarray(np.tile(a, (B // n + 1,) + (1,) * (a.ndim - 1))[:B])  # noqa: E731
    h = hip_backend.HipOcp(humanoid, po, B)
    h.set_refs(rep(ref))
    xs_h, us_h, K_h, st_h = h.solve(rep(x0), rep(xs), rep(us), 4)
    h.close()
    for k in range(n, B):
        assert np.array_equal(xs_h[k], xs_h[k % n]) and np.array_equal(us_h[k], us_h[k % n]) and np.array_equal(K_h[k], K_h[k % n]), k
        assert st_h["iter"][k] == st_h["iter"][k % n]
    o = Oracle(humanoid, po, n)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 4, nthreads=4)
    np.testing.assert_array_equal(st_h["iter"][:n], st_o["iter"])
    assert rel(xs_h[:n], xs_o) < 1e-8 and rel(us_h[:n], us_o) < 1e-7 and rel(K_h[:n], K_o) < 1e-6


def test_full_size_humanoid_with_torque_limits(hip_backend, humanoid):
    """BASELINE configs[4] shape (B = 512, T = 50) with ConstraintModelControlLimit: the ADMM path of the large models at full
    size -- replicas of three instances come out bit-identical, the controls respect the bound, and the three agree with the
    checker (same SQP / ADMM iteration counts)."""
    frame = len(humanoid.frame_names) - 1
    B, T, n = 512, 50, 3
    po0, ref, x0, xs, us = workloads.random_goal_problem(humanoid, T, 0.01, n, seed=17, frame=frame)
    lim = np.full(30, 60.0)
    con = [_abi.ConstraintSpec(_abi.RES_CONTROL, lower=-lim, upper=lim, name="ctrl_limit")]
    po = _abi.PackedOcp(30, [0.01] * T, po0.running, po0.terminal, max_qp_iters=50, running_constraints=con)
    rep = lambda a: np.ascontiguousarray(np.tile(a, (B // n + 1,) + (1,) * (a.ndim - 1))[:B])  # noqa: E731
    h = hip_backend.HipOcp(humanoid, po, B)
    h.set_refs(rep(ref))
    xs_h, us_h, K_h, st_h = h.solve(rep(x0), rep(xs), rep(us), 2)
    h.close()
    for k in range(n, B):
        assert np.array_equal(xs_h[k], xs_h[k % n]) and np.array_equal(us_h[k], us_h[k % n]), k
        assert st_h["qp_iters"][k] == st_h["qp_iters"][k % n]
    assert np.abs(us_h).max() <= 60.0 + 1e-2
    o = Oracle(humanoid, po, n)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 2, nthreads=4)
    assert np.abs(us_o).max() > 0.9 * 60.0  # the bound matters
    np.testing.assert_array_equal(st_h["iter"][:n], st_o["iter"])
    np.testing.assert_array_equal(st_h["qp_iters"][:n], st_o["qp_iters"])
    assert rel(xs_h[:n], xs_o) < 1e-6 and rel(us_h[:n], us_o) < 1e-5
