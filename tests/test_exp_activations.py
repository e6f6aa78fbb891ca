"""colmpc ActivationModelExp / ActivationModelQuadExp on VECTOR residuals.  The reference builds them for any residual
(`ocp_croco_generic.py:118-131`: `colmpc.ActivationModelExp(residual.nr, alpha)`), not only for the scalar collision
distance of `ocp_traj_tracking_collision_avoidance.yaml`.  The forms are recalled (value exp(-|r|^2 / alpha)
resp. exp(-|r| / alpha), diagonal second derivative): parity is HIP against this repository's checker, UNPINNED against colmpc.
Serial chains with such a row take the one-lane-per-node derivative kernel, large models the workgroup kernel."""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _rows(frame, act, kinds):
    """Goal-reaching rows (quadratic) plus one Exp / QuadExp row per kind in `kinds`."""
    running, terminal = workloads.goal_reaching_rows(frame)
    alpha = {_abi.RES_STATE: 40.0, _abi.RES_CONTROL: 400.0, _abi.RES_FRAME_PLACEMENT: 8.0, _abi.RES_FRAME_TRANSLATION: 2.0,
             _abi.RES_FRAME_ROTATION: 6.0}
    for k in kinds:
        row = _abi.RowSpec(k, activation=act, alpha=alpha[k], frame=frame, name=f"exp_{k}")
        running = running + [row]
        if k != _abi.RES_CONTROL:
            terminal = terminal + [row]
    return running, terminal


# weights of the extra rows: small enough that the negative curvature of the activation stays below the quadratic rows'
SCALE = {_abi.RES_STATE: 0.05, _abi.RES_CONTROL: 2e-3, _abi.RES_FRAME_PLACEMENT: 0.2, _abi.RES_FRAME_TRANSLATION: 0.2, _abi.RES_FRAME_ROTATION: 0.2}


def _problem(table, T, B, seed, act, kinds):
    frame = len(table.frame_names) - 1
    running, terminal = _rows(frame, act, kinds)
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=seed, frame=frame, rows=(running, terminal))
    for term, rws in ((False, running), (True, terminal)):
        for i, r in enumerate(rws):
            if r.activation != _abi.ACT_WEIGHTED_QUAD:
                wi, _, _ = po.row_view(ref, term, i)
                wi[...] *= SCALE[r.kind]
    return po, ref, x0, xs, us


def _table(name):
    if name == "panda":
        return rt.panda_table(0.1)
    if name == "tree5":
        return rt.tree_table(5, seed=45)
    if name == "fingers9":
        from test_model_sizes import _model
        return _model(9, "panda_fingers")
    return rt.humanoid30_table()


ALL = (_abi.RES_STATE, _abi.RES_CONTROL, _abi.RES_FRAME_PLACEMENT)
CASES = [("panda", _abi.ACT_QUAD_EXP, ALL), ("panda", _abi.ACT_EXP, ALL), ("panda", _abi.ACT_QUAD_EXP, (_abi.RES_FRAME_TRANSLATION, _abi.RES_FRAME_ROTATION)),
         ("tree5", _abi.ACT_QUAD_EXP, ALL), ("fingers9", _abi.ACT_QUAD_EXP, ALL), ("fingers9", _abi.ACT_EXP, ALL),
         ("humanoid", _abi.ACT_QUAD_EXP, (_abi.RES_FRAME_PLACEMENT,))]  # 30 joints: the reference tile of a node is limited to 256 doubles


@pytest.mark.parametrize("name,act,kinds", CASES)
def test_derivative_tiles(hip_backend, name, act, kinds):
    table = _table(name)
    nv = table.nv
    B, T = 3, 4
    po, ref, x0, xs, us = _problem(table, T, B, 7 + nv, act, kinds)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    got, want = h.calc_diff(), o.calc_diff(ref, None, xs, us)
    for field, s in _abi.tile_slices(nv).items():
        scale = max(np.abs(want[..., s]).max(), 1e-300)
        assert np.abs(got[..., s] - want[..., s]).max() <= 1e-10 * scale + 1e-13, field
    h.close()


@pytest.mark.parametrize("name,act,kinds", [("panda", _abi.ACT_QUAD_EXP, ALL), ("panda", _abi.ACT_EXP, ALL), ("fingers9", _abi.ACT_QUAD_EXP, ALL),
                                            ("humanoid", _abi.ACT_QUAD_EXP, (_abi.RES_FRAME_PLACEMENT,))])
def test_full_solve(hip_backend, name, act, kinds):
    """Same SQP iterations, iterate and gains as the checker with a non-convex activation in the cost."""
    table = _table(name)
    nv = table.nv
    B, T = 2, 8
    po, ref, x0, xs, us = _problem(table, T, B, 70 + nv, act, kinds)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    xs_h, us_h, K_h, st_h = h.solve(x0, xs, us, 8)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 8)
    assert np.array_equal(st_h["iter"], st_o["iter"]) and np.array_equal(st_h["solved"], st_o["solved"])
    np.testing.assert_allclose(xs_h, xs_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(us_h, us_o, rtol=1e-7, atol=1e-7)
    assert np.abs(K_h - K_o).max() <= 1e-6 * np.abs(K_o).max()
    h.close()
