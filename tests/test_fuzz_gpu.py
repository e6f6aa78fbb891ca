"""Seeded random problem definitions against the CPU checker: model size (every compiled capacity: 7 / 16 / 30 / 32 joints after
padding), chain or tree, a random subset of the cost rows of `ocp_croco_generic.py` (State, Control, FramePlacement / Translation /
Rotation with WeightedQuad, Exp or QuadExp activations), non-uniform time steps, optionally torque limits.  Derivative tiles to
1e-10, a two-iteration solve with the checker's iteration counts.  Complements the hand-written cases of the other files: the
row-table lowering, the padding and the kernel selection (8 lanes per node / one lane per node / workgroup per node) are exercised
in combinations nobody wrote down."""
import os

import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

ALPHA = {_abi.RES_STATE: 40.0, _abi.RES_CONTROL: 400.0, _abi.RES_FRAME_PLACEMENT: 8.0, _abi.RES_FRAME_TRANSLATION: 2.0, _abi.RES_FRAME_ROTATION: 6.0}
SCALE = {_abi.RES_STATE: 0.05, _abi.RES_CONTROL: 2e-3, _abi.RES_FRAME_PLACEMENT: 0.2, _abi.RES_FRAME_TRANSLATION: 0.2, _abi.RES_FRAME_ROTATION: 0.2}


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    nv = int(rng.choice([3, 5, 7, 9, 12, 20]))
    chain = bool(rng.integers(2)) or nv == 7
    table = rt.chain_table(nv, seed=seed) if chain else rt.tree_table(nv, seed=seed)
    frame = len(table.frame_names) - 1
    # the quadratic base rows keep the problem convex; extra rows with random kinds / activations on top
    running = [_abi.RowSpec(_abi.RES_CONTROL, name="control_reg"), _abi.RowSpec(_abi.RES_STATE, name="state_reg")]
    terminal = [_abi.RowSpec(_abi.RES_STATE, name="state_reg")]
    extra = []
    n_frame = 0
    for k in rng.permutation([_abi.RES_FRAME_PLACEMENT, _abi.RES_FRAME_TRANSLATION, _abi.RES_FRAME_ROTATION, _abi.RES_STATE, _abi.RES_CONTROL]):
        if rng.random() < 0.5:
            continue
        is_frame = k in (_abi.RES_FRAME_PLACEMENT, _abi.RES_FRAME_TRANSLATION, _abi.RES_FRAME_ROTATION)
        if is_frame and n_frame == 2:
            continue  # the 8-lane kernel stages two frame rows
        if nv > 16 and not is_frame:
            continue  # reference tile of a node: 256 doubles for large models
        n_frame += is_frame
        act = int(rng.choice([_abi.ACT_WEIGHTED_QUAD, _abi.ACT_EXP, _abi.ACT_QUAD_EXP])) if not (is_frame and rng.random() < 0.5) else _abi.ACT_WEIGHTED_QUAD
        extra.append(_abi.RowSpec(int(k), activation=act, alpha=ALPHA[int(k)], frame=frame, name=f"extra_{k}_{act}"))
    running = running + extra
    terminal = terminal + [r for r in extra if r.kind != _abi.RES_CONTROL]
    T = int(rng.integers(3, 8))
    ts = [0.01 if rng.random() < 0.7 else 0.02 for _ in range(T)]
    B = 2
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=seed, frame=frame, rows=(running, terminal), timesteps=ts)
    for term, rws in ((False, running), (True, terminal)):
        for i, r in enumerate(rws):
            if r.activation != _abi.ACT_WEIGHTED_QUAD:
                wi, _, _ = po.row_view(ref, term, i)
                wi[...] *= SCALE[r.kind]
    if rng.random() < 0.4:  # torque limits wide enough to stay well posed
        lim = np.full(nv, 50.0)
        con = [_abi.ConstraintSpec(_abi.RES_CONTROL, lower=-lim, upper=lim, name="ctrl_limit")]
        po = _abi.PackedOcp(nv, ts, running, terminal, max_qp_iters=50, running_constraints=con)
    return table, po, ref, x0, xs, us, B


@pytest.mark.parametrize("seed", range(int(os.environ.get("AGX_FUZZ_SEEDS", "16"))))  # AGX_FUZZ_SEEDS=200: a longer hunt
def test_random_problem_definitions(hip_backend, seed):
    table, po, ref, x0, xs, us, B = _case(seed)
    nv = table.nv
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    h.upload_warmstart(xs, us)
    got, want = h.calc_diff(), o.calc_diff(ref, None, xs, us)
    for field, s in _abi.tile_slices(nv).items():
        scale = max(np.abs(want[..., s]).max(), 1e-300)
        assert np.abs(got[..., s] - want[..., s]).max() <= 1e-10 * scale + 1e-13, (seed, field)
    r_h, r_o = h.solve(x0, xs, us, 2), o.solve(ref, None, x0, xs, us, 2)
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"]), seed
    assert np.array_equal(r_h[3]["flags"], r_o[3]["flags"]), seed
    if not np.any(r_o[3]["flags"]):
        np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
    h.close()


def _constrained_case(seed):
    rng = np.random.default_rng(5000 + seed)
    nv = int(rng.choice([4, 6, 7, 9, 14, 24]))
    chain = bool(rng.integers(2)) or nv == 7
    table = rt.chain_table(nv, seed=100 + seed) if chain else rt.tree_table(nv, seed=100 + seed)
    frame = len(table.frame_names) - 1
    T, B = int(rng.integers(4, 9)), 2
    po0, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, seed=seed, frame=frame)
    o0 = Oracle(table, po0, B)
    p0 = o0.frame_placement(frame, x0[:, :nv])
    con, tcon = [], []
    if rng.random() < 0.6:
        lim = np.full(nv, float(rng.uniform(20.0, 60.0)))
        con.append(_abi.ConstraintSpec(_abi.RES_CONTROL, lower=-lim, upper=lim, name="ctrl_limit"))
    if rng.random() < 0.5:
        xb = np.full(2 * nv, np.inf)
        xb[nv:] = float(rng.uniform(1.0, 3.0))
        c = _abi.ConstraintSpec(_abi.RES_STATE, lower=-xb, upper=xb, name="velocity_limit")
        con.append(c)
        tcon.append(c)
    if rng.random() < 0.5 or not con:
        half = float(rng.uniform(0.05, 0.2))
        c = _abi.ConstraintSpec(_abi.RES_FRAME_TRANSLATION, lower=-half, upper=half, ref=p0[:, 9:].mean(0), frame=frame, name="ee_box")
        con.append(c)
        tcon.append(c)
    po = _abi.PackedOcp(nv, [0.01] * T, po0.running, po0.terminal, max_qp_iters=50, running_constraints=con, terminal_constraints=tcon)
    return table, po, ref, x0, xs, us, B


@pytest.mark.parametrize("seed", range(int(os.environ.get("AGX_FUZZ_SEEDS", "12"))))
def test_random_constraint_sets(hip_backend, seed):
    """Torque limits, velocity bounds and an end-effector box in random combinations on models of every capacity: the ADMM
    paths (8 lanes per node / one lane per node for the constraint evaluation up to 7 joints, workgroup kernels above)."""
    table, po, ref, x0, xs, us, B = _constrained_case(seed)
    h, o = hip_backend.HipOcp(table, po, B), Oracle(table, po, B)
    h.set_refs(ref)
    r_h, r_o = h.solve(x0, xs, us, 2), o.solve(ref, None, x0, xs, us, 2)
    assert np.array_equal(r_h[3]["iter"], r_o[3]["iter"]) and np.array_equal(r_h[3]["qp_iters"], r_o[3]["qp_iters"]), seed
    assert np.array_equal(r_h[3]["flags"], r_o[3]["flags"]), seed
    if not np.any(r_o[3]["flags"]):
        np.testing.assert_allclose(r_h[0], r_o[0], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r_h[1], r_o[1], rtol=1e-5, atol=1e-5)
    h.close()
