"""Synthetic, seeded problem definitions shared by bench.py, smoke() and the tests
(SURVEY.md section 8(d)).  Pure numpy: no solver code lives here."""

from __future__ import annotations

import numpy as np

from . import _abi
from .factory import robot_tables as rt

# q0 of the reference's sine-wave test and pick-and-place example
# (agimus_controller/tests/test_sin_wave_configuration_space.py:31-43).
PANDA_Q0 = np.array(
    [-0.3619834760502907, -1.3575006398318104, 0.969610481368033, -2.6028532848927295,
     0.2040785081450368, 1.9436352693107668, 0.6423896937386857]
)  # fmt: skip


def goal_reaching_rows(frame: int):
    """Row tables of agimus_controller/agimus_controller/ocp/ocp_goal_reaching.yaml."""
    running = [
        _abi.RowSpec(_abi.RES_CONTROL, name="control_reg"),
        _abi.RowSpec(_abi.RES_STATE, name="state_reg"),
        _abi.RowSpec(_abi.RES_FRAME_PLACEMENT, frame=frame, name="goal_tracking"),
    ]
    terminal = [
        _abi.RowSpec(_abi.RES_STATE, name="state_reg"),
        _abi.RowSpec(_abi.RES_FRAME_PLACEMENT, frame=frame, name="goal_tracking"),
    ]
    return running, terminal


def regulation_rows(terminal_weight: float = 1.0):
    """Row tables of the pick-and-place example's ocp_definition_file.yaml
    (control_reg + state_reg; terminal state_reg)."""
    running = [_abi.RowSpec(_abi.RES_CONTROL, name="control_reg"), _abi.RowSpec(_abi.RES_STATE, name="state_reg")]
    terminal = [_abi.RowSpec(_abi.RES_STATE, name="state_reg", weight=terminal_weight)]
    return running, terminal


def collision_avoidance_rows(table, frame: int, pair=("panda_link7_capsule_0", "obstacle"), activation=_abi.ACT_QUAD_EXP, alpha=1e-4):
    """Cost rows of agimus_controller/agimus_controller/ocp/ocp_traj_tracking_collision_avoidance.yaml:
    control_reg, state_reg, goal_tracking, distance (QuadExp alpha = 1e-4 on the pair's signed distance);
    terminal the same minus control_reg.  (Its `collision` constraint is a constraint row, not a cost.)"""
    fa, fb = table.frame_id(pair[0]), table.frame_id(pair[1])
    dist = dict(activation=activation, alpha=alpha, frame=fa, frame_b=fb, name="distance")
    running, terminal = goal_reaching_rows(frame)
    return running + [_abi.RowSpec(_abi.RES_COLLISION, **dist)], terminal + [_abi.RowSpec(_abi.RES_COLLISION, **dist)]


def golden_problem():
    """The reference's only golden case, agimus_controller/tests/test_ocp_croco_base.py:14-155:
    Panda, T = 9, Euler step 1e-3, stateReg 0.1 / ctrlReg 1e-4 / placement 1.0 to (1,1,1),
    terminal stateReg 0.1 + placement 50, zero warm start, x0 = 0, 100 iterations."""
    table = rt.panda_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    running = [
        _abi.RowSpec(_abi.RES_STATE, name="stateReg"),
        _abi.RowSpec(_abi.RES_CONTROL, name="ctrlRegGrav"),
        _abi.RowSpec(_abi.RES_FRAME_PLACEMENT, frame=tcp, name="gripperPoseRM"),
    ]
    terminal = [
        _abi.RowSpec(_abi.RES_STATE, name="stateReg"),
        _abi.RowSpec(_abi.RES_FRAME_PLACEMENT, frame=tcp, name="gripperPose"),
    ]
    T = 9
    po = _abi.PackedOcp(7, [1e-3] * T, running, terminal)
    ref = po.new_ref_tile(1)
    target = np.concatenate([np.eye(3).reshape(9), [1.0, 1.0, 1.0]])
    for terminal_flag, row, w in ((False, 0, 0.1), (False, 1, 1e-4), (False, 2, 1.0), (True, 0, 0.1), (True, 1, 50.0)):
        wi, r, _ = po.row_view(ref, terminal_flag, row)
        wi[...] = w
        r[...] = 0.0
        if r.shape[-1] == 12:
            r[...] = target
    x0 = np.zeros((1, 14))
    xs0 = np.zeros((1, T + 1, 14))
    us0 = np.zeros((1, T, 7))
    return table, po, ref, x0, xs0, us0


def random_goal_problem(table, T, dt, B, seed, frame=None, rows="goal", timesteps=None):
    """Seeded random references / weights / warm start around a random posture."""
    nv = table.nv
    rng = np.random.default_rng(seed)
    if frame is None:
        frame = len(table.frame_names) - 1
    if isinstance(rows, tuple):
        running, terminal = rows  # the caller's own row tables
    elif rows == "goal":
        running, terminal = goal_reaching_rows(frame)
    elif rows == "collision":
        running, terminal = collision_avoidance_rows(table, frame, alpha=0.05)
    elif rows == "collision_exp":
        running, terminal = collision_avoidance_rows(table, frame, activation=_abi.ACT_EXP, alpha=0.1)
    else:
        running, terminal = regulation_rows()
    ts = [dt] * T if timesteps is None else list(timesteps)
    po = _abi.PackedOcp(nv, ts, running, terminal)
    ref = po.new_ref_tile(B)
    lo = np.maximum(table.lower_position_limit, -2.0)
    hi = np.minimum(table.upper_position_limit, 2.0)
    qc = rng.uniform(lo, hi, (B, 1, nv))
    for term, rws in ((False, running), (True, terminal)):
        n = 1 if term else T
        for i, r in enumerate(rws):
            wi, rr, aw = po.row_view(ref, term, i)
            wi[...] = rng.uniform(0.5, 2.0, (B, n))
            aw[...] = rng.uniform(0.1, 2.0, aw.shape)
            if r.kind == _abi.RES_STATE:
                rr[..., :nv] = qc + rng.normal(0, 0.05, (B, n, nv))
                rr[..., nv:] = rng.normal(0, 0.1, (B, n, nv))
            elif r.kind == _abi.RES_COLLISION:
                wi[...] = rng.uniform(0.05, 0.2, (B, n))  # w_collision_avoidance
            elif r.kind == _abi.RES_CONTROL:
                rr[...] = rng.normal(0, 2.0, rr.shape)
                aw[...] = rng.uniform(1e-3, 1e-2, aw.shape)
            elif r.kind == _abi.RES_FRAME_PLACEMENT:
                for b in range(B):
                    for t in range(n):
                        R = rt.rpy(*rng.uniform(-1.0, 1.0, 3))
                        rr[b, t, :9] = R.reshape(9)
                        rr[b, t, 9:] = rng.uniform(-0.5, 0.5, 3) + np.array([0.3, 0.0, 0.5])
            elif r.kind == _abi.RES_FRAME_TRANSLATION:
                rr[...] = rng.uniform(-0.5, 0.5, rr.shape) + np.array([0.3, 0.0, 0.5])
            elif r.kind == _abi.RES_FRAME_ROTATION:
                for b in range(B):
                    for t in range(n):
                        rr[b, t, :9] = rt.rpy(*rng.uniform(-1.0, 1.0, 3)).reshape(9)
    x0 = np.concatenate([qc[:, 0, :] + rng.normal(0, 0.02, (B, nv)), rng.normal(0, 0.1, (B, nv))], axis=1)
    xs = np.repeat(x0[:, None, :], T + 1, axis=1) + rng.normal(0, 0.01, (B, T + 1, 2 * nv))
    us = rng.normal(0, 1.0, (B, T, nv))
    return po, ref, x0, xs, us


def sine_batch_params(B: int, nv: int = 7, seed0: int = 1234, q0=None, lower=None, upper=None):
    """Per-instance sine-wave parameters: instance b draws from
    default_rng(seed0 + b): A_j ~ U(0.05, 0.2), period_j ~ U(2, 6) s, t0 ~ U(0, 4) s,
    q0 perturbed by N(0, 0.02^2) clipped to the joint limits.  Instance 0 is the
    reference's own test case (A 0.1, period 4, t0 0, unperturbed q0)."""
    q0 = PANDA_Q0 if q0 is None else np.asarray(q0, dtype=float)
    out_q0 = np.empty((B, nv))
    amp = np.empty((B, nv))
    puls = np.empty((B, nv))
    scale = np.full((B, nv), 0.2)
    t0 = np.empty(B)
    for b in range(B):
        rng = np.random.default_rng(seed0 + b)
        a = rng.uniform(0.05, 0.2, nv)
        period = rng.uniform(2.0, 6.0, nv)
        tt = rng.uniform(0.0, 4.0)
        qq = q0 + rng.normal(0.0, 0.02, nv)
        if b == 0:
            a, period, tt, qq = np.full(nv, 0.1), np.full(nv, 4.0), 0.0, q0.copy()
        if lower is not None:
            qq = np.clip(qq, lower + a, upper - a)
        out_q0[b], amp[b], puls[b], t0[b] = qq, a, 2.0 * np.pi / period, tt
    return out_q0, amp, puls, scale, t0


def generic_batch_arrays(B: int, n_points: int, dt: float, nv: int = 7, seed0: int = 1234, q0=None, accel=2.0):
    """q, dq, ddq [B][n_points][nv] as the reference's generic-trajectory test builds them
    (tests/test_generic_trajectory.py:147-160 upstream): uniform random accelerations integrated to
    dq and q with the Euler rule; instance b draws from default_rng(seed0 + b)."""
    q0 = np.tile(PANDA_Q0, (B, 1)) if q0 is None else np.asarray(q0, dtype=float).reshape(B, nv)
    ddq = np.empty((B, n_points, nv))
    for b in range(B):
        ddq[b] = accel * (np.random.default_rng(seed0 + b).random((n_points, nv)) - 0.5)
    dq = np.zeros_like(ddq)
    dq[:, 1:] = np.cumsum(ddq[:, :-1], axis=1) * dt
    q = np.empty_like(ddq)
    q[:, 0] = q0
    q[:, 1:] = q0[:, None, :] + np.cumsum(dq[:, 1:], axis=1) * dt
    return q, dq, ddq


def _log6_batch(R, p):
    """pinocchio.log6 of n placements (R [n,3,3], p [n,3]) -> [n,6] (linear | angular); the branch near
    theta = pi of log3 is not needed for inverse-kinematics errors and falls back to the scalar code."""
    from .se3 import SE3, log6

    n = R.shape[0]
    ct = 0.5 * (np.clip(np.trace(R, axis1=1, axis2=2), -1.0, 3.0) - 1.0)
    theta = np.arccos(ct)
    if np.any(theta >= np.pi - 1e-2):
        return np.stack([log6(SE3(R[i], p[i])).vector for i in range(n)])
    small = theta <= 1e-8
    tt = 0.5 * np.where(small, 1.0, theta / np.where(small, 1.0, np.sin(theta)))
    w = tt[:, None] * np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    t2 = np.einsum("ni,ni->n", w, w)
    t = np.sqrt(t2)
    tiny = t2 < 1e-12
    t2s, ts = np.where(tiny, 1.0, t2), np.where(tiny, 1.0, t)
    st, ctt = np.sin(ts), np.cos(ts)
    i22 = 1.0 / np.where(tiny, 1.0, 2.0 * (1.0 - ctt))
    alpha = np.where(tiny, 1.0 - t2 / 12.0, ts * st * i22)
    beta = np.where(tiny, 1.0 / 12.0 + t2 / 720.0, 1.0 / t2s - st / ts * i22)
    v = alpha[:, None] * p - 0.5 * np.cross(w, p) + (beta * np.einsum("ni,ni->n", w, p))[:, None] * w
    return np.concatenate([v, w], axis=1)


def cartesian_sine_batch_arrays(dyn, frame: int, n_points: int, dt: float, q0, amp_xyz, pulsation, scale_duration=0.2,
                                precision=1e-5, it_max=10000):
    """q, dq, ddq [B][n_points][nv] of B sine_wave_cartesian_space trajectories
    (trajectories/sine_wave_cartesian_space.py:62-111 upstream; tests/test_sin_wave_cartesian_space.py:58-62
    for the parameters): the end effector `frame` follows  p0 + A s(t) sin(w t)  with the initial
    orientation, joint positions from the iterative inverse kinematics  dq = -J' (J J')^-1 log6(des^-1 cur)
    (LOCAL Jacobian, warm-started from the previous point), joint velocities from the
    LOCAL_WORLD_ALIGNED Jacobian, zero accelerations.  The B instances advance in lockstep: forward
    kinematics and Jacobians of all of them in one device call (`dyn`: a backend.HipOcp)."""
    q0 = np.asarray(q0, dtype=float)
    B, nv = q0.shape
    amp = np.broadcast_to(np.asarray(amp_xyz, dtype=float), (B, 3))
    w = np.broadcast_to(np.asarray(pulsation, dtype=float), (B, 3))
    P0 = dyn.frame_placement(frame, q0)
    R0, p0 = P0[:, :9].reshape(B, 3, 3), P0[:, 9:]
    q = q0.copy()
    qs, dqs = np.empty((B, n_points, nv)), np.empty((B, n_points, nv))
    d = float(scale_duration)
    for i in range(n_points):
        t = i * dt
        s = min(max(t / d, 0.0), 1.0)
        quint = 10 * s**3 - 15 * s**4 + 6 * s**5
        dquint = (30 * s**2 - 60 * s**3 + 30 * s**4) / d if 0.0 < t < d else 0.0
        des_p = p0 + amp * quint * np.sin(w * t)
        des_v = np.zeros((B, 6))
        des_v[:, :3] = amp * (dquint * np.sin(w * t) + quint * w * np.cos(w * t))
        active = np.arange(B)
        for it in range(it_max + 2):
            P = dyn.frame_placement(frame, q[active])
            R, pp = P[:, :9].reshape(-1, 3, 3), P[:, 9:]
            Ra = R0[active]
            Rrel = np.einsum("nji,njk->nik", Ra, R)  # des^-1 * cur
            prel = np.einsum("nji,nj->ni", Ra, pp - des_p[active])
            err = _log6_batch(Rrel, prel)
            keep = np.linalg.norm(err, axis=1) >= precision
            active, err = active[keep], err[keep]
            if active.size == 0:
                break
            if it > it_max:  # upstream: `if i > it_max: break` after the convergence test
                raise RuntimeError(f"inverse kinematics failed to converge for instances {active.tolist()} at point {i}")
            J = dyn.frame_jacobian(frame, q[active], local=True)
            JJt = np.einsum("nij,nkj->nik", J, J)
            q[active] -= np.einsum("nji,nj->ni", J, np.linalg.solve(JJt, err[:, :, None])[:, :, 0])
        J = dyn.frame_jacobian(frame, q, local=False)
        JJt = np.einsum("nij,nkj->nik", J, J)
        dqs[:, i] = np.einsum("nji,nj->ni", J, np.linalg.solve(JJt, des_v[:, :, None])[:, :, 0])
        qs[:, i] = q
    return qs, dqs, np.zeros_like(qs)


def cartesian_sine_batch_params(B: int, seed0: int = 1234, lower=None, upper=None):
    """Per-instance parameters of the cartesian sine wave: instance 0 is the reference's test case
    (amplitude (0.1, 0.1, 0.0) m, period 4 s, q0 of the Panda tests); instance b draws from
    default_rng(seed0 + b): amplitude scaled by U(0.5, 1.2), period ~ U(2, 6) s per axis, q0 perturbed by
    N(0, 0.02^2) clipped to the joint limits."""
    q0 = np.empty((B, PANDA_Q0.size))
    amp, puls = np.empty((B, 3)), np.empty((B, 3))
    for b in range(B):
        rng = np.random.default_rng(seed0 + b)
        a = np.array([0.1, 0.1, 0.0]) * rng.uniform(0.5, 1.2)
        period = rng.uniform(2.0, 6.0, 3)
        qq = PANDA_Q0 + rng.normal(0.0, 0.02, PANDA_Q0.size)
        if b == 0:
            a, period, qq = np.array([0.1, 0.1, 0.0]), np.full(3, 4.0), PANDA_Q0.copy()
        if lower is not None:
            qq = np.clip(qq, lower, upper)
        q0[b], amp[b], puls[b] = qq, a, 2.0 * np.pi / period
    return q0, amp, puls


# Weights of the reference's sine-wave test (tests/test_sin_wave_configuration_space.py:138-144).
SINE_WEIGHTS = dict(w_q=1.0, w_qdot=0.1, w_effort=3e-4, w_pose=0.1)
