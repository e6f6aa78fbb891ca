"""Collision-distance residual + colmpc-style activations.

The reference goes through colmpc / coal (absent here, no fixture covers them): parity with the
reference binaries is UNPINNED for these rows.  What is pinned: closed-form distances, the analytic
Jacobian against finite differences (CPU checker), and the HIP kernels against the CPU checker.
"""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt


def _sphere_table():
    """one revolute joint about z at the origin carrying a sphere at radius 1 m; a world sphere."""
    t = rt.pendulum_table()
    t = t.with_geometry("ball", 0, rt.se3(None, [0.0, 0.0, -1.0]), 0.1, 0.0)
    return t.with_geometry("post", -1, rt.se3(None, [0.0, 1.5, 0.0]), 0.2, 0.0)


def _oracle(table, po, B=1):
    from oracle.oracle import Oracle

    return Oracle(table, po, B)


def test_sphere_sphere_distance_closed_form():
    t = _sphere_table()
    rows = [_abi.RowSpec(_abi.RES_COLLISION, activation=_abi.ACT_WEIGHTED_QUAD, frame=t.frame_id("ball"), frame_b=t.frame_id("post"))]
    po = _abi.PackedOcp(1, [0.01], rows, rows)
    o = _oracle(t, po)
    ref = po.new_ref_tile(1)[0, 0]
    for q in (0.0, 0.4, -1.1, 2.0):
        # pendulum_table: joint axis y, link along -z  -> ball centre = Ry(q) (0,0,-1)
        c = rt._ry(q) @ np.array([0.0, 0.0, -1.0])
        want = np.linalg.norm(c - np.array([0.0, 1.5, 0.0])) - 0.3
        _, cost, res = o.node_calc(True, 0.0, np.array([q, 0.0]), None, ref)
        assert res[0] == pytest.approx(want, abs=1e-12)
        assert cost == pytest.approx(0.5 * want * want, abs=1e-12)


def test_capsule_capsule_cases():
    """parallel, crossing and end-cap configurations of two world capsules against hand results."""
    base = rt.pendulum_table()
    cases = [
        # (placement a, halflen a, placement b, halflen b, centre distance)
        (rt.se3(None, [0, 0, 0]), 0.5, rt.se3(None, [1.0, 0, 0]), 0.5, 1.0),  # parallel, side by side
        (rt.se3(None, [0, 0, 0]), 0.5, rt.se3(rt._ry(np.pi / 2), [0, 0.7, 0]), 0.5, 0.7),  # crossing at right angle
        (rt.se3(None, [0, 0, 0]), 0.5, rt.se3(None, [0, 0, 2.0]), 0.5, 1.0),  # collinear, end to end
        (rt.se3(None, [0, 0, 0]), 0.5, rt.se3(None, [0.3, 0.4, 1.5]), 0.25, np.sqrt(0.25 + 0.75**2)),  # end cap to end cap
    ]
    for pa, ha, pb, hb, want in cases:
        t = base.with_geometry("a", -1, pa, 0.05, ha).with_geometry("b", -1, pb, 0.07, hb)
        rows = [_abi.RowSpec(_abi.RES_COLLISION, frame=t.frame_id("a"), frame_b=t.frame_id("b"))]
        po = _abi.PackedOcp(1, [0.01], rows, rows)
        o = _oracle(t, po)
        _, _, res = o.node_calc(True, 0.0, np.zeros(2), None, po.new_ref_tile(1)[0, 0])
        assert res[0] == pytest.approx(want - 0.12, abs=1e-12)


def test_box_against_sphere_and_capsule_cases():
    """coal.Box geometry (kept by factory/robot_model.py:296-302) against spheres / capsules: face,
    edge, corner, rotated-box and penetrating configurations against hand results."""
    base = rt.pendulum_table()
    half = (0.5, 0.3, 0.2)
    rz45 = rt.rpy(0.0, 0.0, np.pi / 4)
    cases = [
        # (box placement, other placement, radius, halflen, signed distance)
        (rt.se3(None, [0, 0, 0]), rt.se3(None, [2.0, 0, 0]), 0.1, 0.0, 1.5 - 0.1),  # sphere, face
        (rt.se3(None, [0, 0, 0]), rt.se3(None, [1.5, 1.3, 0]), 0.1, 0.0, np.sqrt(2.0) - 0.1),  # sphere, edge
        (rt.se3(None, [0, 0, 0]), rt.se3(None, [1.5, 1.3, 1.2]), 0.1, 0.0, np.sqrt(3.0) - 0.1),  # sphere, corner
        (rt.se3(None, [0, 0, 0]), rt.se3(None, [0.4, 0.0, 0.0]), 0.05, 0.0, -0.1 - 0.05),  # sphere centre inside: nearest face x
        (rt.se3(None, [0, 0, 0]), rt.se3(None, [1.0, 0, 0]), 0.1, 0.6, 0.5 - 0.1),  # capsule parallel to a face (axis z)
        (rt.se3(None, [0, 0, 0]), rt.se3(rt._ry(np.pi / 2), [0.2, 0, 1.0]), 0.1, 0.5, 0.8 - 0.1),  # capsule along x above the top face
        (rt.se3(None, [0, 0, 0]), rt.se3(rt._ry(np.pi / 2), [2.0, 0, 0]), 0.1, 0.5, 1.0 - 0.1),  # capsule end cap to the x face
        (rt.se3(rz45, [0, 0, 0]), rt.se3(None, [2.0, 0, 0]), 0.1, 0.0, np.hypot(2.0 - 0.8 / np.sqrt(2.0), 0.2 / np.sqrt(2.0)) - 0.1),  # rotated box: vertical edge at R (0.5, -0.3)
    ]
    for swap in (False, True):
        for pb, po_, rad, hl, want in cases:
            t = base.with_geometry("box", -1, pb, box=half).with_geometry("other", -1, po_, rad, hl)
            a, b = ("other", "box") if swap else ("box", "other")
            rows = [_abi.RowSpec(_abi.RES_COLLISION, frame=t.frame_id(a), frame_b=t.frame_id(b))]
            po = _abi.PackedOcp(1, [0.01], rows, rows)
            o = _oracle(t, po)
            _, _, res = o.node_calc(True, 0.0, np.zeros(2), None, po.new_ref_tile(1)[0, 0])
            assert res[0] == pytest.approx(want, abs=1e-9), (swap, want)


@pytest.mark.parametrize("rows,box", [("collision", None), ("collision_exp", None), ("collision", (0.1, 0.15, 0.08))])
def test_collision_cost_gradient_matches_finite_differences(rows, box):
    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.45, 0.1, 0.45), obstacle_radius=0.08, obstacle_length=0.3, obstacle_box=box)
    tcp = table.frame_id("panda_hand_tcp")
    po, ref, x0, xs, us = workloads.random_goal_problem(table, 3, 0.01, 2, 11, frame=tcp, rows=rows)
    o = _oracle(table, po, 2)
    sl = _abi.tile_slices(7)
    for b in range(2):
        x, u, r = xs[b, 1], us[b, 1], ref[b, 1]
        tile, _, _ = o.node_calc_diff(False, 0.01, x, u, r)
        Lx = tile[sl["Lx"]]
        h = 1e-6
        for i in range(7):
            e = np.zeros(14)
            e[i] = h
            cp = o.node_calc(False, 0.01, x + e, u, r)[1]
            cm = o.node_calc(False, 0.01, x - e, u, r)[1]
            assert Lx[i] == pytest.approx((cp - cm) / (2 * h), rel=2e-5, abs=1e-9)


def test_yaml_lowering_of_the_collision_avoidance_definition():
    """ocp_traj_tracking_collision_avoidance.yaml-shaped cost list lowers to the row table (costs only)."""
    from agimus_controller_amd.ocp import ocp_croco_generic as g
    from agimus_controller_amd.factory.robot_model import RobotModelParameters, RobotModels

    table = rt.panda_collision_table(0.1)
    rm = RobotModels(RobotModelParameters(table=table, armature=table.armature,
                                          collision_pairs=[("panda_link7_capsule_0", "obstacle")]))
    diff = g.create_croco_dataclasses({
        "class": "DifferentialActionModelFreeFwdDynamics",
        "costs": [
            {"name": "distance", "update": False, "weight": 1.0,
             "cost": {"class": "CostModelResidual",
                      "activation": {"class": "ActivationModelQuadExp", "alpha": "1e-4"},
                      "residual": {"class": "ResidualDistanceCollision", "collision_pair_id": 0}}},
            {"name": "distance_named", "weight": 2.0,
             "cost": {"class": "CostModelResidual",
                      "activation": {"class": "ActivationModelExp", "alpha": 0.5},
                      "residual": {"class": "ResidualDistanceCollision",
                                   "collision_pair": ["panda_link5_capsule_0", "obstacle"]}}},
        ],
    })
    rows = diff.lower(g.BuildData(rm.robot_model, 7, rm.collision_model))
    assert [r.kind for r in rows] == [_abi.RES_COLLISION] * 2
    assert rows[0].activation == _abi.ACT_QUAD_EXP and rows[0].alpha == pytest.approx(1e-4)
    assert rows[1].activation == _abi.ACT_EXP and rows[1].alpha == pytest.approx(0.5)
    assert (rows[0].frame, rows[0].frame_b) == (table.frame_id("panda_link7_capsule_0"), table.frame_id("obstacle"))
    assert (rows[1].frame, rows[1].frame_b) == (table.frame_id("panda_link5_capsule_0"), table.frame_id("obstacle"))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,box", [("collision", None), ("collision_exp", None), ("collision", (0.1, 0.15, 0.08))])
def test_hip_collision_tiles_and_solve_match_the_checker(rows, box):
    from agimus_controller_amd import backend

    table = rt.panda_collision_table(0.1, obstacle_xyz=(0.45, 0.1, 0.45), obstacle_radius=0.08, obstacle_length=0.3, obstacle_box=box)
    tcp = table.frame_id("panda_hand_tcp")
    B, T = 6, 12
    po, ref, x0, xs, us = workloads.random_goal_problem(table, T, 0.01, B, 23, frame=tcp, rows=rows)
    o = _oracle(table, po, B)
    hb = backend.HipOcp(table, po, B)
    hb.set_refs(ref)
    hb.upload_x0(x0)
    hb.upload_warmstart(xs, us)
    tiles = hb.calc_diff()
    want = o.calc_diff(ref, None, xs, us)
    sl = _abi.tile_slices(7)
    for name in ("Lx", "Lxx", "cost", "Lu", "Luu", "Fx", "Fu"):
        np.testing.assert_allclose(tiles[..., sl[name]], want[..., sl[name]], rtol=1e-9, atol=1e-10, err_msg=name)
    # residual read-back of the distance row (publish_residual path)
    d_hip = hb.residuals(3)
    assert d_hip.shape[-1] == 1
    for b in range(2):
        _, _, res = o.node_calc(False, 0.01, xs[b, 2], us[b, 2], ref[b, 2])
        assert d_hip[b, 2, 0] == pytest.approx(res[21 + 6], abs=1e-10)
    # full solve
    xs_h, us_h, K_h, st_h = hb.solve(x0, xs, us, 20)
    xs_o, us_o, K_o, st_o = o.solve(ref, None, x0, xs, us, 20)
    assert np.array_equal(st_h["iter"], st_o["iter"])
    np.testing.assert_allclose(xs_h, xs_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(us_h, us_o, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(K_h, K_o, rtol=1e-5, atol=1e-5)
    hb.close()


@pytest.mark.gpu
def test_hip_moving_obstacle_placement_update():
    """update_geometry_placement (ocp_base_croco.py:110-132): distances follow the new obstacle pose."""
    from agimus_controller_amd import backend

    table = rt.panda_collision_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    po, ref, x0, xs, us = workloads.random_goal_problem(table, 4, 0.01, 2, 5, frame=tcp, rows="collision")
    hb = backend.HipOcp(table, po, 2)
    hb.set_refs(ref)
    hb.upload_x0(x0)
    hb.upload_warmstart(xs, us)
    d0 = hb.residuals(3).copy()
    new = rt.se3(rt._ry(np.pi / 2), [0.5, 0.0, 0.4])
    hb.set_geom_placement(table.frame_id("obstacle"), new)
    d1 = hb.residuals(3)
    moved = table.with_geometry("obstacle2", -1, new, 0.1, 0.2)
    rows = [_abi.RowSpec(_abi.RES_COLLISION, frame=moved.frame_id("panda_link7_capsule_0"), frame_b=moved.frame_id("obstacle2"))]
    po2 = _abi.PackedOcp(7, [0.01], rows, rows)
    o = _oracle(moved, po2)
    for b in range(2):
        _, _, res = o.node_calc(True, 0.0, xs[b, 1], None, po2.new_ref_tile(1)[0, 0])
        assert d1[b, 1, 0] == pytest.approx(res[0], abs=1e-10)
    assert not np.allclose(d0, d1)
    hb.close()
