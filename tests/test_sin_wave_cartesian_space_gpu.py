"""SinusWaveCartesianSpace, written like the reference's tests/test_sin_wave_cartesian_space.py
(trajectory reached through the inverse kinematics, 3-D mask variant, and its two IK tests -- the
6-D one carries a GOLDEN joint velocity of the Panda kinematics, atol 1e-6 upstream)."""
import numpy as np
import pytest

from agimus_controller_amd import _abi, workloads
from agimus_controller_amd.factory import robot_tables as rt
from agimus_controller_amd.factory.robot_model import panda_robot_models
from agimus_controller_amd.trajectories.sine_wave_cartesian_space import SinusWaveCartesianSpace
from agimus_controller_amd.trajectories.sine_wave_params import SinWaveParams
from agimus_controller_amd.workloads import PANDA_Q0

pytestmark = pytest.mark.gpu

PARAMS = dict(w_q=np.array([1.0]), w_qdot=np.array([0.1]), w_qddot=np.array([0.000001]), w_robot_effort=np.array([0.0003]),
              w_pose=np.array([0.1]), ee_frame_name="panda_hand_tcp")


@pytest.fixture(scope="module")
def dyn(hip_backend):
    table = rt.panda_table(0.1)
    running, terminal = workloads.goal_reaching_rows(table.frame_id("panda_hand_tcp"))
    h = hip_backend.HipOcp(table, _abi.PackedOcp(7, [0.01] * 4, running, terminal), 1)
    yield h
    h.close()


def make(mask):
    sp = SinWaveParams(amplitude=np.array([0.1, 0.1, 0.0]), period=np.array([4.0, 4.0, 4.0]), scale_duration=np.array([0.2, 0.2, 0.2]))
    return sp, SinusWaveCartesianSpace(sine_wave_params=sp, mask=mask, **PARAMS)


def test_frame_jacobian_against_finite_differences(dyn):
    tcp = 0
    table = rt.panda_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    q = PANDA_Q0 + 0.1
    J = dyn.frame_jacobian(tcp, q, local=False)[0]
    Jl = dyn.frame_jacobian(tcp, q, local=True)[0]
    M0 = dyn.frame_placement(tcp, q)[0]
    R0 = M0[:9].reshape(3, 3)
    h = 1e-6
    for j in range(7):
        e = np.zeros(7)
        e[j] = h
        Mp, Mm = dyn.frame_placement(tcp, q + e)[0], dyn.frame_placement(tcp, q - e)[0]
        lin = (Mp[9:] - Mm[9:]) / (2 * h)
        dR = (Mp[:9].reshape(3, 3) - Mm[:9].reshape(3, 3)) / (2 * h)
        W = dR @ R0.T  # [omega]x in world axes
        ang = np.array([W[2, 1], W[0, 2], W[1, 0]])
        np.testing.assert_allclose(J[:3, j], lin, atol=1e-8)
        np.testing.assert_allclose(J[3:, j], ang, atol=1e-8)
        np.testing.assert_allclose(Jl[:3, j], R0.T @ lin, atol=1e-8)
        np.testing.assert_allclose(Jl[3:, j], R0.T @ ang, atol=1e-8)


@pytest.mark.parametrize("mask,ncmp", [([True] * 6, 7), ([True, True, True, False, False, False], 3)])
def test_sin_wave_cartesian_space_trajectory(dyn, mask, ncmp):
    sp, obj = make(mask)
    obj.initialize(panda_robot_models().robot_model, PANDA_Q0, dyn)
    dt = 1e-1
    duration = np.max(sp.scale_duration) + 2 * np.max(sp.period)
    times = np.linspace(0, duration, int(duration / dt))
    traj = [obj.get_traj_point_at_t(t) for t in times]
    for traj_point in traj:
        ik_ee_pos = obj.get_end_effector_pose_from_q(traj_point.point.robot_configuration)
        ee_pos = traj_point.point.end_effector_poses["panda_hand_tcp"].copy()
        np.testing.assert_allclose(ik_ee_pos[:ncmp], ee_pos[:ncmp], atol=1e-3)
    # the end effector really moves by the amplitude in x and y and not in z
    xyz = np.array([p.point.end_effector_poses["panda_hand_tcp"][:3] for p in traj])
    assert np.ptp(xyz[:, 0]) > 0.15 and np.ptp(xyz[:, 1]) > 0.15 and np.ptp(xyz[:, 2]) < 1e-12


def test_ik_6D_golden_joint_velocity(dyn):
    _, obj = make([True] * 6)
    obj.initialize(panda_robot_models().robot_model, PANDA_Q0 + np.array(7 * [0.1]), dyn)
    ee_pos = obj.get_end_effector_pose_from_q_as_se3(PANDA_Q0)
    ik_q, ik_dq = obj.inverse_kinematics(ee_pos, np.array([0.1, 0.2, 0.3, 0.0, 0.0, 0.0]), precision=1e-4)
    ik_ee_pos = obj.get_end_effector_pose_from_q_as_se3(ik_q)
    # golden vector of the reference (tests/test_sin_wave_cartesian_space.py:210-217)
    np.testing.assert_allclose(ik_dq, -np.array([0.640289, -0.419278, 0.146452, -1.156815, 0.21497, 0.43003, 0.108381]), atol=1e-6)
    np.testing.assert_allclose(ik_ee_pos.homogeneous, ee_pos.homogeneous, atol=1e-3)


def test_ik_3D(dyn):
    _, obj = make([True, True, True, False, False, False])
    obj.initialize(panda_robot_models().robot_model, PANDA_Q0 + np.array(7 * [0.1]), dyn)
    ee_pos = obj.get_end_effector_pose_from_q_as_se3(PANDA_Q0)
    ik_q, ik_dq = obj.inverse_kinematics(ee_pos, np.zeros(6))
    ik_ee_pos = obj.get_end_effector_pose_from_q_as_se3(ik_q)
    np.testing.assert_allclose(ik_dq, np.zeros(7), atol=1e-3)
    np.testing.assert_allclose(ik_ee_pos.translation, ee_pos.translation, atol=1e-3)


def test_batch_arrays_match_the_trajectory_class(dyn):
    """workloads.cartesian_sine_batch_arrays (lockstep inverse kinematics of B instances, used by
    bench.py --workload cartesian) against SinusWaveCartesianSpace point by point."""
    table = rt.panda_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    q0, amp, puls = workloads.cartesian_sine_batch_params(3, lower=table.lower_position_limit, upper=table.upper_position_limit)
    n, dt = 40, 0.01
    qs, dqs, ddqs = workloads.cartesian_sine_batch_arrays(dyn, tcp, n, dt, q0, amp, puls)
    assert not np.any(ddqs)
    for b in (0, 2):
        sp = SinWaveParams(amplitude=amp[b], period=2.0 * np.pi / puls[b], scale_duration=np.array([0.2, 0.2, 0.2]))
        obj = SinusWaveCartesianSpace(sine_wave_params=sp, **PARAMS)
        obj.initialize(panda_robot_models().robot_model, q0[b], dyn)
        for i in range(n):
            pt = obj.get_traj_point_at_t(i * dt).point
            np.testing.assert_allclose(qs[b, i], pt.robot_configuration, atol=1e-9)
            np.testing.assert_allclose(dqs[b, i], pt.robot_velocity, atol=1e-8)
    # the end effector of every instance follows its own sine: x moves, z stays
    P = dyn.frame_placement(tcp, qs[1])
    assert np.ptp(P[:, 9]) > 1e-3 and np.ptp(P[:, 11]) < 1e-4


def test_device_generator_matches_the_lockstep_inverse_kinematics(dyn):
    """agx_traj_cartesian_sine_create (every instance's inverse kinematics, point after point, in one kernel) against
    workloads.cartesian_sine_batch_arrays -- which the test above ties to the SinusWaveCartesianSpace class and
    through it to the reference's golden joint velocity."""
    from agimus_controller_amd import _abi, backend

    table = rt.panda_table(0.1)
    tcp = table.frame_id("panda_hand_tcp")
    B, T, n, dt = 5, 8, 60, 0.01
    q0, amp, puls = workloads.cartesian_sine_batch_params(B, lower=table.lower_position_limit, upper=table.upper_position_limit)
    qs, dqs, _ = workloads.cartesian_sine_batch_arrays(dyn, tcp, n, dt, q0, amp, puls)
    running, terminal = workloads.goal_reaching_rows(tcp)
    hip = backend.HipOcp(table, _abi.PackedOcp(table.nv, [dt] * T, running, terminal), B)
    hip.cartesian_sine_trajectory(n, dt, q0, amp, puls, 1.0, 0.1, 3e-4, 0.1, tcp)
    P0 = dyn.frame_placement(tcp, q0)
    for k in (0, 1, 7, 23, n - 1):
        q, v, a, u, pose = hip.traj_point(k)
        np.testing.assert_allclose(q, qs[:, k], atol=1e-9)
        np.testing.assert_allclose(v, dqs[:, k], atol=1e-8)
        assert not np.any(a)
        # the pose reference of the point is the DESIRED pose ee_des_pos (sine_wave_cartesian_space.py:126-133 upstream): the initial
        # orientation and p0 + amp s(t) sin(w t) -- the pose at the inverse-kinematics solution differs by up to `precision`
        t = k * dt
        s = min(max(t / 0.2, 0.0), 1.0)
        quint = 10 * s**3 - 15 * s**4 + 6 * s**5
        np.testing.assert_allclose(pose[:, 9:], P0[:, 9:] + amp * quint * np.sin(puls * t), atol=1e-13)
        np.testing.assert_allclose(pose[:, :9], P0[:, :9], atol=1e-15)
        assert np.all(np.linalg.norm(pose[:, 9:] - dyn.frame_placement(tcp, q)[:, 9:], axis=1) < 1e-5)
    # an unreachable target is reported, not returned
    far = amp.copy()
    far[2] = [5.0, 0.0, 0.0]
    with pytest.raises(backend.HipError, match="inverse kinematics failed to converge: instance 2"):
        hip.cartesian_sine_trajectory(n, dt, q0, far, puls, 1.0, 0.1, 3e-4, 0.1, tcp, it_max=50)
    # ... and leaves no half-built trajectory behind for a later MPC step to consume
    with pytest.raises(backend.HipError, match="no resident trajectory"):
        hip.set_window(0)
    hip.close()
