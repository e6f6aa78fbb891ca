"""Flat robot descriptions ("tables") consumed by the HIP path.

The reference obtains its model from a URDF through Pinocchio
(agimus_controller/agimus_controller/factory/robot_model.py:88-351).  Neither
Pinocchio nor any URDF exists in this environment, so the MI355X path takes the
same information as a plain table: per 1-DoF revolute joint its parent, fixed
placement, axis, and the spatial inertia of the body it carries; plus named
operational frames.  `from_pinocchio` converts a `pin.Model` when Pinocchio is
importable, so reference users keep their URDF workflow.

The Panda table uses the public Franka kinematics (URDF joint origins) and the
dynamic parameters identified by Gaz et al. (RA-L 2019) as commonly distributed
with franka_description, hand and locked fingers lumped into link 7.  SURVEY.md
section 8c expected these to differ from example-robot-data's inertials; they do
not: the table satisfies the dynamics of the reference's golden trajectory
(tests/resources/simple_ocp_croco_results.pkl) to 1e-11 relative and the golden
xs / us / K are reproduced on it to 1e-9 (tests/test_oracle_golden.py,
tests/test_hip_parity.py::test_golden_fixture_through_the_c_abi).
"""

from __future__ import annotations

import dataclasses

import numpy as np


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=float)


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=float)


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=float)


def rpy(r, p, y):
    return _rz(y) @ _ry(p) @ _rx(r)


def se3(R=None, p=None):
    out = np.zeros(12)
    out[:9] = (np.eye(3) if R is None else np.asarray(R, dtype=float)).reshape(9)
    out[9:] = 0.0 if p is None else np.asarray(p, dtype=float)
    return out


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=float)


def lump_inertia(m1, c1, I1, m2, c2, I2):
    """Combine two rigid bodies expressed in the same frame (I about own com)."""
    m = m1 + m2
    c = (m1 * np.asarray(c1) + m2 * np.asarray(c2)) / m
    out = np.zeros((3, 3))
    for mi, ci, Ii in ((m1, c1, I1), (m2, c2, I2)):
        d = np.asarray(ci) - c
        out += np.asarray(Ii) + mi * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, out


@dataclasses.dataclass
class RobotTable:
    name: str
    joint_names: list
    parent: np.ndarray  # [nv] int
    placement: np.ndarray  # [nv,12]
    axis: np.ndarray  # [nv,3]
    mass: np.ndarray  # [nv]
    com: np.ndarray  # [nv,3]
    inertia: np.ndarray  # [nv,9]
    armature: np.ndarray  # [nv]
    effort_limit: np.ndarray  # [nv]
    lower_position_limit: np.ndarray
    upper_position_limit: np.ndarray
    velocity_limit: np.ndarray
    frame_names: list
    frame_parent: np.ndarray  # [nframes] int
    frame_placement: np.ndarray  # [nframes,12]
    gravity: np.ndarray = dataclasses.field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    # collision geometry: a geometry object is a frame with a radius (> 0) and a half length
    # (capsule segment along the frame's z axis; 0 = sphere).  None = no frame carries geometry.
    frame_radius: np.ndarray | None = None
    frame_halflen: np.ndarray | None = None
    frame_box: np.ndarray | None = None  # [nframes,3] half extents of box geometry (coal.Box), zeros = not a box

    @property
    def nv(self) -> int:
        return int(self.parent.size)

    @property
    def nq(self) -> int:
        return self.nv

    def frame_id(self, name) -> int:
        if isinstance(name, (int, np.integer)):
            assert 0 <= int(name) < len(self.frame_names)
            return int(name)
        assert name in self.frame_names, f"Frame '{name}' does not exist!"
        return self.frame_names.index(name)

    def with_geometry(self, name, parent, placement12, radius=0.0, halflen=0.0, box=None) -> "RobotTable":
        """Copy with one more geometry frame: capsule (coal.Capsule(radius, halfLength), the shape
        factory/robot_model.py:261-302 converts cylinders to), sphere, or box (`box` = the three half
        extents; coal.Box, which the reference keeps as is), attached to joint `parent` (-1 = world,
        e.g. an obstacle of the environment)."""
        assert name not in self.frame_names, f"frame '{name}' exists"
        assert (box is None) != (float(radius) <= 0.0), "a geometry is either a box or has a radius"
        n = len(self.frame_names)
        rad = np.zeros(n) if self.frame_radius is None else np.asarray(self.frame_radius, dtype=float)
        hl = np.zeros(n) if self.frame_halflen is None else np.asarray(self.frame_halflen, dtype=float)
        bx = np.zeros((n, 3)) if self.frame_box is None else np.asarray(self.frame_box, dtype=float).reshape(n, 3)
        half = np.zeros(3) if box is None else np.asarray(box, dtype=float).reshape(3)
        assert box is None or np.all(half > 0.0), "box half extents must be positive"
        return dataclasses.replace(
            self,
            frame_names=list(self.frame_names) + [name],
            frame_parent=np.append(np.asarray(self.frame_parent, dtype=np.int32), np.int32(parent)),
            frame_placement=np.vstack([np.asarray(self.frame_placement, dtype=float).reshape(n, 12),
                                       np.asarray(placement12, dtype=float).reshape(1, 12)]),
            frame_radius=np.append(rad, float(radius)),
            frame_halflen=np.append(hl, float(halflen)),
            frame_box=np.vstack([bx, half.reshape(1, 3)]),
        )

    def with_armature(self, armature) -> "RobotTable":
        arm = np.broadcast_to(np.asarray(armature, dtype=float), (self.nv,)).copy()
        return dataclasses.replace(self, armature=arm)


def panda_table(armature=0.1) -> RobotTable:
    """7-DoF Franka Panda, fingers locked and lumped with the hand into link 7."""
    hp = np.pi / 2
    origins = [
        (rpy(0, 0, 0), [0, 0, 0.333]),
        (rpy(-hp, 0, 0), [0, 0, 0]),
        (rpy(hp, 0, 0), [0, -0.316, 0]),
        (rpy(hp, 0, 0), [0.0825, 0, 0]),
        (rpy(-hp, 0, 0), [-0.0825, 0.384, 0]),
        (rpy(hp, 0, 0), [0, 0, 0]),
        (rpy(hp, 0, 0), [0.088, 0, 0]),
    ]
    # mass, com, (ixx, iyy, izz, ixy, ixz, iyz) -- Gaz et al. 2019 as shipped in franka_description
    links = [
        (4.970684, [0.003875, 0.002081, -0.04762], (0.70337, 0.70661, 0.0091170, -0.00013900, 0.0067720, 0.019169)),
        (0.646926, [-0.003141, -0.02872, 0.003495], (0.0079620, 2.8110e-02, 2.5995e-02, -3.9250e-03, 1.0254e-02, 7.0400e-04)),
        (3.228604, [2.7518e-02, 3.9252e-02, -6.6502e-02], (3.7242e-02, 3.6155e-02, 1.0830e-02, -4.7610e-03, -1.1396e-02, -1.2805e-02)),
        (3.587895, [-5.317e-02, 1.04419e-01, 2.7454e-02], (2.5853e-02, 1.9552e-02, 2.8323e-02, 7.7960e-03, -1.3320e-03, 8.6410e-03)),
        (1.225946, [-1.1953e-02, 4.1065e-02, -3.8437e-02], (3.5549e-02, 2.9474e-02, 8.6270e-03, -2.1170e-03, -4.0370e-03, 2.2900e-04)),
        (1.666555, [6.0149e-02, -1.4117e-02, -1.0517e-02], (1.9640e-03, 4.3540e-03, 5.4330e-03, 1.0900e-04, -1.1580e-03, 3.4100e-04)),
        (7.35522e-01, [1.0517e-02, -4.252e-03, 6.1597e-02], (1.2516e-02, 1.0027e-02, 4.8150e-03, -4.2800e-04, -1.1960e-03, -7.4100e-04)),
    ]

    def full(i6):
        ixx, iyy, izz, ixy, ixz, iyz = i6
        return np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])

    mass, com, inertia = [], [], []
    for m, c, i6 in links:
        mass.append(m)
        com.append(np.array(c, dtype=float))
        inertia.append(full(i6))
    # hand (fixed to link 7 through link8: z 0.107, yaw -pi/4) and two locked fingers
    R_hand = rpy(0, 0, -np.pi / 4)
    p_hand = np.array([0.0, 0.0, 0.107])
    hand = (0.73, R_hand @ np.array([-0.01, 0.0, 0.03]) + p_hand, R_hand @ np.diag([0.001, 0.0025, 0.0017]) @ R_hand.T)
    finger_I = np.diag([2.375e-06, 2.375e-06, 7.5e-07])
    f1 = (0.015, R_hand @ np.array([0.0, 0.0, 0.0584]) + p_hand, finger_I)
    m7, c7, I7 = lump_inertia(mass[6], com[6], inertia[6], *hand)
    m7, c7, I7 = lump_inertia(m7, c7, I7, *f1)
    m7, c7, I7 = lump_inertia(m7, c7, I7, *f1)
    mass[6], com[6], inertia[6] = m7, c7, I7

    nv = 7
    frame_names = ["universe", "panda_link0"]
    frame_parent = [-1, -1]
    frame_placement = [se3(), se3()]
    for i in range(nv):
        frame_names += [f"panda_joint{i + 1}", f"panda_link{i + 1}"]
        frame_parent += [i, i]
        frame_placement += [se3(), se3()]
    frame_names += ["panda_link8", "panda_hand", "panda_hand_tcp"]
    frame_parent += [6, 6, 6]
    frame_placement += [
        se3(None, [0, 0, 0.107]),
        se3(R_hand, [0, 0, 0.107]),
        se3(R_hand, [0, 0, 0.107 + 0.1034]),
    ]
    return RobotTable(
        name="panda",
        joint_names=[f"panda_joint{i + 1}" for i in range(nv)],
        parent=np.arange(-1, nv - 1, dtype=np.int32),
        placement=np.stack([se3(R, p) for R, p in origins]),
        axis=np.tile(np.array([0.0, 0.0, 1.0]), (nv, 1)),
        mass=np.array(mass),
        com=np.stack(com),
        inertia=np.stack([i.reshape(9) for i in inertia]),
        armature=np.full(nv, float(armature)) if np.isscalar(armature) else np.asarray(armature, dtype=float),
        effort_limit=np.array([87.0, 87.0, 87.0, 87.0, 12.0, 12.0, 12.0]),
        lower_position_limit=np.array([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973]),
        upper_position_limit=np.array([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973]),
        velocity_limit=np.array([2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61]),
        frame_names=frame_names,
        frame_parent=np.array(frame_parent, dtype=np.int32),
        frame_placement=np.stack(frame_placement),
    )


# Capsules of the arm links (joint frame, segment along the capsule frame's z) and the capsule
# obstacle of the reference's test environment.  The link capsules are SYNTHETIC stand-ins for what
# factory/robot_model.py:261-302 derives from the URDF cylinders (the URDF is not available here):
# name -> (parent joint, placement R|p in the joint frame, radius, half length).
PANDA_CAPSULES = {
    "panda_link2_capsule_0": (1, se3(_rx(np.pi / 2), [0.0, -0.07, 0.0]), 0.06, 0.07),
    "panda_link3_capsule_0": (2, se3(None, [0.0, 0.0, -0.10]), 0.06, 0.08),
    "panda_link4_capsule_0": (3, se3(_rx(np.pi / 2), [-0.0825, 0.06, 0.0]), 0.06, 0.06),
    "panda_link5_capsule_0": (4, se3(None, [0.0, 0.03, -0.18]), 0.055, 0.12),
    "panda_link7_capsule_0": (6, se3(None, [0.0, 0.0, 0.12]), 0.05, 0.08),
}


def panda_collision_table(armature=0.1, obstacle_xyz=(1.535, 0.0, 0.43), obstacle_radius=0.1, obstacle_length=0.4,
                          obstacle_box=None) -> RobotTable:
    """panda_table plus link capsules and the capsule obstacle `obstacle` of the reference's
    tests/resources/environment.xacro:23-24 (xyz 1.535 0 0.43, direction x, radius 0.1, length 0.4).
    obstacle_box = (hx, hy, hz): a box obstacle of those half extents instead (world axes)."""
    t = panda_table(armature)
    for name, (parent, placement, radius, halflen) in PANDA_CAPSULES.items():
        t = t.with_geometry(name, parent, placement, radius, halflen)
    if obstacle_box is not None:
        return t.with_geometry("obstacle", -1, se3(None, list(obstacle_xyz)), box=obstacle_box)
    # capsule axis (local z) along world x
    return t.with_geometry("obstacle", -1, se3(_ry(np.pi / 2), list(obstacle_xyz)), obstacle_radius, obstacle_length / 2)


def _random_body(rng, mass_range=(0.5, 4.0), size=0.15):
    m = rng.uniform(*mass_range)
    c = rng.uniform(-size, size, 3)
    # inertia of a random box-like body: guarantees the triangle inequalities
    ext = rng.uniform(0.05, 0.3, 3)
    Ib = m / 12.0 * np.diag([ext[1] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[1] ** 2])
    Q = rpy(*rng.uniform(-np.pi, np.pi, 3))
    return m, c, Q @ Ib @ Q.T


def chain_table(nv: int, seed: int = 0, armature=0.1, name=None) -> RobotTable:
    """Seeded serial chain with random placements/axes (test model)."""
    rng = np.random.default_rng(seed)
    placement, axis, mass, com, inertia = [], [], [], [], []
    for i in range(nv):
        R = rpy(*rng.uniform(-np.pi, np.pi, 3))
        p = rng.uniform(-0.3, 0.3, 3)
        placement.append(se3(R, p))
        a = rng.normal(size=3)
        axis.append(a / np.linalg.norm(a))
        m, c, Ii = _random_body(rng)
        mass.append(m)
        com.append(c)
        inertia.append(Ii.reshape(9))
    frame_names = ["universe"] + [f"joint{i}" for i in range(nv)] + ["tool"]
    frame_parent = [-1] + list(range(nv)) + [nv - 1]
    frame_placement = [se3()] + [se3() for _ in range(nv)] + [se3(rpy(0.3, -0.2, 0.5), [0.05, -0.02, 0.12])]
    return RobotTable(
        name=name or f"chain{nv}",
        joint_names=[f"joint{i}" for i in range(nv)],
        parent=np.arange(-1, nv - 1, dtype=np.int32),
        placement=np.stack(placement),
        axis=np.stack(axis),
        mass=np.array(mass),
        com=np.stack(com),
        inertia=np.stack(inertia),
        armature=np.full(nv, float(armature)),
        effort_limit=np.full(nv, 100.0),
        lower_position_limit=np.full(nv, -np.pi),
        upper_position_limit=np.full(nv, np.pi),
        velocity_limit=np.full(nv, 3.0),
        frame_names=frame_names,
        frame_parent=np.array(frame_parent, dtype=np.int32),
        frame_placement=np.stack(frame_placement),
    )


def tree_table(nv: int, seed: int = 0, armature=0.1, branching=0.35, name=None) -> RobotTable:
    """Seeded random kinematic TREE (test model for arbitrary model sizes, e.g. a Panda with unlocked fingers or a mobile
    manipulator, factory/robot_model.py:231-257 upstream): joint i hangs on its predecessor or, with probability `branching`,
    on a random earlier joint.  Frames: one per joint plus a tool on the last joint and one on a mid-tree joint."""
    rng = np.random.default_rng(seed)
    base = chain_table(nv, seed=seed + 1000, armature=armature)
    parent = np.empty(nv, dtype=np.int32)
    for i in range(nv):
        parent[i] = i - 1 if (i < 2 or rng.uniform() > branching) else int(rng.integers(0, i - 1))
    if nv > 2 and np.array_equal(parent, np.arange(nv) - 1):
        parent[nv - 1] = 0  # make sure it is not a serial chain
    frame_names = list(base.frame_names) + ["tool_b"]
    frame_parent = np.concatenate([base.frame_parent, [nv // 2]]).astype(np.int32)
    frame_placement = np.concatenate([base.frame_placement, se3(rpy(-0.2, 0.4, 0.1), [0.03, 0.04, -0.06])[None, :]])
    import dataclasses

    return dataclasses.replace(base, name=name or f"tree{nv}", parent=parent, frame_names=frame_names, frame_parent=frame_parent,
                               frame_placement=frame_placement)


def pendulum_table(length=1.0, mass=1.0, armature=0.0) -> RobotTable:
    """Point mass on a massless rod rotating about world Y: tau = m l^2 qdd + m g l sin(q)."""
    return RobotTable(
        name="pendulum",
        joint_names=["pivot"],
        parent=np.array([-1], dtype=np.int32),
        placement=se3()[None, :],
        axis=np.array([[0.0, 1.0, 0.0]]),
        mass=np.array([mass]),
        com=np.array([[0.0, 0.0, -length]]),
        inertia=np.zeros((1, 9)),
        armature=np.array([float(armature)]),
        effort_limit=np.array([10.0]),
        lower_position_limit=np.array([-np.pi]),
        upper_position_limit=np.array([np.pi]),
        velocity_limit=np.array([10.0]),
        frame_names=["universe", "pivot", "bob"],
        frame_parent=np.array([-1, 0, 0], dtype=np.int32),
        frame_placement=np.stack([se3(), se3(), se3(None, [0, 0, -length])]),
    )


def humanoid30_table(seed: int = 7, armature=0.1) -> RobotTable:
    """Synthetic fixed-base 30-DoF humanoid-like tree (BASELINE.json config 5):
    torso 3, head 1, two 7-DoF arms, two 6-DoF legs; seeded inertias."""
    rng = np.random.default_rng(seed)
    ax = {"x": [1.0, 0, 0], "y": [0, 1.0, 0], "z": [0, 0, 1.0]}
    names, parent, placement, axis = [], [], [], []

    def add(name, par, p, a):
        names.append(name)
        parent.append(par)
        placement.append(se3(None, p))
        axis.append(ax[a])
        return len(names) - 1

    t0 = add("torso_yaw", -1, [0, 0, 1.0], "z")
    t1 = add("torso_pitch", t0, [0, 0, 0.1], "y")
    t2 = add("torso_roll", t1, [0, 0, 0.1], "x")
    add("head_yaw", t2, [0, 0, 0.35], "z")
    for side, sy in (("l", 1.0), ("r", -1.0)):
        j = add(f"{side}_shoulder_pitch", t2, [0, 0.2 * sy, 0.25], "y")
        j = add(f"{side}_shoulder_roll", j, [0, 0.05 * sy, 0], "x")
        j = add(f"{side}_shoulder_yaw", j, [0, 0, -0.1], "z")
        j = add(f"{side}_elbow", j, [0.02, 0, -0.2], "y")
        j = add(f"{side}_wrist_yaw", j, [0, 0, -0.15], "z")
        j = add(f"{side}_wrist_pitch", j, [0, 0, -0.1], "y")
        add(f"{side}_wrist_roll", j, [0, 0, -0.05], "x")
    for side, sy in (("l", 1.0), ("r", -1.0)):
        j = add(f"{side}_hip_yaw", t0, [0, 0.1 * sy, -0.1], "z")
        j = add(f"{side}_hip_roll", j, [0, 0, -0.05], "x")
        j = add(f"{side}_hip_pitch", j, [0, 0, -0.05], "y")
        j = add(f"{side}_knee", j, [0, 0, -0.4], "y")
        j = add(f"{side}_ankle_pitch", j, [0, 0, -0.4], "y")
        add(f"{side}_ankle_roll", j, [0, 0, -0.03], "x")
    nv = len(names)
    assert nv == 30
    mass, com, inertia = [], [], []
    for _ in range(nv):
        m, c, Ii = _random_body(rng, (0.4, 5.0), 0.08)
        mass.append(m)
        com.append(c)
        inertia.append(Ii.reshape(9))
    frame_names = ["universe"] + names + ["l_hand", "r_hand"]
    frame_parent = [-1] + list(range(nv)) + [names.index("l_wrist_roll"), names.index("r_wrist_roll")]
    frame_placement = [se3()] + [se3() for _ in range(nv)] + [se3(None, [0, 0, -0.08]), se3(None, [0, 0, -0.08])]
    return RobotTable(
        name="humanoid30",
        joint_names=names,
        parent=np.array(parent, dtype=np.int32),
        placement=np.stack(placement),
        axis=np.array(axis, dtype=float),
        mass=np.array(mass),
        com=np.stack(com),
        inertia=np.stack(inertia),
        armature=np.full(nv, float(armature)),
        effort_limit=np.full(nv, 150.0),
        lower_position_limit=np.full(nv, -2.5),
        upper_position_limit=np.full(nv, 2.5),
        velocity_limit=np.full(nv, 6.0),
        frame_names=frame_names,
        frame_parent=np.array(frame_parent, dtype=np.int32),
        frame_placement=np.stack(frame_placement),
    )


def from_pinocchio(model, armature=None) -> RobotTable:  # pragma: no cover - needs pinocchio
    """Extract a table from a `pinocchio.Model` made of 1-DoF revolute joints."""
    nv = model.nv
    parent, placement, axis, mass, com, inertia = [], [], [], [], [], []
    for j in range(1, model.njoints):
        jm = model.joints[j]
        assert jm.nv == 1 and jm.nq == 1, "only 1-DoF revolute joints are supported"
        short = jm.shortname()
        ax = {"JointModelRX": [1.0, 0, 0], "JointModelRY": [0, 1.0, 0], "JointModelRZ": [0, 0, 1.0]}.get(short)
        if ax is None:
            ax = list(np.asarray(jm.extract().axis).reshape(3))
        parent.append(int(model.parents[j]) - 1)
        M = model.jointPlacements[j]
        placement.append(se3(np.asarray(M.rotation), np.asarray(M.translation)))
        axis.append(ax)
        Y = model.inertias[j]
        mass.append(float(Y.mass))
        com.append(np.asarray(Y.lever).reshape(3))
        inertia.append(np.asarray(Y.inertia).reshape(9))
    frame_names, frame_parent, frame_placement = [], [], []
    for f in model.frames:
        frame_names.append(f.name)
        frame_parent.append(int(f.parentJoint) - 1)
        frame_placement.append(se3(np.asarray(f.placement.rotation), np.asarray(f.placement.translation)))
    arm = np.zeros(nv) if armature is None else np.asarray(armature, dtype=float)
    return RobotTable(
        name=model.name,
        joint_names=list(model.names)[1:],
        parent=np.array(parent, dtype=np.int32),
        placement=np.stack(placement),
        axis=np.array(axis, dtype=float),
        mass=np.array(mass),
        com=np.stack(com),
        inertia=np.stack(inertia),
        armature=arm,
        effort_limit=np.asarray(model.effortLimit, dtype=float),
        lower_position_limit=np.asarray(model.lowerPositionLimit, dtype=float),
        upper_position_limit=np.asarray(model.upperPositionLimit, dtype=float),
        velocity_limit=np.asarray(model.velocityLimit, dtype=float),
        frame_names=frame_names,
        frame_parent=np.array(frame_parent, dtype=np.int32),
        frame_placement=np.stack(frame_placement),
        gravity=np.asarray(model.gravity.linear, dtype=float),
    )
