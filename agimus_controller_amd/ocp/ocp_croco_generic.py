"""YAML-defined OCP on the MI355X path.

Accepts the reference's OCP definition files unchanged
(agimus_controller/agimus_controller/ocp/ocp_croco_generic.py:33-761: a tree of `class:`-tagged
mappings whose class names are Crocoddyl's) and exposes the same `OCPCrocoGeneric` surface
(:764-897).  Where the reference materialises Crocoddyl objects, each cost item is lowered here
to one row of the per-node cost table consumed by the HIP kernels, and `update()` writes the
node's reference/weights into the reference tile that is uploaded once per MPC step.
"""

from __future__ import annotations

import dataclasses
import pathlib
import typing as T

import numpy as np
import yaml

from .. import _abi
from ..factory.robot_model import RobotModels
from ..mpc_data import OCPResults
from ..ocp_base_croco import OCPBaseCroco
from ..ocp_param_base import OCPParamsBaseCroco
from ..se3 import SE3, SE3ToXYZQUAT, XYZQUATToSE3, as_se3_12, quat_to_rot
from ..trajectory import WeightedTrajectoryPoint

# name -> dataclass, the lookup table of the `class:` tags
_CLASSES: dict[str, type] = {}


def add_modules(values: dict):
    """Make extra OCP component classes known to the YAML loader (same role as the
    reference's `add_modules`, which extends its module globals)."""
    for name, obj in values.items():
        if isinstance(obj, type) and dataclasses.is_dataclass(obj):
            _CLASSES[name] = obj


def _yaml_class(cls):
    _CLASSES[cls.__name__] = cls
    return cls


def create_nested_dataclass(cls, values):
    kwargs = {key: create_croco_dataclasses(val) for key, val in values.items()}
    maker = getattr(cls, "from_dict", None)
    return maker(kwargs) if maker is not None else cls(**kwargs)


def create_croco_dataclasses(values):
    """Recursively turn `{class: Name, ...}` mappings into instances of the registered classes."""
    if isinstance(values, dict):
        if "class" in values:
            name = values["class"]
            if name not in _CLASSES:
                raise KeyError(f"unknown OCP component class '{name}'")
            return create_nested_dataclass(_CLASSES[name], {k: v for k, v in values.items() if k != "class"})
        return {k: create_croco_dataclasses(v) for k, v in values.items()}
    if isinstance(values, (list, tuple)):
        return type(values)(create_croco_dataclasses(v) for v in values)
    return values


def as_dict(obj):
    if dataclasses.is_dataclass(obj):
        out = {f.name: as_dict(getattr(obj, f.name)) for f in dataclasses.fields(obj)}
        out["class"] = obj.class_
        return out
    if isinstance(obj, (list, tuple)):
        return type(obj)(as_dict(v) for v in obj)
    return obj


def get_frame_id(model, id: T.Union[str, int]) -> int:
    if isinstance(id, str):
        assert model.existFrame(id), f"Frame '{id}' does not exist!"
        id = model.getFrameId(id)
    assert isinstance(id, (int, np.integer)) and id < model.nframes
    return int(id)


@dataclasses.dataclass
class BuildData:
    """What a component needs while it is lowered / updated."""

    model: T.Any  # TableModel (pinocchio.Model look-alike)
    nv: int
    collision_model: T.Any = None
    # transforms the OCP needs from outside (TF2 in the ROS node), keyed (parent, child)
    transforms: T.Dict[T.Tuple[str, str], T.Optional[T.Any]] = dataclasses.field(default_factory=dict)


def _vec(w, n) -> np.ndarray:
    """Scalar or vector weights -> n-vector (size-1 arrays broadcast like the ROS publisher's)."""
    a = np.asarray(w, dtype=np.float64).reshape(-1)
    if a.size == 1:
        return np.full(n, a[0])
    assert a.size == n, f"expected {n} weights, got {a.size}"
    return a.copy()


# ------------------------------------------------------------------ activations
@dataclasses.dataclass
class ActivationModel:
    pass


@_yaml_class
@dataclasses.dataclass
class ActivationModelWeightedQuad(ActivationModel):
    class_: T.ClassVar[str] = "ActivationModelWeightedQuad"
    weights: T.Union[None, float, T.Any] = None

    kind: T.ClassVar[int] = _abi.ACT_WEIGHTED_QUAD
    alpha_value: T.ClassVar[float] = 1.0

    def initial_weights(self, nr: int) -> np.ndarray:
        return np.ones(nr) if self.weights is None else _vec(self.weights, nr)


@_yaml_class
@dataclasses.dataclass
class ActivationModelExp(ActivationModel):
    class_: T.ClassVar[str] = "ActivationModelExp"
    alpha: float = 1.0
    exponent: int = 1

    def __post_init__(self):
        assert self.exponent in (1, 2)

    @property
    def kind(self) -> int:
        return _abi.ACT_EXP if self.exponent == 1 else _abi.ACT_QUAD_EXP

    @property
    def alpha_value(self) -> float:
        return float(self.alpha)

    def initial_weights(self, nr: int) -> np.ndarray:
        return np.ones(nr)


@_yaml_class
@dataclasses.dataclass
class ActivationModelQuadExp(ActivationModelExp):
    class_: T.ClassVar[str] = "ActivationModelQuadExp"
    exponent: int = 2

    def __post_init__(self):
        assert self.exponent == 2, "ActivationModelQuadExp is the exponent-2 variant of ActivationModelExp."


# -------------------------------------------------------------------- residuals
@dataclasses.dataclass
class ResidualModel:
    """A residual lowers to one row kind; `reference()` gives the static reference from the YAML,
    `update()` the per-node (reference, weights, frame id) from a weighted trajectory point."""

    kind: T.ClassVar[int] = -1

    @staticmethod
    def needs_colmpc_freefwd_dynamics() -> bool:
        return False

    def frame(self, data: BuildData) -> int:
        return 0

    def reference(self, data: BuildData) -> np.ndarray:
        return np.zeros(_abi.row_nref(self.kind, data.nv))


@_yaml_class
@dataclasses.dataclass
class ResidualModelState(ResidualModel):
    class_: T.ClassVar[str] = "ResidualModelState"
    xref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_STATE

    def reference(self, data):
        return np.zeros(2 * data.nv) if self.xref is None else np.asarray(self.xref, dtype=float).reshape(2 * data.nv)

    def update(self, data, pt: WeightedTrajectoryPoint):
        # each half may be a size-1 array (the trajectory publishers' broadcast convention)
        w = np.concatenate([_vec(pt.weights.w_robot_configuration, data.nv), _vec(pt.weights.w_robot_velocity, data.nv)])
        return pt.point.robot_state, w, None


@_yaml_class
@dataclasses.dataclass
class ResidualModelControl(ResidualModel):
    class_: T.ClassVar[str] = "ResidualModelControl"
    uref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_CONTROL

    def reference(self, data):
        return np.zeros(data.nv) if self.uref is None else np.asarray(self.uref, dtype=float).reshape(data.nv)

    def update(self, data, pt):
        return pt.point.robot_effort, pt.weights.w_robot_effort, None


@_yaml_class
@dataclasses.dataclass
class ResidualModelControlGrav(ResidualModel):
    class_: T.ClassVar[str] = "ResidualModelControlGrav"
    kind: T.ClassVar[int] = _abi.RES_CONTROL_GRAV

    def update(self, data, pt):
        return np.zeros(0), pt.weights.w_robot_effort, None


def _single_pose(pt: WeightedTrajectoryPoint, who: str):
    poses = pt.point.end_effector_poses
    assert len(poses) == 1, f"{who} requires exactly one end-effector pose, current is {poses}."
    return next(iter(poses.items()))


def _pref_se3(pref) -> np.ndarray:
    return as_se3_12(SE3.Identity() if pref is None else XYZQUATToSE3(pref))


@dataclasses.dataclass
class _FrameResidual(ResidualModel):
    def frame(self, data):
        key = getattr(self, "id", None)
        if key is None:
            key = getattr(self, "frame_id")
        return get_frame_id(data.model, key)


@_yaml_class
@dataclasses.dataclass
class ResidualModelFramePlacement(_FrameResidual):
    class_: T.ClassVar[str] = "ResidualModelFramePlacement"
    id: T.Union[str, int] = 0
    pref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_FRAME_PLACEMENT

    def reference(self, data):
        return _pref_se3(self.pref)

    def update(self, data, pt):
        name, pose = _single_pose(pt, "ResidualModelFramePlacement")
        return as_se3_12(pose), pt.weights.w_end_effector_poses[name], get_frame_id(data.model, name)


@_yaml_class
@dataclasses.dataclass
class ResidualModelFramePlacementStatic(_FrameResidual):
    """Frame fixed in the YAML (multi end-effector setups)."""

    class_: T.ClassVar[str] = "ResidualModelFramePlacement"
    frame_id: T.Optional[str] = None
    pref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_FRAME_PLACEMENT

    def reference(self, data):
        return _pref_se3(self.pref)

    def update(self, data, pt):
        _single_pose(pt, "ResidualModelFramePlacementStatic")
        assert self.frame_id in pt.point.end_effector_poses, (
            f"ResidualModelFramePlacementStatic: end_effector_poses should contain the key {self.frame_id}"
        )
        return as_se3_12(pt.point.end_effector_poses[self.frame_id]), pt.weights.w_end_effector_poses[self.frame_id], None


@_yaml_class
@dataclasses.dataclass
class ResidualModelFrameTranslation(_FrameResidual):
    class_: T.ClassVar[str] = "ResidualModelFrameTranslation"
    id: T.Union[str, int] = 0
    pref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_FRAME_TRANSLATION

    def reference(self, data):
        return np.zeros(3) if self.pref is None else np.asarray(self.pref, dtype=float)[:3].copy()

    def update(self, data, pt):
        name, pose = _single_pose(pt, "ResidualModelFrameTranslation")
        return as_se3_12(pose)[9:], np.asarray(pt.weights.w_end_effector_poses[name])[:3], get_frame_id(data.model, name)


@_yaml_class
@dataclasses.dataclass
class ResidualModelFrameTranslationStatic(_FrameResidual):
    class_: T.ClassVar[str] = "ResidualModelFrameTranslation"
    frame_id: T.Optional[str] = None
    pref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_FRAME_TRANSLATION

    def reference(self, data):
        return np.zeros(3) if self.pref is None else np.asarray(self.pref, dtype=float)[:3].copy()

    def update(self, data, pt):
        _single_pose(pt, "ResidualModelFrameTranslationStatic")
        assert self.frame_id in pt.point.end_effector_poses, f"end_effector_poses should contains key {self.frame_id}"
        return as_se3_12(pt.point.end_effector_poses[self.frame_id])[9:], np.asarray(pt.weights.w_end_effector_poses[self.frame_id])[:3], None


@_yaml_class
@dataclasses.dataclass
class ResidualModelFrameRotation(_FrameResidual):
    class_: T.ClassVar[str] = "ResidualModelFrameRotation"
    id: T.Union[str, int] = 0
    pref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_FRAME_ROTATION

    def reference(self, data):
        return np.eye(3).reshape(9) if self.pref is None else quat_to_rot(np.asarray(self.pref, dtype=float)[3:]).reshape(9)

    def update(self, data, pt):
        name, pose = _single_pose(pt, "ResidualModelFrameRotation")
        return as_se3_12(pose)[:9], np.asarray(pt.weights.w_end_effector_poses[name])[3:], get_frame_id(data.model, name)


@_yaml_class
@dataclasses.dataclass
class ResidualModelFrameRotationStatic(_FrameResidual):
    class_: T.ClassVar[str] = "ResidualModelFrameRotation"
    frame_id: T.Optional[str] = None
    pref: T.Optional[T.Any] = None
    kind: T.ClassVar[int] = _abi.RES_FRAME_ROTATION

    def reference(self, data):
        return np.eye(3).reshape(9) if self.pref is None else quat_to_rot(np.asarray(self.pref, dtype=float)[3:]).reshape(9)

    def update(self, data, pt):
        _single_pose(pt, "ResidualModelFrameRotationStatic")
        assert self.frame_id in pt.point.end_effector_poses, (
            f"ResidualModelFrameRotationStatic: end_effector_poses should contain the key {self.frame_id}"
        )
        return as_se3_12(pt.point.end_effector_poses[self.frame_id])[:9], np.asarray(pt.weights.w_end_effector_poses[self.frame_id])[3:], None


_REFERENCE_FRAMES = ("WORLD", "LOCAL", "LOCAL_WORLD_ALIGNED")


@_yaml_class
@dataclasses.dataclass
class ResidualModelFrameVelocity(_FrameResidual):
    class_: T.ClassVar[str] = "ResidualModelFrameVelocity"
    id: T.Union[str, int] = 0
    pref: T.Optional[T.Any] = None
    reference_frame: T.Optional[str] = "WORLD"
    kind: T.ClassVar[int] = _abi.RES_FRAME_VELOCITY

    def __post_init__(self):
        assert self.reference_frame in _REFERENCE_FRAMES, (
            "ResidualModelFrameVelocity.reference_frame has to be one of: 'WORLD', 'LOCAL', 'LOCAL_WORLD_ALIGNED'."
        )

    def reference(self, data):
        return np.zeros(6) if self.pref is None else np.asarray(self.pref, dtype=float).reshape(6)

    def reference_frame_id(self) -> int:
        """pinocchio.ReferenceFrame as the row's second index: 0 WORLD, 1 LOCAL, 2 LOCAL_WORLD_ALIGNED."""
        return _REFERENCE_FRAMES.index(self.reference_frame)

    def update(self, data, pt):
        vels = pt.point.end_effector_velocities
        assert len(vels) == 1, f"ResidualModelFrameVelocity requires exactly one end-effector velocity, current is {vels}."
        name, vel = next(iter(vels.items()))
        return np.asarray(getattr(vel, "vector", vel), dtype=float), pt.weights.w_end_effector_velocities[name], get_frame_id(data.model, name)


@_yaml_class
@dataclasses.dataclass
class ResidualModelFrameVelocityStatic(ResidualModelFrameVelocity):
    class_: T.ClassVar[str] = "ResidualModelFrameVelocity"
    frame_id: T.Optional[str] = None

    def frame(self, data):
        return get_frame_id(data.model, self.frame_id)

    def update(self, data, pt):
        vels = pt.point.end_effector_velocities
        assert len(vels) == 1 and self.frame_id in vels, (
            f"ResidualModelFrameVelocityStatic: end_effector_velocities should contain the key {self.frame_id}"
        )
        vel = vels[self.frame_id]
        return np.asarray(getattr(vel, "vector", vel), dtype=float), pt.weights.w_end_effector_velocities[self.frame_id], None


@_yaml_class
@dataclasses.dataclass
class ResidualModelVisualServoing(ResidualModel):
    """wMf_target = wMo_vision * oMf_target: the object pose `wMo` comes from outside through
    `OCPCrocoGeneric.input_transforms[(world_frame, object_frame)]`, the target in the object
    frame from the trajectory point under the key `<robot_frame>_vs`."""

    class_: T.ClassVar[str] = "ResidualModelVisualServoing"
    world_frame: str = ""
    object_frame: str = ""
    robot_frame: str = ""
    kind: T.ClassVar[int] = _abi.RES_FRAME_PLACEMENT

    def frame(self, data):
        wid = get_frame_id(data.model, self.world_frame)
        table = data.model.table
        assert table.frame_parent[wid] == -1, f"Parent joint of world frame ({self.world_frame}) should be 0"
        assert np.allclose(table.frame_placement[wid], as_se3_12(SE3.Identity())), (
            f"Placement of world frame ({self.world_frame}) should be identity"
        )
        self.transforms_key = (self.world_frame, self.object_frame)
        self.input_key = self.robot_frame + "_vs"
        data.transforms.setdefault(self.transforms_key, None)
        return get_frame_id(data.model, self.robot_frame)

    def reference(self, data):
        return as_se3_12(SE3.Identity())

    def update(self, data, pt):
        poses = pt.point.end_effector_poses
        assert len(poses) == 1, f"ResidualModelVisualServoing requires exactly one end-effector, current is {poses}."
        assert self.input_key in poses, f"end_effector_poses should contains key {self.input_key}"
        weights = pt.weights.w_end_effector_poses[self.input_key]
        active = bool(np.any(np.asarray(weights) != 0))
        wMo = data.transforms[self.transforms_key]
        assert not active or wMo is not None, f"Weights are not all zeros and no transform for {self.transforms_key}"
        target = as_se3_12(poses[self.input_key])
        if wMo is not None:
            a = as_se3_12(wMo)
            Ra, pa, Rb, pb = a[:9].reshape(3, 3), a[9:], target[:9].reshape(3, 3), target[9:]
            target = np.concatenate([(Ra @ Rb).reshape(9), Ra @ pb + pa])
        return target, weights, None


@dataclasses.dataclass
class ResidualDistanceCollisionBase(ResidualModel):
    # Ideally paired with ActivationModelExp (reference ocp_croco_generic.py:499-522).
    collision_pair: T.Optional[T.Tuple[str, str]] = None
    # the shipped ocp_traj_tracking_collision_avoidance.yaml:43-56 still uses the pair index
    collision_pair_id: T.Optional[int] = None
    kind: T.ClassVar[int] = _abi.RES_COLLISION

    def geometry_frames(self, data: "BuildData") -> T.Tuple[int, int]:
        cmodel = data.collision_model
        assert cmodel is not None, "the robot model carries no collision geometry"
        if self.collision_pair is not None:
            assert len(self.collision_pair) == 2
            ids = []
            for name in self.collision_pair:
                assert cmodel.existGeometryName(name), f"Geometry object '{name}' not found."
                ids.append(cmodel.getGeometryId(name))
            if not cmodel.existCollisionPair(ids):
                cmodel.addCollisionPair(ids)
            return ids[0], ids[1]
        assert self.collision_pair_id is not None, "collision_pair (or collision_pair_id) is required"
        assert 0 <= int(self.collision_pair_id) < len(cmodel.collisionPairs)
        return tuple(cmodel.collisionPairs[int(self.collision_pair_id)])

    def reference(self, data: "BuildData"):
        return np.zeros(0)


@_yaml_class
@dataclasses.dataclass
class ResidualDistanceCollision(ResidualDistanceCollisionBase):
    class_: T.ClassVar[str] = "ResidualDistanceCollision"


@_yaml_class
@dataclasses.dataclass
class ResidualDistanceCollision2(ResidualDistanceCollisionBase):
    """colmpc.ResidualDistanceCollision2: the same signed distance of the pair, with the geometry
    placements kept in colmpc.StateMultibody upstream.  Here geometry placements always live in the
    device model (update_geometry_placement), so it lowers to the same row as ResidualDistanceCollision."""

    class_: T.ClassVar[str] = "ResidualDistanceCollision2"

    @staticmethod
    def needs_colmpc_freefwd_dynamics() -> bool:
        return True


# ------------------------------------------------------------------------ costs
@dataclasses.dataclass
class CostModel:
    residual: ResidualModel
    activation: T.Optional[ActivationModel] = None


@_yaml_class
@dataclasses.dataclass
class CostModelResidual(CostModel):
    class_: T.ClassVar[str] = "CostModelResidual"


@_yaml_class
@dataclasses.dataclass
class CostModelSumItem:
    class_: T.ClassVar[str] = "CostModelSumItem"
    name: str
    cost: CostModel
    weight: float = 1.0
    active: bool = True
    update: bool = False
    publish_residual: bool = False


@dataclasses.dataclass
class ConstraintModel:
    residual: ResidualModel


@_yaml_class
@dataclasses.dataclass
class ConstraintModelResidual(ConstraintModel):
    class_: T.ClassVar[str] = "ConstraintModelResidual"
    lower: T.Optional[T.Any] = None
    upper: T.Optional[T.Any] = None
    active_on_terminal_node: bool = True

    def bounds(self, nr: int):
        lo = np.full(nr, -np.inf) if self.lower is None else _vec(np.asarray(self.lower, dtype=float), nr)
        up = np.full(nr, np.inf) if self.upper is None else _vec(np.asarray(self.upper, dtype=float), nr)
        return lo, up


@_yaml_class
@dataclasses.dataclass
class ConstraintModelControlLimit(ConstraintModelResidual):
    class_: T.ClassVar[str] = "ConstraintModelControlLimit"
    residual: T.Optional[ResidualModel] = dataclasses.field(default=None, init=False)
    lower: T.Optional[T.Any] = dataclasses.field(default=None, init=False)
    upper: T.Optional[T.Any] = dataclasses.field(default=None, init=False)

    def __post_init__(self):
        self.residual = ResidualModelControl()


@_yaml_class
@dataclasses.dataclass
class ConstraintListItem:
    name: str
    constraint: ConstraintModel
    active: bool = True


# ---------------------------------------------------------------- action models
@dataclasses.dataclass
class DifferentialActionModel:
    pass


@_yaml_class
@dataclasses.dataclass
class DifferentialActionModelFreeFwdDynamics(DifferentialActionModel):
    class_: T.ClassVar[str] = "DifferentialActionModelFreeFwdDynamics"
    costs: T.List[CostModelSumItem]
    constraints: T.List[ConstraintListItem] = dataclasses.field(default_factory=list)

    @classmethod
    def from_dict(cls, kwargs: T.Dict[str, T.Any]):
        kwargs["costs"] = [c if isinstance(c, CostModelSumItem) else create_nested_dataclass(CostModelSumItem, c) for c in kwargs.get("costs", [])]
        kwargs["constraints"] = [
            c if isinstance(c, ConstraintListItem) else create_nested_dataclass(ConstraintListItem, c) for c in kwargs.get("constraints", [])
        ]
        return cls(**kwargs)

    def needs_colmpc_freefwd_dynamics(self) -> bool:
        residuals = [c.cost.residual for c in self.costs] + [c.constraint.residual for c in self.constraints or []]
        return any(r.needs_colmpc_freefwd_dynamics() for r in residuals)

    def lower_constraints(self, data: BuildData, terminal: bool) -> list[_abi.ConstraintSpec]:
        """ConstraintListItem list -> constraint rows (the counterpart of building a
        crocoddyl.ConstraintModelManager, reference ocp_croco_generic.py:700-711)."""
        out = []
        for item in self.constraints or []:
            con = item.constraint
            res = con.residual
            kind = res.kind
            nr = _abi.row_nr(kind, data.nv)
            if isinstance(con, ConstraintModelControlLimit):
                lim = np.asarray(data.model.effortLimit, dtype=float)
                lo, up = -lim, lim
            else:
                lo, up = con.bounds(nr)
            active = bool(item.active) and not (terminal and (kind in (_abi.RES_CONTROL, _abi.RES_CONTROL_GRAV) or not con.active_on_terminal_node))
            fa = fb = 0
            if kind == _abi.RES_COLLISION:
                fa, fb = res.geometry_frames(data)
                ref = None
            else:
                fa = res.frame(data)
                if kind == _abi.RES_FRAME_VELOCITY:
                    fb = res.reference_frame_id()  # 0 WORLD, 1 LOCAL, 2 LOCAL_WORLD_ALIGNED
                ref = np.asarray(res.reference(data), dtype=float).reshape(-1)
            out.append(_abi.ConstraintSpec(kind=kind, lower=lo, upper=up, ref=ref, active=active, frame=fa, frame_b=fb, name=item.name))
        return out

    def lower(self, data: BuildData) -> list[_abi.RowSpec]:
        """Cost items -> row table (the counterpart of building a crocoddyl.CostModelSum)."""
        if len(self.costs) > _abi.AGX_MAX_ROWS:
            raise ValueError(f"at most {_abi.AGX_MAX_ROWS} cost items per node are supported")
        rows = []
        for item in self.costs:
            res, act = item.cost.residual, item.cost.activation
            kind = res.kind
            act_kind = _abi.ACT_WEIGHTED_QUAD if act is None else act.kind
            alpha = 1.0 if act is None or act_kind == _abi.ACT_WEIGHTED_QUAD else act.alpha_value
            if act_kind != _abi.ACT_WEIGHTED_QUAD and kind in (_abi.RES_CONTROL_GRAV, _abi.RES_FRAME_VELOCITY):
                raise NotImplementedError(f"cost '{item.name}': {type(act).__name__} is not implemented for ControlGrav / FrameVelocity residuals")
            if kind == _abi.RES_COLLISION:
                fa, fb = res.geometry_frames(data)
                rows.append(_abi.RowSpec(kind=kind, activation=act_kind, active=bool(item.active), frame=fa, frame_b=fb,
                                         alpha=alpha, name=item.name, weight=float(item.weight)))  # fmt: skip
                continue
            frame_b = res.reference_frame_id() if kind == _abi.RES_FRAME_VELOCITY else 0
            rows.append(_abi.RowSpec(kind=kind, activation=act_kind, active=bool(item.active), frame=res.frame(data),
                                     frame_b=frame_b, alpha=alpha, name=item.name, weight=float(item.weight)))  # fmt: skip
        return rows


@dataclasses.dataclass
class IntegratedActionModelAbstract:
    differential: DifferentialActionModel
    step_time: float = 0.0
    with_cost_residual: bool = True


@_yaml_class
@dataclasses.dataclass
class IntegratedActionModelEuler(IntegratedActionModelAbstract):
    class_: T.ClassVar[str] = "IntegratedActionModelEuler"


@dataclasses.dataclass
class ShootingProblem:
    running_model: IntegratedActionModelAbstract
    terminal_model: IntegratedActionModelAbstract

    def needs_colmpc_state(self) -> bool:
        return self.running_model.differential.needs_colmpc_freefwd_dynamics() or self.terminal_model.differential.needs_colmpc_freefwd_dynamics()

    def __post_init__(self):
        self.running_model = create_croco_dataclasses(self.running_model)
        self.terminal_model = create_croco_dataclasses(self.terminal_model)


# ------------------------------------------------------------------------ the OCP
class OCPCrocoGeneric(OCPBaseCroco):
    def __init__(self, robot_models: RobotModels, params: OCPParamsBaseCroco, yaml_file: T.Union[str, T.IO],
                 expect_rolling_buffer: bool = False, device: int = 0) -> None:  # fmt: skip
        if hasattr(yaml_file, "read"):
            data = yaml.safe_load(yaml_file)
        else:
            with open(yaml_file, "r") as f:
                data = yaml.safe_load(f)
        self._data = ShootingProblem(**data)
        self._build_data_obj = BuildData(robot_models.robot_model, robot_models.table.nv, robot_models.collision_model)
        super().__init__(robot_models, params, use_colmpc_state=self._data.needs_colmpc_state(), device=device)
        self._expect_rolling_buffer = expect_rolling_buffer
        self._first_call = True
        self._init_static_tile()
        self.init_debug_data_attributes()

    @property
    def _build_data(self) -> BuildData:
        return self._build_data_obj

    @property
    def input_transforms(self) -> T.Dict[T.Tuple[str, str], T.Any]:
        """Transforms the OCP needs as input, keyed (parent frame, child frame)."""
        return self._build_data.transforms

    def create_running_model_list(self) -> list[_abi.RowSpec]:
        return self._data.running_model.differential.lower(self._build_data_obj)

    def create_terminal_model(self) -> list[_abi.RowSpec]:
        return self._data.terminal_model.differential.lower(self._build_data_obj)

    def create_constraint_lists(self):
        return (self._data.running_model.differential.lower_constraints(self._build_data_obj, False),
                self._data.terminal_model.differential.lower_constraints(self._build_data_obj, True))

    # -- reference tile -------------------------------------------------------
    def _items(self, terminal: bool) -> list[CostModelSumItem]:
        return (self._data.terminal_model if terminal else self._data.running_model).differential.costs

    def _init_static_tile(self):
        """What the Crocoddyl objects hold right after construction: YAML item weights,
        activation weights and static references on every node."""
        po = self._packed
        for terminal in (False, True):
            for row, item in enumerate(self._items(terminal)):
                wi, ref, aw = po.row_view(self._ref_tile, terminal, row)
                wi[...] = float(item.weight)
                ref[...] = item.cost.residual.reference(self._build_data_obj)
                nr = aw.shape[-1]
                aw[...] = np.ones(nr) if item.cost.activation is None else item.cost.activation.initial_weights(nr)
        self._hip.set_refs(self._ref_tile, self._frames)

    def _update_node(self, terminal: bool, node: int, pt: WeightedTrajectoryPoint):
        po = self._packed
        t = 0 if terminal else node
        frame_slot = po.horizon if terminal else node
        for row, item in enumerate(self._items(terminal)):
            if not item.update:
                continue
            wi, ref, aw = po.row_view(self._ref_tile, terminal, row)
            res = item.cost.residual
            if isinstance(res, ResidualDistanceCollisionBase):
                wi[0, t] = float(pt.weights.w_collision_avoidance)
                continue
            value, weights, frame = res.update(self._build_data_obj, pt)
            if ref.shape[-1]:
                ref[0, t] = np.asarray(value, dtype=np.float64).reshape(-1)
            if item.cost.activation is not None:
                aw[0, t] = _vec(weights, aw.shape[-1])
            if frame is not None:
                self._frames[0, frame_slot, row] = frame

    def set_reference_weighted_trajectory(self, reference_weighted_trajectory: list[WeightedTrajectoryPoint]):
        T_ = self.n_controls
        assert len(reference_weighted_trajectory) == T_ + 1
        if self._expect_rolling_buffer and not self._first_call:
            # problem.circularAppend(runningModels[0]): node i takes over node i+1's data, then only
            # the last running node is refreshed
            self._ref_tile[0, : T_ - 1] = self._ref_tile[0, 1:T_].copy()
            self._frames[0, : T_ - 1] = self._frames[0, 1:T_].copy()
            self._update_node(False, T_ - 1, reference_weighted_trajectory[-2])
        else:
            for node, pt in enumerate(reference_weighted_trajectory[:-1]):
                self._update_node(False, node, pt)
            self._first_call = False
        self._update_node(True, T_, reference_weighted_trajectory[-1])
        self._hip.set_refs(self._ref_tile, self._frames)

    # -- debug data -----------------------------------------------------------
    def init_debug_data_attributes(self) -> None:
        for item in self._items(False):
            if item.update and not isinstance(item.cost.residual, ResidualDistanceCollisionBase):
                self._debug_data.references.append((item.name, None))
            if item.publish_residual:
                self._debug_data.residuals.append((item.name, None))

    def fill_debug_data(self, res, ocp_results: OCPResults) -> None:
        super().fill_debug_data(res=res, ocp_results=ocp_results)
        names = [item.name for item in self._items(False)]
        for idx, (name, _) in enumerate(self._debug_data.references):
            row = names.index(name)
            _, ref, _ = self._packed.row_view(self._ref_tile, False, row)
            value = ref[0, 0].copy()
            if value.size == 12:  # SE3 references are published as xyz + quaternion
                value = SE3ToXYZQUAT(SE3(value[:9].reshape(3, 3), value[9:]))
            self._debug_data.references[idx] = (name, value)
        for idx, (name, _) in enumerate(self._debug_data.residuals):
            self._debug_data.residuals[idx] = (name, self._hip.residuals(names.index(name))[0])

    @staticmethod
    def get_default_yaml_file(basename: str) -> pathlib.Path:
        return pathlib.Path(__file__).parent / basename
