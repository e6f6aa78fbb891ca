"""Robot model holder with the attribute surface the OCP layer uses from the reference's
`RobotModels` (agimus_controller/agimus_controller/factory/robot_model.py:13-351):
`robot_model` (nq, nv, frames, effortLimit, getFrameId/existFrame), `armature`, `params`,
`collision_model`.  URDF parsing stays Pinocchio's job: when Pinocchio is importable a
`pin.Model` can be passed and is converted to a table; otherwise a table is used directly.
"""

from __future__ import annotations

import dataclasses
import typing as T

import numpy as np

from . import robot_tables


class TableModel:
    """Read-only `pinocchio.Model` look-alike over a RobotTable."""

    def __init__(self, table: robot_tables.RobotTable):
        self.table = table
        self.name = table.name
        self.nq = table.nq
        self.nv = table.nv
        self.names = ["universe"] + list(table.joint_names)
        self.njoints = table.nv + 1
        self.nframes = len(table.frame_names)
        self.effortLimit = table.effort_limit
        self.lowerPositionLimit = table.lower_position_limit
        self.upperPositionLimit = table.upper_position_limit
        self.velocityLimit = table.velocity_limit

    def existFrame(self, name: str) -> bool:
        return name in self.table.frame_names

    def getFrameId(self, name: str) -> int:
        return self.table.frame_names.index(name) if name in self.table.frame_names else self.nframes

    def neutral(self) -> np.ndarray:
        return np.zeros(self.nq)


class TableGeometryModel:
    """pinocchio.GeometryModel look-alike over the geometry frames of a RobotTable (capsules /
    spheres / boxes; what factory/robot_model.py:261-302 of the reference leaves in the collision model)."""

    def __init__(self, table: robot_tables.RobotTable, collision_pairs=()):
        self.table = table
        self.collisionPairs = []
        for a, b in collision_pairs:
            self.addCollisionPair((self.getGeometryId(a), self.getGeometryId(b)))

    def existGeometryName(self, name: str) -> bool:
        if name not in self.table.frame_names or self.table.frame_radius is None:
            return False
        i = self.table.frame_names.index(name)
        is_box = self.table.frame_box is not None and float(np.asarray(self.table.frame_box).reshape(-1, 3)[i, 0]) > 0.0
        return float(self.table.frame_radius[i]) > 0.0 or is_box

    def getGeometryId(self, name: str) -> int:
        assert self.existGeometryName(name), f"Geometry object '{name}' not found."
        return self.table.frame_names.index(name)

    def existCollisionPair(self, pair) -> bool:
        return tuple(pair) in self.collisionPairs

    def addCollisionPair(self, pair) -> None:
        if not self.existCollisionPair(pair):
            self.collisionPairs.append(tuple(pair))

    def findCollisionPair(self, pair) -> int:
        return self.collisionPairs.index(tuple(pair))


@dataclasses.dataclass
class RobotModelParameters:
    """Subset of the reference's parameters that is meaningful without a URDF loader.
    `table` (or a `pinocchio.Model` in `pin_model`) replaces robot_urdf/srdf/meshes."""

    table: T.Optional[robot_tables.RobotTable] = None
    pin_model: T.Any = None
    q0: np.ndarray = dataclasses.field(default_factory=lambda: np.array([], dtype=np.float64))
    free_flyer: bool = False
    armature: np.ndarray = dataclasses.field(default_factory=lambda: np.array([], dtype=np.float64))
    collision_as_capsule: bool = False
    self_collision: bool = False
    collision_pairs: T.List[T.Tuple[str, str]] = dataclasses.field(default_factory=list)

    def __post_init__(self):
        if self.free_flyer:
            raise ValueError("free-flyer bases are not supported by the HIP path (1-DoF revolute joints only)")
        if self.table is None and self.pin_model is None:
            raise ValueError("RobotModelParameters needs a RobotTable (`table`) or a pinocchio model (`pin_model`)")
        if self.table is None:
            self.table = robot_tables.from_pinocchio(self.pin_model)
        nv = self.table.nv
        self.q0 = np.asarray(self.q0, dtype=float)
        if self.q0.size == 0:
            self.q0 = np.zeros(nv)
        self.armature = np.asarray(self.armature, dtype=float)
        if self.armature.size == 0:
            self.armature = np.zeros(nv)
        if self.armature.size != nv:
            raise ValueError(f"Armature must have the same shape as the robot velocity (nv = {nv}), got {self.armature.size}.")


class RobotModels:
    def __init__(self, param: RobotModelParameters):
        self._params = param
        self._table = param.table.with_armature(param.armature)
        self._robot_model = TableModel(self._table)
        self._q0 = param.q0
        has_geom = (self._table.frame_radius is not None and np.any(np.asarray(self._table.frame_radius) > 0.0)) or (
            self._table.frame_box is not None and np.any(np.asarray(self._table.frame_box) > 0.0))
        self._collision_model = TableGeometryModel(self._table, param.collision_pairs) if has_geom else None

    @property
    def params(self) -> RobotModelParameters:
        return self._params

    @property
    def table(self) -> robot_tables.RobotTable:
        return self._table

    @property
    def robot_model(self) -> TableModel:
        return self._robot_model

    @property
    def collision_model(self):
        return self._collision_model

    @property
    def visual_model(self):
        return None

    @property
    def armature(self) -> np.ndarray:
        return self._params.armature

    @property
    def q0(self) -> np.ndarray:
        return self._q0


def panda_robot_models(armature=0.1, q0=None) -> RobotModels:
    table = robot_tables.panda_table(armature)
    return RobotModels(RobotModelParameters(table=table, q0=np.zeros(7) if q0 is None else q0, armature=table.armature))
