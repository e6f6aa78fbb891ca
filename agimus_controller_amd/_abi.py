"""ctypes mirror of include/agimus_hip.h (structures, enums, layout helpers).

Shared by the product loader (backend.py -> libagimus_hip.so) and by the test
checker under oracle/.  Nothing in this module computes anything.
"""

from __future__ import annotations

import ctypes as C
import dataclasses
import typing as T

import numpy as np

AGX_MAX_ROWS = 8
AGX_MAX_NV = 32

# agx_residual_kind
RES_STATE = 0
RES_CONTROL = 1
RES_CONTROL_GRAV = 2
RES_FRAME_PLACEMENT = 3
RES_FRAME_TRANSLATION = 4
RES_FRAME_ROTATION = 5
RES_FRAME_VELOCITY = 6
RES_COLLISION = 7

# agx_activation_kind
ACT_WEIGHTED_QUAD = 0
ACT_EXP = 1
ACT_QUAD_EXP = 2

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class CostRow(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("activation", C.c_int32),
        ("active", C.c_int32),
        ("frame", C.c_int32),
        ("frame_b", C.c_int32),
        ("pad_", C.c_int32),
        ("alpha", C.c_double),
        ("weight", C.c_double),
    ]


class ConstraintRow(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("active", C.c_int32),
        ("frame", C.c_int32),
        ("frame_b", C.c_int32),
        ("ref", c_double_p),
        ("lower", c_double_p),
        ("upper", c_double_p),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("nv", C.c_int32),
        ("nframes", C.c_int32),
        ("parent", c_int32_p),
        ("placement", c_double_p),
        ("axis", c_double_p),
        ("mass", c_double_p),
        ("com", c_double_p),
        ("inertia", c_double_p),
        ("armature", c_double_p),
        ("effort_limit", c_double_p),
        ("gravity", c_double_p),
        ("frame_parent", c_int32_p),
        ("frame_placement", c_double_p),
        ("frame_radius", c_double_p),
        ("frame_halflen", c_double_p),
        ("frame_box", c_double_p),
    ]


class OcpDesc(C.Structure):
    _fields_ = [
        ("horizon", C.c_int32),
        ("dt", c_double_p),
        ("n_running_rows", C.c_int32),
        ("running_rows", C.POINTER(CostRow)),
        ("n_terminal_rows", C.c_int32),
        ("terminal_rows", C.POINTER(CostRow)),
        ("termination_tolerance", C.c_double),
        ("max_qp_iters", C.c_int32),
        ("eps_abs", C.c_double),
        ("eps_rel", C.c_double),
        ("mu_dynamic", C.c_double),
        ("mu_constraint", C.c_double),
        ("use_filter_line_search", C.c_int32),
        ("n_running_constraints", C.c_int32),
        ("running_constraints", C.POINTER(ConstraintRow)),
        ("n_terminal_constraints", C.c_int32),
        ("terminal_constraints", C.POINTER(ConstraintRow)),
    ]


class Status(C.Structure):
    _fields_ = [
        ("kkt", C.c_double),
        ("cost", C.c_double),
        ("merit", C.c_double),
        ("gap_norm", C.c_double),
        ("iter", C.c_int32),
        ("qp_iters", C.c_int32),
        ("solved", C.c_int32),
        ("flags", C.c_int32),
    ]


STATUS_DTYPE = np.dtype(
    [
        ("kkt", "f8"),
        ("cost", "f8"),
        ("merit", "f8"),
        ("gap_norm", "f8"),
        ("iter", "i4"),
        ("qp_iters", "i4"),
        ("solved", "i4"),
        ("flags", "i4"),
    ]
)
assert STATUS_DTYPE.itemsize == C.sizeof(Status)


def tile_doubles(nv: int) -> int:
    """AGX_TILE_DOUBLES(nv): Fx|Fu|f|Lx|Lu|Lxx|Lxu|Luu|cost."""
    return 13 * nv * nv + 5 * nv + 1


def tile_slices(nv: int) -> dict[str, slice]:
    nx, nu = 2 * nv, nv
    sizes = [
        ("Fx", nx * nx),
        ("Fu", nx * nu),
        ("f", nx),
        ("Lx", nx),
        ("Lu", nu),
        ("Lxx", nx * nx),
        ("Lxu", nx * nu),
        ("Luu", nu * nu),
        ("cost", 1),
    ]
    out, off = {}, 0
    for name, n in sizes:
        out[name] = slice(off, off + n)
        off += n
    assert off == tile_doubles(nv)
    return out


_NREF = {
    RES_STATE: lambda nv: 2 * nv,
    RES_CONTROL: lambda nv: nv,
    RES_CONTROL_GRAV: lambda nv: 0,
    RES_FRAME_PLACEMENT: lambda nv: 12,
    RES_FRAME_TRANSLATION: lambda nv: 3,
    RES_FRAME_ROTATION: lambda nv: 9,
    RES_FRAME_VELOCITY: lambda nv: 6,
    RES_COLLISION: lambda nv: 0,
}
_NR = {
    RES_STATE: lambda nv: 2 * nv,
    RES_CONTROL: lambda nv: nv,
    RES_CONTROL_GRAV: lambda nv: nv,
    RES_FRAME_PLACEMENT: lambda nv: 6,
    RES_FRAME_TRANSLATION: lambda nv: 3,
    RES_FRAME_ROTATION: lambda nv: 3,
    RES_FRAME_VELOCITY: lambda nv: 6,
    RES_COLLISION: lambda nv: 1,
}


def row_nref(kind: int, nv: int) -> int:
    return _NREF[kind](nv)


def row_nr(kind: int, nv: int) -> int:
    return _NR[kind](nv)


@dataclasses.dataclass
class RowSpec:
    """Python-side description of one cost row (one CostModelSumItem)."""

    kind: int
    activation: int = ACT_WEIGHTED_QUAD
    active: bool = True
    frame: int = 0
    alpha: float = 1.0
    name: str = ""
    frame_b: int = 0
    weight: float = 1.0

    def width(self, nv: int) -> int:
        return 1 + row_nref(self.kind, nv) + row_nr(self.kind, nv)


@dataclasses.dataclass
class ConstraintSpec:
    """Python-side description of one constraint row: lower <= r(x, u) <= upper."""

    kind: int
    lower: T.Any = -np.inf
    upper: T.Any = np.inf
    ref: T.Any = None
    active: bool = True
    frame: int = 0
    frame_b: int = 0
    name: str = ""


def row_offsets(rows: T.Sequence[RowSpec], nv: int) -> list[int]:
    offs, off = [], 0
    for r in rows:
        offs.append(off)
        off += r.width(nv)
    return offs


def ref_stride(running: T.Sequence[RowSpec], terminal: T.Sequence[RowSpec], nv: int) -> int:
    return max(sum(r.width(nv) for r in running), sum(r.width(nv) for r in terminal), 1)


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def _iptr(a: np.ndarray):
    return a.ctypes.data_as(c_int32_p)


class PackedModel:
    """Owns the contiguous arrays an agx_model_desc points to."""

    def __init__(self, table: "T.Any"):
        nv = int(table.nv)
        self.nv = nv
        self.nframes = len(table.frame_names)
        f8 = lambda a, shape: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))  # noqa: E731
        self.parent = np.ascontiguousarray(np.asarray(table.parent, dtype=np.int32))
        self.placement = f8(table.placement, (nv, 12))
        self.axis = f8(table.axis, (nv, 3))
        self.mass = f8(table.mass, (nv,))
        self.com = f8(table.com, (nv, 3))
        self.inertia = f8(table.inertia, (nv, 9))
        self.armature = f8(table.armature, (nv,))
        self.effort_limit = f8(table.effort_limit, (nv,))
        self.gravity = f8(table.gravity, (3,))
        self.frame_parent = np.ascontiguousarray(np.asarray(table.frame_parent, dtype=np.int32).reshape(-1))
        self.frame_placement = f8(table.frame_placement, (max(self.nframes, 0), 12))
        rad = getattr(table, "frame_radius", None)
        hl = getattr(table, "frame_halflen", None)
        self.frame_radius = f8(np.zeros(self.nframes) if rad is None else rad, (self.nframes,))
        self.frame_halflen = f8(np.zeros(self.nframes) if hl is None else hl, (self.nframes,))
        bx = getattr(table, "frame_box", None)
        self.frame_box = f8(np.zeros((self.nframes, 3)) if bx is None else bx, (self.nframes, 3))
        d = ModelDesc()
        d.nv = nv
        d.nframes = self.nframes
        d.parent = _iptr(self.parent)
        d.placement = _dptr(self.placement)
        d.axis = _dptr(self.axis)
        d.mass = _dptr(self.mass)
        d.com = _dptr(self.com)
        d.inertia = _dptr(self.inertia)
        d.armature = _dptr(self.armature)
        d.effort_limit = _dptr(self.effort_limit)
        d.gravity = _dptr(self.gravity)
        d.frame_parent = _iptr(self.frame_parent)
        d.frame_placement = _dptr(self.frame_placement)
        d.frame_radius = _dptr(self.frame_radius)
        d.frame_halflen = _dptr(self.frame_halflen)
        d.frame_box = _dptr(self.frame_box)
        self.desc = d


class PackedOcp:
    """Owns the arrays an agx_ocp_desc points to."""

    def __init__(
        self,
        nv: int,
        timesteps: T.Sequence[float],
        running: T.Sequence[RowSpec],
        terminal: T.Sequence[RowSpec],
        termination_tolerance: float = 1e-3,
        max_qp_iters: int = 200,
        eps_abs: float = 1e-6,
        eps_rel: float = 0.0,
        mu_dynamic: float = 10.0,
        mu_constraint: float = 10.0,
        use_filter_line_search: bool = False,
        running_constraints: T.Sequence[ConstraintSpec] = (),
        terminal_constraints: T.Sequence[ConstraintSpec] = (),
    ):
        assert len(running) <= AGX_MAX_ROWS and len(terminal) <= AGX_MAX_ROWS
        self.nv = nv
        self.running = list(running)
        self.terminal = list(terminal)
        self.dt = np.ascontiguousarray(np.asarray(timesteps, dtype=np.float64))
        self.horizon = int(self.dt.size)
        self._rr = (CostRow * max(len(running), 1))()
        self._tr = (CostRow * max(len(terminal), 1))()
        for arr, rows in ((self._rr, running), (self._tr, terminal)):
            for i, r in enumerate(rows):
                arr[i].kind = r.kind
                arr[i].activation = r.activation
                arr[i].active = 1 if r.active else 0
                arr[i].frame = r.frame
                arr[i].frame_b = r.frame_b
                arr[i].alpha = r.alpha
                arr[i].weight = r.weight
        self.running_constraints = list(running_constraints)
        self.terminal_constraints = list(terminal_constraints)
        self._rc = (ConstraintRow * max(len(self.running_constraints), 1))()
        self._tc = (ConstraintRow * max(len(self.terminal_constraints), 1))()
        self._cbuf = []  # keeps the bound / reference arrays alive
        for arr, cons in ((self._rc, self.running_constraints), (self._tc, self.terminal_constraints)):
            for i, c in enumerate(cons):
                nref, nr = row_nref(c.kind, nv), row_nr(c.kind, nv)
                lo = np.ascontiguousarray(np.broadcast_to(np.asarray(c.lower, dtype=np.float64), (nr,)).copy())
                up = np.ascontiguousarray(np.broadcast_to(np.asarray(c.upper, dtype=np.float64), (nr,)).copy())
                rf = np.zeros(max(nref, 1)) if c.ref is None else np.ascontiguousarray(np.asarray(c.ref, dtype=np.float64).reshape(-1))
                assert rf.size >= nref
                self._cbuf += [lo, up, rf]
                arr[i].kind = c.kind
                arr[i].active = 1 if c.active else 0
                arr[i].frame = c.frame
                arr[i].frame_b = c.frame_b
                arr[i].ref = _dptr(rf)
                arr[i].lower = _dptr(lo)
                arr[i].upper = _dptr(up)
        d = OcpDesc()
        d.horizon = self.horizon
        d.dt = _dptr(self.dt)
        d.n_running_rows = len(running)
        d.running_rows = C.cast(self._rr, C.POINTER(CostRow))
        d.n_terminal_rows = len(terminal)
        d.terminal_rows = C.cast(self._tr, C.POINTER(CostRow))
        d.termination_tolerance = termination_tolerance
        d.max_qp_iters = max_qp_iters
        d.eps_abs = eps_abs
        d.eps_rel = eps_rel
        d.mu_dynamic = mu_dynamic
        d.mu_constraint = mu_constraint
        d.use_filter_line_search = 1 if use_filter_line_search else 0
        d.n_running_constraints = len(self.running_constraints)
        d.running_constraints = C.cast(self._rc, C.POINTER(ConstraintRow))
        d.n_terminal_constraints = len(self.terminal_constraints)
        d.terminal_constraints = C.cast(self._tc, C.POINTER(ConstraintRow))
        self.desc = d
        self.stride = ref_stride(running, terminal, nv)
        self.running_offsets = row_offsets(running, nv)
        self.terminal_offsets = row_offsets(terminal, nv)

    # -- reference tile helpers -------------------------------------------
    def new_ref_tile(self, batch: int) -> np.ndarray:
        """[B][T+1][stride] tile with item weights 1, activation weights 1."""
        tile = np.zeros((batch, self.horizon + 1, self.stride))
        for rows, offs, sl in (
            (self.running, self.running_offsets, slice(0, self.horizon)),
            (self.terminal, self.terminal_offsets, slice(self.horizon, self.horizon + 1)),
        ):
            for r, o in zip(rows, offs):
                nref, nr = row_nref(r.kind, self.nv), row_nr(r.kind, self.nv)
                tile[:, sl, o] = r.weight
                tile[:, sl, o + 1 + nref : o + 1 + nref + nr] = 1.0
                if r.kind in (RES_FRAME_PLACEMENT, RES_FRAME_ROTATION):
                    tile[:, sl, o + 1 : o + 10] = np.eye(3).reshape(9)
        return tile

    def row_view(self, tile: np.ndarray, terminal: bool, row: int):
        """(item_weight, reference, activation_weights) views of one row."""
        rows = self.terminal if terminal else self.running
        offs = self.terminal_offsets if terminal else self.running_offsets
        sl = slice(self.horizon, self.horizon + 1) if terminal else slice(0, self.horizon)
        r, o = rows[row], offs[row]
        nref, nr = row_nref(r.kind, self.nv), row_nr(r.kind, self.nv)
        return (
            tile[:, sl, o],
            tile[:, sl, o + 1 : o + 1 + nref],
            tile[:, sl, o + 1 + nref : o + 1 + nref + nr],
        )

    def default_frames(self, batch: int) -> np.ndarray:
        fr = np.full((batch, self.horizon + 1, AGX_MAX_ROWS), -1, dtype=np.int32)
        return fr
