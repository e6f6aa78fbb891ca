"""Loader and thin object wrapper of libagimus_hip.so (the product path).

There is no CPU fallback: if the shared library is missing, or no HIP device is
visible, creating a `HipOcp` raises.  The CPU checker under oracle/ is never
imported from here.
"""

from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

from . import _abi

_CSRC = pathlib.Path(__file__).resolve().parent / "csrc"
LIB_PATH = pathlib.Path(os.environ.get("AGX_LIB", _CSRC / "libagimus_hip.so"))  # AGX_LIB: development override
_LIB = None

# Every symbol include/agimus_hip.h declares (checked by the CPU test-suite).
EXPORTED_SYMBOLS = [
    "agx_last_error", "agx_device_count", "agx_row_nref", "agx_row_nr", "agx_ref_stride",
    "agx_model_create", "agx_model_destroy", "agx_ocp_create", "agx_ocp_destroy", "agx_ocp_set_stream",
    "agx_ocp_sync", "agx_ocp_set_refs", "agx_ocp_set_refs_device", "agx_ocp_solve", "agx_ocp_upload_x0",
    "agx_ocp_upload_warmstart", "agx_ocp_solve_resident", "agx_ocp_download", "agx_ocp_download_first", "agx_ocp_first_packed", "agx_ocp_set_geom_placement", "agx_ocp_reset_duals", "agx_traj_generic_create", "agx_traj_set_horizon_indexes", "agx_ocp_feedback_rollout", "agx_ocp_download_x0", "agx_model_frame_jacobian",
    "agx_ocp_shift_warmstart", "agx_ocp_x0_from_prediction", "agx_ocp_integrate", "agx_model_rnea",
    "agx_model_frame_placement", "agx_ocp_get_residuals", "agx_ocp_calc_diff", "agx_ocp_direction",
    "agx_ocp_time_kernel", "agx_ocp_profile", "agx_traj_sine_create", "agx_traj_set_window",
    "agx_traj_get_point", "agx_traj_warmstart_from_reference", "agx_ocp_mpc_step", "agx_ocp_qp_tiles", "agx_ocp_set_quorum",
    "agx_traj_cartesian_sine_create", "agx_ocp_set_refs_async", "agx_ocp_refs_activate", "agx_ocp_refs_wait", "agx_host_alloc", "agx_host_free",
    "agx_ocp_download_async", "agx_ocp_download_wait",
]  # fmt: skip


def _run(cmd, verbose):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError(f"{cmd[0]} failed ({res.returncode}): {res.stderr[-2000:]}")


def build(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU).

    The kernels are templates over the model size; one translation unit with every size takes ~6
    minutes on one core, so the source is compiled once per group of sizes (csrc/agx_front.py: GROUPS)
    in parallel processes -- namespace and entry points suffixed per group -- and linked with a
    generated front that forwards each call to the group owning the handle.  AGX_BUILD_SPLIT=0 builds
    the single translation unit instead."""
    hdr = _CSRC.parent.parent / "include" / "agimus_hip.h"
    srcs = [_CSRC / "agimus_hip.hip"] + sorted(_CSRC.glob("*.hpp")) + [hdr, _CSRC / "agx_front.py"]
    if not force and LIB_PATH.exists() and LIB_PATH.stat().st_mtime >= max(s.stat().st_mtime for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    if os.environ.get("AGX_BUILD_SPLIT", "1") == "0":
        _run(base + ["-shared", "-o", str(LIB_PATH), str(srcs[0])], verbose)
        return LIB_PATH
    import importlib.util

    spec = importlib.util.spec_from_file_location("agx_front", _CSRC / "agx_front.py")
    front = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(front)
    obj = _CSRC.parent.parent / "build" / "obj"
    obj.mkdir(parents=True, exist_ok=True)
    names = [n for _, n, _ in front.prototypes(hdr.read_text())]
    procs = []
    for g in front.GROUPS:
        cmd = base + ["-c", f"-I{hdr.parent}"] + front.rename_flags(names, g) + ["-o", str(obj / f"agx_g{g}.o"), str(srcs[0])]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    (obj / "agx_front.cpp").write_text(front.front_source(hdr.read_text()))
    _run(["g++", "-O2", "-std=c++17", "-fPIC", f"-I{hdr.parent}", "-c", "-o", str(obj / "agx_front.o"), str(obj / "agx_front.cpp")], verbose)
    failed = None
    for cmd, p in procs:
        out, err = p.communicate()
        if verbose or p.returncode != 0:
            print(out, err)
        if p.returncode != 0 and failed is None:
            failed = RuntimeError(f"hipcc failed ({p.returncode}) for {cmd[-1]} [{cmd[-3]}]: {err[-2000:]}")
    if failed:
        raise failed
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(obj / f"agx_g{g}.o") for g in front.GROUPS]
         + [str(obj / "agx_front.o")], verbose)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP path has no CPU fallback)"
            )
        _LIB = C.CDLL(str(LIB_PATH))
        _LIB.agx_last_error.restype = C.c_char_p
    return _LIB


def device_count() -> int:
    return int(lib().agx_device_count())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f8(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        assert a.shape == tuple(shape), f"expected shape {tuple(shape)}, got {a.shape}"
    return a


class HipError(RuntimeError):
    pass


class _PinnedBlock:
    """Owner of one page-locked allocation (agx_host_alloc); freed with the last array that views it."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        lib().agx_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        if lib().agx_host_alloc(C.c_size_t(nbytes), C.byref(self.ptr)) != 0:
            raise HipError(lib().agx_last_error().decode())
        self.buf = (C.c_char * nbytes).from_address(self.ptr.value)

    def __del__(self):
        try:
            lib().agx_host_free.argtypes = [C.c_void_p]
            lib().agx_host_free(self.ptr)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def pinned_array(shape, dtype=np.float64) -> np.ndarray:
    """numpy array over page-locked host memory: transfers to and from it run at the rate of the link and asynchronously."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    block = _PinnedBlock(max(n * dtype.itemsize, 1))
    arr = np.frombuffer(block.buf, dtype=dtype, count=n).reshape(shape)
    arr.flags.writeable = True
    _PINNED[id(block)] = block  # np.frombuffer keeps block.buf alive, not the block: tie the block's life to the array's
    import weakref

    weakref.finalize(arr.base if arr.base is not None else arr, _PINNED.pop, id(block), None)
    return arr


_PINNED: dict = {}


def _chk(rc):
    if rc != 0:
        raise HipError(lib().agx_last_error().decode())


class HipOcp:
    """Device-resident batch of B shooting problems sharing one model and cost table."""

    def __init__(self, table, packed_ocp: _abi.PackedOcp, batch: int = 1, device: int = 0):
        L = lib()
        self.table = table
        self.pm = _abi.PackedModel(table)
        self.po = packed_ocp
        self.nv = self.pm.nv
        self.nx, self.nu = 2 * self.nv, self.nv
        self.T, self.B = packed_ocp.horizon, int(batch)
        self.stride = packed_ocp.stride
        self.tile = _abi.tile_doubles(self.nv)
        self._m = C.c_void_p()
        self._h = C.c_void_p()
        _chk(L.agx_model_create(C.byref(self.pm.desc), C.byref(self._m)))
        rc = L.agx_ocp_create(self._m, C.byref(self.po.desc), self.B, int(device), C.byref(self._h))
        if rc != 0:
            msg = L.agx_last_error().decode()
            L.agx_model_destroy(self._m)
            self._m = None
            raise HipError(msg)

    def close(self):
        if getattr(self, "_h", None):
            lib().agx_ocp_destroy(self._h)
            self._h = None
        if getattr(self, "_m", None):
            lib().agx_model_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- references ---------------------------------------------------------
    def set_refs(self, ref_tile, frames=None):
        ref = _f8(ref_tile, (self.B, self.T + 1, self.stride))
        fr = None
        if frames is not None:
            fr = np.ascontiguousarray(frames, dtype=np.int32)
            assert fr.shape == (self.B, self.T + 1, _abi.AGX_MAX_ROWS)
        _chk(lib().agx_ocp_set_refs(self._h, _p(ref), _p(fr)))

    def set_refs_async(self, ref_tile, frames=None):
        """Stages a tile (ideally a `pinned_array`): a second stream copies it while the solver still works on the current one;
        refs_activate() swaps it in.  Keep the buffer untouched until the solve after its activation has returned."""
        ref = np.asarray(ref_tile)
        assert ref.dtype == np.float64 and ref.flags.c_contiguous and ref.shape == (self.B, self.T + 1, self.stride)
        fr = None
        if frames is not None:
            fr = np.asarray(frames)
            assert fr.dtype == np.int32 and fr.flags.c_contiguous and fr.shape == (self.B, self.T + 1, _abi.AGX_MAX_ROWS)
        self._async_keep = (ref, fr)  # the copy reads the buffers after this call returns
        _chk(lib().agx_ocp_set_refs_async(self._h, _p(ref), _p(fr)))

    def refs_wait(self):
        _chk(lib().agx_ocp_refs_wait(self._h))

    def refs_activate(self):
        """The tile staged by set_refs_async becomes the current one (device-side wait, no host stall)."""
        _chk(lib().agx_ocp_refs_activate(self._h))

    def download_async(self, xs=None, us=None, K=None):
        """Full results into caller-owned arrays (ideally `pinned_array`s) behind the solver's back: complete after download_wait()."""
        for a, shape in ((xs, (self.B, self.T + 1, self.nx)), (us, (self.B, self.T, self.nu)), (K, (self.B, self.T, self.nu, self.nx))):
            assert a is None or (a.dtype == np.float64 and a.flags.c_contiguous and a.shape == shape)
        self._dl_keep = (xs, us, K)
        _chk(lib().agx_ocp_download_async(self._h, _p(xs), _p(us), _p(K)))

    def download_wait(self):
        _chk(lib().agx_ocp_download_wait(self._h))

    def set_stream(self, raw_stream: int):
        _chk(lib().agx_ocp_set_stream(self._h, C.c_void_p(raw_stream)))

    def sync(self):
        _chk(lib().agx_ocp_sync(self._h))

    # -- solve --------------------------------------------------------------
    def solve(self, x0, xs_ws, us_ws, max_iter, max_time=0.0):
        x0 = _f8(x0, (self.B, self.nx))
        xs_ws = _f8(xs_ws, (self.B, self.T + 1, self.nx))
        us_ws = _f8(us_ws, (self.B, self.T, self.nu))
        xs = np.empty_like(xs_ws)
        us = np.empty_like(us_ws)
        K = np.empty((self.B, self.T, self.nu, self.nx))
        st = np.zeros(self.B, dtype=_abi.STATUS_DTYPE)
        _chk(lib().agx_ocp_solve(self._h, _p(x0), _p(xs_ws), _p(us_ws), int(max_iter), C.c_double(max_time),
                                 _p(xs), _p(us), _p(K), _p(st)))  # fmt: skip
        return xs, us, K, st

    def upload_x0(self, x0):
        _chk(lib().agx_ocp_upload_x0(self._h, _p(_f8(x0, (self.B, self.nx)))))

    def upload_warmstart(self, xs, us):
        xs = _f8(xs, (self.B, self.T + 1, self.nx))
        us = _f8(us, (self.B, self.T, self.nu))
        _chk(lib().agx_ocp_upload_warmstart(self._h, _p(xs), _p(us)))

    def solve_resident(self, max_iter, max_time=0.0):
        _chk(lib().agx_ocp_solve_resident(self._h, int(max_iter), C.c_double(max_time)))

    def download(self, want_K=True):
        xs = np.empty((self.B, self.T + 1, self.nx))
        us = np.empty((self.B, self.T, self.nu))
        K = np.empty((self.B, self.T, self.nu, self.nx)) if want_K else None
        st = np.zeros(self.B, dtype=_abi.STATUS_DTYPE)
        _chk(lib().agx_ocp_download(self._h, _p(xs), _p(us), _p(K), _p(st)))
        return xs, us, K, st

    def download_first(self, want_status=True, copy=True):
        """us[0], K[0], xs[1] (+ status) of every instance: one packed transfer into the handle's
        pinned buffer.  copy=False returns views into that buffer, valid until the next call."""
        ptr = C.POINTER(C.c_double)()
        stride = C.c_int(0)
        _chk(lib().agx_ocp_first_packed(self._h, C.byref(ptr), C.byref(stride)))
        addr = C.addressof(ptr.contents)
        cached = getattr(self, "_first_view", None)
        if cached is None or cached[0] != addr:
            cached = (addr, np.ctypeslib.as_array(ptr, shape=(self.B, stride.value)))
            self._first_view = cached  # the pinned buffer lives as long as the handle
        buf = cached[1]
        nu, nx = self.nu, self.nx
        nk = nu * nx
        us0 = buf[:, :nu]
        K0 = buf[:, nu:nu + nk].reshape(self.B, nu, nx)
        x1 = buf[:, nu + nk:nu + nk + nx]
        if copy:
            us0, K0, x1 = us0.copy(), K0.copy(), x1.copy()
        st = None
        if want_status:
            q = buf[:, nu + nk + nx:]
            names = ("kkt", "cost", "merit", "gap_norm", "iter", "qp_iters", "solved", "flags")
            if copy:
                st = np.zeros(self.B, dtype=_abi.STATUS_DTYPE)
                for k, name in enumerate(names):
                    st[name] = q[:, k] if k < 4 else q[:, k].astype(np.int32)
            else:
                st = {name: q[:, k] for k, name in enumerate(names)}  # views (float64) into the pinned buffer
        return us0, K0, x1, st

    def set_geom_placement(self, frame: int, se3_12):
        """OCPBaseCroco.update_geometry_placement (ocp_base_croco.py:110-132) for a geometry frame."""
        _chk(lib().agx_ocp_set_geom_placement(self._h, int(frame), _p(_f8(se3_12).reshape(12))))

    def set_quorum(self, sqp_fraction: float = 1.0, qp_fraction: float = 1.0):
        """Batch policy: end the SQP / ADMM loops once this fraction of the instances has finished (see agimus_hip.h)."""
        _chk(lib().agx_ocp_set_quorum(self._h, C.c_double(sqp_fraction), C.c_double(qp_fraction)))

    def reset_duals(self):
        _chk(lib().agx_ocp_reset_duals(self._h))

    def shift_warmstart(self):
        _chk(lib().agx_ocp_shift_warmstart(self._h))

    def x0_from_prediction(self):
        _chk(lib().agx_ocp_x0_from_prediction(self._h))

    def integrate(self, x, u):
        x = _f8(x).reshape(-1, self.nx)
        u = _f8(u).reshape(-1, self.nu)
        out = np.empty_like(x)
        _chk(lib().agx_ocp_integrate(self._h, x.shape[0], _p(x), _p(u), _p(out)))
        return out

    def rnea(self, q, v, a):
        q = _f8(q).reshape(-1, self.nv)
        v = _f8(v).reshape(-1, self.nv)
        a = _f8(a).reshape(-1, self.nv)
        tau = np.empty_like(q)
        _chk(lib().agx_model_rnea(self._h, q.shape[0], _p(q), _p(v), _p(a), _p(tau)))
        return tau

    def frame_placement(self, frame: int, q):
        q = _f8(q).reshape(-1, self.nv)
        out = np.empty((q.shape[0], 12))
        _chk(lib().agx_model_frame_placement(self._h, q.shape[0], int(frame), _p(q), _p(out)))
        return out

    def frame_jacobian(self, frame: int, q, local: bool = False):
        """pinocchio.getFrameJacobian: [n, 6, nv], rows linear | angular; LOCAL_WORLD_ALIGNED or LOCAL."""
        q = _f8(q).reshape(-1, self.nv)
        J = np.empty((q.shape[0], 6, self.nv))
        _chk(lib().agx_model_frame_jacobian(self._h, q.shape[0], int(frame), 1 if local else 0, _p(q), _p(J)))
        return J

    def residuals(self, row: int):
        nr = _abi.row_nr(self.po.running[row].kind, self.nv)
        out = np.empty((self.B, self.T, nr))
        _chk(lib().agx_ocp_get_residuals(self._h, int(row), _p(out)))
        return out

    # -- kernel level -------------------------------------------------------
    def calc_diff(self, want_tiles=True):
        tiles = np.empty((self.B, self.T + 1, self.tile)) if want_tiles else None
        _chk(lib().agx_ocp_calc_diff(self._h, _p(tiles)))
        return tiles

    def qp_tiles(self):
        """QP tile and aux tile of every node at the resident point: dicts of blocks (see agx_ocp_qp_tiles)."""
        qs, as_ = C.c_int(0), C.c_int(0)
        _chk(lib().agx_ocp_qp_tiles(self._h, None, None, C.byref(qs), C.byref(as_)))
        qt = np.empty((self.B, self.T + 1, qs.value))
        aux = np.empty((self.B, self.T + 1, as_.value))
        _chk(lib().agx_ocp_qp_tiles(self._h, _p(qt), _p(aux), C.byref(qs), C.byref(as_)))
        nv, ld = self.nv, (8 if self.nv <= 8 else 32)
        b2 = nv * ld
        blk = lambda a, off: a[..., off:off + b2].reshape(self.B, self.T + 1, nv, ld)[..., :nv]  # noqa: E731
        q = {name: blk(qt, k * b2) for k, name in enumerate(("Hqq", "Hqv", "Hvv", "Hqw", "Hvw", "Hww"))}
        o = 6 * b2
        q["gx"] = qt[..., o:o + 2 * nv]
        q["gw"] = qt[..., o + 2 * ld:o + 2 * ld + nv]
        q["f"] = qt[..., o + 3 * ld:o + 3 * ld + 2 * nv]
        q["cost"] = qt[..., o + 5 * ld]
        a = {name: blk(aux, k * b2) for k, name in enumerate(("M", "tq", "tv", "Lqq"))}
        o = 4 * b2
        a["Lvv"], a["Luu"], a["Lu"] = aux[..., o:o + nv], aux[..., o + ld:o + ld + nv], aux[..., o + 2 * ld:o + 2 * ld + nv]
        return q, a

    def direction(self):
        K = np.empty((self.B, self.T, self.nu, self.nx))
        k = np.empty((self.B, self.T, self.nu))
        dx = np.empty((self.B, self.T + 1, self.nx))
        du = np.empty((self.B, self.T, self.nu))
        kkt = np.empty(self.B)
        _chk(lib().agx_ocp_direction(self._h, _p(K), _p(k), _p(dx), _p(du), _p(kkt)))
        return K, k, dx, du, kkt

    def time_kernel(self, which: int, reps: int) -> float:
        ms = C.c_double()
        _chk(lib().agx_ocp_time_kernel(self._h, int(which), int(reps), C.byref(ms)))
        return ms.value

    def profile(self, enable: bool):
        """Switch in-situ kernel timing on/off; returns (ms_sum[3], count[3]) accumulated so far."""
        ms = (C.c_double * 3)()
        cnt = (C.c_longlong * 3)()
        _chk(lib().agx_ocp_profile(self._h, 1 if enable else 0, ms, cnt))
        return list(ms), list(cnt)

    # -- resident trajectory -------------------------------------------------
    def sine_trajectory(self, n_points, dt, q0, amp, pulsation, scale_duration, t0, w_q, w_qdot, w_effort, w_pose, frame):
        B, nv = self.B, self.nv
        bc = lambda a, shape: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), shape))  # noqa: E731
        args = [bc(q0, (B, nv)), bc(amp, (B, nv)), bc(pulsation, (B, nv)), bc(scale_duration, (B, nv)), bc(t0, (B,)),
                bc(w_q, (nv,)), bc(w_qdot, (nv,)), bc(w_effort, (nv,)), bc(w_pose, (6,))]  # fmt: skip
        _chk(lib().agx_traj_sine_create(self._h, int(n_points), C.c_double(dt), *[_p(a) for a in args], int(frame)))

    def generic_trajectory(self, q, dq, ddq, w_q, w_qdot, w_effort, w_pose, frame):
        """Resident trajectory from samples q, dq, ddq [B][n_points][nv] (GenericTrajectory upstream)."""
        B, nv = self.B, self.nv
        q, dq, ddq = (_f8(a).reshape(B, -1, nv) for a in (q, dq, ddq))
        assert q.shape == dq.shape == ddq.shape
        bc = lambda a, shape: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), shape))  # noqa: E731
        _chk(lib().agx_traj_generic_create(self._h, int(q.shape[1]), _p(q), _p(dq), _p(ddq), _p(bc(w_q, (nv,))), _p(bc(w_qdot, (nv,))),
                                           _p(bc(w_effort, (nv,))), _p(bc(w_pose, (6,))), int(frame)))

    def cartesian_sine_trajectory(self, n_points, dt, q0, amp, pulsation, w_q, w_qdot, w_effort, w_pose, frame,
                                  scale_duration=0.2, precision=1e-5, it_max=10000):
        """Resident trajectory of the Cartesian sine generator (SinusWaveCartesianSpace upstream): inverse kinematics of
        every instance and point on the device.  q0 [B][nv], amp / pulsation [B][3]."""
        B, nv = self.B, self.nv
        bc = lambda a, shape: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), shape))  # noqa: E731
        _chk(lib().agx_traj_cartesian_sine_create(self._h, int(n_points), C.c_double(dt), _p(bc(q0, (B, nv))), _p(bc(amp, (B, 3))),
                                                  _p(bc(pulsation, (B, 3))), C.c_double(scale_duration), C.c_double(precision), int(it_max),
                                                  _p(bc(w_q, (nv,))), _p(bc(w_qdot, (nv,))), _p(bc(w_effort, (nv,))), _p(bc(w_pose, (6,))),
                                                  int(frame)))

    def set_horizon_indexes(self, idx):
        """TrajectoryBuffer.horizon_indexes for the resident trajectory (None = uniform)."""
        a = None if idx is None else np.ascontiguousarray(np.asarray(idx, dtype=np.int32).reshape(self.T + 1))
        _chk(lib().agx_traj_set_horizon_indexes(self._h, _p(a)))

    def set_window(self, k0: int):
        _chk(lib().agx_traj_set_window(self._h, int(k0)))

    def traj_point(self, k: int):
        B, nv = self.B, self.nv
        q, v, a, u = (np.empty((B, nv)) for _ in range(4))
        pose = np.empty((B, 12))
        _chk(lib().agx_traj_get_point(self._h, int(k), _p(q), _p(v), _p(a), _p(u), _p(pose)))
        return q, v, a, u, pose

    def warmstart_from_reference(self):
        _chk(lib().agx_traj_warmstart_from_reference(self._h))

    def mpc_step(self, k0: int, max_iter: int, first):
        """first: True/1 warm start from the reference, False/0 x0 <- previous xs[1], 2 x0 as set by the caller."""
        _chk(lib().agx_ocp_mpc_step(self._h, int(k0), int(max_iter), int(first)))

    def download_x0(self):
        x0 = np.empty((self.B, self.nx))
        _chk(lib().agx_ocp_download_x0(self._h, _p(x0)))
        return x0

    def feedback_rollout(self, n_substeps: int, dt_sub: float, disturbance=None):
        """u = us[0] + K[0] (x0 - x) on the model for n_substeps of dt_sub; the end state becomes x0."""
        d = None if disturbance is None else _f8(disturbance).reshape(self.B, self.nu)
        _chk(lib().agx_ocp_feedback_rollout(self._h, int(n_substeps), C.c_double(dt_sub), _p(d)))
