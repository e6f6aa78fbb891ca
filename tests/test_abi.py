"""C-ABI surface checks that need no GPU: the product library loads, exports every
symbol include/agimus_hip.h declares, and refuses to compute without a HIP device."""
import ctypes as C
import pathlib
import re

import numpy as np
import pytest

from agimus_controller_amd import _abi, backend
from agimus_controller_amd.factory import robot_tables as rt

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    backend.build()
    return backend.lib()


def test_header_and_python_symbol_lists_agree():
    hdr = (ROOT / "include" / "agimus_hip.h").read_text()
    declared = set(re.findall(r"\b(agx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(backend.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in backend.EXPORTED_SYMBOLS:
        assert hasattr(lib, name), name


def test_layout_helpers_match_python(lib):
    for kind in range(8):
        for nv in (1, 7, 30):
            assert lib.agx_row_nref(kind, nv) == _abi.row_nref(kind, nv)
            assert lib.agx_row_nr(kind, nv) == _abi.row_nr(kind, nv)
    po = _abi.PackedOcp(7, [0.01] * 4, [_abi.RowSpec(_abi.RES_CONTROL), _abi.RowSpec(_abi.RES_STATE), _abi.RowSpec(_abi.RES_FRAME_PLACEMENT)],
                        [_abi.RowSpec(_abi.RES_STATE)])
    assert lib.agx_ref_stride(C.byref(po.desc), 7) == po.stride == 15 + 29 + 19
    assert _abi.tile_doubles(7) == 673  # 673 doubles per node at nv = 7
    assert C.sizeof(_abi.Status) == 48


def test_model_create_validates(lib):
    table = rt.panda_table()
    pm = _abi.PackedModel(table)
    h = C.c_void_p()
    assert lib.agx_model_create(C.byref(pm.desc), C.byref(h)) == 0
    lib.agx_model_destroy(h)
    bad = _abi.PackedModel(table)
    bad.parent[3] = 5  # child before parent
    assert lib.agx_model_create(C.byref(bad.desc), C.byref(h)) != 0
    assert b"parent" in lib.agx_last_error()


def test_front_forwards_by_handle_and_refuses_unknown_ones(lib):
    """Split build (backend.build): the generated front covers every declared entry point, routes a
    handle to the translation unit of its model size, and fails loudly on a handle it never issued."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("agx_front", ROOT / "agimus_controller_amd" / "csrc" / "agx_front.py")
    front = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(front)
    protos = front.prototypes((ROOT / "include" / "agimus_hip.h").read_text())
    assert {n for _, n, _ in protos} == set(backend.EXPORTED_SYMBOLS)
    # two translation units (capacities 7 | 30, 32) serve every model size up to AGX_MAX_NV
    assert sorted(nv for sizes in front.GROUPS.values() for nv in sizes) == list(range(1, 33)) and len(front.GROUPS) == 2
    # models of different groups live side by side
    handles = []
    for table in (rt.panda_table(), rt.pendulum_table(), rt.chain_table(4, seed=1)):
        pm = _abi.PackedModel(table)
        h = C.c_void_p()
        assert lib.agx_model_create(C.byref(pm.desc), C.byref(h)) == 0
        handles.append(h)
    for h in handles:
        lib.agx_model_destroy(h)
    bogus = C.c_void_p(0x1234)
    lib.agx_ocp_sync.argtypes = [C.c_void_p]
    assert lib.agx_ocp_sync(bogus) != 0
    assert b"unknown handle" in lib.agx_last_error()
    # a destroyed handle is forgotten
    out = C.c_void_p()
    po = _abi.PackedOcp(7, [0.01] * 2, [_abi.RowSpec(_abi.RES_STATE)], [_abi.RowSpec(_abi.RES_STATE)])
    lib.agx_ocp_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert lib.agx_ocp_create(handles[0], C.byref(po.desc), 1, 0, C.byref(out)) != 0
    assert b"unknown handle" in lib.agx_last_error()


def test_no_silent_cpu_fallback(lib):
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    if backend.device_count() > 0:
        pytest.skip("a HIP device is visible")
    table = rt.panda_table()
    po = _abi.PackedOcp(7, [0.01] * 4, [_abi.RowSpec(_abi.RES_STATE)], [_abi.RowSpec(_abi.RES_STATE)])
    with pytest.raises(backend.HipError, match="no HIP device"):
        backend.HipOcp(table, po, 1)


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "agimus_controller_amd"
    for f in pkg.rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
    for f in list((pkg / "csrc").glob("*.h*")) + list((pkg / "csrc").glob("*.hip")):
        assert "oracle/" not in f.read_text().replace("under oracle/", "").replace("oracle/ uses", ""), f
