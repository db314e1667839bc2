"""CPU: the C-ABI library loads and exports every symbol include/r2hip.h declares; the ctypes binding covers the same
set; the product path refuses to run without a GPU / without the library (no fallback).  No compute calls here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "r2hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"R2_API\s+[A-Za-z_ \*]+?\b(r2_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for s in ("r2_raster_forward", "r2_raster_backward", "r2_mark_visible", "r2_voxel_forward", "r2_voxel_backward",
              "r2_knn_dist2", "r2_abi_version", "r2_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from r2_gaussian_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libr2hip.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), "libr2hip.so does not export %s" % s
    L.r2_abi_version.restype = ctypes.c_int
    assert L.r2_abi_version() == _lib.R2_ABI_VERSION


def test_ctypes_binding_matches_header():
    from r2_gaussian_amd import _lib
    assert sorted(_lib.exported_symbols()) == declared_symbols()
    _lib.lib()   # sets argtypes on every symbol; raises on a missing one


def test_only_declared_symbols_are_exported():
    """-fvisibility=hidden: nothing but the r2_* C ABI leaks out of the library."""
    from r2_gaussian_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert set(declared_symbols()) <= set(exported)
    leaked = [s for s in exported if not s.startswith("r2_") and not s.startswith("_")]
    assert not leaked, leaked


def test_stage_names():
    from r2_gaussian_amd import _lib
    names = _lib.stage_names()
    assert "raster.render_bwd" in names and "raster.render_fwd" in names and len(set(names)) == len(names)


def test_cpu_tensors_are_refused():
    """The product path has no CPU fallback: CPU tensors raise instead of silently computing something else."""
    from r2_gaussian_amd import _C, _lib
    x = torch.zeros(4, 3)
    e = torch.empty(0)
    with pytest.raises(_lib.R2HipError):
        _C.rasterize_gaussians(x, torch.zeros(4, 1), torch.ones(4, 3), torch.ones(4, 4), 1.0, e, torch.eye(4), torch.eye(4),
                               1.0, 1.0, 16, 16, torch.zeros(3), False, 1, False)
    with pytest.raises(_lib.R2HipError):
        _C.distCUDA2(x)
    with pytest.raises(RuntimeError):
        _C.rasterize_gaussians(torch.zeros(4, 2), torch.zeros(4, 1), e, e, 1.0, e, torch.eye(4), torch.eye(4), 1.0, 1.0, 16, 16,
                               torch.zeros(3), False, 1, False)


def test_missing_library_fails_loudly(tmp_path):
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['R2HIP_LIB'] = %r\n"
            "from r2_gaussian_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept _lib.R2HipError as e:\n    print('LOUD', e)\n" % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "LOUD" in out.stdout and "no fallback" in out.stdout, out.stdout + out.stderr


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under r2_gaussian_amd/ (nor the import shims) may reference it."""
    bad = []
    for base in ("r2_gaussian_amd", "simple_knn", "xray_gaussian_rasterization_voxelization"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|r2_oracle|libr2oracle|oracle/_ref", txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_compiled_torch_boundary_loads_and_refuses_cpu_tensors():
    """_r2shim.so (csrc/torch_shim.cpp) is the compiled twin of _C.py's ctypes glue: it must import against the installed
    torch, report the library's ABI version, and refuse CPU tensors like the ctypes path does."""
    import torch
    from r2_gaussian_amd import _C, _lib
    sh = _C._shim()
    if sh is None:
        pytest.skip("_r2shim.so not built (python -m r2_gaussian_amd.build)")
    assert sh.abi_version() == _lib.R2_ABI_VERSION
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "voxelize_gaussians", "voxelize_gaussians_backward"):
        assert hasattr(sh, name)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sh.rasterize_gaussians(torch.zeros(4, 3), torch.ones(4, 1), torch.zeros(4, 3), torch.zeros(4, 4), 1.0,
                               torch.Tensor([]), torch.eye(4), torch.eye(4), 1.0, 1.0, 8, 8, torch.zeros(3), False, 0, False, 0)


def test_path_stat_names_cover_every_counter():
    """r2_path_stats (csrc/dispatch.hpp): every counter has a distinct name of the form operator.chain[.reason]."""
    from r2_gaussian_amd import _lib
    L = _lib.lib()
    n = L.r2_path_stat_count()
    names = [L.r2_path_stat_name(i) for i in range(n)]
    assert n > 20 and all(names) and len(set(names)) == n
    assert all(nm.decode().split(".")[0] in ("raster", "voxel") for nm in names)
    assert L.r2_path_stat_name(n) is None
    st = _lib.path_stats()
    assert set(st) == {nm.decode() for nm in names} and all(v >= 0 for v in st.values())
