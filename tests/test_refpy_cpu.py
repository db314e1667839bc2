"""CPU: oracle/refpy.py -- the reference's own Python made visible to the run-unmodified GPU tests (tests/test_reference_python_gpu.py).
Where the reference tree or the packed copy exists: the tarball holds the reference's files byte for byte (nothing patched), the
stand-ins cover exactly the third-party modules that are missing, and the reference's packages import on top of this repo's drop-in
shims (module import needs no GPU)."""
import hashlib
import importlib
import os
import sys
import tarfile

import pytest

from oracle import refpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_reference_python_is_the_reference_byte_for_byte():
    if not os.path.isdir(os.path.join(refpy.REF, "r2_gaussian")):
        pytest.skip("no reference tree on this machine")
    assert refpy.pack() and os.path.exists(refpy.TARBALL)
    with tarfile.open(refpy.TARBALL, "r:gz") as tf:
        names = [m.name for m in tf.getmembers() if m.isfile()]
        assert "train.py" in names and "r2_gaussian/gaussian/render_query.py" in names and "r2_gaussian/gaussian/gaussian_model.py" in names
        assert not [n for n in names if "submodules" in n or n.endswith((".cu", ".cpp", ".h"))]
        for n in names:
            with open(os.path.join(refpy.REF, n), "rb") as f:
                assert hashlib.sha256(tf.extractfile(n).read()).digest() == hashlib.sha256(f.read()).digest(), n
    # the copy never enters the history
    with open(os.path.join(ROOT, ".gitignore")) as f:
        assert "oracle/_ref/" in f.read().split()


def test_reference_packages_import_on_the_drop_in_shims():
    if not refpy.available():
        pytest.skip("the reference's Python is not on this machine")
    t = refpy.tree()
    stubbed = refpy.stub_missing_third_party()
    assert set(stubbed) <= set(refpy.STUBS)
    sys.path.insert(0, t)
    try:
        for m in ("r2_gaussian.arguments", "r2_gaussian.gaussian", "r2_gaussian.dataset", "r2_gaussian.utils.loss_utils",
                  "r2_gaussian.utils.image_utils"):
            importlib.import_module(m)
        import simple_knn._C as knn
        import xray_gaussian_rasterization_voxelization as drop_in
        assert os.path.realpath(drop_in.__file__).startswith(os.path.realpath(ROOT))
        assert os.path.realpath(knn.__file__).startswith(os.path.realpath(ROOT))
        rq = sys.modules["r2_gaussian.gaussian.render_query"]
        assert rq.GaussianRasterizer is drop_in.GaussianRasterizer and rq.GaussianVoxelizer is drop_in.GaussianVoxelizer
    finally:
        sys.path.remove(t)
