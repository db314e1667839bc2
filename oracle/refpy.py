"""The REFERENCE's own Python, made visible to tests on machines without /root/reference (TEST INFRASTRUCTURE ONLY).

tests/test_reference_python_gpu.py runs the reference's ``train.py`` / ``GaussianModel`` / ``render()`` / ``query()``
UNMODIFIED on top of the drop-in packages.  The GPU box has no /root/reference, so -- exactly like
oracle/_ref/*.so, the reference's kernels compiled for the CPU -- ``pack()`` (called by __graft_entry__.build() in the build
container) writes ``oracle/_ref/refpy.tar.gz`` with the reference's Python files *as they are* (train.py, test.py,
r2_gaussian/{arguments,dataset,gaussian,utils}/*.py; no submodule sources), and ``tree()`` unpacks it into a temporary
directory at test time.  oracle/_ref/ is git-ignored: nothing of the reference is committed; it travels with the gpurun
snapshot only.

``stub_missing_third_party()`` installs empty stand-ins for the third-party modules the reference imports at module scope
that are not installed in this image and are never CALLED on the exercised path: plyfile (ply I/O; the reference itself saves
pickles), open3d / cv2 / skimage (plot_utils: figures for tensorboard, which is absent too).  Nothing of the reference is
patched.
"""
import os
import sys
import tarfile
import tempfile
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
TARBALL = os.path.join(_HERE, "_ref", "refpy.tar.gz")
_PY_DIRS = ("r2_gaussian/arguments", "r2_gaussian/dataset", "r2_gaussian/gaussian", "r2_gaussian/utils")
_TOP = ("train.py", "test.py", "initialize_pcd.py")
_tree = None


def pack():
    """Write oracle/_ref/refpy.tar.gz from /root/reference (no-op and False when the tree is absent)."""
    if not os.path.isdir(os.path.join(REF, "r2_gaussian")):
        return False
    os.makedirs(os.path.dirname(TARBALL), exist_ok=True)
    tmp = TARBALL + ".tmp"
    with tarfile.open(tmp, "w:gz") as tf:
        for f in _TOP:
            tf.add(os.path.join(REF, f), arcname=f)
        for d in _PY_DIRS:
            for f in sorted(os.listdir(os.path.join(REF, d))):
                if f.endswith(".py"):
                    tf.add(os.path.join(REF, d, f), arcname=d + "/" + f)
    os.replace(tmp, TARBALL)
    return True


def available():
    return os.path.isdir(os.path.join(REF, "r2_gaussian")) or os.path.exists(TARBALL)


def tree():
    """Directory that holds train.py and the r2_gaussian package: /root/reference itself when present, else the unpacked
    tarball (a fresh temporary directory per process); None when neither exists."""
    global _tree
    if _tree is None:
        if os.path.isdir(os.path.join(REF, "r2_gaussian")):
            _tree = REF
        elif os.path.exists(TARBALL):
            d = tempfile.mkdtemp(prefix="r2refpy_")
            with tarfile.open(TARBALL, "r:gz") as tf:
                tf.extractall(d)
            _tree = d
    return _tree


STUBS = ("plyfile", "open3d", "cv2", "skimage", "skimage.measure")


def stub_missing_third_party():
    """-> names stubbed.  Only modules that cannot be imported get a stand-in; attribute access on a stand-in that the
    exercised path never performs would raise AttributeError (so a silent dependency on them cannot hide)."""
    done = []
    for name in STUBS:
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except ImportError:
            pass
        m = types.ModuleType(name)
        m.__r2_stub__ = True
        if name == "plyfile":
            m.PlyData = type("PlyData", (), {})
            m.PlyElement = type("PlyElement", (), {})
        if name == "skimage":
            m.measure = types.ModuleType("skimage.measure")
            sys.modules["skimage.measure"] = m.measure
        sys.modules[name] = m
        done.append(name)
    return done
