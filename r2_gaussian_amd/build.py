"""Builds r2_gaussian_amd/libr2hip.so from csrc/*.hip with plain hipcc for gfx950 (in-tree, no JIT cache).

    python -m r2_gaussian_amd.build [--force]

Geometry translation units (``*_geom.hip``, ``knn.hip``) are compiled with ``-ffp-contract=off``: every
float that can feed an integer decision (radius, tile rectangle, sort key) is a separately rounded
IEEE-754 op, which is what makes tile / sort indices bit-exact against the CPU oracle.  The render
kernels keep FMA contraction (they are VALU-bound and tolerance-checked).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libr2hip.so")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function"]
EXACT = ["-ffp-contract=off"]
# -fno-slp-vectorize: hipcc's SLP pass would otherwise pack adjacent scalar f32 ops into v_pk_*_f32 -- with the register
# shuffling it adds around them the render backward ran ~7x slower (532 us against 75, round 1).  The packed instructions
# themselves, placed by hand on operands that already sit in register pairs, issue ~1.7x slower than the plain forms
# ("v_pk"): two numbers for two different things -- the pass, and the instruction.
FAST = ["-ffp-contract=fast", "-fno-slp-vectorize"]

SOURCES = {
    "binning.hip": FAST,
    "radix_sort.hip": FAST,
    "depth_order.hip": FAST,
    "raster_geom.hip": EXACT,
    "raster_render.hip": FAST,
    "raster_api.hip": FAST,
    "raster_tilefirst.hip": FAST,
    "voxel_geom.hip": EXACT,
    "voxel_render.hip": FAST,
    "voxel_api.hip": FAST,
    "voxel_small.hip": FAST,
    "voxel_sticks.hip": FAST,
    "knn.hip": EXACT,
    "loss_ops.hip": FAST,
    "densify_ops.hip": EXACT,
    "fdk.hip": FAST,
    "dispatch.hip": FAST,
}


def _strip_comments(text):
    """C / C++ source without comments and blank lines (string and character literals are left alone)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":   # literal: copy through the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    return "\n".join(ln.rstrip() for ln in "".join(out).splitlines() if ln.strip())


def source_hash():
    """sha256 over the kernel sources (comments and blank lines stripped) + build flags: identifies WHICH kernels a
    measurement (e.g. the PMC summary under profiles/) belongs to, independently of rebuilds and of edits to comments."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp")))
    for f in files:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "r", encoding="utf-8", errors="replace") as fh:
            h.update(_strip_comments(fh.read()).encode())
    with open(os.path.join(HERE, "..", "include", "r2hip.h"), "r", encoding="utf-8", errors="replace") as fh:
        h.update(_strip_comments(fh.read()).encode())
    h.update(repr((COMMON[:7], EXACT, FAST, sorted(SOURCES.items()))).encode())
    return h.hexdigest()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(HERE, "..", "include", "r2hip.h"))
    hdrs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, flags, force, hdr_mtime):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace(".hip", ".o"))
    if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), hdr_mtime):
        return o, False
    cmd = [_hipcc()] + COMMON + flags + ["-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-8000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return o, True


def build(force=False, verbose=False, extra_flags=None, out=None):
    """extra_flags/out: experiment builds (kernel ablations for profiling) into a separate .so."""
    global OBJ, LIB
    if extra_flags or out:
        tag = os.path.splitext(os.path.basename(out or "libr2hip_exp.so"))[0]
        OBJ = os.path.join(HERE, "csrc", "build_" + tag)
        LIB = os.path.join(HERE, os.path.basename(out or "libr2hip_exp.so"))
        COMMON.extend(extra_flags or [])
    os.makedirs(OBJ, exist_ok=True)
    present = {k: v for k, v in SOURCES.items() if os.path.exists(os.path.join(CSRC, k))}
    missing = sorted(set(SOURCES) - set(present))
    if missing:
        raise RuntimeError("missing kernel sources: %s" % missing)
    hdr_mtime = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(present))) as ex:
        res = list(ex.map(lambda kv: _compile(kv[0], kv[1], force, hdr_mtime), present.items()))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        if verbose:
            print("linked", LIB)
    return LIB


SHIM = os.path.join(HERE, "_r2shim.so")


def build_shim(force=False, verbose=False):
    """r2_gaussian_amd/_r2shim.so: the torch boundary (csrc/torch_shim.cpp) as a compiled pybind11 module, plain g++
    against the installed torch headers, linked to libr2hip.so next to it.  Host code only (no GPU needed to build)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "torch_shim.cpp")
    hdr = os.path.join(HERE, "..", "include", "r2hip.h")
    if not os.path.exists(LIB):
        raise RuntimeError("build libr2hip.so first")
    newest = max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(os.path.abspath(__file__)))
    if not force and os.path.exists(SHIM) and os.path.getmtime(SHIM) >= newest:
        return SHIM
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
           "-Wno-unused-function", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi()),
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-DTORCH_EXTENSION_NAME=_r2shim", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"]
    cmd += ["-I" + d for d in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"]]
    cmd += [src, "-o", SHIM] + ["-L" + d for d in ce.library_paths()] + ["-Wl,-rpath," + d for d in ce.library_paths()]
    cmd += ["-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-L" + HERE, "-l:libr2hip.so", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch shim build failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
    if verbose:
        print("built", SHIM)
    return SHIM


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[len("--out="):] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build(force="--force" in sys.argv, verbose=True, extra_flags=extra or None, out=outs[0] if outs else None))
    if not extra and not outs:
        print(build_shim(force="--force" in sys.argv, verbose=True))
