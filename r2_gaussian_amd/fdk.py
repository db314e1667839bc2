"""FDK initialisation without TIGRE: ``recon_volume`` / ``init_pcd`` of the reference
(r2_gaussian/utils/ct_utils.py:17-27, initialize_pcd.py:36-90) on the MI355X kernels of csrc/fdk.hip.

The reference hands its training projections to ``tigre.algorithms.fdk`` (a third-party CUDA toolbox, v2.3, README.md:47-50),
thresholds the reconstructed volume and samples ``n_points`` voxels as initial Gaussian centres.  Here the reconstruction is
two kernels behind the C ABI (``r2_fdk_filter`` + ``r2_fdk_backproject``); the filter design (a 1-D FFT of at most 8192
points, once) and the sampling (numpy, exactly the reference's statements) stay on the host.

Geometry: voxels are projected with the camera matrices the rasterizer renders with (``scene.make_view`` restates
dataset_readers.py:119-191), so the volume comes out in ``query()``'s [nx,ny,nz] layout and neither the vertical flip nor
the transpose of ct_utils.py:20,26 (TIGRE's conventions) appears.  Like TIGRE's FDK the scan is assumed to cover 2 pi with
equally weighted views; ``offDetector`` is ignored, as it is by the reference's cameras.
"""
import math

import numpy as np
import torch

from . import _lib
from . import scene as S
from ._C import _on_device, _require_gpu, _stream

_F32 = torch.float32
FILTERS = ("ram_lak", "shepp_logan", "cosine", "hamming", "hann")


def filter_length(n_u, n_v):
    """TIGRE pads to the next power of two above twice the larger detector side (at least 64)."""
    return int(max(64, 2 ** int(math.ceil(math.log2(2 * max(n_u, n_v))))))


def ramp_taps(n_u, n_v=None, name="ram_lak"):
    """The 2 n_u - 1 spatial taps of TIGRE's row filter (``filtering.py``: |FFT| of the band-limited ramp h[0] = 1/4,
    h[odd n] = -1/(pi n)^2, times 2, times an optional window, cut at Nyquist):  out[i] = sum_j in[j] taps[i - j + n_u - 1].
    The zero-padded circular convolution TIGRE evaluates by FFT only ever uses offsets |i - j| < n_u <= L/2, so the taps
    ``ifft(response)[offset mod L]`` give the identical linear operator.  float64 design, float32 result."""
    if name is None:
        name = "ram_lak"
    if name not in FILTERS:
        raise ValueError("unknown FDK filter %r (one of %s)" % (name, ", ".join(FILTERS)))
    L = filter_length(n_u, n_u if n_v is None else n_v)
    nn = np.arange(-L // 2, L // 2)
    h = np.zeros(L)
    h[L // 2] = 0.25
    odd = (nn % 2) == 1
    h[odd] = -1.0 / (np.pi * nn[odd]) ** 2
    resp = 2.0 * np.abs(np.fft.fft(h))[: L // 2 + 1]
    w = 2.0 * np.pi * np.arange(resp.shape[0]) / L
    if name == "shepp_logan":
        resp[1:] *= np.sin(w[1:] / 2.0) / (w[1:] / 2.0)
    elif name == "cosine":
        resp[1:] *= np.cos(w[1:] / 2.0)
    elif name == "hamming":
        resp[1:] *= 0.54 + 0.46 * np.cos(w[1:])
    elif name == "hann":
        resp[1:] *= 0.5 * (1.0 + np.cos(w[1:]))
    resp[w > np.pi] = 0.0
    full = np.concatenate([resp, resp[1:-1][::-1]])
    k = np.real(np.fft.ifft(full))
    return np.ascontiguousarray(k[np.arange(-(n_u - 1), n_u) % L], dtype=np.float32)


def fdk_filter(projs, du, dv, DSD, DSO, filter_name="ram_lak", cone=True):
    """projs [V,H,W] (GPU) -> pre-weighted, ramp-filtered projections, TRANSPOSED: [V,W,H]."""
    _require_gpu(projs, "projs")
    p = projs.to(_F32).contiguous()
    V, H, W = p.shape
    taps = torch.from_numpy(ramp_taps(W, H, filter_name)).to(p.device)
    out = torch.empty((V, W, H), dtype=_F32, device=p.device)
    scale = ((DSD / DSO) if cone else 1.0) * (2.0 * math.pi / V) / (4.0 * du)
    L = _lib.lib()
    with _on_device(p.device):
        rc = L.r2_fdk_filter(V, H, W, p.data_ptr(), taps.data_ptr(), float(scale), int(bool(cone)), float(DSD), float(du),
                             float(dv), out.data_ptr(), _stream(p.device))
    _lib.check(rc, "r2_fdk_filter")
    return out


def fdk_backproject(filtered_t, full_proj, DSO, nVoxel, sVoxel, center, cone=True):
    """filtered_t [V,W,H] from ``fdk_filter``; full_proj [V,4,4] (``full_proj_transform`` of the views) -> vol [nx,ny,nz]."""
    _require_gpu(filtered_t, "filtered_t")
    q = filtered_t.to(_F32).contiguous()
    V, W, H = q.shape
    M = full_proj.to(device=q.device, dtype=_F32).reshape(V, 16).contiguous()
    nx, ny, nz = (int(n) for n in nVoxel)
    vol = torch.empty((nx, ny, nz), dtype=_F32, device=q.device)
    L = _lib.lib()
    with _on_device(q.device):
        rc = L.r2_fdk_backproject(V, H, W, q.data_ptr(), M.data_ptr(), int(bool(cone)), float(DSO), nx, ny, nz,
                                  float(sVoxel[0]), float(sVoxel[1]), float(sVoxel[2]), float(center[0]), float(center[1]),
                                  float(center[2]), vol.data_ptr(), _stream(q.device))
    _lib.check(rc, "r2_fdk_backproject")
    return vol


def fdk(projs, angles, scanner_cfg, filter_name=None, device="cuda"):
    """FDK volume [nx,ny,nz] (GPU tensor) of projections [V,H,W] taken at ``angles`` with the scanner ``scanner_cfg``
    (the dictionary of ``Scene.scanner_cfg``: mode, DSD, DSO, nDetector, sDetector, nVoxel, sVoxel, offOrigin, filter)."""
    cfg = scanner_cfg
    p = torch.as_tensor(np.ascontiguousarray(projs) if isinstance(projs, np.ndarray) else projs).to(device=device, dtype=_F32)
    V, H, W = p.shape
    cone = cfg["mode"] == "cone"
    scale = 2.0 / max(cfg["sVoxel"])   # make_view works in the normalised scene (dataset_readers.py:62-76): 1 if cfg already is
    views = [S.make_view(float(a), (H, W), cfg) for a in np.asarray(angles).reshape(-1)]
    assert len(views) == V, "one angle per projection"
    M = torch.stack([v.full_proj_transform for v in views])
    if cone:
        sDet = [s * scale for s in cfg["sDetector"]]
        dv, du = sDet[0] / H, sDet[1] / W   # sDetector is [v, u] (dataset_readers.py:130)
    else:
        # parallel beams: the reference's orthographic camera maps view-space [-1, 1] onto the detector whatever sDetector
        # says (identity projection, graphics_utils.py:98-101), so that IS the pixel pitch the projections were taken with
        dv, du = 2.0 / H, 2.0 / W
    name = filter_name if filter_name is not None else cfg.get("filter")
    q = fdk_filter(p * scale, du, dv, cfg["DSD"] * scale, cfg["DSO"] * scale, name, cone)
    return fdk_backproject(q, M, cfg["DSO"] * scale, cfg["nVoxel"], [s * scale for s in cfg["sVoxel"]],
                           [o * scale for o in cfg["offOrigin"]], cone)


def recon_volume(projs, angles, scanner_cfg, recon_method="fdk"):
    """ct_utils.py:17-27 for ``recon_method="fdk"``: numpy volume [nx,ny,nz]."""
    if recon_method != "fdk":
        raise ValueError("Unsupported reconstruction method")
    return fdk(projs, angles, scanner_cfg).cpu().numpy()


def init_pcd(projs, angles, scanner_cfg, n_points=50000, density_thresh=0.05, density_rescale=0.15, recon_method="fdk",
             random_density_max=1.0, rng=np.random, save_path=None):
    """initialize_pcd.py:36-90: [n_points, 4] = positions | densities.  ``rng``: the numpy generator the draws come from
    (the reference seeds the global one with 0)."""
    if recon_method not in ("random", "fdk"):
        raise AssertionError("--recon_method not supported.")
    off, sVoxel = np.array(scanner_cfg["offOrigin"]), np.array(scanner_cfg["sVoxel"])
    if recon_method == "random":
        pos = off[None, ...] + sVoxel[None, ...] * (rng.rand(n_points, 3) - 0.5)
        dens = rng.rand(n_points) * random_density_max
    else:
        vol = recon_volume(projs, angles, scanner_cfg, recon_method)
        valid = np.argwhere(vol > density_thresh)
        dVoxel = sVoxel / np.array(scanner_cfg["nVoxel"])
        assert valid.shape[0] >= n_points, "Valid voxels less than target number of sampling. Check threshold"
        idx = valid[rng.choice(len(valid), n_points, replace=False)]
        pos = idx * dVoxel - sVoxel / 2 + off
        dens = vol[idx[:, 0], idx[:, 1], idx[:, 2]] * density_rescale
    out = np.concatenate([pos, dens[:, None]], axis=-1)
    if save_path is not None:
        np.save(save_path, out)
    return out
