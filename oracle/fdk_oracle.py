"""CPU restatement (numpy, float64) of the FDK reconstruction the reference uses to initialise its Gaussians.

TEST INFRASTRUCTURE ONLY (same rule as oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg may import it).

**Parity unpinned.**  The reference calls ``tigre.algorithms.fdk`` (r2_gaussian/utils/ct_utils.py:17-27, called from
initialize_pcd.py:63); TIGRE v2.3 (README.md:47-50) is a third-party dependency that is absent from /root/reference and
from this image, so no output of it can be produced here.  What is restated is the published algorithm (Feldkamp, Davis &
Kress 1984; Kak & Slaney ch. 3.6) with TIGRE's discretisation as documented in its Python sources
(``tigre/algorithms/single_pass_algorithms.py::FDK``, ``tigre/utilities/filtering.py::{filtering, ramp_flat, filter}``):

1. cosine pre-weight  w(u,v) = DSD / sqrt(DSD^2 + u^2 + v^2)          (u, v: detector coordinates of the pixel centres)
2. ramp filter along detector rows: zero-padded FFT of length L = max(64, 2^ceil(log2(2 * max(nDetector)))), response
   filt = 2 |FFT(h)|, h[0] = 1/4, h[n odd] = -1/(pi n)^2, optional window (shepp_logan / cosine / hamming / hann),
   times  (DSD/DSO) (2 pi / N) / (4 du)
3. voxel-driven back-projection with bilinear detector interpolation (zero outside the detector) and the distance weight
   (DSO / U)^2, U = distance of the voxel from the source along the central ray.

Geometry is NOT TIGRE's: voxels are projected with the reference's own camera matrices (dataset_readers.py:119-191,
graphics_utils.py:81-142 -- the matrices its rasterizer renders with), so the result lives in the [nx,ny,nz] layout of
``query()`` and no flip / transpose of ct_utils.py:20,26 is involved.  The anchor that replaces TIGRE's output is
physical: projections of a known density rendered by the X-ray rasterizer must reconstruct to that density's voxelisation
(tests/test_fdk_gpu.py).
"""
import numpy as np

FILTERS = ("ram_lak", "shepp_logan", "cosine", "hamming", "hann")


def filter_length(n_u, n_v):
    return int(max(64, 2 ** int(np.ceil(np.log2(2 * max(n_u, n_v))))))


def ramp_response(L, name="ram_lak", d=1.0):
    """Frequency response of length L (filtering.py::filter over ramp_flat)."""
    nn = np.arange(-L / 2, L / 2)
    h = np.zeros(L)
    h[L // 2] = 0.25
    odd = (nn % 2) == 1
    h[odd] = -1.0 / (np.pi * nn[odd]) ** 2
    f = np.abs(np.fft.fft(h)) * 2.0
    filt = f[: L // 2 + 1].copy()
    w = 2.0 * np.pi * np.arange(filt.shape[0]) / L
    if name in (None, "ram_lak"):
        pass
    elif name == "shepp_logan":
        filt[1:] *= np.sin(w[1:] / (2 * d)) / (w[1:] / (2 * d))
    elif name == "cosine":
        filt[1:] *= np.cos(w[1:] / (2 * d))
    elif name == "hamming":
        filt[1:] *= 0.54 + 0.46 * np.cos(w[1:] / d)
    elif name == "hann":
        filt[1:] *= (1.0 + np.cos(w[1:] / d)) / 2.0
    else:
        raise ValueError("unknown filter %r" % (name,))
    filt[w > np.pi * d] = 0.0
    return np.concatenate([filt, filt[1:-1][::-1]])


def preweight(n_v, n_u, dv, du, DSD, cone=True):
    if not cone:
        return np.ones((n_v, n_u))
    u = (np.arange(n_u) + 0.5 - n_u / 2.0) * du
    v = (np.arange(n_v) + 0.5 - n_v / 2.0) * dv
    vv, uu = np.meshgrid(v, u, indexing="ij")
    return DSD / np.sqrt(DSD * DSD + uu * uu + vv * vv)


def filter_scale(n_views, du, DSD, DSO, cone=True):
    return ((DSD / DSO) if cone else 1.0) * (2.0 * np.pi / n_views) / (4.0 * du)


def fdk_filter(projs, du, dv, DSD, DSO, name="ram_lak", cone=True):
    """projs [V,H,W] -> filtered [V,H,W] (float64), the FFT formulation."""
    p = np.asarray(projs, dtype=np.float64)
    V, H, W = p.shape
    L = filter_length(W, H)
    resp = ramp_response(L, name)
    pad = (L - W) // 2
    x = np.zeros((V, H, L))
    x[:, :, pad:pad + W] = p * preweight(H, W, dv, du, DSD, cone)[None]
    y = np.real(np.fft.ifft(np.fft.fft(x, axis=2) * resp[None, None, :], axis=2))
    return y[:, :, pad:pad + W] * filter_scale(V, du, DSD, DSO, cone)


def ndc2pix(v, S):
    return ((v + 1.0) * S - 1.0) * 0.5


def fdk_backproject(filtered, full_proj, DSO, nVoxel, sVoxel, center, cone=True, return_abs=False):
    """filtered [V,H,W]; full_proj [V,4,4] as the rasterizer takes it (row-vector convention: p_hom = [x,y,z,1] @ M).
    -> vol [nx,ny,nz] float64 (and the sum of |terms| per voxel when return_abs)."""
    q = np.asarray(filtered, dtype=np.float64)
    V, H, W = q.shape
    nx, ny, nz = [int(n) for n in nVoxel]
    d = [float(s) / n for s, n in zip(sVoxel, (nx, ny, nz))]
    ax = [center[k] - sVoxel[k] / 2.0 + (np.arange(n) + 0.5) * d[k] for k, n in enumerate((nx, ny, nz))]
    X, Y, Z = np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
    vol = np.zeros((nx, ny, nz))
    absum = np.zeros((nx, ny, nz))
    for v in range(V):
        M = np.asarray(full_proj[v], dtype=np.float64)
        hx = X * M[0, 0] + Y * M[1, 0] + Z * M[2, 0] + M[3, 0]
        hy = X * M[0, 1] + Y * M[1, 1] + Z * M[2, 1] + M[3, 1]
        hw = X * M[0, 3] + Y * M[1, 3] + Z * M[2, 3] + M[3, 3]
        inv = 1.0 / (hw + 1e-7)
        fx = ndc2pix(hx * inv, W)
        fy = ndc2pix(hy * inv, H)
        x0 = np.floor(fx)
        y0 = np.floor(fy)
        ax_, ay_ = fx - x0, fy - y0
        x0 = x0.astype(np.int64)
        y0 = y0.astype(np.int64)

        def tap(yy, xx):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(ok, q[v][np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)

        s = ((1 - ay_) * ((1 - ax_) * tap(y0, x0) + ax_ * tap(y0, x0 + 1))
             + ay_ * ((1 - ax_) * tap(y0 + 1, x0) + ax_ * tap(y0 + 1, x0 + 1)))
        wgt = (DSO * inv) ** 2 if cone else 1.0   # p_hom.w is the view-space depth U (projection matrix row 3 = [0,0,1,0])
        vol += s * wgt
        absum += np.abs(s * wgt)
    return (vol, absum) if return_abs else vol


def fdk(projs, full_proj, du, dv, DSD, DSO, nVoxel, sVoxel, center, name="ram_lak", cone=True):
    return fdk_backproject(fdk_filter(projs, du, dv, DSD, DSO, name, cone), full_proj, DSO, nVoxel, sVoxel, center, cone)


def spatial_taps(n_u, n_v, name="ram_lak"):
    """The same filter as 2 n_u - 1 spatial taps: out[i] = sum_j in[j] * taps[i - j + n_u - 1].  The padded circular
    convolution above only ever uses offsets |i - j| < n_u <= L/2, so this is the identical linear operator."""
    L = filter_length(n_u, n_v)
    k = np.real(np.fft.ifft(ramp_response(L, name)))
    off = np.arange(-(n_u - 1), n_u)
    return k[off % L]
