/*
 * r2_oracle.c -- CPU restatement of the R2-Gaussian hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the MI355X HIP kernels in r2_gaussian_amd/csrc/.
 * It is NOT part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path never falls back to it.
 *
 * It restates, function by function, the arithmetic of the reference CUDA submodule
 *   SUB = r2_gaussian/submodules/xray-gaussian-rasterization-voxelization   (Ruyi-Zha/r2_gaussian)
 * in scalar IEEE-754 binary32 with the reference's operation order (glm column-major
 * matrix products expanded left-to-right, the two double-precision spots kept in double).
 * Build with -ffp-contract=off: the HIP kernels are built the same way for every value
 * that feeds an integer decision (radius, tile rect, sort key), so that tile / sort
 * indices are bit-exact between the two.
 *
 * PARITY PINNING.  The reference ships no tests / golden vectors for this path, so
 * the oracle is pinned against THE REFERENCE ITSELF RUN HERE: oracle/_ref is the reference's own
 * CUDA sources (SUB/cuda_rasterizer/*.cu, SUB/cuda_voxelizer/*.cu) compiled for the host CPU from
 * where they lie under /root/reference (oracle/Makefile target `ref`; the shim boundary -- glm,
 * CUB and the CUDA execution model are restated, the kernels are not -- is described there).
 *   (1) tests/test_oracle_vs_ref.py compares this file with oracle/_ref live: every forward
 *       intermediate (radii, tile counts, keys, sorted lists, ranges, n_contrib, image / volume)
 *       is BIT-exact, backward sums agree to 1e-6 (float summation order);
 *   (2) tests/golden/*.npz are outputs of oracle/_ref and of the reference's Python (camera
 *       matrices, covariance, PSNR), committed with their generator tests/golden/make_golden.py,
 *       so the pinning travels to machines without the reference tree (tests/test_oracle_golden.py);
 *   (3) closed-form known-answer tests (tests/test_oracle_kat.py).
 * Residual caveat: nvcc contracts a*b+c into FMAs, g++/-ffp-contract=off does not, so "bit-exact"
 * is between uncontracted builds; a real CUDA run can differ in the last bit of a float that sits
 * exactly on a tile boundary.
 * simple-knn (distCUDA2) is an un-vendored third-party submodule (gitlab.inria.fr/bkerbl/simple-knn,
 * version unknown, source absent): for it parity is UNPINNED; the restatement here is the published
 * algorithm's exact result (mean of the 3 smallest squared distances, self excluded), checked
 * against a float64 brute force.
 *
 * Each function cites the reference file:line it follows (paths relative to SUB unless noted;
 * RAS = cuda_rasterizer, VOX = cuda_voxelizer).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <omp.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define R2O_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ glm restatement */
/* glm::mat3 is column-major: m[c][r].  glm::mat3(a,b,c, d,e,f, g,h,i) fills column 0 with
 * (a,b,c), column 1 with (d,e,f), column 2 with (g,h,i).  operator* computes
 * R[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2], summed left to right. */
typedef struct { float m[3][3]; } mat3;

static mat3 mat3_make(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    mat3 M;
    M.m[0][0] = a; M.m[0][1] = b; M.m[0][2] = c;
    M.m[1][0] = d; M.m[1][1] = e; M.m[1][2] = f;
    M.m[2][0] = g; M.m[2][1] = h; M.m[2][2] = i;
    return M;
}
static mat3 mat3_mul(const mat3 *A, const mat3 *B)
{
    mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A->m[0][r] * B->m[c][0] + A->m[1][r] * B->m[c][1] + A->m[2][r] * B->m[c][2];
    return R;
}
static mat3 mat3_T(const mat3 *A)
{
    mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A->m[r][c];
    return R;
}
static mat3 mat3_scale(float s, const mat3 *A)
{
    mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = s * A->m[c][r];
    return R;
}

/* ------------------------------------------------------------------ shared helpers */
/* RAS/auxiliary.h:62-81 transformPoint4x3 / transformPoint4x4 */
static void xf4x3(const float *p, const float *M, float *o)
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}
static void xf4x4(const float *p, const float *M, float *o)
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
    o[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}
/* RAS/auxiliary.h:45-48 ndc2Pix: evaluated in double, narrowed on return */
static float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static float fminf_(float a, float b) { return a < b ? a : b; } /* CUDA min/max on finite floats */
static float fmaxf_(float a, float b) { return a > b ? a : b; }

/* RAS/forward.cu:161-195 (== VOX/forward.cu:21-55) computeCov3D: Sigma = (S R)^T (S R),
 * quaternion NOT normalised in-kernel. */
static void cov3d_from_scale_rot(const float *scale, float mod, const float *rot, float *cov3D)
{
    mat3 S = mat3_make(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_make(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 M = mat3_mul(&S, &R);
    mat3 Mt = mat3_T(&M);
    mat3 Sigma = mat3_mul(&Mt, &M);
    cov3D[0] = Sigma.m[0][0];
    cov3D[1] = Sigma.m[0][1];
    cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1];
    cov3D[4] = Sigma.m[1][2];
    cov3D[5] = Sigma.m[2][2];
}
R2O_API void r2o_cov3d(int P, const float *scales, float mod, const float *rots, float *cov3D)
{
    for (int i = 0; i < P; i++)
        cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov3D + 6 * i);
}

/* RAS/backward.cu:334-397 (== VOX/backward.cu:21-84) computeCov3D backward. */
static void cov3d_bwd(const float *scale, float mod, const float *rot, const float *dL_dcov3D,
                      float *dL_dscale, float *dL_drot)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_make(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 S = mat3_make(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    mat3 M = mat3_mul(&S, &R);
    mat3 dSigma = mat3_make(
        dL_dcov3D[0], 0.5f * dL_dcov3D[1], 0.5f * dL_dcov3D[2],
        0.5f * dL_dcov3D[1], dL_dcov3D[3], 0.5f * dL_dcov3D[4],
        0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[4], dL_dcov3D[5]);
    /* glm: 2.0f * M * dL_dSigma  ==  (2.0f * M) * dL_dSigma */
    mat3 M2 = mat3_scale(2.0f, &M);
    mat3 dM = mat3_mul(&M2, &dSigma);
    mat3 Rt = mat3_T(&R);
    mat3 dMt = mat3_T(&dM);
    for (int k = 0; k < 3; k++) /* glm::dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z */
        dL_dscale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++)
            dMt.m[k][j] *= s[k];
    float q0 = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
    float q1 = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
    float q2 = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
    float q3 = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
    dL_drot[0] = q0; dL_drot[1] = q1; dL_drot[2] = q2; dL_drot[3] = q3; /* no normalisation Jacobian (:396) */
}

/* RAS/rasterizer_impl.cu:35-50 getHigherMsb */
R2O_API uint32_t r2o_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* cub::DeviceScan::InclusiveSum (RAS/rasterizer_impl.cu:275) */
R2O_API uint32_t r2o_inclusive_scan(int P, const uint32_t *in, uint32_t *out)
{
    uint32_t s = 0;
    for (int i = 0; i < P; i++) { s += in[i]; out[i] = s; }
    return s;
}

/* cub::DeviceRadixSort::SortPairs(begin_bit=0,end_bit) (RAS/rasterizer_impl.cu:301-306):
 * a STABLE sort on the low end_bit bits of the key.  LSD byte-radix, stable by construction. */
R2O_API void r2o_sort_pairs(int64_t R, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, int end_bit)
{
    if (R <= 0) return;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * R), *kb = (uint64_t *)malloc(sizeof(uint64_t) * R);
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * R), *vb = (uint32_t *)malloc(sizeof(uint32_t) * R);
    memcpy(ka, kin, sizeof(uint64_t) * R);
    memcpy(va, vin, sizeof(uint32_t) * R);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1;
        int64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < R; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < R; i++) {
            int64_t dst = cnt[(ka[i] >> shift) & mask]++;
            kb[dst] = ka[i]; vb[dst] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, sizeof(uint64_t) * R);
    memcpy(vout, va, sizeof(uint32_t) * R);
    free(ka); free(kb); free(va); free(vb);
}

/* RAS/rasterizer_impl.cu:116-138 identifyTileRanges (+ memset :308).  ranges = uint2[T]. */
R2O_API void r2o_tile_ranges(int64_t L, const uint64_t *keys, int64_t T, uint32_t *ranges)
{
    memset(ranges, 0, sizeof(uint32_t) * 2 * T);
    for (int64_t idx = 0; idx < L; idx++) {
        uint32_t cur = (uint32_t)(keys[idx] >> 32);
        if (idx == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)idx; ranges[2 * cur] = (uint32_t)idx; }
        }
        if (idx == L - 1) ranges[2 * cur + 1] = (uint32_t)L;
    }
}

/* ================================================================== RASTERIZER ====== */

/* RAS/auxiliary.h:50-60 getRect (BLOCK_X = BLOCK_Y = 16, RAS/config.h:16-17) */
static void get_rect(float px, float py, int max_radius, int gx, int gy, uint32_t *rmin, uint32_t *rmax)
{
    rmin[0] = (uint32_t)imin(gx, imax(0, (int)((px - max_radius) / 16)));
    rmin[1] = (uint32_t)imin(gy, imax(0, (int)((py - max_radius) / 16)));
    rmax[0] = (uint32_t)imin(gx, imax(0, (int)((px + max_radius + 16 - 1) / 16)));
    rmax[1] = (uint32_t)imin(gy, imax(0, (int)((py + max_radius + 16 - 1) / 16)));
}

/* Shared by forward (RAS/forward.cu:77-156) and backward (RAS/backward.cu:165-241):
 * t (clamped), J, W, M = W*J, cov = M^T Vrk^T M. */
typedef struct { float t[3]; mat3 J, W, M, cov; float x_grad_mul, y_grad_mul; } cov2d_ctx;

static void cov2d_common(const float *mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         const float *cov3D, const float *view, int mode, cov2d_ctx *c)
{
    float t[3];
    xf4x3(mean, view, t);
    if (mode == 0) {
        const float limx = 1.3f, limy = 1.3f;
        t[0] = fminf_(limx, fmaxf_(-limx, t[0]));
        t[1] = fminf_(limx, fmaxf_(-limx, t[1])); /* Q1: limx used for y */
        c->x_grad_mul = (t[0] < -limx || t[0] > limx) ? 0.f : 1.f; /* Q2: after clamping => always 1 */
        c->y_grad_mul = (t[1] < -limy || t[1] > limy) ? 0.f : 1.f;
        c->J = mat3_make(focal_x, 0.0f, 0.0f, 0.0f, focal_y, 0.0f, 0.0f, 0.0f, 1.0f);
    } else {
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float txtz = t[0] / t[2], tytz = t[1] / t[2];
        t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
        t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
        c->x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        c->y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float l = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        c->J = mat3_make(
            focal_x / t[2], 0.0f, -(focal_x * t[0]) / (t[2] * t[2]),
            0.0f, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
            t[0] / l, t[1] / l, t[2] / l);
    }
    c->t[0] = t[0]; c->t[1] = t[1]; c->t[2] = t[2];
    c->W = mat3_make(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    c->M = mat3_mul(&c->W, &c->J);
    mat3 Vrk = mat3_make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 Mt = mat3_T(&c->M), Vt = mat3_T(&Vrk);
    mat3 tmp = mat3_mul(&Mt, &Vt);
    c->cov = mat3_mul(&tmp, &c->M);
}

/* mu = sqrt(2*pi*circ/diamond) with the double-precision 2*M_PI of RAS/forward.cu:147-153 */
static float mu_from(float circ, float diamond)
{
    float mu_square = (float)(2 * M_PI * (double)circ / (double)diamond);
    float mu = 0.0f;
    if (mu_square > 0.0f) mu = (float)sqrt(2 * M_PI * (double)circ / (double)diamond);
    return mu;
}

/* RAS/forward.cu:198-289 preprocessCUDA (forward).  Outputs not written by the reference for a
 * rejected Gaussian are left untouched here too (callers zero-initialise them). */
R2O_API void r2o_raster_preprocess(
    int P, const float *means3D, const float *scales, float scale_modifier, const float *rotations,
    const float *opacities, const float *cov3D_precomp, const float *view, const float *proj,
    int W, int H, float tan_fovx, float tan_fovy, int mode,
    int32_t *radii, float *means2D, float *depths, float *cov3Ds, float *conic_opacity, float *mus,
    uint32_t *tiles_touched)
{
    /* RAS/rasterizer_impl.cu:219-220 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float *p_orig = means3D + 3 * idx;
        /* in_frustum, RAS/auxiliary.h:143-168 */
        float p_view[3];
        xf4x3(p_orig, view, p_view);
        if (p_view[2] <= 0.2f) continue;
        float p_hom[4];
        xf4x4(p_orig, proj, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = { p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w };
        const float *cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * idx;
        else {
            cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx);
            cov3D = cov3Ds + 6 * idx; /* Q12: written even if rejected below */
        }
        cov2d_ctx c;
        cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, mode, &c);
        float hata = c.cov.m[0][0], hatb = c.cov.m[0][1], hatc = c.cov.m[0][2];
        float hatd = c.cov.m[1][1], hate = c.cov.m[1][2], hatf = c.cov.m[2][2];
        float diamond = hata * hatd - hatb * hatb;
        float circ = hata * hatd * hatf + 2 * hatb * hatc * hate - hata * hate * hate - hatf * hatb * hatb - hatd * hatc * hatc;
        float mu = mu_from(circ, diamond);
        float covx = hata, covy = hatb, covz = hatd;
        float det = (covx * covz - covy * covy);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = { covz * det_inv, -covy * det_inv, covx * det_inv };
        float mid = 0.5f * (covx + covz);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        float px = ndc2pix(p_proj[0], W), py = ndc2pix(p_proj[1], H);
        uint32_t rmin[2], rmax[2];
        get_rect(px, py, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        depths[idx] = p_view[2];
        radii[idx] = (int32_t)my_radius;
        means2D[2 * idx] = px; means2D[2 * idx + 1] = py;
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
        mus[idx] = mu;
    }
}

/* RAS/rasterizer_impl.cu:54-66 checkFrustum / markVisible */
R2O_API void r2o_mark_visible(int P, const float *means3D, const float *view, const float *proj, uint8_t *present)
{
    (void)proj;
    for (int i = 0; i < P; i++) {
        float pv[3];
        xf4x3(means3D + 3 * i, view, pv);
        present[i] = pv[2] <= 0.2f ? 0 : 1;
    }
}

/* RAS/rasterizer_impl.cu:70-111 duplicateWithKeys */
R2O_API void r2o_raster_duplicate(int P, const float *means2D, const float *depths, const uint32_t *offsets,
                                  const int32_t *radii, int gx, int gy, uint64_t *keys, uint32_t *vals)
{
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (int y = (int)rmin[1]; y < (int)rmax[1]; y++)
                for (int x = (int)rmin[0]; x < (int)rmax[0]; x++) {
                    uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    keys[off] = key;
                    vals[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
}

/* RAS/forward.cu:294-395 renderCUDA (forward).  One "thread" per pixel, list walked in sorted order. */
R2O_API void r2o_raster_render_fwd(const uint32_t *ranges, const uint32_t *point_list, int W, int H,
                                   const float *means2D, const float *conic_opacity, const float *mus,
                                   uint32_t *n_contrib, float *out_color)
{
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < 16; ly++)
            for (int lx = 0; lx < 16; lx++) {
                const int pxi = tx * 16 + lx, pyi = ty * 16 + ly;
                if (!(pxi < W && pyi < H)) continue;
                const float pixx = (float)pxi, pixy = (float)pyi;
                uint32_t contributor = 0, last_contributor = 0;
                float C = 0.f;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t id = point_list[k];
                    const float dx = means2D[2 * id] - pixx, dy = means2D[2 * id + 1] - pixy;
                    const float *co = conic_opacity + 4 * id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = co[3] * mus[id] * expf(power);
                    if (alpha < 0.00001f) continue;
                    C += alpha;
                    last_contributor = contributor;
                }
                n_contrib[pyi * W + pxi] = last_contributor;
                out_color[pyi * W + pxi] = C;
            }
    }
}

/* RAS/backward.cu:447-575 renderCUDA (backward).  The reference accumulates with float atomicAdd in an
 * unspecified order (Q11); here terms are accumulated tile by tile, pixel by pixel, back to front.
 * acc64 != 0 accumulates in double (a tighter "truth" for tolerance tests); 0 mimics float atomics.
 * dL_dmean2D is [P,3] (z untouched), dL_dconic is [P,4] (slots 0,1,3 used). */
R2O_API void r2o_raster_render_bwd(const uint32_t *ranges, const uint32_t *point_list, int W, int H, int P,
                                   const float *means2D, const float *conic_opacity, const float *mus,
                                   const uint32_t *n_contrib, const float *dL_dpixels,
                                   float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dmu, int acc64)
{
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    /* one accumulator slab per thread, reduced in thread order: deterministic for a fixed thread count */
    const int nthr = omp_get_max_threads();
    double *acc = acc64 ? (double *)calloc((size_t)P * 7 * nthr, sizeof(double)) : NULL;
    float *accf = acc64 ? NULL : (float *)calloc((size_t)P * 7 * nthr, sizeof(float));
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
#pragma omp parallel for schedule(static)
    for (int tile = 0; tile < gx * gy; tile++) {
        const size_t slab = (size_t)omp_get_thread_num() * P * 7;
        const int ty = tile / gx, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < 16; ly++)
            for (int lx = 0; lx < 16; lx++) {
                const int pxi = tx * 16 + lx, pyi = ty * 16 + ly;
                if (!(pxi < W && pyi < H)) continue;
                const float pixx = (float)pxi, pixy = (float)pyi;
                const uint32_t last_contributor = n_contrib[pyi * W + pxi];
                const float dL_dpixel = dL_dpixels[pyi * W + pxi];
                uint32_t contributor = r1 - r0;
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t id = point_list[k];
                    const float dx = means2D[2 * id] - pixx, dy = means2D[2 * id + 1] - pixy;
                    const float *co = conic_opacity + 4 * id;
                    const float mu = mus[id];
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = co[3] * mu * G;
                    if (alpha < 0.00001f) continue;
                    float dL_dalpha = 0.0f;
                    dL_dalpha += 1.f * dL_dpixel;
                    const float dL_dG = co[3] * mu * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    const float v[7] = {
                        dL_dG * dG_ddelx * ddelx_dx, dL_dG * dG_ddely * ddely_dy,
                        -0.5f * gdx * dx * dL_dG, -1.0f * gdx * dy * dL_dG, -0.5f * gdy * dy * dL_dG,
                        mu * G * dL_dalpha, co[3] * G * dL_dalpha };
                    if (acc64) for (int q = 0; q < 7; q++) acc[slab + 7 * (size_t)id + q] += (double)v[q];
                    else for (int q = 0; q < 7; q++) accf[slab + 7 * (size_t)id + q] += v[q];
                }
            }
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float v[7];
        for (int q = 0; q < 7; q++) {
            if (acc64) {
                double sum = 0.0;
                for (int t = 0; t < nthr; t++) sum += acc[(size_t)t * P * 7 + 7 * (size_t)i + q];
                v[q] = (float)sum;
            } else {
                float sum = 0.f;
                for (int t = 0; t < nthr; t++) sum += accf[(size_t)t * P * 7 + 7 * (size_t)i + q];
                v[q] = sum;
            }
        }
        dL_dmean2D[3 * i + 0] += v[0]; dL_dmean2D[3 * i + 1] += v[1];
        dL_dconic[4 * i + 0] += v[2]; dL_dconic[4 * i + 1] += v[3]; dL_dconic[4 * i + 3] += v[4];
        dL_dopacity[i] += v[5]; dL_dmu[i] += v[6];
    }
    free(acc); free(accf);
}

/* RAS/backward.cu:145-330 computeCov2DCUDA.  dL_dconics is [P,4] (0,1,3 read), dL_dmeans [P,3]
 * (ASSIGNED in mode 1), dL_dcov [P,6] (+= or zeroed, Q8). */
R2O_API void r2o_raster_cov2d_bwd(int P, const float *means, const int32_t *radii, const float *cov3Ds,
                                  int W, int H, float tan_fovx, float tan_fovy, const float *view,
                                  const float *dL_dconics, const float *dL_dmus, float *dL_dmeans, float *dL_dcov, int mode)
{
    const float h_y = H / (2.0f * tan_fovy), h_x = W / (2.0f * tan_fovx);
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float *cov3D = cov3Ds + 6 * idx;
        const float gx_ = dL_dconics[4 * idx], gy_ = dL_dconics[4 * idx + 1], gz_ = dL_dconics[4 * idx + 3];
        const float dL_dmu = dL_dmus[idx];
        cov2d_ctx c;
        cov2d_common(means + 3 * idx, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, mode, &c);
        const mat3 *M = &c.M, *Wm = &c.W;
        float hata = c.cov.m[0][0], hatb = c.cov.m[0][1], hatc = c.cov.m[0][2];
        float hatd = c.cov.m[1][1], hate = c.cov.m[1][2], hatf = c.cov.m[2][2];
        float dL_dhata = 0, dL_dhatb = 0, dL_dhatc = 0, dL_dhatd = 0, dL_dhate = 0, dL_dhatf = 0;
        float denom = hata * hatd - hatb * hatb;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float diamond = hata * hatd - hatb * hatb;
        float circ = hata * hatd * hatf + 2 * hatb * hatc * hate - hata * hate * hate - hatf * hatb * hatb - hatd * hatc * hatc;
        float mu = mu_from(circ, diamond);
        float pi_mu = (float)(M_PI / (double)(mu + 0.0000001f));
        float circ_diamond = circ / diamond;
        float *o = dL_dcov + 6 * idx;
        if (denom2inv != 0.0f && mu != 0.0f) {
            dL_dhata = denom2inv * (-hatd * hatd * gx_ + hatb * hatd * gy_ + (denom - hata * hatd) * gz_);
            dL_dhatd = denom2inv * (-hata * hata * gz_ + hata * hatb * gy_ + (denom - hata * hatd) * gx_);
            dL_dhatb = denom2inv * (2 * hatb * hatd * gx_ - (denom + 2 * hatb * hatb) * gy_ + 2 * hata * hatb * gz_);

            dL_dhata += pi_mu * ((hatd * hatf - hate * hate) / diamond - hatd * circ_diamond / diamond) * dL_dmu;
            dL_dhatb += pi_mu * ((2 * hatc * hate - 2 * hatf * hatb) / diamond + 2 * hatb * circ_diamond / diamond) * dL_dmu;
            dL_dhatc += pi_mu * ((2 * hatb * hate - 2 * hatd * hatc) / diamond) * dL_dmu;
            dL_dhatd += pi_mu * ((hata * hatf - hatc * hatc) / diamond - hata * circ_diamond / diamond) * dL_dmu;
            dL_dhate += pi_mu * ((2 * hatb * hatc - 2 * hata * hate) / diamond) * dL_dmu;
            dL_dhatf += pi_mu * ((hata * hatd - hatb * hatb) / diamond) * dL_dmu;
#define MM(c_, r_) (M->m[c_][r_])
            o[0] += MM(0,0)*MM(0,0)*dL_dhata + MM(0,0)*MM(1,0)*dL_dhatb + MM(0,0)*MM(2,0)*dL_dhatc + MM(1,0)*MM(1,0)*dL_dhatd + MM(1,0)*MM(2,0)*dL_dhate + MM(2,0)*MM(2,0)*dL_dhatf;
            o[3] += MM(0,1)*MM(0,1)*dL_dhata + MM(0,1)*MM(1,1)*dL_dhatb + MM(0,1)*MM(2,1)*dL_dhatc + MM(1,1)*MM(1,1)*dL_dhatd + MM(1,1)*MM(2,1)*dL_dhate + MM(2,1)*MM(2,1)*dL_dhatf;
            o[5] += MM(0,2)*MM(0,2)*dL_dhata + MM(0,2)*MM(1,2)*dL_dhatb + MM(0,2)*MM(2,2)*dL_dhatc + MM(1,2)*MM(1,2)*dL_dhatd + MM(1,2)*MM(2,2)*dL_dhate + MM(2,2)*MM(2,2)*dL_dhatf;
            o[1] += 2*MM(0,0)*MM(0,1)*dL_dhata + (MM(0,1)*MM(1,0)+MM(0,0)*MM(1,1))*dL_dhatb + (MM(0,1)*MM(2,0)+MM(0,0)*MM(2,1))*dL_dhatc + 2*MM(1,0)*MM(1,1)*dL_dhatd + (MM(1,1)*MM(2,0)+MM(1,0)*MM(2,1))*dL_dhate + 2*MM(2,0)*MM(2,1)*dL_dhatf;
            o[2] += 2*MM(0,0)*MM(0,2)*dL_dhata + (MM(0,2)*MM(1,0)+MM(0,0)*MM(1,2))*dL_dhatb + (MM(0,2)*MM(2,0)+MM(0,0)*MM(2,2))*dL_dhatc + 2*MM(1,0)*MM(1,2)*dL_dhatd + (MM(1,2)*MM(2,0)+MM(1,0)*MM(2,2))*dL_dhate + 2*MM(2,0)*MM(2,2)*dL_dhatf;
            o[4] += 2*MM(0,1)*MM(0,2)*dL_dhata + (MM(0,2)*MM(1,1)+MM(0,1)*MM(1,2))*dL_dhatb + (MM(0,2)*MM(2,1)+MM(0,1)*MM(2,2))*dL_dhatc + 2*MM(1,1)*MM(1,2)*dL_dhatd + (MM(1,2)*MM(2,1)+MM(1,1)*MM(2,2))*dL_dhate + 2*MM(2,1)*MM(2,2)*dL_dhatf;
        } else {
            for (int i = 0; i < 6; i++) o[i] = 0;
        }
        if (mode == 1) {
            float a = cov3D[0], b = cov3D[1], cc = cov3D[2], d = cov3D[3], e = cov3D[4], f = cov3D[5];
            float dL_dM00 = 2*(MM(0,0)*a+MM(0,1)*b + MM(0,2)*cc)*dL_dhata + (MM(1,0)*a+MM(1,1)*b+MM(1,2)*cc)*dL_dhatb + (MM(2,0)*a+MM(2,1)*b+MM(2,2)*cc)*dL_dhatc;
            float dL_dM01 = 2*(MM(0,0)*b+MM(0,1)*d + MM(0,2)*e)*dL_dhata + (MM(1,0)*b+MM(1,1)*d+MM(1,2)*e)*dL_dhatb + (MM(2,0)*b+MM(2,1)*d+MM(2,2)*e)*dL_dhatc;
            float dL_dM02 = 2*(MM(0,0)*cc+MM(0,1)*e + MM(0,2)*f)*dL_dhata + (MM(1,0)*cc+MM(1,1)*e+MM(1,2)*f)*dL_dhatb + (MM(2,0)*cc+MM(2,1)*e+MM(2,2)*f)*dL_dhatc;
            float dL_dM10 = (MM(0,0)*a+MM(0,1)*b+MM(0,2)*cc)*dL_dhatb + 2*(MM(1,0)*a+MM(1,1)*b+MM(1,2)*cc)*dL_dhatd + (MM(2,0)*a+MM(2,1)*b+MM(2,2)*cc)*dL_dhate;
            float dL_dM11 = (MM(0,0)*b+MM(0,1)*d+MM(0,2)*e)*dL_dhatb + 2*(MM(1,0)*b+MM(1,1)*d+MM(1,2)*e)*dL_dhatd + (MM(2,0)*b+MM(2,1)*d+MM(2,2)*e)*dL_dhate;
            float dL_dM12 = (MM(0,0)*cc+MM(0,1)*e+MM(0,2)*f)*dL_dhatb + 2*(MM(1,0)*cc+MM(1,1)*e+MM(1,2)*f)*dL_dhatd + (MM(2,0)*cc+MM(2,1)*e+MM(2,2)*f)*dL_dhate;
            float dL_dM20 = (MM(0,0)*a+MM(0,1)*b+MM(0,2)*cc)*dL_dhatc + (MM(1,0)*a+MM(1,1)*b+MM(1,2)*cc)*dL_dhate + 2*(MM(2,0)*a+MM(2,1)*b+MM(2,2)*cc)*dL_dhatf;
            float dL_dM21 = (MM(0,0)*b+MM(0,1)*d+MM(0,2)*e)*dL_dhatc + (MM(1,0)*b+MM(1,1)*d+MM(1,2)*e)*dL_dhate + 2*(MM(2,0)*b+MM(2,1)*d+MM(2,2)*e)*dL_dhatf;
            float dL_dM22 = (MM(0,0)*cc+MM(0,1)*e+MM(0,2)*f)*dL_dhatc + (MM(1,0)*cc+MM(1,1)*e+MM(1,2)*f)*dL_dhate + 2*(MM(2,0)*cc+MM(2,1)*e+MM(2,2)*f)*dL_dhatf;
#define WW(c_, r_) (Wm->m[c_][r_])
            float dL_dJ00 = WW(0,0)*dL_dM00 + WW(0,1)*dL_dM01 + WW(0,2)*dL_dM02;
            float dL_dJ02 = WW(2,0)*dL_dM00 + WW(2,1)*dL_dM01 + WW(2,2)*dL_dM02;
            float dL_dJ11 = WW(1,0)*dL_dM10 + WW(1,1)*dL_dM11 + WW(1,2)*dL_dM12;
            float dL_dJ12 = WW(2,0)*dL_dM10 + WW(2,1)*dL_dM11 + WW(2,2)*dL_dM12;
            float dL_dJ20 = WW(0,0)*dL_dM20 + WW(0,1)*dL_dM21 + WW(0,2)*dL_dM22;
            float dL_dJ21 = WW(1,0)*dL_dM20 + WW(1,1)*dL_dM21 + WW(1,2)*dL_dM22;
            float dL_dJ22 = WW(2,0)*dL_dM20 + WW(2,1)*dL_dM21 + WW(2,2)*dL_dM22;
#undef WW
            float tx = c.t[0], ty = c.t[1], tz = c.t[2];
            float inv_tz = 1.f / tz;
            float inv_tz2 = inv_tz * inv_tz;
            float inv_tz3 = inv_tz2 * inv_tz;
            float circledcirc = sqrtf(tx * tx + ty * ty + tz * tz);
            float inv_circledcirc3 = 1 / (circledcirc * circledcirc * circledcirc);
            float dL_dtx = c.x_grad_mul * (-h_x*inv_tz2*dL_dJ02 + (1/circledcirc - tx*tx*inv_circledcirc3)*dL_dJ20 - tx*ty*inv_circledcirc3*dL_dJ21 - tx*tz*inv_circledcirc3*dL_dJ22);
            float dL_dty = c.y_grad_mul * (-h_y*inv_tz2*dL_dJ12 - tx*ty*inv_circledcirc3*dL_dJ20 + (1/circledcirc - ty*ty*inv_circledcirc3)*dL_dJ21 - ty*tz*inv_circledcirc3*dL_dJ22);
            float dL_dtz = -h_x*inv_tz2*dL_dJ00 + 2*h_x*tx*inv_tz3*dL_dJ02 - h_y*inv_tz2*dL_dJ11 + 2*h_y*ty*inv_tz3*dL_dJ12 - tx*tz*inv_circledcirc3*dL_dJ20 - ty*tz*inv_circledcirc3*dL_dJ21 + (1/circledcirc-tz*tz*inv_circledcirc3)*dL_dJ22;
            /* transformVec4x3Transpose, RAS/auxiliary.h:93-101 */
            dL_dmeans[3 * idx + 0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
            dL_dmeans[3 * idx + 1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
            dL_dmeans[3 * idx + 2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
        }
#undef MM
    }
}

/* RAS/backward.cu:402-444 preprocessCUDA (backward).  dL_dmean2D is [P,3]. */
R2O_API void r2o_raster_preprocess_bwd(int P, const float *means, const int32_t *radii, const float *scales,
                                       const float *rotations, float scale_modifier, const float *proj,
                                       const float *dL_dmean2D, float *dL_dmeans, const float *dL_dcov3D,
                                       float *dL_dscale, float *dL_drot)
{
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float *m = means + 3 * idx;
        float m_hom[4];
        xf4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float g0 = dL_dmean2D[3 * idx], g1 = dL_dmean2D[3 * idx + 1];
        float dx = (proj[0] * m_w - proj[3] * mul1) * g0 + (proj[1] * m_w - proj[3] * mul2) * g1;
        float dy = (proj[4] * m_w - proj[7] * mul1) * g0 + (proj[5] * m_w - proj[7] * mul2) * g1;
        float dz = (proj[8] * m_w - proj[11] * mul1) * g0 + (proj[9] * m_w - proj[11] * mul2) * g1;
        dL_dmeans[3 * idx + 0] += dx;
        dL_dmeans[3 * idx + 1] += dy;
        dL_dmeans[3 * idx + 2] += dz;
        if (scales)
            cov3d_bwd(scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D + 6 * idx,
                      dL_dscale + 3 * idx, dL_drot + 4 * idx);
    }
}

/* ================================================================== VOXELIZER ======= */

/* VOX/auxiliary.h:27-39 getCube (8x8x8 tiles, VOX/config.h:16-18); radii are floats here */
static void get_cube(const float *p, const float *rad, const int *g, uint32_t *cmin, uint32_t *cmax)
{
    for (int k = 0; k < 3; k++) {
        cmin[k] = (uint32_t)imin(g[k], imax(0, (int)((p[k] - rad[k]) / 8)));
        cmax[k] = (uint32_t)imin(g[k], imax(0, (int)((p[k] + rad[k] + 8 - 1) / 8)));
    }
}

static void vox_cov(const float *cov3D, float dvx, float dvy, float dvz, mat3 *Mout, float *h)
{
    mat3 Vrk = mat3_make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 M = mat3_make(1.f / dvx, 0.0f, 0.0f, 0.0f, 1.f / dvy, 0.0f, 0.0f, 0.0f, 1.f / dvz);
    mat3 Mt = mat3_T(&M), Vt = mat3_T(&Vrk);
    mat3 tmp = mat3_mul(&Mt, &Vt);
    mat3 cov = mat3_mul(&tmp, &M);
    h[0] = cov.m[0][0]; h[1] = cov.m[0][1]; h[2] = cov.m[0][2];
    h[3] = cov.m[1][1]; h[4] = cov.m[1][2]; h[5] = cov.m[2][2];
    if (Mout) *Mout = M;
}

/* VOX/forward.cu:58-178 preprocessCUDA.  conic_opacity is [P,7]; means3D_norm [P,3]. */
R2O_API void r2o_voxel_preprocess(
    int P, const float *means3D, const float *scales, float scale_modifier, const float *rotations,
    const float *opacities, const float *cov3D_precomp,
    int nx, int ny, int nz, float sx, float sy, float sz, float cx, float cy, float cz,
    int32_t *radii_x, int32_t *radii_y, int32_t *radii_z, float *points_vol, float *depths,
    float *cov3Ds, float *conic_opacity, uint32_t *tiles_touched)
{
    const int g[3] = { (nx + 7) / 8, (ny + 7) / 8, (nz + 7) / 8 };
    for (int idx = 0; idx < P; idx++) {
        radii_x[idx] = 0; radii_y[idx] = 0; radii_z[idx] = 0;
        tiles_touched[idx] = 0;
        float dvx = sx / (float)nx, dvy = sy / (float)ny, dvz = sz / (float)nz;
        const float *p = means3D + 3 * idx;
        const float *cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * idx;
        else {
            cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx);
            cov3D = cov3Ds + 6 * idx;
        }
        float h[6];
        vox_cov(cov3D, dvx, dvy, dvz, NULL, h);
        float hata = h[0], hatb = h[1], hatc = h[2], hatd = h[3], hate = h[4], hatf = h[5];
        float det = hata * hatd * hatf + 2 * hatb * hatc * hate - hata * hate * hate - hatf * hatb * hatb - hatd * hatc * hatc;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float inv_a = (hatd * hatf - hate * hate) * det_inv;
        float inv_b = (hatc * hate - hatb * hatf) * det_inv;
        float inv_c = (hatb * hate - hatc * hatd) * det_inv;
        float inv_d = (hata * hatf - hatc * hatc) * det_inv;
        float inv_e = (hatb * hatc - hata * hate) * det_inv;
        float inv_f = (hata * hatd - hatb * hatb) * det_inv;
        const float *scale = scales + 3 * idx; /* Q5: raw scales, no modifier */
        float max_scale = fmaxf_(fmaxf_(scale[0], scale[1]), scale[2]);
        float rad[3] = { ceilf((3.f * max_scale) / dvx), ceilf((3.f * max_scale) / dvy), ceilf((3.f * max_scale) / dvz) };
        float pv[3] = { (p[0] - cx + sx / 2) / dvx, (p[1] - cy + sy / 2) / dvy, (p[2] - cz + sz / 2) / dvz };
        if (pv[0] + rad[0] < 0 || pv[1] + rad[1] < 0 || pv[2] + rad[2] < 0 ||
            pv[0] - rad[0] > (float)nx || pv[1] - rad[1] > (float)ny || pv[2] - rad[2] > (float)nz)
            continue;
        uint32_t cmin[3], cmax[3];
        get_cube(pv, rad, g, cmin, cmax);
        if ((cmax[0] - cmin[0]) * (cmax[1] - cmin[1]) * (cmax[2] - cmin[2]) == 0) continue;
        radii_x[idx] = (int32_t)rad[0]; radii_y[idx] = (int32_t)rad[1]; radii_z[idx] = (int32_t)rad[2];
        tiles_touched[idx] = (cmax[2] - cmin[2]) * (cmax[1] - cmin[1]) * (cmax[0] - cmin[0]);
        depths[idx] = p[2];
        points_vol[3 * idx] = pv[0]; points_vol[3 * idx + 1] = pv[1]; points_vol[3 * idx + 2] = pv[2];
        float *co = conic_opacity + 7 * idx;
        co[0] = inv_a; co[1] = inv_b; co[2] = inv_c; co[3] = inv_d; co[4] = inv_e; co[5] = inv_f;
        co[6] = opacities[idx];
    }
}

/* VOX/voxelizer_impl.cu:54-101 duplicateWithKeys */
R2O_API void r2o_voxel_duplicate(int P, const float *points_vol, const float *depths, const uint32_t *offsets,
                                 const int32_t *radii_x, const int32_t *radii_y, const int32_t *radii_z,
                                 int gx, int gy, int gz, uint64_t *keys, uint32_t *vals)
{
    const int g[3] = { gx, gy, gz };
    for (int idx = 0; idx < P; idx++) {
        if (radii_x[idx] > 0 && radii_y[idx] > 0 && radii_z[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
            float rad[3] = { (float)radii_x[idx], (float)radii_y[idx], (float)radii_z[idx] };
            uint32_t cmin[3], cmax[3];
            get_cube(points_vol + 3 * idx, rad, g, cmin, cmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (int z = (int)cmin[2]; z < (int)cmax[2]; z++)
                for (int y = (int)cmin[1]; y < (int)cmax[1]; y++)
                    for (int x = (int)cmin[0]; x < (int)cmax[0]; x++) {
                        uint64_t key = (uint64_t)(uint32_t)(z * gy * gx + y * gx + x);
                        key <<= 32;
                        key |= dbits;
                        keys[off] = key;
                        vals[off] = (uint32_t)idx;
                        off++;
                    }
        }
    }
}

/* the mixed double/float "power" of VOX/forward.cu:274 (Q6) */
static float vox_power(const float *co, float dx, float dy, float dz)
{
    return (float)(-0.5 * (double)(co[0] * dx * dx + co[3] * dy * dy + co[5] * dz * dz)
                   - (double)(co[1] * dx * dy) - (double)(co[2] * dx * dz) - (double)(co[4] * dy * dz));
}

/* VOX/forward.cu:183-315 renderCUDA (forward).  out index = x*ny*nz + y*nz + z. */
R2O_API void r2o_voxel_render_fwd(const uint32_t *ranges, const uint32_t *point_list, int nx, int ny, int nz,
                                  const float *points_vol, const float *conic_opacity,
                                  uint32_t *n_contrib, float *out_volume)
{
    const int gx = (nx + 7) / 8, gy = (ny + 7) / 8, gz = (nz + 7) / 8;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy * gz; tile++) {
        const int tz = tile / (gy * gx), ty = (tile / gx) % gy, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int lz = 0; lz < 8; lz++)
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++) {
                    const int vx = tx * 8 + lx, vy = ty * 8 + ly, vz = tz * 8 + lz;
                    if (!(vx < nx && vy < ny && vz < nz)) continue;
                    const size_t vid = (size_t)nz * ny * vx + (size_t)nz * vy + vz;
                    const float fx = (float)vx + 0.5f, fy = (float)vy + 0.5f, fz = (float)vz + 0.5f;
                    uint32_t contributor = 0, last_contributor = 0;
                    float C = 0.f;
                    for (uint32_t k = r0; k < r1; k++) {
                        contributor++;
                        const uint32_t id = point_list[k];
                        const float dx = points_vol[3 * id] - fx, dy = points_vol[3 * id + 1] - fy, dz = points_vol[3 * id + 2] - fz;
                        const float *co = conic_opacity + 7 * id;
                        const float power = vox_power(co, dx, dy, dz);
                        if (power > 0.0f) continue;
                        const float alpha = co[6] * expf(power);
                        if (alpha < 0.000001f) continue;
                        C += alpha;
                        last_contributor = contributor;
                    }
                    n_contrib[vid] = last_contributor;
                    out_volume[vid] = C;
                }
    }
}

/* VOX/backward.cu:216-374 renderCUDA (backward).  dL_dmean3D_norm [P,3], dL_dconic3D [P,6]. */
R2O_API void r2o_voxel_render_bwd(const uint32_t *ranges, const uint32_t *point_list, int nx, int ny, int nz,
                                  float sx, float sy, float sz, int P,
                                  const float *points_vol, const float *conic_opacity, const uint32_t *n_contrib,
                                  const float *dL_dpixels, float *dL_dmean3D_norm, float *dL_dconic3D,
                                  float *dL_dopacity, int acc64)
{
    const int gx = (nx + 7) / 8, gy = (ny + 7) / 8, gz = (nz + 7) / 8;
    const float dvx = sx / (float)nx, dvy = sy / (float)ny, dvz = sz / (float)nz;
    const int nthr = omp_get_max_threads();
    double *acc = acc64 ? (double *)calloc((size_t)P * 10 * nthr, sizeof(double)) : NULL;
    float *accf = acc64 ? NULL : (float *)calloc((size_t)P * 10 * nthr, sizeof(float));
#pragma omp parallel for schedule(static)
    for (int tile = 0; tile < gx * gy * gz; tile++) {
        const size_t slab = (size_t)omp_get_thread_num() * P * 10;
        const int tz = tile / (gy * gx), ty = (tile / gx) % gy, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int lz = 0; lz < 8; lz++)
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++) {
                    const int vx = tx * 8 + lx, vy = ty * 8 + ly, vz = tz * 8 + lz;
                    if (!(vx < nx && vy < ny && vz < nz)) continue;
                    const size_t vid = (size_t)nz * ny * vx + (size_t)nz * vy + vz;
                    const float fx = (float)vx + 0.5f, fy = (float)vy + 0.5f, fz = (float)vz + 0.5f;
                    const uint32_t last_contributor = n_contrib[vid];
                    const float dL_dpixel = dL_dpixels[vid];
                    uint32_t contributor = r1 - r0;
                    for (uint32_t k = r1; k-- > r0;) {
                        contributor--;
                        if (contributor >= last_contributor) continue;
                        const uint32_t id = point_list[k];
                        const float dx = points_vol[3 * id] - fx, dy = points_vol[3 * id + 1] - fy, dz = points_vol[3 * id + 2] - fz;
                        const float *co = conic_opacity + 7 * id;
                        const float opa = co[6];
                        const float power = vox_power(co, dx, dy, dz);
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        float alpha = opa * G;
                        if (alpha < 0.000001f) continue;
                        float dL_dalpha = 0.0f;
                        dL_dalpha += 1.0f * dL_dpixel;
                        const float dL_dG = opa * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy, gdz = G * dz;
                        const float dG_ddelx = -co[0] * gdx - co[1] * gdy - co[2] * gdz;
                        const float dG_ddely = -co[3] * gdy - co[1] * gdx - co[4] * gdz;
                        const float dG_ddelz = -co[5] * gdz - co[2] * gdx - co[4] * gdy;
                        const float v[10] = {
                            dL_dG * dG_ddelx * dvx, dL_dG * dG_ddely * dvy, dL_dG * dG_ddelz * dvz, /* Q4 */
                            (float)(-0.5 * (double)gdx * (double)dx * (double)dL_dG), /* Q6: double literals */
                            (float)(-1.0 * (double)gdx * (double)dy * (double)dL_dG),
                            (float)(-1.0 * (double)gdx * (double)dz * (double)dL_dG),
                            (float)(-0.5 * (double)gdy * (double)dy * (double)dL_dG),
                            (float)(-1.0 * (double)gdy * (double)dz * (double)dL_dG),
                            (float)(-0.5 * (double)gdz * (double)dz * (double)dL_dG),
                            G * dL_dalpha };
                        if (acc64) for (int q = 0; q < 10; q++) acc[slab + 10 * (size_t)id + q] += (double)v[q];
                        else for (int q = 0; q < 10; q++) accf[slab + 10 * (size_t)id + q] += v[q];
                    }
                }
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float v[10];
        for (int q = 0; q < 10; q++) {
            if (acc64) {
                double sum = 0.0;
                for (int t = 0; t < nthr; t++) sum += acc[(size_t)t * P * 10 + 10 * (size_t)i + q];
                v[q] = (float)sum;
            } else {
                float sum = 0.f;
                for (int t = 0; t < nthr; t++) sum += accf[(size_t)t * P * 10 + 10 * (size_t)i + q];
                v[q] = sum;
            }
        }
        for (int q = 0; q < 3; q++) dL_dmean3D_norm[3 * i + q] += v[q];
        for (int q = 0; q < 6; q++) dL_dconic3D[6 * i + q] += v[3 + q];
        dL_dopacity[i] += v[9];
    }
    free(acc); free(accf);
}

/* VOX/backward.cu:86-177 computeCov3DCUDA */
R2O_API void r2o_voxel_cov3d_bwd(int P, const int32_t *radii_x, const int32_t *radii_y, const int32_t *radii_z,
                                 const float *cov3Ds, int nx, int ny, int nz, float sx, float sy, float sz,
                                 const float *dL_dconic3D, float *dL_dcov)
{
    for (int idx = 0; idx < P; idx++) {
        if (!(radii_x[idx] > 0) || !(radii_y[idx] > 0) || !(radii_z[idx] > 0)) continue;
        float dvx = sx / (float)nx, dvy = sy / (float)ny, dvz = sz / (float)nz;
        const float *g = dL_dconic3D + 6 * idx;
        const float dL_dconic_a = g[0], dL_dconic_b = g[1], dL_dconic_c = g[2], dL_dconic_d = g[3], dL_dconic_e = g[4], dL_dconic_f = g[5];
        mat3 Mm; float h[6];
        vox_cov(cov3Ds + 6 * idx, dvx, dvy, dvz, &Mm, h);
        const mat3 *M = &Mm;
        float hata = h[0], hatb = h[1], hatc = h[2], hatd = h[3], hate = h[4], hatf = h[5];
        float denom = hata * hatd * hatf + 2 * hatb * hatc * hate - hata * hate * hate - hatf * hatb * hatb - hatd * hatc * hatc;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_dhata = 0, dL_dhatb = 0, dL_dhatc = 0, dL_dhatd = 0, dL_dhate = 0, dL_dhatf = 0;
        float *o = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            float denom_da = hatd * hatf - hate * hate;
            float denom_db = 2 * hatc * hate - 2 * hatf * hatb;
            float denom_dc = 2 * hatb * hate - 2 * hatd * hatc;
            float denom_dd = hata * hatf - hatc * hatc;
            float denom_de = 2 * hatb * hatc - 2 * hata * hate;
            float denom_df = hata * hatd - hatb * hatb;
            float ce_bf = hatc * hate - hatb * hatf;
            float be_cd = hatb * hate - hatc * hatd;
            float bc_ae = hatb * hatc - hata * hate;
            dL_dhata = denom2inv * (-denom_da*denom_da*dL_dconic_a - ce_bf*denom_da*dL_dconic_b - be_cd*denom_da*dL_dconic_c + (hatf*denom-denom_dd*denom_da)*dL_dconic_d + (-hate*denom-bc_ae*denom_da)*dL_dconic_e + (hatd*denom-denom_df*denom_da)*dL_dconic_f);
            dL_dhatb = denom2inv * (-denom_da*denom_db*dL_dconic_a + (-hatf*denom-ce_bf*denom_db)*dL_dconic_b + (hate*denom-be_cd*denom_db)*dL_dconic_c - denom_dd*denom_db*dL_dconic_d + (hatc*denom-bc_ae*denom_db)*dL_dconic_e + (-2*hatb*denom-denom_df*denom_db)*dL_dconic_f);
            dL_dhatc = denom2inv * (-denom_da*denom_dc*dL_dconic_a + (hate*denom-ce_bf*denom_dc)*dL_dconic_b + (-hatd*denom-be_cd*denom_dc)*dL_dconic_c + (-2*hatc*denom-denom_dd*denom_dc)*dL_dconic_d + (hatb*denom-bc_ae*denom_dc)*dL_dconic_e - denom_df*denom_dc*dL_dconic_f);
            dL_dhatd = denom2inv * ((hatf*denom-denom_da*denom_dd)*dL_dconic_a - ce_bf*denom_dd*dL_dconic_b +(-hatc*denom-be_cd*denom_dd)*dL_dconic_c - denom_dd*denom_dd*dL_dconic_d - bc_ae*denom_dd*dL_dconic_e + (hata*denom-denom_df*denom_dd)*dL_dconic_f);
            dL_dhate = denom2inv * ((-2*hate*denom-denom_da*denom_de)*dL_dconic_a + (hatc*denom-ce_bf*denom_de)*dL_dconic_b + (hatb*denom-be_cd*denom_de)*dL_dconic_c - denom_dd*denom_de*dL_dconic_d + (-hata*denom-bc_ae*denom_de)*dL_dconic_e + -denom_df*denom_de*dL_dconic_f);
            dL_dhatf = denom2inv * ((hatd*denom-denom_da*denom_df)*dL_dconic_a + (-hatb*denom-ce_bf*denom_df)*dL_dconic_b - be_cd*denom_df*dL_dconic_c + (hata*denom-denom_dd*denom_df)*dL_dconic_d - bc_ae*denom_df*dL_dconic_e - denom_df*denom_df*dL_dconic_f);
#define MM(c_, r_) (M->m[c_][r_])
            o[0] += MM(0,0)*MM(0,0)*dL_dhata + MM(0,0)*MM(1,0)*dL_dhatb + MM(0,0)*MM(2,0)*dL_dhatc + MM(1,0)*MM(1,0)*dL_dhatd + MM(1,0)*MM(2,0)*dL_dhate + MM(2,0)*MM(2,0)*dL_dhatf;
            o[3] += MM(0,1)*MM(0,1)*dL_dhata + MM(0,1)*MM(1,1)*dL_dhatb + MM(0,1)*MM(2,1)*dL_dhatc + MM(1,1)*MM(1,1)*dL_dhatd + MM(1,1)*MM(2,1)*dL_dhate + MM(2,1)*MM(2,1)*dL_dhatf;
            o[5] += MM(0,2)*MM(0,2)*dL_dhata + MM(0,2)*MM(1,2)*dL_dhatb + MM(0,2)*MM(2,2)*dL_dhatc + MM(1,2)*MM(1,2)*dL_dhatd + MM(1,2)*MM(2,2)*dL_dhate + MM(2,2)*MM(2,2)*dL_dhatf;
            o[1] += 2*MM(0,0)*MM(0,1)*dL_dhata + (MM(0,1)*MM(1,0)+MM(0,0)*MM(1,1))*dL_dhatb + (MM(0,1)*MM(2,0)+MM(0,0)*MM(2,1))*dL_dhatc + 2*MM(1,0)*MM(1,1)*dL_dhatd + (MM(1,1)*MM(2,0)+MM(1,0)*MM(2,1))*dL_dhate + 2*MM(2,0)*MM(2,1)*dL_dhatf;
            o[2] += 2*MM(0,0)*MM(0,2)*dL_dhata + (MM(0,2)*MM(1,0)+MM(0,0)*MM(1,2))*dL_dhatb + (MM(0,2)*MM(2,0)+MM(0,0)*MM(2,2))*dL_dhatc + 2*MM(1,0)*MM(1,2)*dL_dhatd + (MM(1,2)*MM(2,0)+MM(1,0)*MM(2,2))*dL_dhate + 2*MM(2,0)*MM(2,2)*dL_dhatf;
            o[4] += 2*MM(0,1)*MM(0,2)*dL_dhata + (MM(0,2)*MM(1,1)+MM(0,1)*MM(1,2))*dL_dhatb + (MM(0,2)*MM(2,1)+MM(0,1)*MM(2,2))*dL_dhatc + 2*MM(1,1)*MM(1,2)*dL_dhatd + (MM(1,2)*MM(2,1)+MM(1,1)*MM(2,2))*dL_dhate + 2*MM(2,1)*MM(2,2)*dL_dhatf;
#undef MM
        } else {
            for (int i = 0; i < 6; i++) o[i] = 0;
        }
    }
}

/* VOX/backward.cu:180-213 preprocessCUDA (backward) */
R2O_API void r2o_voxel_preprocess_bwd(int P, const int32_t *radii_x, const int32_t *radii_y, const int32_t *radii_z,
                                      const float *scales, const float *rotations, float scale_modifier,
                                      const float *dL_dmean3D_norm, float *dL_dmeans, const float *dL_dcov3D,
                                      float *dL_dscale, float *dL_drot)
{
    for (int idx = 0; idx < P; idx++) {
        if (!(radii_x[idx] > 0) || !(radii_y[idx] > 0) || !(radii_z[idx] > 0)) continue;
        for (int k = 0; k < 3; k++) dL_dmeans[3 * idx + k] += dL_dmean3D_norm[3 * idx + k];
        if (scales)
            cov3d_bwd(scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D + 6 * idx,
                      dL_dscale + 3 * idx, dL_drot + 4 * idx);
    }
}

/* ================================================================== SIMPLE-KNN ====== */
/* distCUDA2 of the un-vendored simple-knn submodule (r2_gaussian/gaussian/gaussian_model.py:21,147):
 * for every point, the mean of the 3 smallest squared distances to OTHER points (self excluded by
 * index).  The upstream Morton-box search is exact, so exhaustive search is the same function.
 * Parity unpinned (see header). */
R2O_API void r2o_knn_dist2(int P, const float *pts, float *out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
        const float rx = pts[3 * i], ry = pts[3 * i + 1], rz = pts[3 * i + 2];
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = pts[3 * j] - rx, dy = pts[3 * j + 1] - ry, dz = pts[3 * j + 2] - rz;
            float dist = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++)
                if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}


/* ====================================================================================== AUDIT (test bookkeeping)
 * The render kernels' cut-off tests (power > 0, alpha < 1e-5 / 1e-6) are discontinuities: an implementation that
 * evaluates alpha with different rounding (exp2 of a pre-scaled conic, a row recurrence, FMA contraction -- or nvcc's
 * own contraction of the reference) may take the other branch for a pair whose alpha sits ON a threshold.  The functions
 * below restate the same loops as the render kernels above and additionally book, per output element, how much of the
 * result hangs on such borderline pairs, so that a parity test can assert the pure relative tolerance everywhere and
 * attribute every excess to a counted, bounded set of cut-off flips instead of hiding them under an absolute floor.
 * A pair is BORDERLINE when
 *   (a) |ln(alpha) - ln(cutoff)| <= dlt, or   (b) |power| <= dlt   (the power > 0 test, RAS/forward.cu:369),
 * dlt = 32 eps_f32 * (sum of the magnitudes of the terms of `power` + |ln(opacity*mu)| + 1): a few dozen roundings of
 * the largest intermediate, evaluated in double.  Its budget is the alpha it would add.
 * The backward audit accumulates, per Gaussian and per raw gradient sum, the double-precision sum, the sum of absolute
 * terms (the honest scale of a float sum with cancellation) and the absolute terms of borderline pairs; it accumulates
 * per list instance first and reduces over instances in list order (deterministic, no per-thread slabs). */
static int r2o_borderline(double power, double mag, double lnw, double ln_cut, double *alpha_if_added)
{
    const double dlt = 32.0 * (double)FLT_EPSILON * (mag + fabs(lnw) + 1.0);
    *alpha_if_added = exp(lnw + (power < 0.0 ? power : 0.0));
    if (fabs(power) <= dlt && lnw >= ln_cut - dlt) return 1;                 /* (b) sign of power undecided */
    if (power <= dlt && fabs(lnw + power - ln_cut) <= dlt) return 1;         /* (a) alpha on the cut-off */
    return 0;
}

R2O_API void r2o_raster_render_fwd_audit(const uint32_t *ranges, const uint32_t *point_list, int W, int H,
                                         const float *means2D, const float *conic_opacity, const float *mus,
                                         float *flip_budget, uint32_t *n_border)
{
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    const double ln_cut = log((double)0.00001f);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < 16; ly++)
            for (int lx = 0; lx < 16; lx++) {
                const int pxi = tx * 16 + lx, pyi = ty * 16 + ly;
                if (!(pxi < W && pyi < H)) continue;
                double budget = 0.0;
                uint32_t nb = 0;
                for (uint32_t k = r0; k < r1; k++) {
                    const uint32_t id = point_list[k];
                    const double dx = (double)means2D[2 * id] - pxi, dy = (double)means2D[2 * id + 1] - pyi;
                    const float *co = conic_opacity + 4 * id;
                    const double t0 = 0.5 * co[0] * dx * dx, t1 = 0.5 * co[2] * dy * dy, t2 = (double)co[1] * dx * dy;
                    const double w = (double)co[3] * (double)mus[id];
                    if (!(w > 0.0)) continue;
                    double a;
                    if (r2o_borderline(-t0 - t1 - t2, fabs(t0) + fabs(t1) + fabs(t2), log(w), ln_cut, &a)) { budget += a; nb++; }
                }
                flip_budget[pyi * W + pxi] = (float)(budget * 1.0001);
                n_border[pyi * W + pxi] = nb;
            }
    }
}

/* sum / abssum / flip: [P*7] doubles, order {mean2D.x, mean2D.y, conic.x, conic.y, conic.w, opacity, mu}; R = list length */
R2O_API void r2o_raster_render_bwd_audit(const uint32_t *ranges, const uint32_t *point_list, int W, int H, int P, int64_t R,
                                         const float *means2D, const float *conic_opacity, const float *mus,
                                         const uint32_t *n_contrib, const float *dL_dpixels,
                                         double *sum, double *abssum, double *flip,
                                         int64_t pair_cap, uint32_t *pair_id, double *pair_val, int64_t *pair_count)
{
    /* pair list (optional, pair_cap > 0): every borderline pair as (Gaussian id, the 7 terms it would ADD to that Gaussian's
     * sums if an implementation decided the cut-off the other way: +v when the reference left it out, -v when it took it).
     * pair_count = how many exist (may exceed pair_cap: the caller then knows the list is truncated). */
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    const double ln_cut = log((double)0.00001f);
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
    int64_t npairs = 0;
    double *inst = (double *)calloc((size_t)R * 21 + 1, sizeof(double));   /* per list instance: 7 sums, 7 abs, 7 flip */
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (uint32_t k = r0; k < r1; k++) {
            const uint32_t id = point_list[k];
            const float *co = conic_opacity + 4 * id;
            const float mu = mus[id];
            double *acc = inst + (size_t)k * 21;
            const double w = (double)co[3] * (double)mu;
            for (int ly = 0; ly < 16; ly++)
                for (int lx = 0; lx < 16; lx++) {
                    const int pxi = tx * 16 + lx, pyi = ty * 16 + ly;
                    if (!(pxi < W && pyi < H)) continue;
                    const float dL_dpixel = dL_dpixels[pyi * W + pxi];
                    const float dx = means2D[2 * id] - (float)pxi, dy = means2D[2 * id + 1] - (float)pyi;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    const float G = expf(power <= 0.0f ? power : 0.0f);
                    const float dL_dG = co[3] * mu * dL_dpixel;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    const float v[7] = {
                        dL_dG * dG_ddelx * ddelx_dx, dL_dG * dG_ddely * ddely_dy,
                        -0.5f * gdx * dx * dL_dG, -1.0f * gdx * dy * dL_dG, -0.5f * gdy * dy * dL_dG,
                        mu * G * dL_dpixel, co[3] * G * dL_dpixel };
                    /* the reference's tests (RAS/backward.cu:523-543) */
                    const int passes = (k - r0) < n_contrib[pyi * W + pxi] && !(power > 0.0f) && !(co[3] * mu * G < 0.00001f);
                    if (passes) for (int q = 0; q < 7; q++) { acc[q] += (double)v[q]; acc[7 + q] += fabs((double)v[q]); }
                    if (w > 0.0) {
                        const double ddx = dx, ddy = dy;
                        const double t0 = 0.5 * co[0] * ddx * ddx, t1 = 0.5 * co[2] * ddy * ddy, t2 = (double)co[1] * ddx * ddy;
                        double a;
                        if (r2o_borderline(-t0 - t1 - t2, fabs(t0) + fabs(t1) + fabs(t2), log(w), ln_cut, &a)) {
                            for (int q = 0; q < 7; q++) acc[14 + q] += fabs((double)v[q]) * 1.0001;
                            if (pair_cap > 0) {
                                int64_t slot;
#pragma omp atomic capture
                                slot = npairs++;
                                if (slot < pair_cap) {
                                    pair_id[slot] = id;
                                    for (int q = 0; q < 7; q++) pair_val[7 * slot + q] = passes ? -(double)v[q] : (double)v[q];
                                }
                            }
                        }
                    }
                }
        }
    }
    memset(sum, 0, sizeof(double) * 7 * (size_t)P);
    memset(abssum, 0, sizeof(double) * 7 * (size_t)P);
    memset(flip, 0, sizeof(double) * 7 * (size_t)P);
    for (int64_t k = 0; k < R; k++) {
        const uint32_t id = point_list[k];
        for (int q = 0; q < 7; q++) {
            sum[7 * (size_t)id + q] += inst[(size_t)k * 21 + q];
            abssum[7 * (size_t)id + q] += inst[(size_t)k * 21 + 7 + q];
            flip[7 * (size_t)id + q] += inst[(size_t)k * 21 + 14 + q];
        }
    }
    free(inst);
    if (pair_count) *pair_count = npairs;
}

static void vox_terms(const float *co, double dx, double dy, double dz, double *power, double *mag)
{
    const double t[6] = { 0.5 * co[0] * dx * dx, 0.5 * co[3] * dy * dy, 0.5 * co[5] * dz * dz,
                          (double)co[1] * dx * dy, (double)co[2] * dx * dz, (double)co[4] * dy * dz };
    *power = 0.0; *mag = 0.0;
    for (int i = 0; i < 6; i++) { *power -= t[i]; *mag += fabs(t[i]); }
}

R2O_API void r2o_voxel_render_fwd_audit(const uint32_t *ranges, const uint32_t *point_list, int nx, int ny, int nz,
                                        const float *points_vol, const float *conic_opacity,
                                        float *flip_budget, uint32_t *n_border)
{
    const int gx = (nx + 7) / 8, gy = (ny + 7) / 8, gz = (nz + 7) / 8;
    const double ln_cut = log((double)0.000001f);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy * gz; tile++) {
        const int tz = tile / (gy * gx), ty = (tile / gx) % gy, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int lz = 0; lz < 8; lz++)
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++) {
                    const int vx = tx * 8 + lx, vy = ty * 8 + ly, vz = tz * 8 + lz;
                    if (!(vx < nx && vy < ny && vz < nz)) continue;
                    const size_t vid = (size_t)nz * ny * vx + (size_t)nz * vy + vz;
                    const float fx = (float)vx + 0.5f, fy = (float)vy + 0.5f, fz = (float)vz + 0.5f;
                    double budget = 0.0;
                    uint32_t nb = 0;
                    for (uint32_t k = r0; k < r1; k++) {
                        const uint32_t id = point_list[k];
                        const float dx = points_vol[3 * id] - fx, dy = points_vol[3 * id + 1] - fy, dz = points_vol[3 * id + 2] - fz;
                        const float *co = conic_opacity + 7 * id;
                        if (!(co[6] > 0.0f)) continue;
                        double power, mag, a;
                        vox_terms(co, dx, dy, dz, &power, &mag);
                        if (r2o_borderline(power, mag, log((double)co[6]), ln_cut, &a)) { budget += a; nb++; }
                    }
                    flip_budget[vid] = (float)(budget * 1.0001);
                    n_border[vid] = nb;
                }
    }
}

/* sum / abssum / flip: [P*10] doubles, order {mean.x, mean.y, mean.z, conic[6], opacity} */
R2O_API void r2o_voxel_render_bwd_audit(const uint32_t *ranges, const uint32_t *point_list, int nx, int ny, int nz,
                                        float sx, float sy, float sz, int P, int64_t R,
                                        const float *points_vol, const float *conic_opacity, const uint32_t *n_contrib,
                                        const float *dL_dpixels, double *sum, double *abssum, double *flip,
                                        int64_t pair_cap, uint32_t *pair_id, double *pair_val, int64_t *pair_count)
{
    const int gx = (nx + 7) / 8, gy = (ny + 7) / 8, gz = (nz + 7) / 8;
    const float dvx = sx / (float)nx, dvy = sy / (float)ny, dvz = sz / (float)nz;
    const double ln_cut = log((double)0.000001f);
    int64_t npairs = 0;   /* pair list: see r2o_raster_render_bwd_audit */
    double *inst = (double *)calloc((size_t)R * 30 + 1, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy * gz; tile++) {
        const int tz = tile / (gy * gx), ty = (tile / gx) % gy, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (uint32_t k = r0; k < r1; k++) {
            const uint32_t id = point_list[k];
            const float *co = conic_opacity + 7 * id;
            const float opa = co[6];
            double *acc = inst + (size_t)k * 30;
            for (int lz = 0; lz < 8; lz++)
                for (int ly = 0; ly < 8; ly++)
                    for (int lx = 0; lx < 8; lx++) {
                        const int vx = tx * 8 + lx, vy = ty * 8 + ly, vz = tz * 8 + lz;
                        if (!(vx < nx && vy < ny && vz < nz)) continue;
                        const size_t vid = (size_t)nz * ny * vx + (size_t)nz * vy + vz;
                        const float fx = (float)vx + 0.5f, fy = (float)vy + 0.5f, fz = (float)vz + 0.5f;
                        const float dL_dpixel = dL_dpixels[vid];
                        const float dx = points_vol[3 * id] - fx, dy = points_vol[3 * id + 1] - fy, dz = points_vol[3 * id + 2] - fz;
                        const float power = vox_power(co, dx, dy, dz);
                        const float G = expf(power <= 0.0f ? power : 0.0f);
                        const float dL_dG = opa * dL_dpixel;
                        const float gdx = G * dx, gdy = G * dy, gdz = G * dz;
                        const float dG_ddelx = -co[0] * gdx - co[1] * gdy - co[2] * gdz;
                        const float dG_ddely = -co[3] * gdy - co[1] * gdx - co[4] * gdz;
                        const float dG_ddelz = -co[5] * gdz - co[2] * gdx - co[4] * gdy;
                        const float v[10] = {
                            dL_dG * dG_ddelx * dvx, dL_dG * dG_ddely * dvy, dL_dG * dG_ddelz * dvz,
                            (float)(-0.5 * (double)gdx * (double)dx * (double)dL_dG),
                            (float)(-1.0 * (double)gdx * (double)dy * (double)dL_dG),
                            (float)(-1.0 * (double)gdx * (double)dz * (double)dL_dG),
                            (float)(-0.5 * (double)gdy * (double)dy * (double)dL_dG),
                            (float)(-1.0 * (double)gdy * (double)dz * (double)dL_dG),
                            (float)(-0.5 * (double)gdz * (double)dz * (double)dL_dG),
                            G * dL_dpixel };
                        const int passes = (k - r0) < n_contrib[vid] && !(power > 0.0f) && !(opa * G < 0.000001f);
                        if (passes) for (int q = 0; q < 10; q++) { acc[q] += (double)v[q]; acc[10 + q] += fabs((double)v[q]); }
                        if (opa > 0.0f) {
                            double pw, mag, a;
                            vox_terms(co, dx, dy, dz, &pw, &mag);
                            if (r2o_borderline(pw, mag, log((double)opa), ln_cut, &a)) {
                                for (int q = 0; q < 10; q++) acc[20 + q] += fabs((double)v[q]) * 1.0001;
                                if (pair_cap > 0) {
                                    int64_t slot;
#pragma omp atomic capture
                                    slot = npairs++;
                                    if (slot < pair_cap) {
                                        pair_id[slot] = id;
                                        for (int q = 0; q < 10; q++) pair_val[10 * slot + q] = passes ? -(double)v[q] : (double)v[q];
                                    }
                                }
                            }
                        }
                    }
        }
    }
    memset(sum, 0, sizeof(double) * 10 * (size_t)P);
    memset(abssum, 0, sizeof(double) * 10 * (size_t)P);
    memset(flip, 0, sizeof(double) * 10 * (size_t)P);
    for (int64_t k = 0; k < R; k++) {
        const uint32_t id = point_list[k];
        for (int q = 0; q < 10; q++) {
            sum[10 * (size_t)id + q] += inst[(size_t)k * 30 + q];
            abssum[10 * (size_t)id + q] += inst[(size_t)k * 30 + 10 + q];
            flip[10 * (size_t)id + q] += inst[(size_t)k * 30 + 20 + q];
        }
    }
    free(inst);
    if (pair_count) *pair_count = npairs;
}

R2O_API int r2o_abi_version(void) { return 2; }
R2O_API int r2o_num_threads(void) { return omp_get_max_threads(); }
R2O_API void r2o_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
