// densify_ops.hip -- adaptive density control on the device: the densification statistics of a view, and
// densify (clone / split) + prune with the optimizer-state surgery, as a handful of kernels and ONE host read instead of
// ~30 torch ops with boolean-mask indexing (each a device synchronisation) and two full passes of optimizer-state
// re-allocation.  Reference: r2_gaussian/gaussian/gaussian_model.py:320-556 (densify_and_clone, densify_and_split,
// prune_points, cat_tensors_to_optimizer, _prune_optimizer, add_densification_stats), train.py:151-168.
//
// Result layout = the reference's, element for element: [surviving originals in order | surviving clones | surviving first
// split children | surviving second split children] (densification_postfix appends clones, then the 2 x selected children as
// two repeated blocks; the split parents and the pruned rows are then removed by stable masks).
#include "r2_common.hpp"

namespace r2 {
namespace {

struct DensifyCfg {
    float grad_thr, scale_thr, density_min;
    float lo[3], hi[3];          // bounding box
    float s_lo, s_hi;            // bounded-sigmoid scaling activation; s_lo >= s_hi: exp activation (no bound)
    int do_densify;
    float max_screen, max_scale; // optional prune thresholds (gaussian_model.py:540-545); <= 0: off (the reference's None)
};

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch.nn.Softplus()
__device__ __forceinline__ float inv_softplus_f(float y) { return logf(expf(y) - 1.0f); }        // utils/gaussian_utils.py:5-6
__device__ __forceinline__ float scale_act(float x, const DensifyCfg &c)
{
    return c.s_lo < c.s_hi ? 1.0f / (1.0f + expf(-x)) * (c.s_hi - c.s_lo) + c.s_lo : expf(x);
}
__device__ __forceinline__ float scale_inv(float y, const DensifyCfg &c)
{
    if (!(c.s_lo < c.s_hi)) return logf(y);
    const float t = fmaxf((y - c.s_lo) / (c.s_hi - c.s_lo), 0.f);
    return logf(t / (1.0f - t));
}
__device__ __forceinline__ bool outside(const float *p, const DensifyCfg &c)
{
    return p[0] < c.lo[0] || p[0] > c.hi[0] || p[1] < c.lo[1] || p[1] > c.hi[1] || p[2] < c.lo[2] || p[2] > c.hi[2];
}

struct Decision {
    bool clone, split, keep_orig, keep_clone, keep_child[2];
    float density_raw_orig;       // raw density of the original after the clone step (halved if cloned)
    float child_xyz[2][3], child_density_raw, child_scaling_raw[3];
};

// everything densify_and_prune decides about Gaussian i (gaussian_model.py:430-550), recomputed by both passes
__device__ __forceinline__ Decision decide(int i, int P, const float *__restrict__ xyz, const float *__restrict__ density,
                                           const float *__restrict__ scaling, const float *__restrict__ rotation,
                                           const float *__restrict__ max_radii, const float *__restrict__ grad_accum,
                                           const float *__restrict__ denom, const float *__restrict__ normals, const DensifyCfg &c)
{
    Decision d;
    const float p[3] = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
    const float s[3] = { scale_act(scaling[3 * i], c), scale_act(scaling[3 * i + 1], c), scale_act(scaling[3 * i + 2], c) };
    const float smax = fmaxf(fmaxf(s[0], s[1]), s[2]);
    const float dens = softplus_f(density[i]);
    float g = grad_accum[i] / denom[i];
    if (g != g) g = 0.f;                                   // grads[grads.isnan()] = 0
    const bool hot = c.do_densify && g >= c.grad_thr;
    d.clone = hot && smax <= c.scale_thr;                  // :474-483
    d.split = hot && smax > c.scale_thr;                   // :430-441
    d.density_raw_orig = d.clone ? inv_softplus_f(dens * 0.5f) : density[i];   // :486-493: both copies get half the density
    // the prune mask is evaluated AFTER clone / split on all rows (:524-546); a clone shares its original's scale and
    // max_radii2D, split children inherit the parent's max_radii2D (:468) and get scale / 1.6
    const bool big_screen = c.max_screen > 0.f && max_radii[i] > c.max_screen;   // :540-542
    const bool pruned = softplus_f(d.density_raw_orig) < c.density_min || outside(p, c) || big_screen ||
                        (c.max_scale > 0.f && smax > c.max_scale);               // :543-545
    d.keep_orig = !d.split && !pruned;
    d.keep_clone = d.clone && !pruned;
    d.keep_child[0] = d.keep_child[1] = false;
    if (d.split) {
        // rotation matrix of the normalised quaternion (utils/gaussian_utils.py:49-84)
        float q0 = rotation[4 * i], q1 = rotation[4 * i + 1], q2 = rotation[4 * i + 2], q3 = rotation[4 * i + 3];
        const float n = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        q0 /= n; q1 /= n; q2 /= n; q3 /= n;
        const float R[3][3] = { { 1 - 2 * (q2 * q2 + q3 * q3), 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2) },
                                { 2 * (q1 * q2 + q0 * q3), 1 - 2 * (q1 * q1 + q3 * q3), 2 * (q2 * q3 - q0 * q1) },
                                { 2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 1 - 2 * (q1 * q1 + q2 * q2) } };
        d.child_density_raw = inv_softplus_f(dens * 0.5f);
        for (int k = 0; k < 3; ++k) d.child_scaling_raw[k] = scale_inv(s[k] / 1.6f, c);   // / (0.8 N), N = 2
        bool low = softplus_f(d.child_density_raw) < c.density_min || big_screen;
        if (c.max_scale > 0.f)
            low = low || fmaxf(fmaxf(scale_act(d.child_scaling_raw[0], c), scale_act(d.child_scaling_raw[1], c)),
                               scale_act(d.child_scaling_raw[2], c)) > c.max_scale;
        for (int ch = 0; ch < 2; ++ch) {
            const float *nn = normals + ((size_t)ch * P + i) * 3;
            const float v[3] = { nn[0] * s[0], nn[1] * s[1], nn[2] * s[2] };   // N(0, scale) in the local frame
            for (int r = 0; r < 3; ++r) d.child_xyz[ch][r] = R[r][0] * v[0] + R[r][1] * v[1] + R[r][2] * v[2] + p[r];
            d.keep_child[ch] = !low && !outside(d.child_xyz[ch], c);
        }
    }
    return d;
}

__global__ void __launch_bounds__(256) densify_classify_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ density,
                                                               const float *__restrict__ scaling, const float *__restrict__ rotation,
                                                               const float *__restrict__ max_radii,
                                                               const float *__restrict__ grad_accum, const float *__restrict__ denom,
                                                               const float *__restrict__ normals, DensifyCfg c,
                                                               uint32_t *__restrict__ keep /* [4][P] */)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const Decision d = decide(i, P, xyz, density, scaling, rotation, max_radii, grad_accum, denom, normals, c);
    keep[i] = d.keep_orig;
    keep[(size_t)P + i] = d.keep_clone;
    keep[2 * (size_t)P + i] = d.keep_child[0];
    keep[3 * (size_t)P + i] = d.keep_child[1];
}

struct Rows {   // the four parameters, their two Adam moments (nullable), the per-Gaussian statistics
    const float *p[4];        // xyz[3], density[1], scaling[3], rotation[4]
    const float *m[4], *v[4]; // exp_avg, exp_avg_sq
    float *po[4], *mo[4], *vo[4];
    const float *max_radii;
    float *max_radii_out, *grad_accum_out, *denom_out;
};
__device__ const int ROW_W[4] = { 3, 1, 3, 4 };

__global__ void __launch_bounds__(256) densify_emit_kernel(int P, Rows r, const float *__restrict__ grad_accum,
                                                           const float *__restrict__ denom, const float *__restrict__ normals,
                                                           DensifyCfg c, const uint32_t *__restrict__ pos /* [4][P] inclusive scans */)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const Decision d = decide(i, P, r.p[0], r.p[1], r.p[2], r.p[3], r.max_radii, grad_accum, denom, normals, c);
    const uint32_t n0 = pos[P - 1], n1 = pos[2 * (size_t)P - 1], n2 = pos[3 * (size_t)P - 1];
    auto write_stats = [&](size_t o) {
        r.max_radii_out[o] = r.max_radii[i];
        // densification_postfix resets the statistics (:423-425); a prune-only call (P >= max_num_gaussians) keeps them
        r.grad_accum_out[o] = c.do_densify ? 0.f : grad_accum[i];
        r.denom_out[o] = c.do_densify ? 0.f : denom[i];
    };
    if (d.keep_orig) {   // the original: parameters and Adam moments move with it
        const size_t o = pos[i] - 1u;
        for (int a = 0; a < 4; ++a)
            for (int k = 0; k < ROW_W[a]; ++k) {
                const size_t src = (size_t)i * ROW_W[a] + k, dst = o * ROW_W[a] + k;
                r.po[a][dst] = (a == 1) ? d.density_raw_orig : r.p[a][src];
                if (r.mo[a]) { r.mo[a][dst] = r.m[a][src]; r.vo[a][dst] = r.v[a][src]; }
            }
        write_stats(o);
    }
    auto write_new = [&](size_t o, const float *xyz_new, float dens_raw, const float *scal_raw) {   // fresh rows: zero moments
        for (int k = 0; k < 3; ++k) r.po[0][o * 3 + k] = xyz_new[k];
        r.po[1][o] = dens_raw;
        for (int k = 0; k < 3; ++k) r.po[2][o * 3 + k] = scal_raw[k];
        for (int k = 0; k < 4; ++k) r.po[3][o * 4 + k] = r.p[3][(size_t)i * 4 + k];
        for (int a = 0; a < 4; ++a)
            if (r.mo[a])
                for (int k = 0; k < ROW_W[a]; ++k) { r.mo[a][o * ROW_W[a] + k] = 0.f; r.vo[a][o * ROW_W[a] + k] = 0.f; }
        write_stats(o);
    };
    if (d.keep_clone) {
        const float p3[3] = { r.p[0][3 * i], r.p[0][3 * i + 1], r.p[0][3 * i + 2] };
        const float s3[3] = { r.p[2][3 * i], r.p[2][3 * i + 1], r.p[2][3 * i + 2] };
        write_new((size_t)n0 + pos[(size_t)P + i] - 1u, p3, d.density_raw_orig, s3);
    }
    if (d.keep_child[0]) write_new((size_t)n0 + n1 + pos[2 * (size_t)P + i] - 1u, d.child_xyz[0], d.child_density_raw, d.child_scaling_raw);
    if (d.keep_child[1]) write_new((size_t)n0 + n1 + n2 + pos[3 * (size_t)P + i] - 1u, d.child_xyz[1], d.child_density_raw, d.child_scaling_raw);
}

// train.py:151-154 + add_densification_stats (gaussian_model.py:552-556), one launch, no boolean-mask indexing
__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const int *__restrict__ radii, const float *__restrict__ g2d,
                                                            float *__restrict__ max_radii, float *__restrict__ grad_accum,
                                                            float *__restrict__ denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int rad = radii[i];
    if (!(rad > 0)) return;
    max_radii[i] = fmaxf(max_radii[i], (float)rad);
    const float gx = g2d[3 * i], gy = g2d[3 * i + 1];
    grad_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}

}  // namespace
}  // namespace r2

extern "C" int r2_densify_stats(int P, const int *radii, const float *dL_dmeans2D, float *max_radii2D, float *grad_accum,
                                float *denom, void *stream)
{
    if (P == 0) return 0;
    if (P < 0 || !radii || !dL_dmeans2D || !max_radii2D || !grad_accum || !denom) {
        r2::set_error("r2_densify_stats: invalid argument");
        return R2_ERR_INVALID;
    }
    r2::densify_stats_kernel<<<dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(P, radii, dL_dmeans2D, max_radii2D,
                                                                                          grad_accum, denom);
    R2_STAGE_CHECK(0, (hipStream_t)stream, "densification statistics");
    return 0;
}

extern "C" size_t r2_densify_scratch_bytes(int P)
{
    return 2 * 4 * (size_t)P * sizeof(uint32_t) + r2::scan_temp_bytes(4 * P) + 1024;
}

static r2::DensifyCfg make_cfg(float grad_thr, float scale_thr, float density_min, const float *bbox, float scale_lo, float scale_hi,
                               int do_densify, float max_screen_size, float max_scale)
{
    r2::DensifyCfg c;
    c.grad_thr = grad_thr; c.scale_thr = scale_thr; c.density_min = density_min;
    for (int k = 0; k < 3; ++k) { c.lo[k] = bbox[k]; c.hi[k] = bbox[3 + k]; }
    c.s_lo = scale_lo; c.s_hi = scale_hi; c.do_densify = do_densify;
    c.max_screen = max_screen_size; c.max_scale = max_scale;
    return c;
}

// pass 1: decide and count.  counts_host[4] = surviving {originals, clones, first children, second children}; the call
// synchronises the stream once to read them (the caller sizes the outputs with them).
extern "C" int r2_densify_classify(int P, const float *xyz, const float *density, const float *scaling, const float *rotation,
                                   const float *max_radii2D, const float *grad_accum, const float *denom, const float *normals,
                                   float grad_thr, float scale_thr, float density_min, const float *bbox_host, float scale_lo,
                                   float scale_hi, int do_densify, float max_screen_size, float max_scale, void *scratch,
                                   unsigned int *counts_host, void *stream)
{
    using namespace r2;
    if (P <= 0 || !xyz || !density || !scaling || !rotation || !max_radii2D || !grad_accum || !denom || !normals || !bbox_host ||
        !scratch || !counts_host) {
        set_error("r2_densify_classify: invalid argument");
        return R2_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    const DensifyCfg c = make_cfg(grad_thr, scale_thr, density_min, bbox_host, scale_lo, scale_hi, do_densify, max_screen_size, max_scale);
    uint32_t *keep = reinterpret_cast<uint32_t *>(scratch), *pos = keep + 4 * (size_t)P;
    char *scan_temp = reinterpret_cast<char *>(pos + 4 * (size_t)P);
    densify_classify_kernel<<<dim3((P + 255) / 256), dim3(256), 0, s>>>(P, xyz, density, scaling, rotation, max_radii2D, grad_accum,
                                                                       denom, normals, c, keep);
    for (int a = 0; a < 4; ++a) {
        const int rc = inclusive_scan_u32(scan_temp, scan_temp_bytes(4 * P), keep + (size_t)a * P, pos + (size_t)a * P, P, s);
        if (rc) return rc;
    }
    uint32_t h[4];
    for (int a = 0; a < 4; ++a)
        R2_HIP_TRY(hipMemcpyAsync(&h[a], pos + (size_t)(a + 1) * P - 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    R2_HIP_TRY(hipStreamSynchronize(s));
    for (int a = 0; a < 4; ++a) counts_host[a] = h[a];
    return 0;
}

// pass 2: write the new parameter / Adam-moment / statistics arrays (sized sum(counts) rows) from the decisions of pass 1
// (recomputed: same inputs, same code).  params / exp_avg / exp_avg_sq: 4 pointers each in the order xyz, density, scaling,
// rotation; the moment arrays may be all NULL (no optimizer state yet).
extern "C" int r2_densify_emit(int P, const float *const *params, const float *const *exp_avg, const float *const *exp_avg_sq,
                               const float *max_radii2D, const float *grad_accum, const float *denom, const float *normals,
                               float grad_thr, float scale_thr, float density_min, const float *bbox_host, float scale_lo,
                               float scale_hi, int do_densify, float max_screen_size, float max_scale, const void *scratch,
                               float *const *params_out,
                               float *const *exp_avg_out, float *const *exp_avg_sq_out, float *max_radii2D_out,
                               float *grad_accum_out, float *denom_out, void *stream)
{
    using namespace r2;
    if (P <= 0 || !params || !params_out || !max_radii2D || !max_radii2D_out || !grad_accum_out || !denom_out || !scratch) {
        set_error("r2_densify_emit: invalid argument");
        return R2_ERR_INVALID;
    }
    Rows r;
    for (int a = 0; a < 4; ++a) {
        r.p[a] = params[a]; r.po[a] = params_out[a];
        r.m[a] = exp_avg ? exp_avg[a] : nullptr; r.v[a] = exp_avg_sq ? exp_avg_sq[a] : nullptr;
        r.mo[a] = (exp_avg_out && r.m[a]) ? exp_avg_out[a] : nullptr; r.vo[a] = (exp_avg_sq_out && r.v[a]) ? exp_avg_sq_out[a] : nullptr;
        if (!r.p[a] || !r.po[a] || ((r.mo[a] == nullptr) != (r.vo[a] == nullptr))) {
            set_error("r2_densify_emit: NULL parameter array / inconsistent moment arrays");
            return R2_ERR_INVALID;
        }
    }
    r.max_radii = max_radii2D; r.max_radii_out = max_radii2D_out; r.grad_accum_out = grad_accum_out; r.denom_out = denom_out;
    const DensifyCfg c = make_cfg(grad_thr, scale_thr, density_min, bbox_host, scale_lo, scale_hi, do_densify, max_screen_size, max_scale);
    const uint32_t *pos = reinterpret_cast<const uint32_t *>(scratch) + 4 * (size_t)P;
    densify_emit_kernel<<<dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(P, r, grad_accum, denom, normals, c, pos);
    R2_STAGE_CHECK(0, (hipStream_t)stream, "densify emit");
    return 0;
}
