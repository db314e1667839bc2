// ref_entry_raster.cpp -- extern "C" door into the REFERENCE's CudaRasterizer::Rasterizer (compiled for the CPU
// by oracle/Makefile from /root/reference/.../cuda_rasterizer/*.cu through cuda_on_cpu.h).  It plays the role of
// SUB/rasterize_points.cu (the torch boundary) without torch: zero-initialised outputs, malloc-backed state
// buffers, and getters for the private intermediates so tests can compare them bit for bit.
// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
#include "cuda_on_cpu.h"
#include "rasterizer.h"
#include "rasterizer_impl.h"

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
std::vector<char> g_geom, g_bin, g_img;
int g_P = 0, g_R = 0, g_W = 0, g_H = 0;
std::function<char *(size_t)> resizer(std::vector<char> &v)
{
    return [&v](size_t n) { v.assign(n + 256, 0); return v.data(); };
}
}  // namespace

REF_API int r2ref_raster_forward(int P, int W, int H, const float *means3D, const float *opacities, const float *scales,
                                 float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                 const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                                 float tan_fovy, int mode, float *out_color, int *radii)
{
    g_P = P; g_W = W; g_H = H; g_R = 0;
    memset(out_color, 0, sizeof(float) * W * H);            // SUB/rasterize_points.cu:58
    memset(radii, 0, sizeof(int) * P);                      // SUB/rasterize_points.cu:59
    if (P == 0) return 0;                                   // SUB/rasterize_points.cu:70
    g_R = CudaRasterizer::Rasterizer::forward(resizer(g_geom), resizer(g_bin), resizer(g_img), P, W, H, means3D, opacities,
                                              scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                              campos, tan_fovx, tan_fovy, false, mode, out_color, radii, false);
    return g_R;
}

// which: 0 depths f32[P] 1 means2D f32[2P] 2 cov3D f32[6P] 3 conic_opacity f32[4P] 4 mus f32[P] 5 tiles_touched u32[P]
// 6 point_offsets u32[P] 7 keys_unsorted u64[R] 8 vals_unsorted u32[R] 9 keys u64[R] 10 point_list u32[R]
// 11 ranges u32[2T] 12 n_contrib u32[W*H]
REF_API int r2ref_raster_get(int which, void *dst)
{
    using namespace CudaRasterizer;
    char *gp = g_geom.data(), *bp = g_bin.data(), *ip = g_img.data();
    if (g_P == 0) return 0;
    GeometryState g = GeometryState::fromChunk(gp, g_P);
    BinningState b = BinningState::fromChunk(bp, g_R);
    ImageState im = ImageState::fromChunk(ip, (size_t)g_W * g_H);
    const size_t P = g_P, R = g_R, T = (size_t)((g_W + 15) / 16) * ((g_H + 15) / 16);
    switch (which) {
    case 0: memcpy(dst, g.depths, 4 * P); break;
    case 1: memcpy(dst, g.means2D, 8 * P); break;
    case 2: memcpy(dst, g.cov3D, 24 * P); break;
    case 3: memcpy(dst, g.conic_opacity, 16 * P); break;
    case 4: memcpy(dst, g.mus, 4 * P); break;
    case 5: memcpy(dst, g.tiles_touched, 4 * P); break;
    case 6: memcpy(dst, g.point_offsets, 4 * P); break;
    case 7: memcpy(dst, b.point_list_keys_unsorted, 8 * R); break;
    case 8: memcpy(dst, b.point_list_unsorted, 4 * R); break;
    case 9: memcpy(dst, b.point_list_keys, 8 * R); break;
    case 10: memcpy(dst, b.point_list, 4 * R); break;
    case 11: memcpy(dst, im.ranges, 8 * T); break;
    case 12: memcpy(dst, im.n_contrib, 4 * (size_t)g_W * g_H); break;
    default: return -1;
    }
    return 0;
}

// gradient outputs are zero-initialised here, as SUB/rasterize_points.cu:124-131 does
REF_API void r2ref_raster_backward(const float *means3D, const float *scales, float scale_modifier, const float *rotations,
                                   const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                                   const float *campos, float tan_fovx, float tan_fovy, const int *radii, int mode,
                                   const float *dL_dpix, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity,
                                   float *dL_dmu, float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot)
{
    const size_t P = g_P;
    memset(dL_dmean2D, 0, 12 * P); memset(dL_dconic, 0, 16 * P); memset(dL_dopacity, 0, 4 * P); memset(dL_dmu, 0, 4 * P);
    memset(dL_dmean3D, 0, 12 * P); memset(dL_dcov3D, 0, 24 * P); memset(dL_dscale, 0, 12 * P); memset(dL_drot, 0, 16 * P);
    if (P == 0) return;                                     // SUB/rasterize_points.cu:133
    CudaRasterizer::Rasterizer::backward((int)P, g_R, g_W, g_H, means3D, scales, scale_modifier, rotations, cov3D_precomp,
                                         viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, g_geom.data(),
                                         g_bin.data(), g_img.data(), dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dmu,
                                         dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, mode, false);
}

REF_API void r2ref_mark_visible(int P, float *means3D, float *viewmatrix, float *projmatrix, unsigned char *present)
{
    static_assert(sizeof(bool) == 1, "bool is one byte");
    if (P) CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, (bool *)present);
}
