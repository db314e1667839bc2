// ref_entry_voxel.cpp -- extern "C" door into the REFERENCE's CudaVoxelizer::Voxelizer (compiled for the CPU by
// oracle/Makefile from /root/reference/.../cuda_voxelizer/*.cu through cuda_on_cpu.h); stands in for
// SUB/voxelize_points.cu without torch.  TEST INFRASTRUCTURE ONLY (oracle/_ref build).
#include "cuda_on_cpu.h"
#include "voxelizer.h"
#include "voxelizer_impl.h"

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
std::vector<char> g_geom, g_bin, g_img;
int g_P = 0, g_R = 0, g_n[3] = { 0, 0, 0 };
std::function<char *(size_t)> resizer(std::vector<char> &v)
{
    return [&v](size_t n) { v.assign(n + 256, 0); return v.data(); };
}
}  // namespace

REF_API int r2ref_voxel_forward(int P, int nx, int ny, int nz, float sx, float sy, float sz, float cx, float cy, float cz,
                                const float *means3D, const float *opacities, const float *scales, float scale_modifier,
                                const float *rotations, const float *cov3D_precomp, float *out_volume, int *radii_x,
                                int *radii_y, int *radii_z)
{
    g_P = P; g_R = 0; g_n[0] = nx; g_n[1] = ny; g_n[2] = nz;
    memset(out_volume, 0, sizeof(float) * (size_t)nx * ny * nz);   // SUB/voxelize_points.cu:58
    memset(radii_x, 0, 4 * (size_t)P); memset(radii_y, 0, 4 * (size_t)P); memset(radii_z, 0, 4 * (size_t)P);
    if (P == 0) return 0;
    g_R = CudaVoxelizer::Voxelizer::forward(resizer(g_geom), resizer(g_bin), resizer(g_img), P, nx, ny, nz, sx, sy, sz, cx, cy,
                                            cz, means3D, opacities, scales, scale_modifier, rotations, cov3D_precomp, false,
                                            out_volume, radii_x, radii_y, radii_z, false);
    return g_R;
}

// which: 0 depths f32[P] 1 means3D_norm f32[3P] 2 cov3D f32[6P] 3 conic_opacity f32[7P] 5 tiles_touched u32[P]
// 6 point_offsets u32[P] 7 keys_unsorted u64[R] 8 vals_unsorted u32[R] 9 keys u64[R] 10 point_list u32[R]
// 11 ranges u32[2T] 12 n_contrib u32[nx*ny*nz]
REF_API int r2ref_voxel_get(int which, void *dst)
{
    using namespace CudaVoxelizer;
    if (g_P == 0) return 0;
    char *gp = g_geom.data(), *bp = g_bin.data(), *ip = g_img.data();
    const size_t P = g_P, R = g_R, N = (size_t)g_n[0] * g_n[1] * g_n[2];
    const size_t T = (size_t)((g_n[0] + 7) / 8) * ((g_n[1] + 7) / 8) * ((g_n[2] + 7) / 8);
    GeometryState g = GeometryState::fromChunk(gp, P);
    BinningState b = BinningState::fromChunk(bp, R);
    ImageState im = ImageState::fromChunk(ip, N);
    switch (which) {
    case 0: memcpy(dst, g.depths, 4 * P); break;
    case 1: memcpy(dst, g.means3D_norm, 12 * P); break;
    case 2: memcpy(dst, g.cov3D, 24 * P); break;
    case 3: memcpy(dst, g.conic_opacity, 28 * P); break;
    case 5: memcpy(dst, g.tiles_touched, 4 * P); break;
    case 6: memcpy(dst, g.point_offsets, 4 * P); break;
    case 7: memcpy(dst, b.point_list_keys_unsorted, 8 * R); break;
    case 8: memcpy(dst, b.point_list_unsorted, 4 * R); break;
    case 9: memcpy(dst, b.point_list_keys, 8 * R); break;
    case 10: memcpy(dst, b.point_list, 4 * R); break;
    case 11: memcpy(dst, im.ranges, 8 * T); break;
    case 12: memcpy(dst, im.n_contrib, 4 * N); break;
    default: return -1;
    }
    return 0;
}

// gradient outputs zero-initialised as SUB/voxelize_points.cu:130-136 does
REF_API void r2ref_voxel_backward(float sx, float sy, float sz, float cx, float cy, float cz, const float *means3D,
                                  const float *scales, float scale_modifier, const float *rotations,
                                  const float *cov3D_precomp, const int *radii_x, const int *radii_y, const int *radii_z,
                                  const float *dL_dvol, float *dL_dmean3D_norm, float *dL_dconic3D, float *dL_dopacity,
                                  float *dL_dmean3D, float *dL_dcov3D, float *dL_dscale, float *dL_drot)
{
    const size_t P = g_P;
    memset(dL_dmean3D_norm, 0, 12 * P); memset(dL_dconic3D, 0, 24 * P); memset(dL_dopacity, 0, 4 * P);
    memset(dL_dmean3D, 0, 12 * P); memset(dL_dcov3D, 0, 24 * P); memset(dL_dscale, 0, 12 * P); memset(dL_drot, 0, 16 * P);
    if (P == 0) return;
    CudaVoxelizer::Voxelizer::backward((int)P, g_R, g_n[0], g_n[1], g_n[2], sx, sy, sz, cx, cy, cz, means3D, scales,
                                       scale_modifier, rotations, cov3D_precomp, radii_x, radii_y, radii_z, g_geom.data(),
                                       g_bin.data(), g_img.data(), dL_dvol, dL_dmean3D_norm, dL_dconic3D, dL_dopacity,
                                       dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, false);
}
