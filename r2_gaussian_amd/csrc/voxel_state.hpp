// voxel_state.hpp -- private layout of the voxelizer's opaque state buffers (the reference's
// GeometryState / BinningState / ImageState, VOX/voxelizer_impl.h, VOX/voxelizer_impl.cu:130-169).
#pragma once
#include "r2_common.hpp"

namespace r2 {

struct VoxelGeom {
    float4 *rec;              // [3P] {x,y,z (voxel units), opacity} {a2,b2,c2,d2} {e2,f2,-,-}: inverse covariance
                              //      (xx,xy,xz,yy,yz,zz) pre-scaled by -log2e/2 (diagonal) or -log2e (off-diagonal)
    float *depths;            // [P]  world z, the arbitrary low sort word of the reference (Q10)
    float *cov3D;             // [6P]
    uint32_t *tiles_touched;  // [P]
    uint32_t *offsets;        // [P]
    char *scan_temp;
    size_t scan_bytes;
    size_t bytes;
    static VoxelGeom carve(char *chunk, int P)
    {
        VoxelGeom g;
        Bump b(chunk);
        g.rec = b.take<float4>(3 * (size_t)P);
        g.depths = b.take<float>(P);
        g.cov3D = b.take<float>(6 * (size_t)P);
        g.tiles_touched = b.take<uint32_t>(P);
        g.offsets = b.take<uint32_t>(P);
        g.scan_bytes = scan_temp_bytes(P);
        g.scan_temp = b.take<char>(g.scan_bytes);
        g.bytes = b.total();
        return g;
    }
};

struct VoxelBinning {
    uint64_t *keys_unsorted, *keys;
    uint32_t *vals_unsorted, *point_list;
    char *sort_temp;
    size_t sort_bytes;
    size_t bytes;
    static VoxelBinning carve(char *chunk, size_t R)
    {
        VoxelBinning s;
        Bump b(chunk);
        s.keys_unsorted = b.take<uint64_t>(R);
        s.keys = b.take<uint64_t>(R);
        s.vals_unsorted = b.take<uint32_t>(R);
        s.point_list = b.take<uint32_t>(R);
        s.sort_bytes = sort_temp_bytes(R);
        s.sort_temp = b.take<char>(s.sort_bytes);
        s.bytes = b.total();
        return s;
    }
};

struct VoxelImage {
    uint2 *ranges;        // [T3]
    uint32_t *n_contrib;  // [V], debug only
    size_t bytes;
    static VoxelImage carve(char *chunk, size_t T, size_t V, bool with_ncontrib)
    {
        VoxelImage s;
        Bump b(chunk);
        s.ranges = b.take<uint2>(T);
        s.n_contrib = b.take<uint32_t>(with_ncontrib ? V : 0);
        s.bytes = b.total();
        return s;
    }
};

struct VoxelGrid {
    int nx, ny, nz;
    float sx, sy, sz;
    float cx, cy, cz;
    int gx, gy, gz;
};

int launch_voxel_preprocess(const VoxelGeom &g, const VoxelGrid &v, int P, const float *means3D, const float *scales,
                            float scale_modifier, const float *rotations, const float *opacities,
                            const float *cov3D_precomp, int *radii_x, int *radii_y, int *radii_z, hipStream_t s);
int launch_voxel_duplicate(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, int P, const int *radii_x,
                           const int *radii_y, const int *radii_z, hipStream_t s);
int launch_voxel_geom_backward(const VoxelGrid &v, int P, const int *radii_x, const int *radii_y, const int *radii_z,
                               const float *cov3D, const float *scales, const float *rotations, float scale_modifier,
                               const float *dL_dconic3D, const float *dL_dmean3D_norm, float *dL_dmean3D,
                               float *dL_dcov3D, float *dL_dscale, float *dL_drot, hipStream_t s);
int launch_voxel_render_forward(const VoxelGeom &g, const VoxelBinning &b, const VoxelImage &im, const VoxelGrid &v,
                                float *out_volume, bool write_ncontrib, hipStream_t s);
int launch_voxel_render_backward(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, size_t R,
                                 const float *dL_dvol, float *dL_dmean3D_norm, float *dL_dconic3D, float *dL_dopacity,
                                 hipStream_t s);

}  // namespace r2
