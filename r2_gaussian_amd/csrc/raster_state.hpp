// raster_state.hpp -- private layout of the three opaque state buffers of the rasterizer
// (the reference's GeometryState / BinningState / ImageState, RAS/rasterizer_impl.h:33-67,
// RAS/rasterizer_impl.cu:155-192).  The layout is ours: 32-byte packed render records instead of
// three separate arrays, and tile ranges sized per TILE, not per pixel (survey quirk Q7).
#pragma once
#include "r2_common.hpp"

namespace r2 {

struct RasterGeom {
    float4 *rec;              // [2P]  {px, py, A2, B2} {C2, op*mu, op, mu}   (A2,B2,C2: conic * -log2e/2, -log2e, -log2e/2)
    float *depths;            // [P]   view-space z (sort key low word)
    float *cov3D;             // [6P]
    uint32_t *tiles_touched;  // [P]
    uint32_t *offsets;        // [P]   inclusive scan of tiles_touched
    char *scan_temp;
    size_t scan_bytes;
    size_t bytes;
    static RasterGeom carve(char *chunk, int P)
    {
        RasterGeom g;
        Bump b(chunk);
        g.rec = b.take<float4>(2 * (size_t)P);
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

struct RasterBinning {
    uint64_t *keys_unsorted;  // [R]
    uint64_t *keys;           // [R]
    uint32_t *vals_unsorted;  // [R]
    uint32_t *point_list;     // [R]
    char *sort_temp;
    size_t sort_bytes;
    size_t bytes;
    static RasterBinning carve(char *chunk, size_t R)
    {
        RasterBinning s;
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

struct RasterImage {
    uint2 *ranges;        // [T]
    uint32_t *n_contrib;  // [N]  last contributing list position per pixel; only written in debug mode
    size_t bytes;
    static RasterImage carve(char *chunk, size_t T, size_t N)
    {
        RasterImage s;
        Bump b(chunk);
        s.ranges = b.take<uint2>(T);
        s.n_contrib = b.take<uint32_t>(N);
        s.bytes = b.total();
        return s;
    }
};

// launchers (raster_geom.hip, raster_render.hip)
int launch_raster_preprocess(const RasterGeom &g, int P, const float *means3D, const float *scales, float scale_modifier,
                             const float *rotations, const float *opacities, const float *cov3D_precomp,
                             const float *view, const float *proj, int W, int H, float tan_fovx, float tan_fovy,
                             int mode, int *radii, hipStream_t s);
int launch_raster_duplicate(const RasterGeom &g, const RasterBinning &b, int P, const int *radii, int W, int H,
                            hipStream_t s);
int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s);
int launch_raster_geom_backward(int P, const float *means3D, const int *radii, const float *cov3D, const float *scales,
                                const float *rotations, float scale_modifier, int W, int H, float tan_fovx,
                                float tan_fovy, const float *view, const float *proj, const float *dL_dconic,
                                const float *dL_dmu, const float *dL_dmean2D, float *dL_dmean3D, float *dL_dcov3D,
                                float *dL_dscale, float *dL_drot, int mode, hipStream_t s);
int launch_raster_render_forward(const RasterGeom &g, const RasterBinning &b, const RasterImage &im, int W, int H,
                                 float *out_color, bool write_ncontrib, hipStream_t s);
int launch_raster_render_backward(const RasterGeom &g, const RasterBinning &b, const RasterImage &im, int W, int H,
                                  size_t R, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                                  float *dL_dopacity, float *dL_dmu, hipStream_t s);

}  // namespace r2
