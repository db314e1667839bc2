// voxel_render.hip -- per-tile evaluation of the Gaussian mixture on the voxel grid and its backward.
//
// Reference: renderCUDA forward VOX/forward.cu:183-315, renderCUDA backward VOX/backward.cu:216-374.
// VALU/exp-bound (512 voxel-Gaussian pairs per 48 bytes gathered): FMA contraction ON, tolerance-checked.
// The reference evaluates `power` partly in double (quirk Q6); here it is a float FMA chain on the
// log2e-pre-scaled inverse covariance -- the difference is ~1e-7 relative, inside the 1e-4 budget.
//
// Forward : a tile's list is cut into work items of VOX_CHUNK instances; one workgroup = one work item =
//           8 waves over the 8x8x8 tile.  Lanes are mapped z-fastest (lane = y*8+z, wave = x) so the volume is
//           written in 32-byte runs of [nx,ny,nz] (the reference's x-fastest thread order strides by ny*nz
//           floats between lanes).  Records (48 bytes packed) are staged through LDS in 512-record batches;
//           partial sums per work item are added in list order by a second kernel (deterministic volume).
// Backward: loop nest inverted as in raster_render.hip: one LANE owns one (tile, Gaussian) instance and walks
//           the 512 voxels of its tile; dL/dvol of the tile is staged once per wave in LDS.  The 10 gradient
//           sums of the reference are linear in 10 moments of w = G*dL: sum w, sum w d_i, sum w d_i d_j.
//           No atomics: each instance stores its moment row at its emission index, the geometry backward
//           reduces each Gaussian's contiguous run in a fixed order (bit-reproducible gradients).
#include "voxel_state.hpp"

namespace r2 {

constexpr float ALPHA_MIN_3D = 0.000001f;   // VOX/forward.cu:293

template <bool NCONTRIB>
__global__ void __launch_bounds__(512) voxel_render_forward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ chunk_base, const uint32_t *__restrict__ work_tile,
    uint32_t T, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, VoxelGrid v,
    float *__restrict__ partial, uint32_t *__restrict__ partial_last)
{
    const uint32_t w = blockIdx.x;
    if (w >= chunk_base[T]) return;
    const uint32_t tile = work_tile[w];
    const uint32_t j0 = (w - chunk_base[tile]) * VOX_CHUNK;
    const uint2 range = ranges[tile];
    const uint32_t beg = range.x + j0, end = min(range.y, beg + VOX_CHUNK);
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const int tid = threadIdx.x;
    // lane = y*8+z, wave = x: z-fastest, see the header comment
    const float fx = (float)(tx * TILE3D + (tid >> 6)) + 0.5f, fy = (float)(ty * TILE3D + ((tid >> 3) & 7)) + 0.5f,
                fz = (float)(tz * TILE3D + (tid & 7)) + 0.5f;

    __shared__ float4 s0[512];
    __shared__ float4 s1[512];
    __shared__ float2 s2[512];

    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t base = beg; base < end; base += 512) {
        __syncthreads();
        const uint32_t k = base + tid;
        if (k < end) {
            const uint32_t id = point_list[k];
            s0[tid] = rec[3 * id];
            s1[tid] = rec[3 * id + 1];
            s2[tid] = *reinterpret_cast<const float2 *>(&rec[3 * id + 2]);
        }
        __syncthreads();
        const int n = min(512u, end - base);
#pragma unroll 2
        for (int j = 0; j < n; ++j) {
            const float4 p = s0[j];    // x y z opacity
            const float4 q = s1[j];    // a2 b2 c2 d2
            const float2 r = s2[j];    // e2 f2
            const float dx = p.x - fx, dy = p.y - fy, dz = p.z - fz;
            const float p2 = dx * (q.x * dx + q.y * dy + q.z * dz) + dy * (q.w * dy + r.x * dz) + (r.y * dz) * dz;
            const float alpha = p.w * __builtin_amdgcn_exp2f(p2);
            const bool ok = (p2 <= 0.0f) && (alpha >= ALPHA_MIN_3D);
            C += ok ? alpha : 0.f;
            if (NCONTRIB) last = ok ? (base - range.x) + (uint32_t)j + 1u : last;
        }
    }
    partial[(size_t)w * 512 + tid] = C;
    if (NCONTRIB) partial_last[(size_t)w * 512 + tid] = last;
}

// adds a tile's partial sums in list order and writes the volume (zeros for empty tiles)
template <bool NCONTRIB>
__global__ void __launch_bounds__(512) voxel_combine_kernel(
    const uint32_t *__restrict__ chunk_base, const float *__restrict__ partial,
    const uint32_t *__restrict__ partial_last, VoxelGrid v, float *__restrict__ out, uint32_t *__restrict__ n_contrib)
{
    const uint32_t tile = blockIdx.x;
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const int tid = threadIdx.x;
    const int vx = tx * TILE3D + (tid >> 6), vy = ty * TILE3D + ((tid >> 3) & 7), vz = tz * TILE3D + (tid & 7);
    const uint32_t w0 = chunk_base[tile], w1 = chunk_base[tile + 1];
    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t w = w0; w < w1; ++w) {
        C += partial[(size_t)w * 512 + tid];
        if (NCONTRIB) {
            const uint32_t l = partial_last[(size_t)w * 512 + tid];
            last = l ? l : last;
        }
    }
    if (vx < v.nx && vy < v.ny && vz < v.nz) {
        const size_t vid = ((size_t)vx * v.ny + vy) * v.nz + vz;
        out[vid] = C;
        if (NCONTRIB) n_contrib[vid] = last;
    }
}

__device__ __forceinline__ void voxel_moments(const float4 p, float ry2, float dz, float k0, float k1, float g,
                                              float &r0, float &r1, float &r2)
{
    const float p2 = dz * (k1 + ry2 * dz) + k0;
    const float G = __builtin_amdgcn_exp2f(p2);
    const bool ok = (p2 <= 0.0f) && (p.w * G >= ALPHA_MIN_3D);
    const float w = ok ? G * g : 0.f;
    const float wdz = w * dz;
    r0 += w;
    r1 += wdz;
    r2 += wdz * dz;
}

__device__ __forceinline__ void row_to_moments(float dx, float dy, float r0, float r1, float r2, float *S)
{
    const float wx = dx * r0, wy = dy * r0;
    S[0] += r0;
    S[1] += wx;         // sum w dx
    S[2] += wy;         // sum w dy
    S[3] += r1;         // sum w dz
    S[4] += dx * wx;    // xx
    S[5] += dx * wy;    // xy
    S[6] += dx * r1;    // xz
    S[7] += dy * wy;    // yy
    S[8] += dy * r1;    // yz
    S[9] += r2;         // zz
}

// moments over one 8x8x8 tile whose dL/dvol block sits in this wave's LDS slab (zeros outside the volume):
// 128 float4, index (x*8+y)*2 + z/4; every read is a wave-uniform broadcast.
__device__ __forceinline__ void tile_moments3_uniform(const float4 p, const float4 q, const float4 r,
                                                      const float4 *__restrict__ gt, int x0, int y0, int z0, float *S)
{
    const float dz0 = p.z - ((float)z0 + 0.5f);
    for (int ix = 0; ix < TILE3D; ++ix) {
        const float dx = p.x - ((float)(x0 + ix) + 0.5f);
#pragma unroll 2
        for (int iy = 0; iy < TILE3D; ++iy) {
            const float dy = p.y - ((float)(y0 + iy) + 0.5f);
            const float k0 = dx * (q.x * dx + q.y * dy) + (q.w * dy) * dy;   // a2 dx^2 + b2 dx dy + d2 dy^2
            const float k1 = q.z * dx + r.x * dy;                            // c2 dx + e2 dy
            const float4 g0 = gt[(ix * 8 + iy) * 2], g1 = gt[(ix * 8 + iy) * 2 + 1];
            const float g[8] = { g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w };
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int iz = 0; iz < TILE3D; ++iz) voxel_moments(p, r.y, dz0 - (float)iz, k0, k1, g[iz], r0, r1, r2);
            row_to_moments(dx, dy, r0, r1, r2, S);
        }
    }
}

// lanes of the wave sit in many different sparse tiles: each lane gathers dL/dvol of its own tile
__device__ __forceinline__ void tile_moments3_gather(const float4 p, const float4 q, const float4 r,
                                                     const float *__restrict__ dL, const VoxelGrid &v, int x0, int y0,
                                                     int z0, float *S)
{
    const float dz0 = p.z - ((float)z0 + 0.5f);
    const int ncx = min(TILE3D, v.nx - x0), ncy = min(TILE3D, v.ny - y0), ncz = min(TILE3D, v.nz - z0);
    for (int ix = 0; ix < ncx; ++ix) {
        const float dx = p.x - ((float)(x0 + ix) + 0.5f);
        for (int iy = 0; iy < ncy; ++iy) {
            const float dy = p.y - ((float)(y0 + iy) + 0.5f);
            const float k0 = dx * (q.x * dx + q.y * dy) + (q.w * dy) * dy;
            const float k1 = q.z * dx + r.x * dy;
            const float *__restrict__ row = dL + ((size_t)(x0 + ix) * v.ny + (y0 + iy)) * v.nz + z0;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
            for (int iz = 0; iz < ncz; ++iz) voxel_moments(p, r.y, dz0 - (float)iz, k0, k1, row[iz], r0, r1, r2);
            row_to_moments(dx, dy, r0, r1, r2, S);
        }
    }
}

__global__ void __launch_bounds__(256) voxel_render_backward_kernel(
    const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ point_list,
    const float4 *__restrict__ rec, uint32_t R, VoxelGrid v, uint32_t nchunks, const float *__restrict__ dL_dvol,
    float4 *__restrict__ part)
{
    __shared__ float4 gtile[4][128];   // one 8x8x8 dL/dvol block per wave
    const uint32_t chunk = xcd_remap(blockIdx.x, nchunks);
    if (chunk >= nchunks) return;
    const uint32_t k = chunk * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = k < R;
    uint32_t tile = 0xffffffffu, id = 0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p, r = p;
    if (live) {
        tile = tiles[k];
        id = point_list[k];
        p = rec[3 * id];
        q = rec[3 * id + 1];
        r = rec[3 * id + 2];
    }
    float S[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) S[i] = 0.f;
    float4 *gt = gtile[wave];
    const uint32_t gxy = (uint32_t)(v.gx * v.gy);

    const uint32_t prev_tile = __shfl_up(tile, 1);
    const unsigned long long heads = __ballot(live && (lane == 0 || tile != prev_tile));
    if (__popcll(heads) > 3) {
        if (live)
            tile_moments3_gather(p, q, r, dL_dvol, v, (int)(tile % v.gx) * TILE3D, (int)((tile / v.gx) % v.gy) * TILE3D,
                                 (int)(tile / gxy) * TILE3D, S);
    } else {
        unsigned long long todo = __ballot(live);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t t = __builtin_amdgcn_readfirstlane(__shfl(tile, leader));
            const bool mine = live && tile == t;
            todo &= ~__ballot(mine);
            const int x0 = (int)(t % v.gx) * TILE3D, y0 = (int)((t / v.gx) % v.gy) * TILE3D, z0 = (int)(t / gxy) * TILE3D;
            {   // stage the tile's dL/dvol: lane -> (x = lane/8, y = lane%8), its 8 z values as two float4
                const int vx = x0 + (lane >> 3), vy = y0 + (lane & 7);
                float g[8];
#pragma unroll
                for (int iz = 0; iz < 8; ++iz) g[iz] = 0.f;
                if (vx < v.nx && vy < v.ny) {
                    const float *__restrict__ src = dL_dvol + ((size_t)vx * v.ny + vy) * v.nz + z0;
                    if (z0 + 7 < v.nz && (v.nz & 3) == 0) {
                        const float4 a = *reinterpret_cast<const float4 *>(src);
                        const float4 b = *reinterpret_cast<const float4 *>(src + 4);
                        g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
                    } else {
#pragma unroll
                        for (int iz = 0; iz < 8; ++iz)
                            if (z0 + iz < v.nz) g[iz] = src[iz];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                gt[lane * 2] = make_float4(g[0], g[1], g[2], g[3]);
                gt[lane * 2 + 1] = make_float4(g[4], g[5], g[6], g[7]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (mine) tile_moments3_uniform(p, q, r, gt, x0, y0, z0, S);
        }
    }
    if (live) {
        // scratch row = sorted position (coalesced); the geometry backward gathers through the inverse permutation
        part[3 * (size_t)k] = make_float4(S[0], S[1], S[2], S[3]);
        part[3 * (size_t)k + 1] = make_float4(S[4], S[5], S[6], S[7]);
        part[3 * (size_t)k + 2] = make_float4(S[8], S[9], 0.f, 0.f);
    }
}

int launch_voxel_render_forward(const VoxelGeom &g, const VoxelBinning &b, const VoxelImage &im, const VoxelGrid &v,
                                float *out_volume, bool write_ncontrib, hipStream_t s)
{
    const uint32_t T = (uint32_t)v.gx * v.gy * v.gz;
    if (im.NW > 0) {
        if (write_ncontrib)
            voxel_render_forward_kernel<true><<<dim3((unsigned)im.NW), dim3(512), 0, s>>>(
                im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, v, im.partial, im.partial_last);
        else
            voxel_render_forward_kernel<false><<<dim3((unsigned)im.NW), dim3(512), 0, s>>>(
                im.ranges, im.chunk_base, im.work_tile, T, b.point_list, g.rec, v, im.partial, im.partial_last);
    }
    if (write_ncontrib)
        voxel_combine_kernel<true><<<dim3(T), dim3(512), 0, s>>>(im.chunk_base, im.partial, im.partial_last, v,
                                                                 out_volume, im.n_contrib);
    else
        voxel_combine_kernel<false><<<dim3(T), dim3(512), 0, s>>>(im.chunk_base, im.partial, im.partial_last, v,
                                                                  out_volume, im.n_contrib);
    return 0;
}

int launch_voxel_render_backward(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, size_t R,
                                 const float *dL_dvol, hipStream_t s)
{
    if (R == 0) return 0;
    const uint32_t nchunks = (uint32_t)((R + 255) / 256);
    const uint32_t grid = ((nchunks + 7u) >> 3) << 3;
    voxel_render_backward_kernel<<<dim3(grid), dim3(256), 0, s>>>(b.tiles, b.point_list, g.rec, (uint32_t)R, v,
                                                                  nchunks, dL_dvol, reinterpret_cast<float4 *>(b.part));
    return 0;
}

}  // namespace r2
