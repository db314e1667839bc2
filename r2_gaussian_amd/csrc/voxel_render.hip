// voxel_render.hip -- per-tile evaluation of the Gaussian mixture on the voxel grid and its backward.
//
// Reference: renderCUDA forward VOX/forward.cu:183-315, renderCUDA backward VOX/backward.cu:216-374.
// VALU/exp-bound (512 voxel-Gaussian pairs per 48 bytes gathered): FMA contraction ON, tolerance-checked.
// The reference evaluates `power` partly in double (quirk Q6); here it is a float FMA chain on the
// log2e-pre-scaled inverse covariance -- the difference is ~1e-7 relative, inside the 1e-4 budget.
//
// Forward : one workgroup = one 8x8x8 tile = 8 waves.  Lanes are mapped z-fastest (lane = y*8+z, wave = x)
//           so each wave writes eight 32-byte runs of the [nx,ny,nz] volume instead of 64 scattered words
//           (the reference's x-fastest thread order strides by ny*nz floats between lanes).
//           The tile list is staged through LDS in 512-record batches of 48-byte packed records.
// Backward: loop nest inverted as in raster_render.hip: one LANE owns one (tile, Gaussian) instance and walks
//           the 512 voxels of its tile; dL/dvol is wave-uniform (scalar loads of 8 contiguous z values).
//           The 10 gradient terms of the reference are linear in 10 moments of w = G*dL: sum w, sum w d_i,
//           sum w d_i d_j; 10 atomics per INSTANCE instead of 10 per contributing pair.
#include "voxel_state.hpp"

namespace r2 {

constexpr float ALPHA_MIN_3D = 0.000001f;   // VOX/forward.cu:293

template <bool NCONTRIB>
__global__ void __launch_bounds__(512) voxel_render_forward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
    VoxelGrid v, uint32_t T, float *__restrict__ out, uint32_t *__restrict__ n_contrib)
{
    const uint32_t tile = xcd_remap(blockIdx.x, T);
    if (tile >= T) return;
    const int tx = tile % v.gx, ty = (tile / v.gx) % v.gy, tz = tile / (v.gx * v.gy);
    const int tid = threadIdx.x;
    const int vx = tx * TILE3D + (tid >> 6), vy = ty * TILE3D + ((tid >> 3) & 7), vz = tz * TILE3D + (tid & 7);
    const bool inside = vx < v.nx && vy < v.ny && vz < v.nz;
    const float fx = (float)vx + 0.5f, fy = (float)vy + 0.5f, fz = (float)vz + 0.5f;
    const uint2 range = ranges[tile];

    __shared__ float4 s0[512];
    __shared__ float4 s1[512];
    __shared__ float2 s2[512];

    float C = 0.f;
    uint32_t last = 0;
    for (uint32_t base = range.x; base < range.y; base += 512) {
        __syncthreads();
        const uint32_t k = base + tid;
        if (k < range.y) {
            const uint32_t id = point_list[k];
            s0[tid] = rec[3 * id];
            s1[tid] = rec[3 * id + 1];
            const float4 c = rec[3 * id + 2];
            s2[tid] = make_float2(c.x, c.y);
        }
        __syncthreads();
        const int n = min(512u, range.y - base);
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
    if (inside) {
        const size_t vid = ((size_t)vx * v.ny + vy) * v.nz + vz;
        out[vid] = C;
        if (NCONTRIB) n_contrib[vid] = last;
    }
}

template <bool FULLZ>
__device__ __forceinline__ void tile_moments3(const float4 p, const float4 q, const float4 r, const float *__restrict__ dL,
                                              const VoxelGrid &v, int x0, int y0, int z0, int ncx, int ncy, int ncz,
                                              float *S)
{
    const float dz0 = p.z - ((float)z0 + 0.5f);
    for (int ix = 0; ix < ncx; ++ix) {
        const float dx = p.x - ((float)(x0 + ix) + 0.5f);
        for (int iy = 0; iy < ncy; ++iy) {
            const float dy = p.y - ((float)(y0 + iy) + 0.5f);
            const float k0 = dx * (q.x * dx + q.y * dy) + (q.w * dy) * dy;   // a2 dx^2 + b2 dx dy + d2 dy^2
            const float k1 = q.z * dx + r.x * dy;                            // c2 dx + e2 dy
            const float *__restrict__ row = dL + ((size_t)(x0 + ix) * v.ny + (y0 + iy)) * v.nz + z0;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int iz = 0; iz < TILE3D; ++iz) {
                float g;   // wave-uniform address -> scalar load
                if (FULLZ) g = row[iz];
                else g = (iz < ncz) ? row[min(iz, ncz - 1)] : 0.f;
                const float dz = dz0 - (float)iz;
                const float p2 = dz * (k1 + r.y * dz) + k0;
                const float G = __builtin_amdgcn_exp2f(p2);
                const bool ok = (p2 <= 0.0f) && (p.w * G >= ALPHA_MIN_3D);
                const float w = ok ? G * g : 0.f;
                const float wdz = w * dz;
                r0 += w;
                r1 += wdz;
                r2 += wdz * dz;
            }
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
    }
}

__global__ void __launch_bounds__(256) voxel_render_backward_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
    uint32_t R, VoxelGrid v, uint32_t nchunks, const float *__restrict__ dL_dvol, float *__restrict__ dL_dmean3D_norm,
    float *__restrict__ dL_dconic3D, float *__restrict__ dL_dopacity)
{
    const uint32_t chunk = xcd_remap(blockIdx.x, nchunks);
    if (chunk >= nchunks) return;
    const uint32_t k = chunk * 256u + threadIdx.x;
    const bool live = k < R;
    uint32_t tile = 0xffffffffu, id = 0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p, r = p;
    if (live) {
        tile = (uint32_t)(keys[k] >> 32);
        id = point_list[k];
        p = rec[3 * id];
        q = rec[3 * id + 1];
        r = rec[3 * id + 2];
    }
    float S[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) S[i] = 0.f;

    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t t = __builtin_amdgcn_readfirstlane(__shfl(tile, leader));
        const bool mine = live && tile == t;
        todo &= ~__ballot(mine);
        const int tx = t % v.gx, ty = (t / v.gx) % v.gy, tz = t / (v.gx * v.gy);
        const int x0 = tx * TILE3D, y0 = ty * TILE3D, z0 = tz * TILE3D;
        const int ncx = min(TILE3D, v.nx - x0), ncy = min(TILE3D, v.ny - y0), ncz = min(TILE3D, v.nz - z0);
        if (mine) {
            if (ncz == TILE3D) tile_moments3<true>(p, q, r, dL_dvol, v, x0, y0, z0, ncx, ncy, ncz, S);
            else tile_moments3<false>(p, q, r, dL_dvol, v, x0, y0, z0, ncx, ncy, ncz, S);
        }
    }
    if (live) {
        const float dvx = v.sx / (float)v.nx, dvy = v.sy / (float)v.ny, dvz = v.sz / (float)v.nz;
        const float opa = p.w;
        // undo the log2e pre-scale of the inverse covariance
        const float a = q.x * (-2.0f * LN2), b = q.y * (-LN2), c = q.z * (-LN2);
        const float d = q.w * (-2.0f * LN2), e = r.x * (-LN2), f = r.y * (-2.0f * LN2);
        // mean gradient is scaled by dVoxel exactly as the reference does (quirk Q4, VOX/backward.cu:359-361)
        unsafeAtomicAdd(&dL_dmean3D_norm[3 * id + 0], opa * dvx * (-a * S[1] - b * S[2] - c * S[3]));
        unsafeAtomicAdd(&dL_dmean3D_norm[3 * id + 1], opa * dvy * (-d * S[2] - b * S[1] - e * S[3]));
        unsafeAtomicAdd(&dL_dmean3D_norm[3 * id + 2], opa * dvz * (-f * S[3] - c * S[1] - e * S[2]));
        unsafeAtomicAdd(&dL_dconic3D[6 * id + 0], -0.5f * opa * S[4]);
        unsafeAtomicAdd(&dL_dconic3D[6 * id + 1], -opa * S[5]);
        unsafeAtomicAdd(&dL_dconic3D[6 * id + 2], -opa * S[6]);
        unsafeAtomicAdd(&dL_dconic3D[6 * id + 3], -0.5f * opa * S[7]);
        unsafeAtomicAdd(&dL_dconic3D[6 * id + 4], -opa * S[8]);
        unsafeAtomicAdd(&dL_dconic3D[6 * id + 5], -0.5f * opa * S[9]);
        unsafeAtomicAdd(&dL_dopacity[id], S[0]);
    }
}

int launch_voxel_render_forward(const VoxelGeom &g, const VoxelBinning &b, const VoxelImage &im, const VoxelGrid &v,
                                float *out_volume, bool write_ncontrib, hipStream_t s)
{
    const uint32_t T = (uint32_t)v.gx * v.gy * v.gz;
    const uint32_t grid = ((T + 7u) >> 3) << 3;
    if (write_ncontrib)
        voxel_render_forward_kernel<true><<<dim3(grid), dim3(512), 0, s>>>(im.ranges, b.point_list, g.rec, v, T,
                                                                           out_volume, im.n_contrib);
    else
        voxel_render_forward_kernel<false><<<dim3(grid), dim3(512), 0, s>>>(im.ranges, b.point_list, g.rec, v, T,
                                                                            out_volume, im.n_contrib);
    return 0;
}

int launch_voxel_render_backward(const VoxelGeom &g, const VoxelBinning &b, const VoxelGrid &v, size_t R,
                                 const float *dL_dvol, float *dL_dmean3D_norm, float *dL_dconic3D, float *dL_dopacity,
                                 hipStream_t s)
{
    if (R == 0) return 0;
    const uint32_t nchunks = (uint32_t)((R + 255) / 256);
    const uint32_t grid = ((nchunks + 7u) >> 3) << 3;
    voxel_render_backward_kernel<<<dim3(grid), dim3(256), 0, s>>>(b.keys, b.point_list, g.rec, (uint32_t)R, v, nchunks,
                                                                  dL_dvol, dL_dmean3D_norm, dL_dconic3D, dL_dopacity);
    return 0;
}

}  // namespace r2
