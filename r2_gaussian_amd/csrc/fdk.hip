// fdk.hip -- cone-beam FDK reconstruction: the volume the reference thresholds and samples to initialise its
// Gaussians (initialize_pcd.py:36-90 -> r2_gaussian/utils/ct_utils.py:17-27 -> tigre.algorithms.fdk, a third-party CUDA
// toolbox that is not part of the reference tree).  Two kernels:
//
//   fdk_filter_kernel       cosine pre-weight + ramp filter along detector rows, as a direct convolution with the 2W-1
//                           spatial taps of the (windowed) ramp -- identical to the zero-padded FFT product TIGRE forms,
//                           because that circular convolution never wraps (offsets < W <= L/2).  W^2 FMAs per row:
//                           VALU-bound, register-tiled 8 outputs x 8 inputs per step out of LDS (64 FMAs per 6 ds_read_b128).
//                           The result is stored TRANSPOSED, [view][u][v]: the scan axis is z, so a column of voxels walks
//                           along v and the back-projection's gathers become runs.
//   fdk_backproject_kernel  voxel-driven: lane = z, 4 y per thread, loop over the views with the view's matrix in scalar
//                           registers; bilinear detector interpolation with a zero border, FDK distance weight (DSO/U)^2;
//                           fixed view order (bit-reproducible), the volume is written once, coalesced along z.
//
// Voxels are projected with the SAME 4x4 matrices the rasterizer renders with (full_proj_transform, row-vector convention),
// so the reconstruction is registered to the rasterizer / voxelizer by construction: voxel (i,j,k) has its centre at
// center - sVoxel/2 + (i+0.5, j+0.5, k+0.5) * dVoxel, the voxelizer's convention (VOX/forward.cu:145-147,206), and a pixel is
// ndc2Pix of the projected point (RAS/auxiliary.h:45-48).
#include "r2_common.hpp"
#include <math.h>

namespace r2 {

namespace {

constexpr int FB = 256;   // threads per block, both kernels

// ------------------------------------------------------------------------------------------------------------ filter
// LDS: rows[RB][Wp + 4] (pre-weighted input rows, stride padded so that 16 rows hit 64 distinct banks) | T[Wp + W + PAD]
// (taps, T[x + PAD] = taps[x]); PAD makes the first tap of every 8x8 step 16-byte aligned.
template <int RB>
__global__ void __launch_bounds__(FB) fdk_filter_kernel(int H, int W, int Wp, int PAD, const float *__restrict__ projs,
                                                        const float *__restrict__ taps, float scale, int cone, float DSD,
                                                        float du, float dv, float *__restrict__ out_t /* [V][W][H] */)
{
    extern __shared__ float4 lds4[];
    float *rows = reinterpret_cast<float *>(lds4);
    const int RS = Wp + 4;
    float *T = rows + RB * RS;
    const int view = blockIdx.y, v0 = blockIdx.x * RB, tid = threadIdx.x;
    const float *src = projs + (size_t)view * H * W;
    for (int i = tid; i < RB * RS; i += FB) {
        const int r = i / RS, c = i - r * RS, v = v0 + r;
        float x = 0.f;
        if (c < W && v < H) {
            x = src[(size_t)v * W + c];
            if (cone) {
                const float uu = ((float)c + 0.5f - 0.5f * (float)W) * du, vv = ((float)v + 0.5f - 0.5f * (float)H) * dv;
                x *= DSD / sqrtf(DSD * DSD + uu * uu + vv * vv);
            }
        }
        rows[i] = x;
    }
    const int TN = Wp + W + PAD;
    for (int i = tid; i < TN; i += FB) {
        const int x = i - PAD;
        T[i] = (x >= 0 && x < 2 * W - 1) ? taps[x] * scale : 0.f;
    }
    __syncthreads();
    const int groups = RB * (Wp / 8);
    float *dst = out_t + (size_t)view * W * H;
    for (int g = tid; g < groups; g += FB) {
        const int r = g % RB, i0 = (g / RB) * 8;
        const float *row = rows + r * RS;
        const float *tb = T + (i0 + W - 8 + PAD);   // tap of (output i0+q, input j0+jj) = tb[-j0 + q - jj + 7]
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        for (int j0 = 0; j0 < Wp; j0 += 8) {
            float x[8], t[16];
            const float4 xa = *reinterpret_cast<const float4 *>(row + j0), xb = *reinterpret_cast<const float4 *>(row + j0 + 4);
            x[0] = xa.x; x[1] = xa.y; x[2] = xa.z; x[3] = xa.w; x[4] = xb.x; x[5] = xb.y; x[6] = xb.z; x[7] = xb.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 tt = *reinterpret_cast<const float4 *>(tb - j0 + 4 * k);
                t[4 * k] = tt.x; t[4 * k + 1] = tt.y; t[4 * k + 2] = tt.z; t[4 * k + 3] = tt.w;
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(x[jj], t[q - jj + 7], acc[q]);
        }
        const int v = v0 + r;
        if (v < H)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + q < W) dst[(size_t)(i0 + q) * H + v] = acc[q];
    }
}

// ------------------------------------------------------------------------------------------------------ back-projection
constexpr int BY = 4;   // y values per thread

__device__ __forceinline__ float tap_t(const float *__restrict__ q, int u, int v, int W, int H)
{
    const bool ok = (u >= 0) & (u < W) & (v >= 0) & (v < H);
    const int uc = min(max(u, 0), W - 1), vc = min(max(v, 0), H - 1);
    const float x = q[(size_t)uc * H + vc];
    return ok ? x : 0.f;
}

__global__ void __launch_bounds__(FB) fdk_backproject_kernel(int V, int H, int W, const float *__restrict__ filt_t,
                                                             const float *__restrict__ mats, int cone, float DSO, int nx,
                                                             int ny, int nz, float3 d, float3 o /* centre of voxel 0 */,
                                                             float *__restrict__ vol)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int z = blockIdx.x * 64 + lane;
    const int y0 = (blockIdx.y * (FB / 64) + wave) * BY;
    const int x = blockIdx.z;
    if (y0 >= ny) return;
    const float X = o.x + (float)x * d.x, Z = o.z + (float)min(z, nz - 1) * d.z;
    float acc[BY];
#pragma unroll
    for (int k = 0; k < BY; ++k) acc[k] = 0.f;
    const float hW = 0.5f * (float)W, hH = 0.5f * (float)H;
    for (int v = 0; v < V; ++v) {
        const float *M = mats + 16 * v;
        const float *q = filt_t + (size_t)v * W * H;
        // p_hom = [X, Y, Z, 1] M   (columns 0, 1, 3)
        const float bx = X * M[0] + Z * M[8] + M[12], by = X * M[1] + Z * M[9] + M[13], bw = X * M[3] + Z * M[11] + M[15];
#pragma unroll
        for (int k = 0; k < BY; ++k) {
            const float Y = o.y + (float)min(y0 + k, ny - 1) * d.y;
            const float hx = fmaf(Y, M[4], bx), hy = fmaf(Y, M[5], by), hw = fmaf(Y, M[7], bw);
            const float inv = 1.0f / (hw + 0.0000001f);
            const float fx = (hx * inv + 1.0f) * hW - 0.5f, fy = (hy * inv + 1.0f) * hH - 0.5f;   // ndc2Pix
            const float flx = floorf(fx), fly = floorf(fy);
            const float ax = fx - flx, ay = fy - fly;
            // clamp before the conversion: points far outside the detector (or behind the source) must not overflow an int
            const int u0 = (int)fminf(fmaxf(flx, -2.f), (float)W), v0 = (int)fminf(fmaxf(fly, -2.f), (float)H);
            const float s00 = tap_t(q, u0, v0, W, H), s01 = tap_t(q, u0 + 1, v0, W, H);
            const float s10 = tap_t(q, u0, v0 + 1, W, H), s11 = tap_t(q, u0 + 1, v0 + 1, W, H);
            const float s = (1.f - ay) * ((1.f - ax) * s00 + ax * s01) + ay * ((1.f - ax) * s10 + ax * s11);
            const float wi = DSO * inv;
            acc[k] += cone ? s * (wi * wi) : s;
        }
    }
    if (z < nz)
#pragma unroll
        for (int k = 0; k < BY; ++k)
            if (y0 + k < ny) vol[((size_t)x * ny + (y0 + k)) * nz + z] = acc[k];
}

}  // namespace

}  // namespace r2

extern "C" int r2_fdk_filter(int V, int H, int W, const float *projs, const float *taps, float scale, int cone, float DSD,
                             float du, float dv, float *filtered_t, void *stream)
{
    using namespace r2;
    if (V <= 0 || H <= 0 || W <= 0 || !projs || !taps || !filtered_t || !(DSD > 0.f) || !(du > 0.f) || !(dv > 0.f)) {
        set_error("r2_fdk_filter: invalid argument");
        return R2_ERR_INVALID;
    }
    if (W > 4096) {
        set_error("r2_fdk_filter: detector rows longer than 4096 pixels are not supported (W = %d)", W);
        return R2_ERR_INVALID;
    }
    const int Wp = (W + 7) & ~7, PAD = 8 + ((4 - (W & 3)) & 3);
    const int RB = W <= 1024 ? 16 : (W <= 2048 ? 8 : 4);
    const size_t lds = ((size_t)RB * (Wp + 4) + (size_t)(Wp + W + PAD)) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((H + RB - 1) / RB, V);
    if (RB == 16) {
        R2_HIP_TRY(hipFuncSetAttribute((const void *)fdk_filter_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fdk_filter_kernel<16><<<grid, dim3(FB), lds, s>>>(H, W, Wp, PAD, projs, taps, scale, cone, DSD, du, dv, filtered_t);
    } else if (RB == 8) {
        R2_HIP_TRY(hipFuncSetAttribute((const void *)fdk_filter_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fdk_filter_kernel<8><<<grid, dim3(FB), lds, s>>>(H, W, Wp, PAD, projs, taps, scale, cone, DSD, du, dv, filtered_t);
    } else {
        R2_HIP_TRY(hipFuncSetAttribute((const void *)fdk_filter_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fdk_filter_kernel<4><<<grid, dim3(FB), lds, s>>>(H, W, Wp, PAD, projs, taps, scale, cone, DSD, du, dv, filtered_t);
    }
    R2_STAGE_CHECK(0, s, "fdk filter");
    return 0;
}

extern "C" int r2_fdk_backproject(int V, int H, int W, const float *filtered_t, const float *projmatrices, int cone, float DSO,
                                  int nx, int ny, int nz, float sx, float sy, float sz, float cx, float cy, float cz, float *vol,
                                  void *stream)
{
    using namespace r2;
    if (V <= 0 || H <= 0 || W <= 0 || nx <= 0 || ny <= 0 || nz <= 0 || !filtered_t || !projmatrices || !vol) {
        set_error("r2_fdk_backproject: invalid argument");
        return R2_ERR_INVALID;
    }
    if (nx > 65535 || (ny + 15) / 16 > 65535) {
        set_error("r2_fdk_backproject: volume too large for one launch (%d x %d x %d)", nx, ny, nz);
        return R2_ERR_INVALID;
    }
    const float3 d = make_float3(sx / (float)nx, sy / (float)ny, sz / (float)nz);
    const float3 o = make_float3(cx - 0.5f * sx + 0.5f * d.x, cy - 0.5f * sy + 0.5f * d.y, cz - 0.5f * sz + 0.5f * d.z);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((nz + 63) / 64, (ny + (FB / 64) * BY - 1) / ((FB / 64) * BY), nx);
    fdk_backproject_kernel<<<grid, dim3(FB), 0, s>>>(V, H, W, filtered_t, projmatrices, cone, DSO, nx, ny, nz, d, o, vol);
    R2_STAGE_CHECK(0, s, "fdk backproject");
    return 0;
}
