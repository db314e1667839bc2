// r2_math.hpp -- device-side per-Gaussian algebra shared by the rasterizer and voxelizer geometry kernels.
//
// Every value computed here can feed an integer decision (radius, tile rectangle, sort key), so the
// translation units that include this header are compiled with floating-point contraction OFF: each
// operation is one correctly rounded IEEE-754 binary32 op, evaluated in the order the reference's
// glm expressions imply (column-major mat3, products summed left to right).  hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt keeps '/' and sqrtf correctly rounded.
#pragma once
#include "r2_common.hpp"

namespace r2 {

// column-major 3x3, m[c][r] -- the storage convention of the reference's matrix type, so that the
// index patterns of RAS/backward.cu:258-300 can be followed one to one.
struct M3 {
    float m[3][3];
};

__device__ __forceinline__ M3 m3(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    M3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}
__device__ __forceinline__ M3 mul(const M3 &A, const M3 &B)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            r.m[c][k] = A.m[0][k] * B.m[c][0] + A.m[1][k] * B.m[c][1] + A.m[2][k] * B.m[c][2];
    return r;
}
__device__ __forceinline__ M3 tr(const M3 &A)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) r.m[c][k] = A.m[k][c];
    return r;
}

__device__ __forceinline__ M3 quat_to_rot(float r, float x, float y, float z)
{
    return m3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
              2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
              2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

// Sigma = (S R)^T (S R), packed (xx,xy,xz,yy,yz,zz).  RAS/forward.cu:161-195 == VOX/forward.cu:21-55.
__device__ __forceinline__ void cov3d_from_scale_rot(float sx, float sy, float sz, float mod, float4 q, float *cov)
{
    M3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * sx;
    S.m[1][1] = mod * sy;
    S.m[2][2] = mod * sz;
    const M3 R = quat_to_rot(q.x, q.y, q.z, q.w);
    const M3 M = mul(S, R);
    const M3 Sig = mul(tr(M), M);
    cov[0] = Sig.m[0][0];
    cov[1] = Sig.m[0][1];
    cov[2] = Sig.m[0][2];
    cov[3] = Sig.m[1][1];
    cov[4] = Sig.m[1][2];
    cov[5] = Sig.m[2][2];
}

// d(Sigma) -> d(scale), d(quaternion).  RAS/backward.cu:334-397 == VOX/backward.cu:21-84.
__device__ __forceinline__ void cov3d_backward(float sx, float sy, float sz, float mod, float4 q, const float *dcov,
                                               float *dscale, float4 *drot)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const M3 R = quat_to_rot(r, x, y, z);
    M3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    const float s[3] = { mod * sx, mod * sy, mod * sz };
    S.m[0][0] = s[0];
    S.m[1][1] = s[1];
    S.m[2][2] = s[2];
    const M3 M = mul(S, R);
    const M3 dSig = m3(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                       0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
    M3 M2;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) M2.m[c][k] = 2.0f * M.m[c][k];
    const M3 dM = mul(M2, dSig);
    const M3 Rt = tr(R);
    M3 dMt = tr(dM);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        dscale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) dMt.m[k][j] *= s[k];
    float4 g;
    g.x = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
    g.y = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
    g.z = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
    g.w = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
    *drot = g;   // gradient w.r.t. the quaternion as given (no normalisation Jacobian, RAS/backward.cu:396)
}

// 6 quadratic forms d(cov3D) += f(M, d(hat a..f)) common to RAS/backward.cu:258-271 and VOX/backward.cu:157-170.
__device__ __forceinline__ void dcov_from_dhat(const M3 &M, float da, float db, float dc, float dd, float de, float df,
                                               float *o)
{
#define MM(c_, r_) (M.m[c_][r_])
    o[0] += MM(0,0)*MM(0,0)*da + MM(0,0)*MM(1,0)*db + MM(0,0)*MM(2,0)*dc + MM(1,0)*MM(1,0)*dd + MM(1,0)*MM(2,0)*de + MM(2,0)*MM(2,0)*df;
    o[3] += MM(0,1)*MM(0,1)*da + MM(0,1)*MM(1,1)*db + MM(0,1)*MM(2,1)*dc + MM(1,1)*MM(1,1)*dd + MM(1,1)*MM(2,1)*de + MM(2,1)*MM(2,1)*df;
    o[5] += MM(0,2)*MM(0,2)*da + MM(0,2)*MM(1,2)*db + MM(0,2)*MM(2,2)*dc + MM(1,2)*MM(1,2)*dd + MM(1,2)*MM(2,2)*de + MM(2,2)*MM(2,2)*df;
    o[1] += 2*MM(0,0)*MM(0,1)*da + (MM(0,1)*MM(1,0)+MM(0,0)*MM(1,1))*db + (MM(0,1)*MM(2,0)+MM(0,0)*MM(2,1))*dc + 2*MM(1,0)*MM(1,1)*dd + (MM(1,1)*MM(2,0)+MM(1,0)*MM(2,1))*de + 2*MM(2,0)*MM(2,1)*df;
    o[2] += 2*MM(0,0)*MM(0,2)*da + (MM(0,2)*MM(1,0)+MM(0,0)*MM(1,2))*db + (MM(0,2)*MM(2,0)+MM(0,0)*MM(2,2))*dc + 2*MM(1,0)*MM(1,2)*dd + (MM(1,2)*MM(2,0)+MM(1,0)*MM(2,2))*de + 2*MM(2,0)*MM(2,2)*df;
    o[4] += 2*MM(0,1)*MM(0,2)*da + (MM(0,2)*MM(1,1)+MM(0,1)*MM(1,2))*db + (MM(0,2)*MM(2,1)+MM(0,1)*MM(2,2))*dc + 2*MM(1,1)*MM(1,2)*dd + (MM(1,2)*MM(2,1)+MM(1,1)*MM(2,2))*de + 2*MM(2,1)*MM(2,2)*df;
#undef MM
}

__device__ __forceinline__ float3 xform4x3(float3 p, const float *__restrict__ M)
{
    return make_float3(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                       M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14]);
}
__device__ __forceinline__ float4 xform4x4(float3 p, const float *__restrict__ M)
{
    return make_float4(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
                       M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14], M[3] * p.x + M[7] * p.y + M[11] * p.z + M[15]);
}

}  // namespace r2
