// glm/glm.hpp -- minimal RESTATEMENT of the subset of g-truc/glm that the reference's kernels use
// (vec3, vec4, mat3, transpose, dot, length, max).  TEST INFRASTRUCTURE ONLY (oracle/_ref build).
//
// The reference vendors glm as a git submodule (SUB/.gitmodules:1-3) that is EMPTY in /root/reference
// (version unknown), so the real header cannot be used.  Semantics restated here follow glm 0.9.9's
// published definitions: column-major storage m[col][row]; mat3(a..i) fills column 0 with (a,b,c), ...;
// operator*(mat3,mat3): R[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] (summed left to
// right); dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z; length = sqrt(dot(v,v)).
#pragma once
#include <math.h>

namespace glm {

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    template <class A, class B, class C> vec3(A a, B b, C c) : x((float)a), y((float)b), z((float)c) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
    vec3 &operator+=(const vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3 &operator-=(const vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    vec3 &operator+=(float s) { x += s; y += s; z += s; return *this; }
    vec3 &operator-=(float s) { x -= s; y -= s; z -= s; return *this; }
};
inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, const vec3 &a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator+(const vec3 &a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }

struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    template <class A, class B, class C, class D> vec4(A a, B b, C c, D d) : x((float)a), y((float)b), z((float)c), w((float)d) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
inline vec4 operator*(const vec4 &a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator/(const vec4 &a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }

inline float dot(const vec3 &a, const vec3 &b) { vec3 t(a * b); return t.x + t.y + t.z; }
inline float dot(const vec4 &a, const vec4 &b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
inline float length(const vec3 &v) { return sqrtf(dot(v, v)); }
inline float length(const vec4 &v) { return sqrtf(dot(v, v)); }
inline vec3 max(const vec3 &v, float s) { return vec3(v.x > s ? v.x : s, v.y > s ? v.y : s, v.z > s ? v.z : s); }

struct mat3 {
    vec3 c[3];
    mat3() {}
    explicit mat3(float d) { c[0] = vec3(d, 0, 0); c[1] = vec3(0, d, 0); c[2] = vec3(0, 0, d); }
    mat3(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2)
    {
        c[0] = vec3(x0, y0, z0); c[1] = vec3(x1, y1, z1); c[2] = vec3(x2, y2, z2);
    }
    vec3 &operator[](int i) { return c[i]; }
    const vec3 &operator[](int i) const { return c[i]; }
};
inline mat3 transpose(const mat3 &m)
{
    return mat3(m[0][0], m[1][0], m[2][0], m[0][1], m[1][1], m[2][1], m[0][2], m[1][2], m[2][2]);
}
inline mat3 operator*(const mat3 &A, const mat3 &B)
{
    mat3 R;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            R[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
    return R;
}
inline mat3 operator*(const mat3 &A, float s) { mat3 R; for (int c = 0; c < 3; ++c) R[c] = A[c] * s; return R; }
inline mat3 operator*(float s, const mat3 &A) { mat3 R; for (int c = 0; c < 3; ++c) R[c] = A[c] * s; return R; }
inline vec3 operator*(const mat3 &m, const vec3 &v)
{
    return vec3(m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z, m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
                m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z);
}
inline vec3 operator*(const vec3 &v, const mat3 &m) { return vec3(dot(m[0], v), dot(m[1], v), dot(m[2], v)); }

}  // namespace glm
