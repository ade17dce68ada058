/*
 * glm/glm.hpp -- stand-in for the subset of GLM (g-truc/glm; the reference does not vendor or pin it:
 * RAST/setup.py:29 points at an absent third_party/glm, R/packages.txt:1 asks for libglm-dev) that the
 * reference rasterizer uses: vec3, vec4, mat3, length, dot, max, transpose and the arithmetic operators.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/_ref recipe).  Each operation restates GLM's published definition
 * (glm 0.9.9, detail/type_vec3.inl, type_mat3x3.inl, func_geometric.inl, func_common.inl) including its
 * evaluation order, which is what fixes the float32 result:
 *   mat3 is column-major, m[c] is column c, mat3(a,b,c, d,e,f, g,h,i) has columns (a,b,c),(d,e,f),(g,h,i);
 *   (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2]      (type_mat3x3.inl operator*)
 *   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z  via tmp = a*b; tmp.x + tmp.y + tmp.z  (func_geometric.inl)
 *   length(v) = sqrt(dot(v,v));   max(v, s) = component-wise (x < s) ? s : x        (func_common.inl)
 *   s*m = mat3(m[0]*s, m[1]*s, m[2]*s);  v/s divides each component.
 */
#ifndef LUCID_REF_GLM_SUBSET_HPP
#define LUCID_REF_GLM_SUBSET_HPP
#include <math.h>

/* host build (oracle/ref_shim): plain inline; device build (oracle/ref_shim_hip includes this file with hipcc): both sides */
#if defined(__HIPCC__)
#define LR_GLM_FUNC __host__ __device__ inline
#else
#define LR_GLM_FUNC inline
#endif

namespace glm {

struct vec3 {
    float x, y, z;
    LR_GLM_FUNC vec3() : x(0), y(0), z(0) {}
    template <class X, class Y, class Z> LR_GLM_FUNC vec3(X x_, Y y_, Z z_) : x((float)x_), y((float)y_), z((float)z_) {}
    LR_GLM_FUNC explicit vec3(float s) : x(s), y(s), z(s) {}
    LR_GLM_FUNC float& operator[](int i) { return (&x)[i]; }
    LR_GLM_FUNC const float& operator[](int i) const { return (&x)[i]; }
    LR_GLM_FUNC vec3& operator+=(const vec3& v) { x += v.x; y += v.y; z += v.z; return *this; }
    LR_GLM_FUNC vec3& operator+=(float s) { x += s; y += s; z += s; return *this; }
    LR_GLM_FUNC vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};
struct vec4 {
    float x, y, z, w;
    LR_GLM_FUNC vec4() : x(0), y(0), z(0), w(0) {}
    LR_GLM_FUNC vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};

LR_GLM_FUNC vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
LR_GLM_FUNC vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
LR_GLM_FUNC vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
LR_GLM_FUNC vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
LR_GLM_FUNC vec3 operator*(float s, const vec3& v) { return vec3(s * v.x, s * v.y, s * v.z); }
LR_GLM_FUNC vec3 operator*(const vec3& v, float s) { return vec3(v.x * s, v.y * s, v.z * s); }
LR_GLM_FUNC vec3 operator/(const vec3& v, float s) { return vec3(v.x / s, v.y / s, v.z / s); }

LR_GLM_FUNC float dot(const vec3& a, const vec3& b) { vec3 tmp(a * b); return tmp.x + tmp.y + tmp.z; }
LR_GLM_FUNC float length(const vec3& v) { return sqrtf(dot(v, v)); }
LR_GLM_FUNC vec3 max(const vec3& v, float s) { return vec3((v.x < s) ? s : v.x, (v.y < s) ? s : v.y, (v.z < s) ? s : v.z); }

struct mat3 {
    vec3 col[3];
    LR_GLM_FUNC mat3() { col[0] = vec3(1, 0, 0); col[1] = vec3(0, 1, 0); col[2] = vec3(0, 0, 1); }
    LR_GLM_FUNC explicit mat3(float s) { col[0] = vec3(s, 0, 0); col[1] = vec3(0, s, 0); col[2] = vec3(0, 0, s); }
    template <class X1, class Y1, class Z1, class X2, class Y2, class Z2, class X3, class Y3, class Z3>
    LR_GLM_FUNC mat3(X1 x1, Y1 y1, Z1 z1, X2 x2, Y2 y2, Z2 z2, X3 x3, Y3 y3, Z3 z3)
    { col[0] = vec3(x1, y1, z1); col[1] = vec3(x2, y2, z2); col[2] = vec3(x3, y3, z3); }
    LR_GLM_FUNC mat3(const vec3& c0, const vec3& c1, const vec3& c2) { col[0] = c0; col[1] = c1; col[2] = c2; }
    LR_GLM_FUNC vec3& operator[](int i) { return col[i]; }
    LR_GLM_FUNC const vec3& operator[](int i) const { return col[i]; }
};

LR_GLM_FUNC mat3 operator*(const mat3& m1, const mat3& m2)
{
    const float A00 = m1[0][0], A01 = m1[0][1], A02 = m1[0][2];
    const float A10 = m1[1][0], A11 = m1[1][1], A12 = m1[1][2];
    const float A20 = m1[2][0], A21 = m1[2][1], A22 = m1[2][2];
    const float B00 = m2[0][0], B01 = m2[0][1], B02 = m2[0][2];
    const float B10 = m2[1][0], B11 = m2[1][1], B12 = m2[1][2];
    const float B20 = m2[2][0], B21 = m2[2][1], B22 = m2[2][2];
    mat3 R(0.0f);
    R[0][0] = A00 * B00 + A10 * B01 + A20 * B02;
    R[0][1] = A01 * B00 + A11 * B01 + A21 * B02;
    R[0][2] = A02 * B00 + A12 * B01 + A22 * B02;
    R[1][0] = A00 * B10 + A10 * B11 + A20 * B12;
    R[1][1] = A01 * B10 + A11 * B11 + A21 * B12;
    R[1][2] = A02 * B10 + A12 * B11 + A22 * B12;
    R[2][0] = A00 * B20 + A10 * B21 + A20 * B22;
    R[2][1] = A01 * B20 + A11 * B21 + A21 * B22;
    R[2][2] = A02 * B20 + A12 * B21 + A22 * B22;
    return R;
}
LR_GLM_FUNC mat3 operator*(float s, const mat3& m) { return mat3(m[0] * s, m[1] * s, m[2] * s); }
LR_GLM_FUNC mat3 operator*(const mat3& m, float s) { return mat3(m[0] * s, m[1] * s, m[2] * s); }
LR_GLM_FUNC mat3 transpose(const mat3& m)
{
    mat3 R(0.0f);
    R[0][0] = m[0][0]; R[0][1] = m[1][0]; R[0][2] = m[2][0];
    R[1][0] = m[0][1]; R[1][1] = m[1][1]; R[1][2] = m[2][1];
    R[2][0] = m[0][2]; R[2][1] = m[1][2]; R[2][2] = m[2][2];
    return R;
}

}  // namespace glm
#endif
