// gauss_bwd.hip -- per-Gaussian backward stage for gfx950.
//
// Replaces computeCov2DCUDA (RAST/cuda_rasterizer/backward.cu:144-274) and the backward
// preprocessCUDA (:346-396) with its helpers computeColorFromSH (:20-139) and computeCov3D
// (:278-341) -- fused into ONE streaming pass: read the 48-byte GradRec produced by the blend
// backward, recompute the forward intermediates from the inputs (cheaper than storing cov3D /
// re-reading it: HBM-bound kernel).  Only visible Gaussians (radii > 0) do any work or touch memory.
// Two output modes per tensor (bit in `accum_mask`):
//   write      : rows of visible Gaussians are stored; the library zero-fills the tensor first with one
//                multi-segment kernel (k_zero_segments), so callers still never pre-zero anything
//                (the reference memsets 300 B/Gaussian per backward, rasterize_points.cu:154-162);
//   accumulate : rows of visible Gaussians are ADDED to what is already there and culled rows are not
//                touched at all -- this fuses autograd's `grad += new_grad` pass (another 3 x 236 B per
//                Gaussian per view of HBM traffic) into the kernel when several views are accumulated.
#include "common.h"

namespace lr {

namespace {

__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f };
__device__ constexpr float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f };

struct M3 { float c[3][3]; };   // column-major like glm::mat3: c[col][row]
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B)
{
    M3 R;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            R.c[j][i] = A.c[0][i] * B.c[j][0] + A.c[1][i] * B.c[j][1] + A.c[2][i] * B.c[j][2];
    return R;
}
__device__ __forceinline__ M3 m3_t(const M3& A)
{
    M3 R;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) R.c[j][i] = A.c[i][j];
    return R;
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Row store of `n` floats at dst (n*4 bytes per row).  Uses 16-byte stores when the row is aligned.
template <int NMAX>
__device__ __forceinline__ void store_row(float* __restrict__ dst, const float (&v)[NMAX], int n)
{
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < NMAX / 4; q++)
            if (4 * q < n) reinterpret_cast<float4*>(dst)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
#pragma unroll
        for (int q = 0; q < NMAX; q++)
            if (q < n) dst[q] = v[q];
    }
}

// SH backward for one Gaussian (backward.cu:20-139).  Fills dsh[0..3K) and returns dL/ddir.
template <int DEG>
__device__ __forceinline__ V3 sh_backward(const float* __restrict__ sh_row, V3 dir, V3 dL_dRGB, float (&dsh)[48])
{
    constexpr int K = (DEG + 1) * (DEG + 1);
    V3 sh[K];
#pragma unroll
    for (int k = 0; k < K; k++) sh[k] = { sh_row[3 * k], sh_row[3 * k + 1], sh_row[3 * k + 2] };
    float basis[K];
    const float x = dir.x, y = dir.y, z = dir.z;
    V3 dRGBdx = { 0, 0, 0 }, dRGBdy = { 0, 0, 0 }, dRGBdz = { 0, 0, 0 };
    basis[0] = SH_C0;
    if (DEG > 0) {
        basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
        dRGBdx = -SH_C1 * sh[3];
        dRGBdy = -SH_C1 * sh[1];
        dRGBdz = SH_C1 * sh[2];
        if (DEG > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz; basis[6] = SH_C2[2] * (2.f * zz - xx - yy);
            basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
            dRGBdx = dRGBdx + ((SH_C2[0] * y) * sh[4] + (SH_C2[2] * 2.f * -x) * sh[6] + (SH_C2[3] * z) * sh[7] + (SH_C2[4] * 2.f * x) * sh[8]);
            dRGBdy = dRGBdy + ((SH_C2[0] * x) * sh[4] + (SH_C2[1] * z) * sh[5] + (SH_C2[2] * 2.f * -y) * sh[6] + (SH_C2[4] * 2.f * -y) * sh[8]);
            dRGBdz = dRGBdz + ((SH_C2[1] * y) * sh[5] + (SH_C2[2] * 2.f * 2.f * z) * sh[6] + (SH_C2[3] * x) * sh[7]);
            if (DEG > 2) {
                basis[9] = SH_C3[0] * y * (3.f * xx - yy);
                basis[10] = SH_C3[1] * xy * z;
                basis[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                basis[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                basis[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                basis[14] = SH_C3[5] * z * (xx - yy);
                basis[15] = SH_C3[6] * x * (xx - 3.f * yy);
                dRGBdx = dRGBdx + ((SH_C3[0] * 3.f * 2.f * xy) * sh[9] + (SH_C3[1] * yz) * sh[10] + (SH_C3[2] * -2.f * xy) * sh[11]
                                   + (SH_C3[3] * -3.f * 2.f * xz) * sh[12] + (SH_C3[4] * (-3.f * xx + 4.f * zz - yy)) * sh[13]
                                   + (SH_C3[5] * 2.f * xz) * sh[14] + (SH_C3[6] * 3.f * (xx - yy)) * sh[15]);
                dRGBdy = dRGBdy + ((SH_C3[0] * 3.f * (xx - yy)) * sh[9] + (SH_C3[1] * xz) * sh[10]
                                   + (SH_C3[2] * (-3.f * yy + 4.f * zz - xx)) * sh[11] + (SH_C3[3] * -3.f * 2.f * yz) * sh[12]
                                   + (SH_C3[4] * -2.f * xy) * sh[13] + (SH_C3[5] * -2.f * yz) * sh[14]
                                   + (SH_C3[6] * -3.f * 2.f * xy) * sh[15]);
                dRGBdz = dRGBdz + ((SH_C3[1] * xy) * sh[10] + (SH_C3[2] * 4.f * 2.f * yz) * sh[11]
                                   + (SH_C3[3] * 3.f * (2.f * zz - xx - yy)) * sh[12] + (SH_C3[4] * 4.f * 2.f * xz) * sh[13]
                                   + (SH_C3[5] * (xx - yy)) * sh[14]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        dsh[3 * k] = basis[k] * dL_dRGB.x; dsh[3 * k + 1] = basis[k] * dL_dRGB.y; dsh[3 * k + 2] = basis[k] * dL_dRGB.z;
    }
    return { dot(dRGBdx, dL_dRGB), dot(dRGBdy, dL_dRGB), dot(dRGBdz, dL_dRGB) };
}

// Backward of ONE visible Gaussian.
__device__ __forceinline__ void
gauss_backward_one(const int idx, const ViewParams& vp, const float* __restrict__ means3D, const float* __restrict__ scales,
            const float* __restrict__ rotations, const float* __restrict__ shs,
            const float* __restrict__ cov3D_precomp,
            const uint8_t* __restrict__ clamped, const float4 g0, const float4 g1, const float4 g2,
            float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
            float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
            float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
            uint32_t accum_mask)
{
    const size_t i = (size_t)idx;
    const float* __restrict__ V = vp.view;
    const float* __restrict__ Pm = vp.proj;
    const int shrow = vp.M * 3;

    float o_m2d[3] = { 0, 0, 0 }, o_col[3] = { 0, 0, 0 }, o_m3d[3] = { 0, 0, 0 }, o_scale[3] = { 0, 0, 0 };
    float o_conic[4] = { 0, 0, 0, 0 }, o_rot[4] = { 0, 0, 0, 0 };
    float o_cov[6] = { 0, 0, 0, 0, 0, 0 };
    float o_op = 0.f;
    float dsh[48];
#pragma unroll
    for (int k = 0; k < 48; k++) dsh[k] = 0.f;

    {
        const float gmx = g0.x, gmy = g0.y;                 // dL/dmean2D
        const float gca = g0.z, gcb = g0.w, gcc = g1.x;     // dL/dconic (x, y, w)
        o_op = g1.y;
        o_col[0] = g1.z; o_col[1] = g1.w; o_col[2] = g2.x;
        o_m2d[0] = gmx; o_m2d[1] = gmy;
        o_conic[0] = gca; o_conic[1] = gcb; o_conic[3] = gcc;

        const float mx = means3D[3 * i], my = means3D[3 * i + 1], mz = means3D[3 * i + 2];

        // ---- recompute cov3D (forward.cu:118-152) ----
        float c3[6];
        float sx = 0, sy = 0, sz = 0, qr = 0, qx = 0, qy = 0, qz = 0;
        M3 Rm = {}, Mm = {};
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * i + k];
        } else {
            sx = vp.scale_modifier * scales[3 * i]; sy = vp.scale_modifier * scales[3 * i + 1];
            sz = vp.scale_modifier * scales[3 * i + 2];
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            qr = q.x; qx = q.y; qy = q.z; qz = q.w;
            M3 S = { { { sx, 0, 0 }, { 0, sy, 0 }, { 0, 0, sz } } };
            Rm = { { { 1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qr * qz), 2.f * (qx * qz + qr * qy) },
                     { 2.f * (qx * qy + qr * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qr * qx) },
                     { 2.f * (qx * qz - qr * qy), 2.f * (qy * qz + qr * qx), 1.f - 2.f * (qx * qx + qy * qy) } } };
            Mm = m3_mul(S, Rm);
            M3 Sig = m3_mul(m3_t(Mm), Mm);
            c3[0] = Sig.c[0][0]; c3[1] = Sig.c[0][1]; c3[2] = Sig.c[0][2];
            c3[3] = Sig.c[1][1]; c3[4] = Sig.c[1][2]; c3[5] = Sig.c[2][2];
        }

        // ---- K8: conic gradient -> cov3D gradient and mean gradient (backward.cu:159-273) ----
        const float vx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
        const float vy = V[1] * mx + V[5] * my + V[9] * mz + V[13];
        const float vz = V[2] * mx + V[6] * my + V[10] * mz + V[14];
        const float limx = 1.3f * vp.tan_fovx, limy = 1.3f * vp.tan_fovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float h_x = vp.focal_x, h_y = vp.focal_y;
        M3 J = { { { h_x / vz, 0.0f, -(h_x * tx) / (vz * vz) },
                   { 0.0f, h_y / vz, -(h_y * ty) / (vz * vz) },
                   { 0, 0, 0 } } };
        M3 Wm = { { { V[0], V[4], V[8] }, { V[1], V[5], V[9] }, { V[2], V[6], V[10] } } };
        M3 Vrk = { { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } } };
        M3 T = m3_mul(Wm, J);
        M3 cov2D = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
        const float a = cov2D.c[0][0] + 0.3f, b = cov2D.c[0][1], c = cov2D.c[1][1] + 0.3f;
        const float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * gca + 2 * b * c * gcb + (denom - a * c) * gcc);
            dL_dc = denom2inv * (-a * a * gcc + 2 * a * b * gcb + (denom - a * c) * gca);
            dL_db = denom2inv * 2 * (b * c * gca - (denom + 2 * b * b) * gcb + a * b * gcc);
            o_cov[0] = (T.c[0][0] * T.c[0][0] * dL_da + T.c[0][0] * T.c[1][0] * dL_db + T.c[1][0] * T.c[1][0] * dL_dc);
            o_cov[3] = (T.c[0][1] * T.c[0][1] * dL_da + T.c[0][1] * T.c[1][1] * dL_db + T.c[1][1] * T.c[1][1] * dL_dc);
            o_cov[5] = (T.c[0][2] * T.c[0][2] * dL_da + T.c[0][2] * T.c[1][2] * dL_db + T.c[1][2] * T.c[1][2] * dL_dc);
            o_cov[1] = 2 * T.c[0][0] * T.c[0][1] * dL_da + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_db + 2 * T.c[1][0] * T.c[1][1] * dL_dc;
            o_cov[2] = 2 * T.c[0][0] * T.c[0][2] * dL_da + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_db + 2 * T.c[1][0] * T.c[1][2] * dL_dc;
            o_cov[4] = 2 * T.c[0][2] * T.c[0][1] * dL_da + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_db + 2 * T.c[1][1] * T.c[1][2] * dL_dc;
        }
        // dL/dT (upper 2x3), dL/dJ, dL/dt
        float tv0[3], tv1[3];   // T[0][k]*Vrk[j][k] sums
#pragma unroll
        for (int j = 0; j < 3; j++) {
            tv0[j] = T.c[0][0] * Vrk.c[j][0] + T.c[0][1] * Vrk.c[j][1] + T.c[0][2] * Vrk.c[j][2];
            tv1[j] = T.c[1][0] * Vrk.c[j][0] + T.c[1][1] * Vrk.c[j][1] + T.c[1][2] * Vrk.c[j][2];
        }
        const float dL_dT00 = 2 * tv0[0] * dL_da + tv1[0] * dL_db;
        const float dL_dT01 = 2 * tv0[1] * dL_da + tv1[1] * dL_db;
        const float dL_dT02 = 2 * tv0[2] * dL_da + tv1[2] * dL_db;
        const float dL_dT10 = 2 * tv1[0] * dL_dc + tv0[0] * dL_db;
        const float dL_dT11 = 2 * tv1[1] * dL_dc + tv0[1] * dL_db;
        const float dL_dT12 = 2 * tv1[2] * dL_dc + tv0[2] * dL_db;
        const float dL_dJ00 = Wm.c[0][0] * dL_dT00 + Wm.c[0][1] * dL_dT01 + Wm.c[0][2] * dL_dT02;
        const float dL_dJ02 = Wm.c[2][0] * dL_dT00 + Wm.c[2][1] * dL_dT01 + Wm.c[2][2] * dL_dT02;
        const float dL_dJ11 = Wm.c[1][0] * dL_dT10 + Wm.c[1][1] * dL_dT11 + Wm.c[1][2] * dL_dT12;
        const float dL_dJ12 = Wm.c[2][0] * dL_dT10 + Wm.c[2][1] * dL_dT11 + Wm.c[2][2] * dL_dT12;
        const float tz = 1.f / vz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 + (2 * h_y * ty) * tz3 * dL_dJ12;
        // transformVec4x3Transpose (auxiliary.h:89-97); assigned, not accumulated (backward.cu:273)
        o_m3d[0] = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
        o_m3d[1] = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
        o_m3d[2] = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;

        // ---- K9: screen-space mean gradient through the projection (backward.cu:370-387) ----
        const float hw = Pm[3] * mx + Pm[7] * my + Pm[11] * mz + Pm[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (Pm[0] * mx + Pm[4] * my + Pm[8] * mz + Pm[12]) * m_w * m_w;
        const float mul2 = (Pm[1] * mx + Pm[5] * my + Pm[9] * mz + Pm[13]) * m_w * m_w;
        o_m3d[0] += (Pm[0] * m_w - Pm[3] * mul1) * gmx + (Pm[1] * m_w - Pm[3] * mul2) * gmy;
        o_m3d[1] += (Pm[4] * m_w - Pm[7] * mul1) * gmx + (Pm[5] * m_w - Pm[7] * mul2) * gmy;
        o_m3d[2] += (Pm[8] * m_w - Pm[11] * mul1) * gmx + (Pm[9] * m_w - Pm[11] * mul2) * gmy;

        // ---- SH backward (backward.cu:20-139) ----
        if (shs != nullptr) {
            const V3 dir_orig = { mx - vp.campos[0], my - vp.campos[1], mz - vp.campos[2] };
            const float len = sqrtf(dot(dir_orig, dir_orig));
            const V3 dir = { dir_orig.x / len, dir_orig.y / len, dir_orig.z / len };
            const uint8_t cb = clamped[idx];
            const V3 dL_dRGB = { (cb & 1) ? 0.f : o_col[0], (cb & 2) ? 0.f : o_col[1], (cb & 4) ? 0.f : o_col[2] };
            const float* sh_row = shs + i * shrow;
            V3 dL_ddir;
            switch (vp.D) {
                case 0: dL_ddir = sh_backward<0>(sh_row, dir, dL_dRGB, dsh); break;
                case 1: dL_ddir = sh_backward<1>(sh_row, dir, dL_dRGB, dsh); break;
                case 2: dL_ddir = sh_backward<2>(sh_row, dir, dL_dRGB, dsh); break;
                default: dL_ddir = sh_backward<3>(sh_row, dir, dL_dRGB, dsh); break;
            }
            // dnormvdv (auxiliary.h:107-117)
            const V3 v = dir_orig, dv = dL_ddir;
            const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            o_m3d[0] += ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
            o_m3d[1] += (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
            o_m3d[2] += (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
        }

        // ---- cov3D -> scale / rotation (backward.cu:278-341) ----
        if (scales != nullptr) {
            M3 dSig = { { { o_cov[0], 0.5f * o_cov[1], 0.5f * o_cov[2] },
                          { 0.5f * o_cov[1], o_cov[3], 0.5f * o_cov[4] },
                          { 0.5f * o_cov[2], 0.5f * o_cov[4], o_cov[5] } } };
            M3 M2 = Mm;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 3; k++) M2.c[j][k] = 2.0f * Mm.c[j][k];
            M3 dL_dM = m3_mul(M2, dSig);
            M3 Rt = m3_t(Rm);
            M3 dMt = m3_t(dL_dM);
            o_scale[0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
            o_scale[1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
            o_scale[2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
#pragma unroll
            for (int k = 0; k < 3; k++) { dMt.c[0][k] *= sx; dMt.c[1][k] *= sy; dMt.c[2][k] *= sz; }
#define Q(a_, b_) dMt.c[a_][b_]
            o_rot[0] = 2 * qz * (Q(0, 1) - Q(1, 0)) + 2 * qy * (Q(2, 0) - Q(0, 2)) + 2 * qx * (Q(1, 2) - Q(2, 1));
            o_rot[1] = 2 * qy * (Q(1, 0) + Q(0, 1)) + 2 * qz * (Q(2, 0) + Q(0, 2)) + 2 * qr * (Q(1, 2) - Q(2, 1)) - 4 * qx * (Q(2, 2) + Q(1, 1));
            o_rot[2] = 2 * qx * (Q(1, 0) + Q(0, 1)) + 2 * qr * (Q(2, 0) - Q(0, 2)) + 2 * qz * (Q(1, 2) + Q(2, 1)) - 4 * qy * (Q(2, 2) + Q(0, 0));
            o_rot[3] = 2 * qr * (Q(0, 1) - Q(1, 0)) + 2 * qx * (Q(2, 0) + Q(0, 2)) + 2 * qy * (Q(1, 2) + Q(2, 1)) - 4 * qz * (Q(1, 1) + Q(0, 0));
#undef Q
        }
    }

    // ---- store (or accumulate) the rows of this visible Gaussian ----
#define LR_OUT(bit, ptr, val) do { float* p__ = (ptr); *p__ = ((accum_mask >> (bit)) & 1u) ? (*p__ + (val)) : (val); } while (0)
    LR_OUT(ACC_MEAN2D, dL_dmean2D + 3 * i, o_m2d[0]); LR_OUT(ACC_MEAN2D, dL_dmean2D + 3 * i + 1, o_m2d[1]);
    if (!((accum_mask >> ACC_MEAN2D) & 1u)) dL_dmean2D[3 * i + 2] = 0.f;
    if (dL_dconic != nullptr) {
        LR_OUT(ACC_CONIC, dL_dconic + 4 * i, o_conic[0]); LR_OUT(ACC_CONIC, dL_dconic + 4 * i + 1, o_conic[1]);
        LR_OUT(ACC_CONIC, dL_dconic + 4 * i + 3, o_conic[3]);
    }
    LR_OUT(ACC_OPACITY, dL_dopacity + i, o_op);
#pragma unroll
    for (int k = 0; k < 3; k++) if (dL_dcolor != nullptr) LR_OUT(ACC_COLOR, dL_dcolor + 3 * i + k, o_col[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) LR_OUT(ACC_MEAN3D, dL_dmean3D + 3 * i + k, o_m3d[k]);
#pragma unroll
    for (int k = 0; k < 6; k++) if (dL_dcov3D != nullptr) LR_OUT(ACC_COV3D, dL_dcov3D + 6 * i + k, o_cov[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) if (dL_dscale != nullptr) LR_OUT(ACC_SCALE, dL_dscale + 3 * i + k, o_scale[k]);
    if (dL_drot != nullptr) {
        float4* pr = reinterpret_cast<float4*>(dL_drot) + idx;
        float4 v = make_float4(o_rot[0], o_rot[1], o_rot[2], o_rot[3]);
        if ((accum_mask >> ACC_ROT) & 1u) { const float4 o = *pr; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *pr = v;
    }
#undef LR_OUT
    if (dL_dsh != nullptr && shrow > 0) {
        float* dst = dL_dsh + i * shrow;
        const int nk = 3 * (vp.D + 1) * (vp.D + 1);            // only the active bands are non-zero
        const bool acc = (accum_mask >> ACC_SH) & 1u;
        if ((shrow & 3) == 0 && (reinterpret_cast<uintptr_t>(dL_dsh) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < 12; q++) {
                if (4 * q < nk) {
                    float4 v = make_float4(dsh[4 * q], dsh[4 * q + 1], dsh[4 * q + 2], dsh[4 * q + 3]);
                    float4* pq = reinterpret_cast<float4*>(dst) + q;
                    if (acc) { const float4 o = *pq; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *pq = v;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 48; k++)
                if (k < nk) dst[k] = acc ? dst[k] + dsh[k] : dsh[k];
        }
    }
}

// Each single-wave workgroup owns GB_CHUNK consecutive Gaussians: it compacts the visible ones (radii > 0 and at
// least one instance) into LDS with ballots and then runs the heavy per-Gaussian backward densely over that list.  With few visible Gaussians
// (camera paths see ~10 % of a scene) a thread-per-Gaussian launch would run the full 200-VGPR body in
// every wave for a handful of live lanes; here the body runs once per CHUNK/64 as many Gaussians, index
// order (and with it coalescing of the row reads/writes) is preserved, and no global atomics are needed.
constexpr int GB_CHUNK = 256;        // Gaussians per workgroup
constexpr int GB_THREADS = 64;       // ONE wave per workgroup: the heavy body needs ~200 VGPRs (2 waves/SIMD), so small
                                     // single-wave groups keep 4x more independent chunks in flight than 256-thread ones

__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(GB_THREADS)
k_gauss_bwd(ViewParams vp, const float* __restrict__ means3D, const float* __restrict__ scales,
            const float* __restrict__ rotations, const float* __restrict__ shs,
            const float* __restrict__ cov3D_precomp, const int* __restrict__ radii,
            const uint8_t* __restrict__ clamped, const uint32_t* __restrict__ tiles_touched,
            const uint32_t* __restrict__ goff, const char* __restrict__ bin_base, const GeomHeader* __restrict__ hdr,
            float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
            float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
            float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
            uint32_t accum_mask)
{
    constexpr uint32_t SERIAL_MAX = 24;      // instances summed by the owning lane; more -> whole wave helps
    __shared__ uint32_t s_list[GB_CHUNK];
    const int base = blockIdx.x * GB_CHUNK;
    const int lane = threadIdx.x;
    uint32_t s_count = 0;                                   // wave-uniform running count
#pragma unroll
    for (int r = 0; r < GB_CHUNK / GB_THREADS; r++) {
        const int idx = base + r * GB_THREADS + lane;
        // a visible Gaussian whose tiles were all culled has an all-zero gradient: nothing to do for it
        const bool vis = idx < vp.P && radii[idx] > 0 && tiles_touched[idx] != 0;
        const uint64_t m = __ballot(vis);
        if (vis) s_list[s_count + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)idx;
        s_count += (uint32_t)__popcll(m);
    }
    __syncthreads();
    const uint32_t n = s_count;
    if (n == 0) return;
    // per-instance partial sums written by k_render_bwd (48-byte slots, contiguous per Gaussian in emission
    // order); slots at or beyond num_sorted were never built (async-mode overflow) and are ignored
    const float4* __restrict__ inst_grad =
        reinterpret_cast<const float4*>(bin_base + bin_layout((long long)hdr->bin_bound).inst_grad);
    const uint32_t n_slots = hdr->num_sorted;

    for (uint32_t t0 = 0; t0 < n; t0 += GB_THREADS) {
        const uint32_t t = t0 + threadIdx.x;
        const bool live = t < n;
        const int idx = live ? (int)s_list[t] : 0;
        const uint32_t tt = live ? tiles_touched[idx] : 0u;
        const uint32_t off = live ? goff[idx] : 0u;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        if (tt <= SERIAL_MAX) {
            for (uint32_t j = 0; j < tt; j++) {
                if (off + j >= n_slots) break;
                const float4* slot = inst_grad + 3 * (size_t)(off + j);
                add4(g0, slot[0]); add4(g1, slot[1]); add4(g2, slot[2]);
            }
        }
        uint64_t big = __ballot(tt > SERIAL_MAX);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const uint32_t b_tt = __shfl(tt, src), b_off = __shfl(off, src);
            float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
            for (uint32_t j = lane; j < b_tt; j += 64) {
                if (b_off + j >= n_slots) break;
                const float4* slot = inst_grad + 3 * (size_t)(b_off + j);
                add4(p0, slot[0]); add4(p1, slot[1]); add4(p2, slot[2]);
            }
            p0.x = wave_sum(p0.x); p0.y = wave_sum(p0.y); p0.z = wave_sum(p0.z); p0.w = wave_sum(p0.w);
            p1.x = wave_sum(p1.x); p1.y = wave_sum(p1.y); p1.z = wave_sum(p1.z); p1.w = wave_sum(p1.w);
            p2.x = wave_sum(p2.x);
            if (lane == src) { g0 = p0; g1 = p1; g2 = p2; }
        }
        if (live)
            gauss_backward_one(idx, vp, means3D, scales, rotations, shs, cov3D_precomp, clamped, g0, g1, g2,
                               dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                               dL_drot, accum_mask);
    }
}

struct ZeroSegs { float4* p[9]; unsigned long long n4[9]; unsigned long long off[10]; int count; };

// One launch zero-fills up to nine output tensors (16-byte stores, grid-stride over the concatenation).
__global__ void __launch_bounds__(256)
k_zero_segments(ZeroSegs z)
{
    const unsigned long long total = z.off[z.count];
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (unsigned long long)gridDim.x * blockDim.x) {
        int sgi = 0;
#pragma unroll
        for (int k = 1; k < 9; k++) if (k < z.count && t >= z.off[k]) sgi = k;
        z.p[sgi][t - z.off[sgi]] = zero;
    }
}

__global__ void k_zero_tail(float* p, unsigned long long from, unsigned long long to)
{
    const unsigned long long t = from + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < to) p[t] = 0.f;
}

}  // namespace

void launch_zero_outputs(float* const* ptrs, const unsigned long long* nfloats, int count, hipStream_t s)
{
    ZeroSegs z;
    z.count = 0;
    z.off[0] = 0;
    for (int k = 0; k < count && z.count < 9; k++) {
        if (ptrs[k] == nullptr || nfloats[k] == 0) continue;
        const unsigned long long n4 = nfloats[k] / 4;          // all torch allocations are >= 16-byte aligned
        if (nfloats[k] % 4) hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, s, ptrs[k], n4 * 4, nfloats[k]);
        if (n4 == 0) continue;
        z.p[z.count] = reinterpret_cast<float4*>(ptrs[k]);
        z.n4[z.count] = n4;
        z.off[z.count + 1] = z.off[z.count] + n4;
        z.count++;
    }
    if (z.count == 0) return;
    for (int k = z.count; k < 9; k++) { z.p[k] = nullptr; z.n4[k] = 0; z.off[k + 1] = z.off[z.count]; }
    const unsigned long long total = z.off[z.count];
    unsigned long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_zero_segments, dim3((unsigned)blocks), dim3(256), 0, s, z);
}

void launch_gauss_bwd(const ViewParams& vp, const float* means3D, const float* scales, const float* rotations,
                      const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                      const int* radii, const uint8_t* clamped, const uint32_t* tiles_touched,
                      const uint32_t* goff, const char* bin_base, const GeomHeader* hdr,
                      float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                      uint32_t accum_mask, hipStream_t s)
{
    (void)colors_precomp;
    if (vp.P <= 0) return;
    hipLaunchKernelGGL(k_gauss_bwd, dim3((vp.P + GB_CHUNK - 1) / GB_CHUNK), dim3(GB_THREADS), 0, s, vp, means3D, scales, rotations, shs,
                       cov3D_precomp, radii, clamped, tiles_touched, goff, bin_base, hdr, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,
                       dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, accum_mask);
}

}  // namespace lr
