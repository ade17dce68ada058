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
#include <algorithm>

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

// SH basis values and their derivatives w.r.t. the (normalised) view direction, band by band
// (the coefficients of backward.cu:20-139: dRGB/dx = sum_k dbx[k] * sh[k], dL/dsh[k] = b[k] * dL/dRGB).
template <int DEG>
__device__ __forceinline__ void sh_basis(V3 dir, float (&b)[16], float (&bx)[16], float (&by)[16], float (&bz)[16])
{
    const float x = dir.x, y = dir.y, z = dir.z;
#pragma unroll
    for (int k = 0; k < 16; k++) { b[k] = 0.f; bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; }
    b[0] = SH_C0;
    if (DEG > 0) {
        b[1] = -SH_C1 * y; by[1] = -SH_C1;
        b[2] = SH_C1 * z;  bz[2] = SH_C1;
        b[3] = -SH_C1 * x; bx[3] = -SH_C1;
        if (DEG > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy;                    bx[4] = SH_C2[0] * y;        by[4] = SH_C2[0] * x;
            b[5] = SH_C2[1] * yz;                    by[5] = SH_C2[1] * z;        bz[5] = SH_C2[1] * y;
            b[6] = SH_C2[2] * (2.f * zz - xx - yy);  bx[6] = SH_C2[2] * 2.f * -x; by[6] = SH_C2[2] * 2.f * -y;
            bz[6] = SH_C2[2] * 2.f * 2.f * z;
            b[7] = SH_C2[3] * xz;                    bx[7] = SH_C2[3] * z;        bz[7] = SH_C2[3] * x;
            b[8] = SH_C2[4] * (xx - yy);             bx[8] = SH_C2[4] * 2.f * x;  by[8] = SH_C2[4] * 2.f * -y;
            if (DEG > 2) {
                b[9] = SH_C3[0] * y * (3.f * xx - yy);
                bx[9] = SH_C3[0] * 3.f * 2.f * xy;   by[9] = SH_C3[0] * 3.f * (xx - yy);
                b[10] = SH_C3[1] * xy * z;
                bx[10] = SH_C3[1] * yz;              by[10] = SH_C3[1] * xz;      bz[10] = SH_C3[1] * xy;
                b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                bx[11] = SH_C3[2] * -2.f * xy;       by[11] = SH_C3[2] * (-3.f * yy + 4.f * zz - xx);
                bz[11] = SH_C3[2] * 4.f * 2.f * yz;
                b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                bx[12] = SH_C3[3] * -3.f * 2.f * xz; by[12] = SH_C3[3] * -3.f * 2.f * yz;
                bz[12] = SH_C3[3] * 3.f * (2.f * zz - xx - yy);
                b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                bx[13] = SH_C3[4] * (-3.f * xx + 4.f * zz - yy); by[13] = SH_C3[4] * -2.f * xy;
                bz[13] = SH_C3[4] * 4.f * 2.f * xz;
                b[14] = SH_C3[5] * z * (xx - yy);
                bx[14] = SH_C3[5] * 2.f * xz;        by[14] = SH_C3[5] * -2.f * yz; bz[14] = SH_C3[5] * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.f * yy);
                bx[15] = SH_C3[6] * 3.f * (xx - yy); by[15] = SH_C3[6] * -3.f * 2.f * xy;
            }
        }
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum within each 16-lane row; every lane of the row ends with the row total
__device__ __forceinline__ float row_sum(float v)
{
    v += dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);           // row_half_mirror
    v += dpp<0x140>(v);           // row_mirror
    return v;
}

// Backward of ONE visible Gaussian, everything except the spherical-harmonics rows (those are handled 16 lanes per
// Gaussian by the caller, which passes the resulting dL/d(view direction) in `dL_ddir` when `have_sh`).
// RAW (lr_backward_raw) is a template parameter, not a run-time branch: with `if (vp.raw)` blocks in this function
// ROCm 7.2 hipcc produced wrong dL/dcov3D in the NON-raw path (verified on hardware by removing either block).
template <bool RAW>
__device__ __forceinline__ void
gauss_backward_one(const int idx, const ViewParams& vp, const float* __restrict__ means3D, const float* __restrict__ scales,
            const float* __restrict__ rotations, const bool have_sh, const V3 dL_ddir,
            const float* __restrict__ cov3D_precomp, const float4 g0, const float4 g1, const float4 g2,
            float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
            float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
            float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
            uint32_t accum_mask, float* __restrict__ acc16)
{
    const size_t i = (size_t)idx;
    const float* __restrict__ V = vp.view;
    const float* __restrict__ Pm = vp.proj;

    float o_m2d[3] = { 0, 0, 0 }, o_col[3] = { 0, 0, 0 }, o_m3d[3] = { 0, 0, 0 }, o_scale[3] = { 0, 0, 0 };
    float o_conic[4] = { 0, 0, 0, 0 }, o_rot[4] = { 0, 0, 0, 0 };
    float o_cov[6] = { 0, 0, 0, 0, 0, 0 };
    float o_op = 0.f;

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
        float raw_s[3] = { 1.f, 1.f, 1.f }, raw_inv = 1.f;      // raw mode: activated scales, 1/max(|q|, eps)
        M3 Rm = {}, Mm = {};
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * i + k];
        } else {
            sx = scales[3 * i]; sy = scales[3 * i + 1]; sz = scales[3 * i + 2];
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            qr = q.x; qx = q.y; qy = q.z; qz = q.w;
            if (RAW) {
                sx = act_scale(sx); sy = act_scale(sy); sz = act_scale(sz);
                raw_s[0] = sx; raw_s[1] = sy; raw_s[2] = sz;
                raw_inv = act_quat_inv_norm(qr, qx, qy, qz);
                qr *= raw_inv; qx *= raw_inv; qy *= raw_inv; qz *= raw_inv;
            }
            sx *= vp.scale_modifier; sy *= vp.scale_modifier; sz *= vp.scale_modifier;
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

        // ---- view-direction gradient of the SH colour -> mean (backward.cu:128-138) ----
        if (have_sh) {
            const V3 v = { mx - vp.campos[0], my - vp.campos[1], mz - vp.campos[2] }, dv = dL_ddir;
            // dnormvdv (auxiliary.h:107-117)
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
            if (RAW) {
                // through exp: d/ds_raw = d/ds * exp(s_raw); through q = r/|r|: (g - q (q.g)) / |r|
                o_scale[0] *= raw_s[0]; o_scale[1] *= raw_s[1]; o_scale[2] *= raw_s[2];
                const float qg = qr * o_rot[0] + qx * o_rot[1] + qy * o_rot[2] + qz * o_rot[3];
                o_rot[0] = (o_rot[0] - qr * qg) * raw_inv; o_rot[1] = (o_rot[1] - qx * qg) * raw_inv;
                o_rot[2] = (o_rot[2] - qy * qg) * raw_inv; o_rot[3] = (o_rot[3] - qz * qg) * raw_inv;
            }
        }
        if (RAW) {
            const float o = act_opacity(vp.opacity_raw[i]);
            o_op *= o * (1.0f - o);
        }
    }

    // ---- interleaved accumulation (lr_views_accumulate): the five small rows of a Gaussian -- mean2D (2 of 3 floats),
    // opacity, mean3D, scale, rotation: 13 floats scattered over five arrays, each a 12-16 byte read-modify-write that
    // moves a 32-byte sector both ways -- live in ONE 64-byte row of acc16 [P][16] during the step: one full line in, one
    // out.  k_uninterleave_add hands them to the caller's tensors once per step.
    if (acc16 != nullptr) {
        float4* row = reinterpret_cast<float4*>(acc16 + 16 * i);
        float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
        r0.x += o_m2d[0]; r0.y += o_m2d[1]; r0.z += o_op; r0.w += o_m3d[0];
        r1.x += o_m3d[1]; r1.y += o_m3d[2]; r1.z += o_scale[0]; r1.w += o_scale[1];
        r2.x += o_scale[2]; r2.y += o_rot[0]; r2.z += o_rot[1]; r2.w += o_rot[2];
        r3.x += o_rot[3];
        row[0] = r0; row[1] = r1; row[2] = r2; row[3] = r3;
        return;
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
}

// Persistent single-wave workgroups walk the index-ordered list of emitting Gaussians that the forward's compaction
// left in the geometry buffer (vis_list, hdr->num_compact entries), 64 Gaussians per round, rounds dealt round-robin.
// Only Gaussians that own at least one tile instance do any work or touch memory (camera paths see ~10-20 % of a
// scene; a thread-per-Gaussian launch would run the heavy body in every wave for a handful of live lanes), index
// order -- and with it the locality of the row reads/writes -- is preserved, and no global atomics are needed.
// A round is a chain of dependent memory operations (list -> slots -> rows -> LDS -> SH rows), i.e. latency bound:
// what matters is many independent rounds in flight, hence single-wave groups and a register budget of 128.
constexpr int GB_THREADS = 64;
#ifndef LR_GB_WAVES
#define LR_GB_WAVES 3
#endif
constexpr int GB_MAX_GROUPS = 256 * 4 * LR_GB_WAVES;      // CUs x SIMDs x resident waves

__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <bool RAW>
__global__ void __launch_bounds__(GB_THREADS) __attribute__((amdgpu_waves_per_eu(LR_GB_WAVES, 8)))
k_gauss_bwd(ViewParams vp, const float* __restrict__ means3D, const float* __restrict__ scales,
            const float* __restrict__ rotations, const float* __restrict__ shs,
            const float* __restrict__ cov3D_precomp, const uint32_t* __restrict__ vis_list,
            const uint8_t* __restrict__ clamped, const uint32_t* __restrict__ offsets,
            const char* __restrict__ bin_base, const GeomHeader* __restrict__ hdr,
            float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
            float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D,
            float* __restrict__ dL_dsh, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
            uint32_t accum_mask, float* __restrict__ acc16)
{
    constexpr uint32_t SERIAL_MAX = 24;      // instances summed by the owning lane; more -> whole wave helps
    constexpr int BST = 17;                  // LDS row stride (floats) of the per-Gaussian basis rows: odd -> no conflicts
    __shared__ uint32_t s_idx[GB_THREADS];
    __shared__ float s_b[4][32 * BST];           // basis, d/dx, d/dy, d/dz of 32 Gaussians (half a round)
    __shared__ float s_rgb[3][GB_THREADS];       // dL/dRGB after the clamp mask
    __shared__ float s_ddir[3][GB_THREADS];      // dL/d(view direction)
    const int lane = threadIdx.x;
    const uint32_t n = hdr->num_compact;
    if (n == 0) return;
    if (hdr->overflow != 0u) return;         // async mode: an overflowed view contributes nothing (see k_render_bwd)
    // per-instance partial sums written by k_render_bwd (48-byte slots, contiguous per Gaussian in emission
    // order); slots at or beyond num_sorted were never built (async-mode overflow) and are ignored
    const float4* __restrict__ inst_grad =
        reinterpret_cast<const float4*>(bin_base + bin_layout((long long)hdr->bin_bound).inst_grad);
    const uint32_t n_slots = hdr->num_sorted;

    for (uint32_t t0 = blockIdx.x * GB_THREADS; t0 < n; t0 += gridDim.x * GB_THREADS) {
        const uint32_t t = t0 + threadIdx.x;
        const bool live = t < n;
        const int idx = live ? (int)vis_list[t] : 0;
        s_idx[lane] = (uint32_t)idx;
        // first instance slot and instance count from the rank-ordered `offsets` (dense reads: neighbouring lanes read
        // neighbouring words) instead of goff[idx] / tiles_touched[idx] -- two more 64-byte lines per visible Gaussian for
        // 4 useful bytes each when ~9 % of the Gaussians are visible
        const uint32_t off = live ? offsets[t] : 0u;
        const uint32_t tt = live ? ((t + 1 < n) ? offsets[t + 1] : hdr->num_instances) - off : 0u;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        if (tt <= SERIAL_MAX && off < n_slots) {
            // four slots per step with independent loads (a one-slot loop pays one memory latency per instance)
            const uint32_t cnt = min(tt, n_slots - off);
            const float4* first = inst_grad + 3 * (size_t)off;
            for (uint32_t j = 0; j < cnt; j += 4) {
                float4 a[4][3];
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const float4* slot = first + 3 * (size_t)min(j + q, cnt - 1);
                    a[q][0] = slot[0]; a[q][1] = slot[1]; a[q][2] = slot[2];
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; q++)
                    if (j + q < cnt) { add4(g0, a[q][0]); add4(g1, a[q][1]); add4(g2, a[q][2]); }
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
        // ---- spherical harmonics: rows are 3*M floats per Gaussian.  One lane per Gaussian would make every
        // load/store instruction touch 64 different rows (64 cache lines for 16 useful bytes each, and the working
        // set of a wave overflows the L1); instead each lane computes the 16 basis values and their direction
        // derivatives of ITS Gaussian into LDS, and then 16 LANES share one Gaussian, lane k owning coefficient k:
        // a wave reads/updates 4 complete rows per step with fully used cache lines.
        V3 dL_ddir = { 0.f, 0.f, 0.f };
        const bool have_sh = shs != nullptr && dL_dsh != nullptr;
        if (have_sh) {
            const int k = lane & 15, sub = lane >> 4;
            const int K = (vp.D + 1) * (vp.D + 1);
            const size_t shrow = (size_t)vp.M * 3;
            const bool acc = (accum_mask >> ACC_SH) & 1u;
            const bool no_fill = !acc && (accum_mask >> 31) != 0u;          // LR_ACC_NO_ZERO_FILL (lucid_raster.h)
            const int n_here = (int)min((uint32_t)GB_THREADS, n - t0);
            if (live) {
                const uint8_t cb = clamped[idx];
                s_rgb[0][lane] = (cb & 1) ? 0.f : g1.z;
                s_rgb[1][lane] = (cb & 2) ? 0.f : g1.w;
                s_rgb[2][lane] = (cb & 4) ? 0.f : g2.x;
            }
            // two half rounds of 32 Gaussians keep the LDS footprint (and with it the occupancy limit) small
            for (int half = 0; half < 2; half++) {
                if (half * 32 >= n_here) break;
                if (live && (lane >> 5) == half) {
                    const size_t i = (size_t)idx;
                    const V3 d0 = { means3D[3 * i] - vp.campos[0], means3D[3 * i + 1] - vp.campos[1], means3D[3 * i + 2] - vp.campos[2] };
                    const float len = sqrtf(dot(d0, d0));
                    const V3 dir = { d0.x / len, d0.y / len, d0.z / len };
                    float b[16], bx[16], by[16], bz[16];
                    switch (vp.D) {
                        case 0: sh_basis<0>(dir, b, bx, by, bz); break;
                        case 1: sh_basis<1>(dir, b, bx, by, bz); break;
                        case 2: sh_basis<2>(dir, b, bx, by, bz); break;
                        default: sh_basis<3>(dir, b, bx, by, bz); break;
                    }
                    const int o = (lane & 31) * BST;
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        s_b[0][o + q] = b[q]; s_b[1][o + q] = bx[q]; s_b[2][o + q] = by[q]; s_b[3][o + q] = bz[q];
                    }
                }
                lds_barrier();
                // four steps (16 Gaussians) at a time: all row loads are issued before the first store, which the
                // compiler cannot do across steps by itself (the accumulate loads may alias the previous stores)
                for (int it0 = 0; it0 < 8; it0 += 4) {
                    if (half * 32 + it0 * 4 >= n_here) break;
                    float sv[4][3], dv[4][3];
                    size_t rowv[4];
                    bool onv[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int gl = (it0 + q) * 4 + sub, g = half * 32 + gl;
                        onv[q] = g < n_here && k < K;
                        const size_t gi = (size_t)s_idx[onv[q] ? g : 0];
                        const int kk = onv[q] ? k : 0;
                        // raw mode: coefficient 0 lives in features_dc [P,3], the others in features_rest [P,M-1,3]
                        rowv[q] = !RAW ? gi * shrow + 3 * kk : (kk == 0 ? gi * 3 : gi * (shrow - 3) + 3 * (kk - 1));
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const bool in_rest = RAW && onv[q] && k != 0;      // idle lanes re-read row 0 of features_dc
                        const float* sp = (in_rest ? vp.sh_rest : shs) + rowv[q];
                        sv[q][0] = sp[0]; sv[q][1] = sp[1]; sv[q][2] = sp[2];
                        if (acc) { const float* dp = (in_rest ? vp.dL_dsh_rest : dL_dsh) + rowv[q]; dv[q][0] = dp[0]; dv[q][1] = dp[1]; dv[q][2] = dp[2]; }
                        else { dv[q][0] = 0.f; dv[q][1] = 0.f; dv[q][2] = 0.f; }
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int gl = (it0 + q) * 4 + sub, g = half * 32 + gl;
                        const int gs = onv[q] ? g : 0, gls = onv[q] ? gl : 0, ks = onv[q] ? k : 0;
                        const float r0 = s_rgb[0][gs], r1 = s_rgb[1][gs], r2 = s_rgb[2][gs];
                        const float bk = s_b[0][gls * BST + ks];
                        if (onv[q]) {
                            float* dp = ((RAW && k != 0) ? vp.dL_dsh_rest : dL_dsh) + rowv[q];   // onv[q] holds here
                            dp[0] = dv[q][0] + bk * r0; dp[1] = dv[q][1] + bk * r1; dp[2] = dv[q][2] + bk * r2;
                        } else if (no_fill && g < n_here && k >= K && k < vp.M) {
                            // LR_ACC_NO_ZERO_FILL: nobody zero-filled the tensor, and a VISITED Gaussian's rows are read by the
                            // masked optimizer step -- its coefficients above the active degree are written as the zeros they are
                            const size_t gi = (size_t)s_idx[g];
                            float* dp = !RAW ? dL_dsh + gi * shrow + 3 * k
                                             : (k == 0 ? dL_dsh + gi * 3 : vp.dL_dsh_rest + gi * (shrow - 3) + 3 * (size_t)(k - 1));
                            dp[0] = 0.f; dp[1] = 0.f; dp[2] = 0.f;
                        }
                        const float sd = onv[q] ? sv[q][0] * r0 + sv[q][1] * r1 + sv[q][2] * r2 : 0.f;
                        const float px = row_sum(s_b[1][gls * BST + ks] * sd);
                        const float py = row_sum(s_b[2][gls * BST + ks] * sd);
                        const float pz = row_sum(s_b[3][gls * BST + ks] * sd);
                        if (k == 0 && g < n_here) { s_ddir[0][g] = px; s_ddir[1][g] = py; s_ddir[2][g] = pz; }
                    }
                }
                lds_barrier();
            }
            if (live) dL_ddir = { s_ddir[0][lane], s_ddir[1][lane], s_ddir[2][lane] };
        }
        if (live)
            gauss_backward_one<RAW>(idx, vp, means3D, scales, rotations, have_sh, dL_ddir, cov3D_precomp, g0, g1, g2,
                               dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dscale,
                               dL_drot, accum_mask, acc16);
        lds_barrier();                     // the LDS planes are rewritten by the next round
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
                      const uint32_t* vis_list, const uint8_t* clamped, const uint32_t* offsets,
                      const char* bin_base, const GeomHeader* hdr,
                      float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                      uint32_t accum_mask, float* acc16, hipStream_t s)
{
    (void)colors_precomp;
    if (vp.P <= 0) return;
    const int groups = std::min((vp.P + GB_THREADS - 1) / GB_THREADS, GB_MAX_GROUPS);
    if (vp.raw)
        hipLaunchKernelGGL(k_gauss_bwd<true>, dim3(groups), dim3(GB_THREADS), 0, s, vp, means3D, scales, rotations, shs,
                           cov3D_precomp, vis_list, clamped, offsets, bin_base, hdr, dL_dmean2D, dL_dconic,
                           dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, accum_mask, acc16);
    else
        hipLaunchKernelGGL(k_gauss_bwd<false>, dim3(groups), dim3(GB_THREADS), 0, s, vp, means3D, scales, rotations, shs,
                           cov3D_precomp, vis_list, clamped, offsets, bin_base, hdr, dL_dmean2D, dL_dconic,
                           dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, accum_mask, acc16);
}

namespace {
// acc16 [P][16] -> the caller's accumulators (lr_views_accumulate, once per step): rows that no view touched are all
// zero and are skipped (nothing read or written on the caller's side)
__global__ void __launch_bounds__(256)
k_uninterleave_add(int P, const float* __restrict__ acc16, float* __restrict__ mean2D, float* __restrict__ opacity,
                   float* __restrict__ mean3D, float* __restrict__ scale, float* __restrict__ rot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4* row = reinterpret_cast<const float4*>(acc16 + 16 * (size_t)i);
    const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
    const bool any = r0.x != 0.f || r0.y != 0.f || r0.z != 0.f || r0.w != 0.f || r1.x != 0.f || r1.y != 0.f || r1.z != 0.f ||
                     r1.w != 0.f || r2.x != 0.f || r2.y != 0.f || r2.z != 0.f || r2.w != 0.f || r3.x != 0.f;
    if (!any) return;
    const size_t k = (size_t)i;
    mean2D[3 * k] += r0.x; mean2D[3 * k + 1] += r0.y;
    opacity[k] += r0.z;
    mean3D[3 * k] += r0.w; mean3D[3 * k + 1] += r1.x; mean3D[3 * k + 2] += r1.y;
    if (scale != nullptr) { scale[3 * k] += r1.z; scale[3 * k + 1] += r1.w; scale[3 * k + 2] += r2.x; }
    if (rot != nullptr) {
        float4* pr = reinterpret_cast<float4*>(rot) + i;
        float4 v = *pr;
        v.x += r2.y; v.y += r2.z; v.z += r2.w; v.w += r3.x;
        *pr = v;
    }
}
}  // namespace

void launch_uninterleave_add(int P, const float* acc16, float* mean2D, float* opacity, float* mean3D, float* scale, float* rot,
                             hipStream_t s)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_uninterleave_add, dim3((P + 255) / 256), dim3(256), 0, s, P, acc16, mean2D, opacity, mean3D, scale, rot);
}

}  // namespace lr
