// preprocess.hip -- per-Gaussian forward stage for gfx950.
//
// Replaces FORWARD::preprocess / preprocessCUDA (RAST/cuda_rasterizer/forward.cu:155-256) and
// checkFrustum (RAST/cuda_rasterizer/rasterizer_impl.cu:54-66).  One streaming pass over the
// Gaussian attribute arrays: frustum cull, projection, 3D->2D covariance (EWA), conic, radius,
// tile rectangle, SH->RGB; output is ONE packed 48-byte GaussRec per Gaussian, its instance count after exact tile
// culling and, for the binning, the outcome of that culling as a bit mask over the tile rectangle (common.h HitRec) --
// the reference scatters the per-Gaussian data over seven arrays.
//
// This translation unit is compiled with -ffp-contract=off: every product/sum is rounded in the
// operand order the reference source uses (GLM column-major operator order), so that radii, tile
// rectangles, depths and conics are bit-identical to the CPU oracle and the discrete outputs
// (radii, num_rendered, per-tile order) can be compared exactly.  The kernel is HBM-bound, so the
// missing FMAs cost nothing.
#include <cstddef>
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
__device__ __forceinline__ V3 sh_fma(V3 acc, float s, const V3& c, bool sub = false)
{
    // acc (+|-) s * c, each product rounded separately (contraction is off in this TU)
    V3 r;
    if (sub) { r.x = acc.x - s * c.x; r.y = acc.y - s * c.y; r.z = acc.z - s * c.z; }
    else     { r.x = acc.x + s * c.x; r.y = acc.y + s * c.y; r.z = acc.z + s * c.z; }
    return r;
}

// Load K coefficient triples of Gaussian idx.  M==16 rows are 192 B (16-byte aligned): 12 x dwordx4.
template <int K>
__device__ __forceinline__ void load_sh(const float* __restrict__ shs, size_t idx, int M, V3 (&sh)[16])
{
    const float* base = shs + idx * (size_t)M * 3;
    constexpr int NF = 3 * K;
    float f[(NF + 3) / 4 * 4];
    if (((M * 3) & 3) == 0 && ((reinterpret_cast<uintptr_t>(shs) & 15) == 0)) {
        const float4* b4 = reinterpret_cast<const float4*>(base);
#pragma unroll
        for (int q = 0; q < (NF + 3) / 4; q++) {
            float4 v = b4[q];
            f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NF; q++) f[q] = base[q];
    }
#pragma unroll
    for (int k = 0; k < K; k++) { sh[k].x = f[3 * k]; sh[k].y = f[3 * k + 1]; sh[k].z = f[3 * k + 2]; }
}

// raw mode: coefficient 0 from features_dc [P,3], coefficients 1.. from features_rest [P,M-1,3]
template <int K>
__device__ __forceinline__ void load_sh_split(const float* __restrict__ dc, const float* __restrict__ rest, size_t idx, int M,
                                              V3 (&sh)[16])
{
    sh[0].x = dc[3 * idx]; sh[0].y = dc[3 * idx + 1]; sh[0].z = dc[3 * idx + 2];
    const float* base = rest + idx * (size_t)(M - 1) * 3;
#pragma unroll
    for (int k = 1; k < K; k++) { sh[k].x = base[3 * (k - 1)]; sh[k].y = base[3 * (k - 1) + 1]; sh[k].z = base[3 * (k - 1) + 2]; }
}

template <int DEG>
__device__ __forceinline__ V3 eval_sh(const float* __restrict__ shs, const float* __restrict__ sh_rest, size_t idx, int M, V3 dir,
                                      uint8_t& clamp_bits)
{
    V3 sh[16];
    if (sh_rest != nullptr) load_sh_split<(DEG + 1) * (DEG + 1)>(shs, sh_rest, idx, M, sh);
    else load_sh<(DEG + 1) * (DEG + 1)>(shs, idx, M, sh);
    V3 res = { SH_C0 * sh[0].x, SH_C0 * sh[0].y, SH_C0 * sh[0].z };
    if (DEG > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        res = sh_fma(res, SH_C1 * y, sh[1], true);
        res = sh_fma(res, SH_C1 * z, sh[2]);
        res = sh_fma(res, SH_C1 * x, sh[3], true);
        if (DEG > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = sh_fma(res, SH_C2[0] * xy, sh[4]);
            res = sh_fma(res, SH_C2[1] * yz, sh[5]);
            res = sh_fma(res, SH_C2[2] * (2.0f * zz - xx - yy), sh[6]);
            res = sh_fma(res, SH_C2[3] * xz, sh[7]);
            res = sh_fma(res, SH_C2[4] * (xx - yy), sh[8]);
            if (DEG > 2) {
                res = sh_fma(res, SH_C3[0] * y * (3.0f * xx - yy), sh[9]);
                res = sh_fma(res, SH_C3[1] * xy * z, sh[10]);
                res = sh_fma(res, SH_C3[2] * y * (4.0f * zz - xx - yy), sh[11]);
                res = sh_fma(res, SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), sh[12]);
                res = sh_fma(res, SH_C3[4] * x * (4.0f * zz - xx - yy), sh[13]);
                res = sh_fma(res, SH_C3[5] * z * (xx - yy), sh[14]);
                res = sh_fma(res, SH_C3[6] * x * (xx - 3.0f * yy), sh[15]);
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamp_bits = (uint8_t)((res.x < 0 ? 1 : 0) | (res.y < 0 ? 2 : 0) | (res.z < 0 ? 4 : 0));
    res.x = fmaxf(res.x, 0.0f); res.y = fmaxf(res.y, 0.0f); res.z = fmaxf(res.z, 0.0f);
    return res;
}

// Tile rectangle of a splat (RAST/cuda_rasterizer/auxiliary.h:46-56).
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy,
                                          int& minx, int& miny, int& maxx, int& maxy)
{
    minx = min(gx, max(0, (int)((px - radius) / TILE_X)));
    miny = min(gy, max(0, (int)((py - radius) / TILE_Y)));
    maxx = min(gx, max(0, (int)((px + radius + TILE_X - 1) / TILE_X)));
    maxy = min(gy, max(0, (int)((py + radius + TILE_Y - 1) / TILE_Y)));
}

// RAW (lr_forward_raw) is a template parameter so that the standard path keeps its register budget (100 VGPRs,
// 4 waves/SIMD; the split SH loader of raw mode needs 130)
constexpr int PP_THREADS = 128;                 // Gaussians per workgroup (measured: 64 -> 0.047, 128 -> 0.039, 256 -> 0.042, 512 -> 0.042 ms on C3)
constexpr int PP_WAVES = PP_THREADS / 64;

template <bool RAW>
__global__ void __launch_bounds__(PP_THREADS)
k_preprocess(ViewParams vp, const float* __restrict__ means3D, const float* __restrict__ scales,
             const float* __restrict__ rotations, const float* __restrict__ opacities,
             const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
             const float* __restrict__ colors_precomp, int prefiltered,
             int* __restrict__ radii, GaussRec* __restrict__ rec, uint8_t* __restrict__ clamped,
             uint32_t* __restrict__ tiles_touched, uint4* __restrict__ hitrec,
             uint32_t* __restrict__ depth_key, GeomHeader* hdr, uint32_t binning_capacity,
             uint32_t* __restrict__ chunk_sums)
{
    // Phase 1, one thread per Gaussian: the near-plane test (auxiliary.h:152-162).  Survivors are compacted, in index
    // order, into LDS; phase 2 runs the ~600-instruction projection / covariance / SH body on dense lanes only (on a
    // camera path about half of a scene is behind the camera, so half of the waves of a block skip it entirely).
    __shared__ uint32_t s_list[PP_THREADS];
    __shared__ uint32_t s_wcnt[PP_WAVES];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    // the prefilter trap lives behind the chunk sums, in the library's own scratch (zero between forwards: api.hip)
    uint32_t* const trap = chunk_sums != nullptr ? chunk_sums + 4 * ((size_t)(vp.P + SCAN_TILE - 1) / SCAN_TILE) : &hdr->prefilter_trap;
    const float* __restrict__ V = vp.view;
    const float* __restrict__ Pm = vp.proj;
    {
        const bool in_range = gid < vp.P;
        const size_t li = in_range ? (size_t)gid : 0;
        const float vz1 = V[2] * means3D[3 * li] + V[6] * means3D[3 * li + 1] + V[10] * means3D[3 * li + 2] + V[14];
        const bool pass = in_range && !(vz1 <= 0.2f);
        if (in_range && !pass) {
            if (prefiltered) *trap = 1;
            radii[gid] = 0; tiles_touched[gid] = 0;
            if (depth_key) depth_key[gid] = 0xFFFFFFFFu;
        }
        const uint64_t m = __ballot(pass);
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        if (l == 0) s_wcnt[w] = (uint32_t)__popcll(m);
        lds_barrier();
        uint32_t before = 0;
        for (int i = 0; i < w; i++) before += s_wcnt[i];
        if (pass) s_list[before + __popcll(m & ((1ull << l) - 1ull))] = (uint32_t)gid;
        lds_barrier();
    }
    uint32_t n_pass = 0;
#pragma unroll
    for (int i = 0; i < PP_WAVES; i++) n_pass += s_wcnt[i];
    if ((threadIdx.x & ~63u) >= n_pass) return;             // whole wave has nothing to do
    const bool live = threadIdx.x < n_pass;
    const int idx = live ? (int)s_list[threadIdx.x] : 0;

    int radius_out = 0;
    uint32_t tiles_out = 0;               // tile instances this Gaussian will emit (after exact tile culling)
    uint32_t area_ref = 0;                // the reference's tiles_touched (rectangle area)
    uint32_t key_out = 0xFFFFFFFFu;       // culled Gaussians sort to the end and emit nothing

    const size_t li = (size_t)idx;
    const float px_w = means3D[3 * li], py_w = means3D[3 * li + 1], pz_w = means3D[3 * li + 2];
    // view-space point (auxiliary.h:58-66)
    const float vx = V[0] * px_w + V[4] * py_w + V[8] * pz_w + V[12];
    const float vy = V[1] * px_w + V[5] * py_w + V[9] * pz_w + V[13];
    const float vz = V[2] * px_w + V[6] * py_w + V[10] * pz_w + V[14];

    do {
        if (!live) break;
        // clip-space projection (forward.cu:196-200)
        const float hx = Pm[0] * px_w + Pm[4] * py_w + Pm[8] * pz_w + Pm[12];
        const float hy = Pm[1] * px_w + Pm[5] * py_w + Pm[9] * pz_w + Pm[13];
        const float hw = Pm[3] * px_w + Pm[7] * py_w + Pm[11] * pz_w + Pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float ndcx = hx * p_w, ndcy = hy * p_w;

        // 3D covariance (forward.cu:118-152), quaternion used as given
        float c3[6];
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            float sx = scales[3 * (size_t)idx], sy = scales[3 * (size_t)idx + 1], sz = scales[3 * (size_t)idx + 2];
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            float r = q.x, x = q.y, y = q.z, z = q.w;
            if (RAW) {
                sx = act_scale(sx); sy = act_scale(sy); sz = act_scale(sz);
                const float inv = act_quat_inv_norm(r, x, y, z);
                r *= inv; x *= inv; y *= inv; z *= inv;
            }
            M3 S = { { { vp.scale_modifier * sx, 0, 0 }, { 0, vp.scale_modifier * sy, 0 }, { 0, 0, vp.scale_modifier * sz } } };
            M3 R = { { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
                       { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
                       { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } } };
            M3 Mm = m3_mul(S, R);
            M3 Sig = m3_mul(m3_t(Mm), Mm);
            c3[0] = Sig.c[0][0]; c3[1] = Sig.c[0][1]; c3[2] = Sig.c[0][2];
            c3[3] = Sig.c[1][1]; c3[4] = Sig.c[1][2]; c3[5] = Sig.c[2][2];
        }

        // EWA 2D covariance (forward.cu:74-113)
        const float limx = 1.3f * vp.tan_fovx, limy = 1.3f * vp.tan_fovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        M3 J = { { { vp.focal_x / vz, 0.0f, -(vp.focal_x * tx) / (vz * vz) },
                   { 0.0f, vp.focal_y / vz, -(vp.focal_y * ty) / (vz * vz) },
                   { 0, 0, 0 } } };
        M3 Wm = { { { V[0], V[4], V[8] }, { V[1], V[5], V[9] }, { V[2], V[6], V[10] } } };
        M3 T = m3_mul(Wm, J);
        M3 Vrk = { { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } } };
        M3 cov = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
        const float ca = cov.c[0][0] + 0.3f, cb = cov.c[0][1], cc = cov.c[1][1] + 0.3f;

        const float det = ca * cc - cb * cb;                   // forward.cu:219-223
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float con_a = cc * det_inv, con_b = -cb * det_inv, con_c = ca * det_inv;

        const float mid = 0.5f * (ca + cc);                    // forward.cu:229-232
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        // ndc2Pix evaluates in double (auxiliary.h:41-44)
        const float pix = (float)(((ndcx + 1.0) * vp.W - 1.0) * 0.5);
        const float piy = (float)(((ndcy + 1.0) * vp.H - 1.0) * 0.5);
        int minx, miny, maxx, maxy;
        tile_rect(pix, piy, (int)my_radius, vp.gx, vp.gy, minx, miny, maxx, maxy);
        const uint32_t area = (uint32_t)(maxx - minx) * (uint32_t)(maxy - miny);
        if (area == 0) break;

        V3 rgb;
        uint8_t cbits = 0;
        if (colors_precomp != nullptr) {
            rgb.x = colors_precomp[3 * (size_t)idx]; rgb.y = colors_precomp[3 * (size_t)idx + 1];
            rgb.z = colors_precomp[3 * (size_t)idx + 2];
        } else {
            V3 dir = { px_w - vp.campos[0], py_w - vp.campos[1], pz_w - vp.campos[2] };
            const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
            dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
            switch (vp.D) {
                case 0: rgb = eval_sh<0>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
                case 1: rgb = eval_sh<1>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
                case 2: rgb = eval_sh<2>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
                default: rgb = eval_sh<3>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
            }
        }
        clamped[idx] = cbits;

        GaussRec g;
        g.x = pix; g.y = piy; g.ca = con_a; g.cb = con_b;
        g.cc = con_c; g.opacity = RAW ? act_opacity(opacities[idx]) : opacities[idx]; g.r = rgb.x; g.g = rgb.y;
        g.b = rgb.z; g.depth = vz; g.qmax = cull_qmax(g.opacity); g.pad1 = 0.f;
        float4* dst = reinterpret_cast<float4*>(rec + idx);
        dst[0] = make_float4(g.x, g.y, g.ca, g.cb);
        dst[1] = make_float4(g.cc, g.opacity, g.r, g.g);
        dst[2] = make_float4(g.b, g.depth, g.qmax, 0.f);

        radius_out = (int)my_radius;
        area_ref = area;
        key_out = __float_as_uint(vz);                         // vz > 0.2: bit order == float order
        // exact tile culling (common.h): count the tiles of the rectangle that can actually matter
        // ... and leave the outcome per tile for the binning (common.h HitRec)
        unsigned long long mask = 0ull;
        if (area > CULL_MAX_TILES) {
            tiles_out = area;
        } else {
            const float qmax = g.qmax;
            const float r_c = -con_b / con_c, r_a = -con_b / con_a;
            uint32_t cnt = 0, j = 0;
            for (int ty = miny; ty < maxy; ty++)
                for (int tx = minx; tx < maxx; tx++, j++)
                    if (tile_hit(pix, piy, con_a, con_b, con_c, r_c, r_a, qmax, tx, ty)) { cnt++; mask |= 1ull << (j & 63u); }
            tiles_out = cnt;
        }
        if (tiles_out != 0)
            hitrec[idx] = make_uint4((uint32_t)mask, (uint32_t)(mask >> 32), hit_geo(minx, miny, maxx - minx, area, vp.hit_origin_limit), key_out);
    } while (false);

    // this wave's share of the compaction's chunk sums (tilebin.hip k_compact_write): emitting Gaussians, instances,
    // the reference's rectangle areas.  Integer atomics: the result does not depend on their order.  All 128 Gaussians
    // of the workgroup lie in one SCAN_TILE chunk (zero when the forward starts: api.hip StreamScratch); dead lanes carry zeros.
    if (chunk_sums != nullptr) {
        uint32_t inst = tiles_out, ref = area_ref;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { inst += __shfl_xor(inst, off); ref += __shfl_xor(ref, off); }
        const uint32_t cnt = (uint32_t)__popcll(__ballot(tiles_out != 0));
        if ((threadIdx.x & 63) == 0 && ref != 0) {
            uint32_t* dst = chunk_sums + 4 * ((size_t)blockIdx.x * PP_THREADS / SCAN_TILE);
            if (cnt) { atomicAdd(dst, cnt); atomicAdd(dst + 1, inst); }
            atomicAdd(dst + 2, ref);
        }
    }
    if (!live) return;
    radii[idx] = radius_out;
    tiles_touched[idx] = tiles_out;
    if (depth_key) depth_key[idx] = key_out;
}

// ---------------------------------------------------------------------------------------------------------------
// Pooled version (round 3), for views that see a small part of the scene.  On a camera path ~50 % of the Gaussians pass
// the near plane and ~9 % end up with a non-empty tile rectangle, so with one thread per Gaussian the covariance chain runs
// on half-empty waves and the spherical-harmonics / tile-count part on lanes 18 % live.  Here a workgroup owns a POOL of
// 512 consecutive Gaussians and compacts twice through LDS:
//   phase 1   all 512: near-plane test (auxiliary.h:152-162) -> dense list of the passers
//   phase 2   passers, dense lanes: projection, 3D -> 2D covariance, conic, radius, tile rectangle (forward.cu:196-232);
//             the ones with a non-empty rectangle park their screen-space values in LDS
//   phase 3a  parked survivors, dense lanes: SH -> RGB (12 x dwordx4 per lane), GaussRec store
//   phase 3b  exact tile culling with ONE (Gaussian, tile) pair per thread: a lane looping over its own rectangle makes
//             its wave wait for the largest rectangle among 64, and the ~46 survivors of a pool fill one of the four waves
//   finally   radii and tiles_touched of the whole pool leave in two fully coalesced passes, chunk sums by three integer
//             atomics per workgroup.
// Arithmetic is the same code in the same order as in k_preprocess (project_gaussian / colour_and_record are its loop body
// cut in two), so every record stays bit-identical to the oracle.  Measured on MI355X (profiles/r03g_ab_preprocess.json):
// C3 (9 % visible) 36.5 -> 28.1 us; the thread-per-Gaussian kernel stays ahead when most of the scene is in view (dense
// 1 M cloud 73 vs 79 us: the compactions buy nothing there) and below ~400 k Gaussians (too few pools to hide a pool's
// longer chain: 100 k Gaussians 25 vs 35 us) -- launch_preprocess picks per call.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PL_THREADS = 256;
constexpr int PL_POOL = 512;
constexpr int PL_WAVES = PL_THREADS / 64;
constexpr int PL_ROUNDS = PL_POOL / PL_THREADS;

struct Projected { float pix, piy, con_a, con_b, con_c, vz; int radius; };

// forward.cu:196-232 for one Gaussian that passed the near plane.  false: culled (det == 0 or empty tile rectangle).
template <bool RAW>
__device__ __forceinline__ bool project_gaussian(const ViewParams& vp, int idx, const float* __restrict__ means3D,
                                                 const float* __restrict__ scales, const float* __restrict__ rotations,
                                                 const float* __restrict__ cov3D_precomp, Projected& out)
{
    const float* __restrict__ V = vp.view;
    const float* __restrict__ Pm = vp.proj;
    const size_t li = (size_t)idx;
    const float px_w = means3D[3 * li], py_w = means3D[3 * li + 1], pz_w = means3D[3 * li + 2];
    // view-space point (auxiliary.h:58-66)
    const float vx = V[0] * px_w + V[4] * py_w + V[8] * pz_w + V[12];
    const float vy = V[1] * px_w + V[5] * py_w + V[9] * pz_w + V[13];
    const float vz = V[2] * px_w + V[6] * py_w + V[10] * pz_w + V[14];
    // clip-space projection (forward.cu:196-200)
    const float hx = Pm[0] * px_w + Pm[4] * py_w + Pm[8] * pz_w + Pm[12];
    const float hy = Pm[1] * px_w + Pm[5] * py_w + Pm[9] * pz_w + Pm[13];
    const float hw = Pm[3] * px_w + Pm[7] * py_w + Pm[11] * pz_w + Pm[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float ndcx = hx * p_w, ndcy = hy * p_w;

    // 3D covariance (forward.cu:118-152), quaternion used as given
    float c3[6];
    if (cov3D_precomp != nullptr) {
#pragma unroll
        for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * (size_t)idx + i];
    } else {
        float sx = scales[3 * (size_t)idx], sy = scales[3 * (size_t)idx + 1], sz = scales[3 * (size_t)idx + 2];
        const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
        float r = q.x, x = q.y, y = q.z, z = q.w;
        if (RAW) {
            sx = act_scale(sx); sy = act_scale(sy); sz = act_scale(sz);
            const float inv = act_quat_inv_norm(r, x, y, z);
            r *= inv; x *= inv; y *= inv; z *= inv;
        }
        M3 S = { { { vp.scale_modifier * sx, 0, 0 }, { 0, vp.scale_modifier * sy, 0 }, { 0, 0, vp.scale_modifier * sz } } };
        M3 R = { { { 1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y) },
                   { 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x) },
                   { 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y) } } };
        M3 Mm = m3_mul(S, R);
        M3 Sig = m3_mul(m3_t(Mm), Mm);
        c3[0] = Sig.c[0][0]; c3[1] = Sig.c[0][1]; c3[2] = Sig.c[0][2];
        c3[3] = Sig.c[1][1]; c3[4] = Sig.c[1][2]; c3[5] = Sig.c[2][2];
    }

    // EWA 2D covariance (forward.cu:74-113)
    const float limx = 1.3f * vp.tan_fovx, limy = 1.3f * vp.tan_fovy;
    const float txtz = vx / vz, tytz = vy / vz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    M3 J = { { { vp.focal_x / vz, 0.0f, -(vp.focal_x * tx) / (vz * vz) },
               { 0.0f, vp.focal_y / vz, -(vp.focal_y * ty) / (vz * vz) },
               { 0, 0, 0 } } };
    M3 Wm = { { { V[0], V[4], V[8] }, { V[1], V[5], V[9] }, { V[2], V[6], V[10] } } };
    M3 T = m3_mul(Wm, J);
    M3 Vrk = { { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } } };
    M3 cov = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
    const float ca = cov.c[0][0] + 0.3f, cb = cov.c[0][1], cc = cov.c[1][1] + 0.3f;

    const float det = ca * cc - cb * cb;                   // forward.cu:219-223
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    out.con_a = cc * det_inv; out.con_b = -cb * det_inv; out.con_c = ca * det_inv;

    const float mid = 0.5f * (ca + cc);                    // forward.cu:229-232
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    // ndc2Pix evaluates in double (auxiliary.h:41-44)
    out.pix = (float)(((ndcx + 1.0) * vp.W - 1.0) * 0.5);
    out.piy = (float)(((ndcy + 1.0) * vp.H - 1.0) * 0.5);
    out.radius = (int)my_radius;
    out.vz = vz;
    int minx, miny, maxx, maxy;
    tile_rect(out.pix, out.piy, out.radius, vp.gx, vp.gy, minx, miny, maxx, maxy);
    return (uint32_t)(maxx - minx) * (uint32_t)(maxy - miny) != 0u;
}

// forward.cu:236-255 for a survivor: colour and record.  Returns the tile rectangle (origin, width, area) and the cull
// threshold for the exact tile count that follows.
template <bool RAW>
__device__ __forceinline__ void colour_and_record(const ViewParams& vp, int idx, const Projected& pj,
                                                  const float* __restrict__ means3D, const float* __restrict__ opacities,
                                                  const float* __restrict__ shs, const float* __restrict__ colors_precomp,
                                                  GaussRec* __restrict__ rec, uint8_t* __restrict__ clamped,
                                                  int& minx, int& miny, int& width, uint32_t& area, float& qmax)
{
    int maxx, maxy;
    tile_rect(pj.pix, pj.piy, pj.radius, vp.gx, vp.gy, minx, miny, maxx, maxy);
    width = maxx - minx;
    area = (uint32_t)(maxx - minx) * (uint32_t)(maxy - miny);
    V3 rgb;
    uint8_t cbits = 0;
    if (colors_precomp != nullptr) {
        rgb.x = colors_precomp[3 * (size_t)idx]; rgb.y = colors_precomp[3 * (size_t)idx + 1];
        rgb.z = colors_precomp[3 * (size_t)idx + 2];
    } else {
        const size_t li = (size_t)idx;
        V3 dir = { means3D[3 * li] - vp.campos[0], means3D[3 * li + 1] - vp.campos[1], means3D[3 * li + 2] - vp.campos[2] };
        const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
        dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
        switch (vp.D) {
            case 0: rgb = eval_sh<0>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
            case 1: rgb = eval_sh<1>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
            case 2: rgb = eval_sh<2>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
            default: rgb = eval_sh<3>(shs, RAW ? vp.sh_rest : nullptr, idx, vp.M, dir, cbits); break;
        }
    }
    clamped[idx] = cbits;

    const float opacity = RAW ? act_opacity(opacities[idx]) : opacities[idx];
    qmax = cull_qmax(opacity);
    float4* dst = reinterpret_cast<float4*>(rec + idx);
    dst[0] = make_float4(pj.pix, pj.piy, pj.con_a, pj.con_b);
    dst[1] = make_float4(pj.con_c, opacity, rgb.x, rgb.y);
    dst[2] = make_float4(rgb.z, pj.vz, qmax, 0.f);
}

// 5 workgroups per CU (29.7 KB of LDS each; 80 VGPRs, no scratch); the split SH loader of raw mode needs 126 registers
template <bool RAW>
__global__ void __launch_bounds__(PL_THREADS) __attribute__((amdgpu_waves_per_eu(RAW ? 4 : 5, 8)))
k_preprocess_pool(ViewParams vp, const float* __restrict__ means3D, const float* __restrict__ scales,
                  const float* __restrict__ rotations, const float* __restrict__ opacities,
                  const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                  const float* __restrict__ colors_precomp, int prefiltered,
                  int* __restrict__ radii, GaussRec* __restrict__ rec, uint8_t* __restrict__ clamped,
                  uint32_t* __restrict__ tiles_touched, uint4* __restrict__ hitrec, GeomHeader* hdr,
                  uint32_t binning_capacity, uint32_t* __restrict__ chunk_sums)
{
    __shared__ uint16_t s_near[PL_POOL];                     // pool-local ids of the near-plane passers, index order
    __shared__ int s_radius[PL_POOL];                        // results of the whole pool (0 for everything culled)
    __shared__ uint32_t s_tiles[PL_POOL];
    __shared__ float s_mid[6][PL_POOL];                      // parked survivors of phase 2: pix, piy, conic a b c, vz (3b: qmax)
    __shared__ uint32_t s_mid_id[PL_POOL];                   // ... and their pool-local ids
    __shared__ uint32_t s_tests[PL_POOL];                    // tile tests per survivor -> inclusive prefix
    __shared__ uint2 s_rect[PL_POOL];                        // tile rectangle of a survivor: min x | min y << 16, width
    __shared__ uint32_t s_mask[PL_POOL][2];                  // outcome of its tile tests (common.h HitRec), rectangles <= 64 tiles
    __shared__ uint32_t s_wcnt[PL_ROUNDS * PL_WAVES];
    __shared__ uint32_t s_nmid;
    __shared__ uint32_t s_sum[3];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int base = blockIdx.x * PL_POOL;
    uint32_t* const trap = chunk_sums != nullptr ? chunk_sums + 4 * ((size_t)(vp.P + SCAN_TILE - 1) / SCAN_TILE) : &hdr->prefilter_trap;
    if (tid == 0) { s_nmid = 0u; s_sum[0] = 0u; s_sum[1] = 0u; s_sum[2] = 0u; }
    const float* __restrict__ V = vp.view;

    // ---- phase 1: near-plane test of the whole pool
    bool pass[PL_ROUNDS];
    uint64_t m[PL_ROUNDS];
#pragma unroll
    for (int r = 0; r < PL_ROUNDS; r++) {
        const int loc = r * PL_THREADS + tid, gid = base + loc;
        const bool in_range = gid < vp.P;
        const size_t li = in_range ? (size_t)gid : 0;
        const float vz1 = V[2] * means3D[3 * li] + V[6] * means3D[3 * li + 1] + V[10] * means3D[3 * li + 2] + V[14];
        pass[r] = in_range && !(vz1 <= 0.2f);
        if (in_range && !pass[r] && prefiltered) *trap = 1;
        s_radius[loc] = 0; s_tiles[loc] = 0u;
        m[r] = __ballot(pass[r]);
        if (l == 0) s_wcnt[r * PL_WAVES + w] = (uint32_t)__popcll(m[r]);
    }
    lds_barrier();
    uint32_t n_near = 0;
#pragma unroll
    for (int r = 0; r < PL_ROUNDS; r++) {
        uint32_t before = n_near;
#pragma unroll
        for (int i = 0; i < PL_WAVES; i++) { if (i < w) before += s_wcnt[r * PL_WAVES + i]; n_near += s_wcnt[r * PL_WAVES + i]; }
        if (pass[r]) s_near[before + __popcll(m[r] & ((1ull << l) - 1ull))] = (uint16_t)(r * PL_THREADS + tid);
    }
    lds_barrier();

    // ---- phase 2: projection / covariance / rectangle on dense lanes; survivors are parked (in arrival order: nothing
    // observable depends on it -- every output is per Gaussian, the chunk sums are integer sums)
    for (uint32_t i0 = 0; i0 < n_near; i0 += PL_THREADS) {
        const uint32_t i = i0 + (uint32_t)tid;
        Projected pj;
        bool ok = false;
        uint32_t loc = 0;
        if (i < n_near) {
            loc = s_near[i];
            ok = project_gaussian<RAW>(vp, base + (int)loc, means3D, scales, rotations, cov3D_precomp, pj);
        }
        const uint64_t mk = __ballot(ok);
        uint32_t wbase = 0;
        if (l == 0 && mk != 0ull) wbase = atomicAdd(&s_nmid, (uint32_t)__popcll(mk));
        wbase = (uint32_t)__shfl((int)wbase, 0);
        if (ok) {
            const uint32_t slot = wbase + (uint32_t)__popcll(mk & ((1ull << l) - 1ull));
            s_mid[0][slot] = pj.pix; s_mid[1][slot] = pj.piy; s_mid[2][slot] = pj.con_a; s_mid[3][slot] = pj.con_b;
            s_mid[4][slot] = pj.con_c; s_mid[5][slot] = pj.vz;
            s_mid_id[slot] = loc; s_radius[loc] = pj.radius;
        }
    }
    lds_barrier();

    // ---- phase 3a: colour and record on dense lanes; every survivor leaves its rectangle for the count
    const uint32_t n_mid = s_nmid;
    uint32_t my_ref = 0;
    for (uint32_t i0 = 0; i0 < n_mid; i0 += PL_THREADS) {
        const uint32_t i = i0 + (uint32_t)tid;
        uint32_t tests = 0;
        if (i < n_mid) {
            Projected pj;
            pj.pix = s_mid[0][i]; pj.piy = s_mid[1][i]; pj.con_a = s_mid[2][i]; pj.con_b = s_mid[3][i];
            pj.con_c = s_mid[4][i]; pj.vz = s_mid[5][i];
            const uint32_t loc = s_mid_id[i];
            pj.radius = s_radius[loc];
            int minx, miny, width; uint32_t area; float qmax;
            colour_and_record<RAW>(vp, base + (int)loc, pj, means3D, opacities, shs, colors_precomp, rec, clamped,
                                   minx, miny, width, area, qmax);
            my_ref += area;
            // rectangles of more than CULL_MAX_TILES tiles are emitted unculled (common.h): nothing to test
            if (area > CULL_MAX_TILES) s_tiles[loc] = area; else tests = area;
            // the binning's record: origin, width and depth bits now, the mask after the tests (two 8-byte halves)
            reinterpret_cast<uint2*>(hitrec + (base + (int)loc))[1] =
                make_uint2(hit_geo(minx, miny, width, area, vp.hit_origin_limit), __float_as_uint(pj.vz));
            s_mask[i][0] = 0u; s_mask[i][1] = 0u;
            s_mid[5][i] = qmax;                                   // vz is in the record now; the slot carries the threshold
            // gx, gy <= 65535 (launcher); width | reciprocal << 16, the latter only meaningful for tested rectangles (<= 96 tiles)
            s_rect[i] = make_uint2((uint32_t)minx | ((uint32_t)miny << 16),
                                   (uint32_t)width | ((tests != 0u ? 32768u / (uint32_t)width + 1u : 0u) << 16));
        }
        s_tests[i0 + tid] = tests;                                // entries between n_mid and the end of the round: zero
    }
    lds_barrier();
    // ---- phase 3b: exact tile culling (common.h): the (Gaussian, tile) pairs of the whole pool laid end to end --
    // inclusive scan of the test counts, every thread finds the owner of its pair by bisection -- and counted with integer
    // LDS atomics (order-free)
    {
        const uint32_t n_scan = (n_mid + PL_THREADS - 1) / PL_THREADS * PL_THREADS;     // <= PL_POOL
        uint32_t run = 0;
        for (uint32_t i0 = 0; i0 < n_scan; i0 += PL_THREADS) {
            const uint32_t v = s_tests[i0 + tid];
            uint32_t inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (l >= off) inc += t; }
            if (l == 63) s_wcnt[w] = inc;
            lds_barrier();
            uint32_t before = run;
#pragma unroll
            for (int q = 0; q < PL_WAVES; q++) { if (q < w) before += s_wcnt[q]; run += s_wcnt[q]; }
            s_tests[i0 + tid] = before + inc;                     // inclusive prefix
            lds_barrier();
        }
        const uint32_t total = run;
        for (uint32_t t = (uint32_t)tid; t < total; t += PL_THREADS) {
            uint32_t lo = 0, hi = n_mid - 1;                      // first entry whose inclusive prefix exceeds t
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_tests[mid] > t) hi = mid; else lo = mid + 1; }
            const uint32_t j = t - (lo ? s_tests[lo - 1] : 0u);
            const uint2 rc = s_rect[lo];
            // j / width without a division: (j * (32768 / width + 1)) >> 15 is exact for j < 96, width <= 96 (checked
            // exhaustively); the reciprocal (<= 32769) rides in the upper half of the width word (phase 3a)
            const uint32_t wdt = rc.y & 0xffffu, jr = (j * (rc.y >> 16)) >> 15;
            const int ty = (int)(rc.x >> 16) + (int)jr, tx = (int)(rc.x & 0xffffu) + (int)(j - jr * wdt);
            const float con_a = s_mid[2][lo], con_b = s_mid[3][lo], con_c = s_mid[4][lo];
            const float r_c = -con_b / con_c, r_a = -con_b / con_a;
            if (tile_hit(s_mid[0][lo], s_mid[1][lo], con_a, con_b, con_c, r_c, r_a, s_mid[5][lo], tx, ty)) {
                atomicAdd(&s_tiles[s_mid_id[lo]], 1u);           // still zero for these (phase 1)
                atomicOr(&s_mask[lo][(j >> 5) & 1u], 1u << (j & 31u));   // (bits of 65..96-tile rectangles alias: mask unused)
            }
        }
    }
    lds_barrier();
    for (uint32_t i = (uint32_t)tid; i < n_mid; i += PL_THREADS)
        reinterpret_cast<uint2*>(hitrec + (base + (int)s_mid_id[i]))[0] = make_uint2(s_mask[i][0], s_mask[i][1]);
    uint32_t my_cnt = 0, my_inst = 0;
#pragma unroll
    for (int r = 0; r < PL_ROUNDS; r++) {
        const uint32_t tt = s_tiles[r * PL_THREADS + tid];
        my_cnt += tt != 0 ? 1u : 0u; my_inst += tt;
    }
    // this workgroup's share of the compaction's chunk sums (tilebin.hip k_compact_write): emitting Gaussians, instances,
    // the reference's rectangle areas.  Integer atomics: the result does not depend on their order.  The pool lies in one
    // SCAN_TILE chunk (zero when the forward starts: api.hip StreamScratch).
    if (chunk_sums != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            my_cnt += __shfl_xor(my_cnt, off); my_inst += __shfl_xor(my_inst, off); my_ref += __shfl_xor(my_ref, off);
        }
        if (l == 0 && (my_ref | my_inst) != 0) { atomicAdd(&s_sum[0], my_cnt); atomicAdd(&s_sum[1], my_inst); atomicAdd(&s_sum[2], my_ref); }
    }
    lds_barrier();
    if (chunk_sums != nullptr && tid == 0 && s_sum[2] != 0) {
        uint32_t* dst = chunk_sums + 4 * ((size_t)blockIdx.x * PL_POOL / SCAN_TILE);
        if (s_sum[0]) { atomicAdd(dst, s_sum[0]); atomicAdd(dst + 1, s_sum[1]); }
        atomicAdd(dst + 2, s_sum[2]);
    }
    // ---- the pool's radii and instance counts, coalesced
#pragma unroll
    for (int r = 0; r < PL_ROUNDS; r++) {
        const int loc = r * PL_THREADS + tid, gid = base + loc;
        if (gid < vp.P) { radii[gid] = s_radius[loc]; tiles_touched[gid] = s_tiles[loc]; }
    }
}

__global__ void __launch_bounds__(256)
k_mark_visible(int P, const float* __restrict__ means3D, const float* __restrict__ V, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float x = means3D[3 * (size_t)idx], y = means3D[3 * (size_t)idx + 1], z = means3D[3 * (size_t)idx + 2];
    const float vz = V[2] * x + V[6] * y + V[10] * z + V[14];
    present[idx] = (vz <= 0.2f) ? 0 : 1;                       // auxiliary.h:152-162 with prefiltered=false
}

}  // namespace

void launch_preprocess(const ViewParams& vp, const float* means3D, const float* scales, const float* rotations,
                       const float* opacities, const float* shs, const float* cov3D_precomp,
                       const float* colors_precomp, bool prefiltered, int* radii, GaussRec* rec,
                       uint8_t* clamped, uint32_t* tiles_touched, uint4* hitrec, uint32_t* depth_key,
                       GeomHeader* hdr, uint32_t binning_capacity, uint32_t* chunk_sums, bool sparse_view_hint, hipStream_t s)
{
    static_assert(SCAN_TILE % PP_THREADS == 0, "a preprocess workgroup must lie inside one compaction chunk");
    static_assert(SCAN_TILE % PL_POOL == 0, "a preprocess pool must lie inside one compaction chunk");
    if (vp.P <= 0) return;
    // Two kernels with identical results.  The pooled one pays when the view sees a small part of a large scene (see its
    // header); `sparse_view_hint` is what the caller knows from earlier views of this scene (api.hip view_hint: visible
    // fraction of the last forward whose counts reached the host).  lr_tune_set("preprocess", 0 / 1) forces one (A/B runs).
    const int forced = tune_get(TUNE_PREPROCESS);
    const bool pooled = forced >= 0 ? forced != 0 : (sparse_view_hint && vp.P >= 400000);
    if (pooled && depth_key == nullptr && vp.gx <= 65535 && vp.gy <= 65535) {
        dim3 grid((vp.P + PL_POOL - 1) / PL_POOL), block(PL_THREADS);
        if (vp.raw)
            hipLaunchKernelGGL(k_preprocess_pool<true>, grid, block, 0, s, vp, means3D, scales, rotations, opacities, shs,
                               cov3D_precomp, colors_precomp, prefiltered ? 1 : 0, radii, rec, clamped, tiles_touched, hitrec,
                               hdr, binning_capacity, chunk_sums);
        else
            hipLaunchKernelGGL(k_preprocess_pool<false>, grid, block, 0, s, vp, means3D, scales, rotations, opacities, shs,
                               cov3D_precomp, colors_precomp, prefiltered ? 1 : 0, radii, rec, clamped, tiles_touched, hitrec,
                               hdr, binning_capacity, chunk_sums);
        return;
    }
    dim3 grid((vp.P + PP_THREADS - 1) / PP_THREADS), block(PP_THREADS);
    if (vp.raw)
        hipLaunchKernelGGL(k_preprocess<true>, grid, block, 0, s, vp, means3D, scales, rotations, opacities, shs,
                           cov3D_precomp, colors_precomp, prefiltered ? 1 : 0, radii, rec, clamped, tiles_touched,
                           hitrec, depth_key, hdr, binning_capacity, chunk_sums);
    else
        hipLaunchKernelGGL(k_preprocess<false>, grid, block, 0, s, vp, means3D, scales, rotations, opacities, shs,
                           cov3D_precomp, colors_precomp, prefiltered ? 1 : 0, radii, rec, clamped, tiles_touched,
                           hitrec, depth_key, hdr, binning_capacity, chunk_sums);
}

void launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present,
                         hipStream_t s)
{
    (void)proj;   // the reference computes the clip-space point but tests only view-space z
    if (P <= 0) return;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
}

}  // namespace lr
