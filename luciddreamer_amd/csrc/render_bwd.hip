// render_bwd.hip -- backward of the per-tile alpha blend, for gfx950.
//
// Replaces BACKWARD::render / renderCUDA (RAST/cuda_rasterizer/backward.cu:399-586).  The
// per-pixel recursion is the reference's (back-to-front, T un-blended by division, colour
// recursion through accum_rec, background term, 0.99 clamp not differentiated, depth gradient
// ignored).  What differs is how the nine per-(pixel,Gaussian) gradient terms reach memory:
//
//   reference: 9 global float atomicAdd per contributing pixel x Gaussian pair (:537-583)
//   here:      wave64 DPP reduction (quad_perm / row_mirror / row_bcast, no LDS traffic)
//              -> one LDS float add per term per wave into a per-batch accumulator
//              -> one global atomic per term per Gaussian per TILE, issued by 256 threads in
//                 parallel into one 48-byte GradRec (a single cache line) instead of four arrays.
//   That is 256x fewer global atomics, and Gaussians no pixel of the wave can reach (same
//   conservative exponent test as the forward) or that lie behind every pixel's last contributor
//   are skipped wave-uniformly before any of that work.
#include "common.h"

namespace lr {

namespace {

constexpr int BATCH = 256;

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// Sum over the 64 lanes; the total is valid in lanes 48..63 (read it from lane 63).
__device__ __forceinline__ float wave_sum_hi(float v)
{
    v += dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);           // row_half_mirror
    v += dpp<0x140>(v);           // row_mirror          -> every lane: sum of its 16-lane row
    v += dpp<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

__device__ __forceinline__ int swizzled_tile(int num_tiles)
{
    const int per = (num_tiles + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}

__global__ void __launch_bounds__(256)
k_render_bwd(int W, int H, int gx, int num_tiles, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const GaussRec* __restrict__ rec,
             const float* __restrict__ bg, const float* __restrict__ final_Ts,
             const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
             GradRec* __restrict__ grad)
{
    __shared__ float4 s_q0[BATCH];      // x, y, conic a, conic b
    __shared__ float4 s_q1[BATCH];      // conic c, reject threshold, opacity, -
    __shared__ float4 s_q2[BATCH];      // r, g, b, -
    __shared__ uint32_t s_id[BATCH];
    __shared__ float s_acc[BATCH][9];   // per-batch gradient accumulator (4 waves add into it)
    __shared__ uint32_t s_touched[BATCH];

    const int tile = swizzled_tile(num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int px = tx * TILE_X + (w & 1) * 8 + (l & 7);
    const int py = ty * TILE_Y + (w >> 1) * 8 + (l >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix = (size_t)py * W + px;
    const size_t N = (size_t)W * H;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    const uint32_t wave_last = wave_max_u32(last_contributor);       // nothing behind this matters to the wave
    float dLr = 0.f, dLg = 0.f, dLb = 0.f;
    if (inside) { dLr = dL_dpix[pix]; dLg = dL_dpix[N + pix]; dLb = dL_dpix[2 * N + pix]; }
    const float bg_dot = bg[0] * dLr + bg[1] * dLg + bg[2] * dLb;
    float acr = 0.f, acg = 0.f, acb = 0.f;      // accum_rec
    float last_alpha = 0.f, lcr = 0.f, lcg = 0.f, lcb = 0.f;
    const float ddelx_dx = (float)(0.5 * W);     // backward.cu:473-474 (double product, rounded once)
    const float ddely_dy = (float)(0.5 * H);

    // block-uniform: the deepest contributor of any pixel in the tile; batches entirely behind it are skipped
    __shared__ uint32_t s_wlast[4];
    if (l == 0) s_wlast[w] = wave_last;
    __syncthreads();
    const uint32_t tile_last = max(max(s_wlast[0], s_wlast[1]), max(s_wlast[2], s_wlast[3]));

    for (int base = 0; base < total; base += BATCH) {
        // staged element i <-> list position pos = total-1-base-i (back to front)
        const int cnt = min(BATCH, total - base);
        const int pos_hi = total - 1 - base;             // position of staged element 0
        const int pos_lo = pos_hi - (cnt - 1);
        if ((uint32_t)pos_lo >= tile_last) continue;     // whole batch lies behind every last contributor
        __syncthreads();                                  // previous batch fully consumed / flushed
        if (tid < cnt) {
            const uint32_t id = point_list[range.x + (pos_hi - tid)];
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            const float thr = -__logf(255.0f * b.y) - 0.01f;
            s_q0[tid] = a;
            s_q1[tid] = make_float4(b.x, thr, b.y, 0.f);
            s_q2[tid] = make_float4(b.z, b.w, c.x, 0.f);
            s_id[tid] = id;
        }
#pragma unroll
        for (int k = 0; k < 9; k++) s_acc[tid][k] = 0.f;
        s_touched[tid] = 0;
        __syncthreads();

        for (int j = 0; j < cnt; j++) {
            const uint32_t pos = (uint32_t)(pos_hi - j);
            if (pos >= wave_last) continue;               // wave-uniform (backward.cu:500-502)
            const float4 a = s_q0[j];
            const float4 b = s_q1[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const bool cand = pos < last_contributor && power <= 0.0f && power >= b.y;
            if (__ballot(cand) == 0) continue;

            float g_dmx = 0.f, g_dmy = 0.f, g_dca = 0.f, g_dcb = 0.f, g_dcc = 0.f, g_dop = 0.f;
            float g_dr = 0.f, g_dg = 0.f, g_db = 0.f;
            bool contrib = false;
            if (cand) {
                const float G = expf(power);
                const float alpha = fminf(0.99f, b.z * G);
                if (alpha >= 1.0f / 255.0f) {
                    contrib = true;
                    const float4 c = s_q2[j];
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.f;
                    acr = last_alpha * lcr + (1.f - last_alpha) * acr; lcr = c.x;
                    dL_dalpha += (c.x - acr) * dLr; g_dr = dchannel_dcolor * dLr;
                    acg = last_alpha * lcg + (1.f - last_alpha) * acg; lcg = c.y;
                    dL_dalpha += (c.y - acg) * dLg; g_dg = dchannel_dcolor * dLg;
                    acb = last_alpha * lcb + (1.f - last_alpha) * acb; lcb = c.z;
                    dL_dalpha += (c.z - acb) * dLb; g_db = dchannel_dcolor * dLb;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;

                    const float dL_dG = b.z * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * a.z - gdy * a.w;
                    const float dG_ddely = -gdy * b.x - gdx * a.w;
                    g_dmx = dL_dG * dG_ddelx * ddelx_dx;
                    g_dmy = dL_dG * dG_ddely * ddely_dy;
                    g_dca = -0.5f * gdx * dx * dL_dG;
                    g_dcb = -0.5f * gdx * dy * dL_dG;
                    g_dcc = -0.5f * gdy * dy * dL_dG;
                    g_dop = G * dL_dalpha;
                }
            }
            if (__ballot(contrib) == 0) continue;
            g_dmx = wave_sum_hi(g_dmx); g_dmy = wave_sum_hi(g_dmy);
            g_dca = wave_sum_hi(g_dca); g_dcb = wave_sum_hi(g_dcb); g_dcc = wave_sum_hi(g_dcc);
            g_dop = wave_sum_hi(g_dop);
            g_dr = wave_sum_hi(g_dr); g_dg = wave_sum_hi(g_dg); g_db = wave_sum_hi(g_db);
            if (l == 63) {
                float* dst = s_acc[j];
                atomicAdd(dst + 0, g_dmx); atomicAdd(dst + 1, g_dmy);
                atomicAdd(dst + 2, g_dca); atomicAdd(dst + 3, g_dcb); atomicAdd(dst + 4, g_dcc);
                atomicAdd(dst + 5, g_dop);
                atomicAdd(dst + 6, g_dr); atomicAdd(dst + 7, g_dg); atomicAdd(dst + 8, g_db);
                s_touched[j] = 1;
            }
        }
        __syncthreads();
        if (tid < cnt && s_touched[tid]) {
            float* dst = reinterpret_cast<float*>(grad + s_id[tid]);
#pragma unroll
            for (int k = 0; k < 9; k++) atomicAdd(dst + k, s_acc[tid][k]);
        }
    }
}

}  // namespace

void launch_render_bwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const GaussRec* rec, const float* bg, const float* final_T,
                       const uint32_t* n_contrib, const float* dL_dpix, GradRec* grad, hipStream_t s)
{
    const int num_tiles = gx * gy;
    if (num_tiles <= 0) return;
    const int grid = ((num_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(k_render_bwd, dim3(grid), dim3(256), 0, s, W, H, gx, num_tiles, ranges, point_list, rec, bg,
                       final_T, n_contrib, dL_dpix, grad);
}

}  // namespace lr
