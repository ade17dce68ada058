// render_fwd.hip -- per-tile front-to-back alpha blending with depth output, for gfx950.
//
// Replaces FORWARD::render / renderCUDA (RAST/cuda_rasterizer/forward.cu:261-391).  Per-pixel
// semantics are the reference's exactly (same skip tests, same 0.99 clamp, same T < 1e-4 stop,
// same depth normalisation).  The kernel is VALU-issue bound (measured: ~100 % VALU busy, HBM idle), so
// the execution shape is chosen to minimise wave-instructions per pixel x Gaussian pair:
//   * one 128-thread workgroup per 16x16 tile = 2 wave64; a wave owns a 16x8 half tile and every LANE owns
//     TWO pixels (same row, 8 columns apart).  All per-pixel arithmetic is written on 2-vectors and compiles
//     to packed FP32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  A packed instruction costs twice the issue
//     cycles of a scalar one on MI355X (tools/valu_microbench.hip: 453 vs 912 G wave-instr/s), so the flops are
//     the same; what two pixels per lane buys is half the per-candidate overhead per pixel (candidate walk, LDS
//     reads and, in the backward, the cross-lane reduction).  The blend is branch-free: a pixel that skips a
//     Gaussian blends it with weight 0, which is arithmetically identical to the reference's `continue`.
//   * every field the inner loop touches is staged in LDS (the reference re-reads colour and depth
//     from global memory per pixel per Gaussian, forward.cu:359, 364);
//   * two-level loop per wave.  CULL: 64 staged Gaussians at a time, one per LANE, are tested against the
//     wave's 16x8 pixel box with the exact box-minimum of the conic quadratic (common.h box_hit);
//     __ballot turns the result into a 64-bit candidate mask.  BLEND: the wave walks only the set bits
//     (s_ff1), in list order, all lanes evaluating the same Gaussian from broadcast ds_reads.  A
//     non-candidate cannot reach alpha >= 1/255 on any pixel of the box (margin in common.h), so skipping
//     it is exactly the reference's `continue` (forward.cu:338-339);
//   * per-wave early termination via 64-bit __ballot (the reference only stops per block);
//   * tiles are assigned to workgroups so that each XCD (its own 4 MiB L2) renders a contiguous
//     band of the image and re-uses the GaussRecs of Gaussians that straddle neighbouring tiles.
#include "common.h"

namespace lr {

namespace {

constexpr int BATCH = 128;          // staged Gaussians per round = threads per workgroup
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int swizzled_tile(int num_tiles)
{
    // workgroup b lands on XCD (b % 8); give XCD x the contiguous tile band [x*per, (x+1)*per)
    const int per = (num_tiles + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}

__global__ void __launch_bounds__(BATCH)
k_render_fwd(int W, int H, int gx, int num_tiles, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ inst_gid,
             const GaussRec* __restrict__ rec,
             const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             float* __restrict__ out_color, float* __restrict__ out_depth)
{
    __shared__ float4 s_q0[BATCH];      // x, y, Ap = -0.5 conic a, Bp = -conic b      (common.h gauss_power)
    __shared__ float4 s_q1[BATCH];      // Cp = -0.5 conic c, qmax (cull threshold), opacity, depth
    __shared__ float4 s_q2[BATCH];      // r, g, b, -
    __shared__ float2 s_q3[BATCH];      // -b/c, -b/a (edge minimiser slopes for box_hit)
    __shared__ int s_wdone[2];

    const int tile = swizzled_tile(num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y + w * 8;          // this wave's 16x8 box
    const int pxA = x0 + (l & 7), pxB = pxA + 8, py = y0 + (l >> 3);
    const bool insA = pxA < W && py < H, insB = pxB < W && py < H;
    const v2f pxf = { (float)pxA, (float)pxB };
    const float pyf = (float)py;
    const float bx0 = (float)x0, bx1 = (float)(x0 + 15), by0 = (float)y0, by1 = (float)(y0 + 7);

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    v2f T = { 1.0f, 1.0f };
    v2f Cr = { 0.f, 0.f }, Cg = Cr, Cb = Cr, Dacc = Cr, acc = { 0.000001f, 0.000001f };
    uint32_t lastA = 0, lastB = 0;
    bool doneA = !insA, doneB = !insB;
    bool wave_done = __ballot(!doneA || !doneB) == 0;

    for (int base = 0; base < total; base += BATCH) {
        // both half tiles finished?  (also the barrier that protects the LDS planes of the previous batch)
        if (l == 0) s_wdone[w] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wdone[0] + s_wdone[1] == 2) break;
        const int cnt = min(BATCH, total - base);
        if (tid < cnt) {
            const uint32_t id = inst_gid[point_list[range.x + base + tid]];   // list holds emission indices
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            s_q0[tid] = make_float4(a.x, a.y, -0.5f * a.z, -a.w);
            s_q1[tid] = make_float4(-0.5f * b.x, c.z, b.y, c.y);
            s_q2[tid] = make_float4(b.z, b.w, c.x, 0.f);
            s_q3[tid] = make_float2(-a.w / b.x, -a.w / a.z);
        }
        __syncthreads();
        if (wave_done) continue;

        for (int sb = 0; sb < cnt; sb += 64) {
            // CULL: lane l tests staged Gaussian sb+l against this wave's 16x8 box
            bool hit = false;
            {
                const int j = sb + l;
                if (j < cnt) {
                    const float4 a = s_q0[j];
                    const float4 b = s_q1[j];
                    const float2 r = s_q3[j];
                    hit = box_hit(a.x, a.y, -2.0f * a.z, -a.w, -2.0f * b.x, r.x, r.y, b.y, bx0, bx1, by0, by1);   // exact inverses
                }
            }
            uint64_t mask = __ballot(hit);
            // BLEND: walk the candidates in list order; two pixels per lane, packed, branch-free
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int j = sb + k;
                const float4 a = s_q0[j];
                const float4 b = s_q1[j];
                const float4 c = s_q2[j];
                const v2f dx = a.x - pxf;
                const float dy = a.y - pyf;
                const v2f power = gauss_power(a.z, a.w, b.x, dx, dy);
                const v2f G = { __expf(power.x), __expf(power.y) };
                const v2f alpha = __builtin_elementwise_min(v2f{ 0.99f, 0.99f }, b.z * G);
                const v2f test_T = T * (1.0f - alpha);
                // reference order of tests (forward.cu:331-347): power > 0 -> skip; alpha < 1/255 -> skip;
                // T*(1-alpha) < 1e-4 -> pixel done (this Gaussian is NOT blended)
                const bool passA = !doneA && power.x <= 0.0f && alpha.x >= 1.0f / 255.0f;
                const bool passB = !doneB && power.y <= 0.0f && alpha.y >= 1.0f / 255.0f;
                const bool stopA = passA && test_T.x < 0.0001f;
                const bool stopB = passB && test_T.y < 0.0001f;
                doneA = doneA || stopA;
                doneB = doneB || stopB;
                const bool useA = passA != stopA, useB = passB != stopB;     // stop implies pass: xor of the lane masks
                const v2f wgt = { useA ? alpha.x * T.x : 0.f, useB ? alpha.y * T.y : 0.f };
                Cr += c.x * wgt; Cg += c.y * wgt; Cb += c.z * wgt;
                Dacc += b.w * wgt;
                acc += wgt;
                T.x = useA ? test_T.x : T.x;
                T.y = useB ? test_T.y : T.y;
                const uint32_t pos1 = (uint32_t)(base + j + 1);
                lastA = useA ? pos1 : lastA;
                lastB = useB ? pos1 : lastB;
            }
            if (__ballot(!doneA || !doneB) == 0) { wave_done = true; break; }   // all 128 pixels are finished
        }
    }

    const size_t N = (size_t)W * H;
    if (insA) {
        const size_t pix = (size_t)py * W + pxA;
        final_T[pix] = T.x;
        n_contrib[pix] = lastA;
        out_color[pix] = Cr.x + T.x * bg[0];
        out_color[N + pix] = Cg.x + T.x * bg[1];
        out_color[2 * N + pix] = Cb.x + T.x * bg[2];
        out_depth[pix] = (acc.x > 0.5f) ? Dacc.x / acc.x : 0.0f;         // forward.cu:384-388
    }
    if (insB) {
        const size_t pix = (size_t)py * W + pxB;
        final_T[pix] = T.y;
        n_contrib[pix] = lastB;
        out_color[pix] = Cr.y + T.y * bg[0];
        out_color[N + pix] = Cg.y + T.y * bg[1];
        out_color[2 * N + pix] = Cb.y + T.y * bg[2];
        out_depth[pix] = (acc.y > 0.5f) ? Dacc.y / acc.y : 0.0f;
    }
}

}  // namespace

void launch_render_fwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const uint32_t* inst_gid, const GaussRec* rec, const float* bg, float* final_T,
                       uint32_t* n_contrib, float* out_color, float* out_depth, hipStream_t s)
{
    const int num_tiles = gx * gy;
    if (num_tiles <= 0) return;
    const int grid = ((num_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(BATCH), 0, s, W, H, gx, num_tiles, ranges, point_list, inst_gid,
                       rec, bg, final_T, n_contrib, out_color, out_depth);
}

}  // namespace lr
