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

// Per-pixel state of the blend, kept in SCALAR registers: a packed FP32 instruction costs the issue cycles of two scalar
// ones on MI355X (tools/valu_microbench.hip), so nothing is lost by not packing, and a step that touches only ONE of
// a lane's two pixels costs half.
struct PixState { float T, Cr, Cg, Cb, D, acc; uint32_t last; bool done; };

__device__ __forceinline__ void fwd_pixel(PixState& p, const float Ap, const float Bd, const float Cdd, const float dx,
                                          const float op, const float cr, const float cg, const float cb, const float depth,
                                          const uint32_t pos1)
{
    const float power = gauss_power1(Ap, Bd, Cdd, dx);
    const float alpha = fminf(0.99f, op * __expf(power));
    const float test_T = p.T * (1.0f - alpha);
    // reference order of tests (forward.cu:331-347): power > 0 -> skip; alpha < 1/255 -> skip;
    // T*(1-alpha) < 1e-4 -> pixel done (this Gaussian is NOT blended)
    const bool pass = !p.done && power <= 0.0f && alpha >= 1.0f / 255.0f;
    const bool stop = pass && test_T < 0.0001f;
    const bool use = pass != stop;                                      // stop implies pass
    p.done = p.done || stop;
    const float wgt = use ? alpha * p.T : 0.f;
    p.Cr += cr * wgt; p.Cg += cg * wgt; p.Cb += cb * wgt; p.D += depth * wgt; p.acc += wgt;
    p.T = use ? test_T : p.T;
    p.last = use ? pos1 : p.last;
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
    const float pxfA = (float)pxA, pxfB = (float)pxB;
    const float pyf = (float)py;
    const float bx0 = (float)x0, bx1 = (float)(x0 + 15), by0 = (float)y0, by1 = (float)(y0 + 7);

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    PixState A = { 1.0f, 0.f, 0.f, 0.f, 0.f, 0.000001f, 0u, !insA }, B = { 1.0f, 0.f, 0.f, 0.f, 0.f, 0.000001f, 0u, !insB };
    bool wave_done = __ballot(!A.done || !B.done) == 0;

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
            // CULL: lane l tests staged Gaussian sb+l against the two 8x8 quadrants of this wave's 16x8 box (pixel A of
            // every lane lies in the left quadrant, pixel B in the right one)
            bool hitL = false, hitR = false;
            {
                const int j = sb + l;
                if (j < cnt) {
                    const float4 a = s_q0[j];
                    const float4 b = s_q1[j];
                    const float2 r = s_q3[j];
                    const float ca = -2.0f * a.z, cb = -a.w, cc = -2.0f * b.x;                  // exact inverses
                    hitL = box_hit(a.x, a.y, ca, cb, cc, r.x, r.y, b.y, bx0, bx0 + 7.0f, by0, by1);
                    hitR = box_hit(a.x, a.y, ca, cb, cc, r.x, r.y, b.y, bx0 + 8.0f, bx1, by0, by1);
                }
            }
            const uint64_t maskL = __ballot(hitL), maskR = __ballot(hitR);
            uint64_t mask = maskL | maskR;
            // BLEND: walk the candidates in list order; a candidate that missed a quadrant skips that pixel of every lane
            // (exactly the reference's `continue`: no pixel there can reach alpha >= 1/255)
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int j = sb + k;
                const float4 a = s_q0[j];
                const float4 b = s_q1[j];
                const float4 c = s_q2[j];
                const uint32_t pos1 = (uint32_t)(base + j + 1);
                const float dy = a.y - pyf;
                const float Bd = a.w * dy, Cdd = (b.x * dy) * dy;                              // common.h gauss_power
                if ((maskL >> k) & 1ull) fwd_pixel(A, a.z, Bd, Cdd, a.x - pxfA, b.z, c.x, c.y, c.z, b.w, pos1);
                if ((maskR >> k) & 1ull) fwd_pixel(B, a.z, Bd, Cdd, a.x - pxfB, b.z, c.x, c.y, c.z, b.w, pos1);
            }
            if (__ballot(!A.done || !B.done) == 0) { wave_done = true; break; }   // all 128 pixels are finished
        }
    }

    const size_t N = (size_t)W * H;
    if (insA) {
        const size_t pix = (size_t)py * W + pxA;
        final_T[pix] = A.T;
        n_contrib[pix] = A.last;
        out_color[pix] = A.Cr + A.T * bg[0];
        out_color[N + pix] = A.Cg + A.T * bg[1];
        out_color[2 * N + pix] = A.Cb + A.T * bg[2];
        out_depth[pix] = (A.acc > 0.5f) ? A.D / A.acc : 0.0f;         // forward.cu:384-388
    }
    if (insB) {
        const size_t pix = (size_t)py * W + pxB;
        final_T[pix] = B.T;
        n_contrib[pix] = B.last;
        out_color[pix] = B.Cr + B.T * bg[0];
        out_color[N + pix] = B.Cg + B.T * bg[1];
        out_color[2 * N + pix] = B.Cb + B.T * bg[2];
        out_depth[pix] = (B.acc > 0.5f) ? B.D / B.acc : 0.0f;
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
