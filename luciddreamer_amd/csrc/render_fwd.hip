// render_fwd.hip -- per-tile front-to-back alpha blending with depth output, for gfx950.
//
// Replaces FORWARD::render / renderCUDA (RAST/cuda_rasterizer/forward.cu:261-391).  Per-pixel
// semantics are the reference's exactly (same skip tests, same 0.99 clamp, same T < 1e-4 stop,
// same depth normalisation); the execution shape is CDNA4's (the kernel is VALU-issue bound, so the
// design minimises wave-instructions per pixel x Gaussian pair):
//   * one 256-thread workgroup per 16x16 tile = 4 wave64, each wave owning an 8x8 pixel quadrant;
//   * every field the inner loop touches is staged in LDS (the reference re-reads colour and depth
//     from global memory per pixel per Gaussian, forward.cu:359, 364);
//   * two-level loop per wave.  CULL: 64 staged Gaussians at a time, one per LANE, are tested against
//     the wave's 8x8 quadrant with the exact box-minimum of the conic quadratic (common.h box_hit);
//     __ballot turns the result into a 64-bit candidate mask.  BLEND: the wave walks only the set
//     bits (s_ff1), in list order, with all 64 pixels evaluating the same Gaussian from broadcast
//     ds_reads.  A non-candidate costs < 1 VALU instruction per wave instead of a full exponent
//     evaluation per pixel; it cannot reach alpha >= 1/255 on any pixel of the quadrant (margin in
//     common.h), so skipping it is exactly the reference's `continue` (forward.cu:338-339);
//   * per-wave early termination via 64-bit __ballot (the reference only stops per block);
//   * tiles are assigned to workgroups so that each XCD (its own 4 MiB L2) renders a contiguous
//     band of the image and re-uses the GaussRecs of Gaussians that straddle neighbouring tiles.
#include "common.h"

namespace lr {

namespace {

constexpr int BATCH = 256;

__device__ __forceinline__ int swizzled_tile(int num_tiles)
{
    // workgroup b lands on XCD (b % 8); give XCD x the contiguous tile band [x*per, (x+1)*per)
    const int per = (num_tiles + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}

__global__ void __launch_bounds__(256)
k_render_fwd(int W, int H, int gx, int num_tiles, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ inst_gid,
             const GaussRec* __restrict__ rec,
             const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             float* __restrict__ out_color, float* __restrict__ out_depth)
{
    __shared__ float4 s_q0[BATCH];      // x, y, conic a, conic b
    __shared__ float4 s_q1[BATCH];      // conic c, qmax (cull threshold), opacity, depth
    __shared__ float4 s_q2[BATCH];      // r, g, b, -
    __shared__ float2 s_q3[BATCH];      // -b/c, -b/a (edge minimiser slopes for box_hit)
    __shared__ int s_wdone[4];

    const int tile = swizzled_tile(num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int qx = tx * TILE_X + (w & 1) * 8, qy = ty * TILE_Y + (w >> 1) * 8;     // quadrant origin
    const int px = qx + (l & 7), py = qy + (l >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)qx, bx1 = (float)(qx + 7), by0 = (float)qy, by1 = (float)(qy + 7);

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f;
    float Cr = 0.f, Cg = 0.f, Cb = 0.f, Dacc = 0.f, acc = 0.000001f;
    uint32_t last_contributor = 0;
    bool done = !inside;
    bool wave_done = __ballot(!done) == 0;

    for (int base = 0; base < total; base += BATCH) {
        // all four quadrants finished?  (also the barrier that protects the LDS planes of the previous batch)
        if (l == 0) s_wdone[w] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wdone[0] + s_wdone[1] + s_wdone[2] + s_wdone[3] == 4) break;
        const int cnt = min(BATCH, total - base);
        if (tid < cnt) {
            const uint32_t id = inst_gid[point_list[range.x + base + tid]];   // list holds emission indices
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            s_q0[tid] = a;
            s_q1[tid] = make_float4(b.x, c.z, b.y, c.y);
            s_q2[tid] = make_float4(b.z, b.w, c.x, 0.f);
            s_q3[tid] = make_float2(-a.w / b.x, -a.w / a.z);
        }
        __syncthreads();
        if (wave_done) continue;

        for (int sb = 0; sb < cnt; sb += 64) {
            // CULL: lane l tests staged Gaussian sb+l against this wave's 8x8 quadrant
            bool hit = false;
            {
                const int j = sb + l;
                if (j < cnt) {
                    const float4 a = s_q0[j];
                    const float4 b = s_q1[j];
                    const float2 r = s_q3[j];
                    hit = box_hit(a.x, a.y, a.z, a.w, b.x, r.x, r.y, b.y, bx0, bx1, by0, by1);
                }
            }
            uint64_t mask = __ballot(hit);
            // BLEND: walk the candidates in list order
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int j = sb + k;
                const float4 a = s_q0[j];
                const float4 b = s_q1[j];
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                if (!done && power <= 0.0f) {
                    const float alpha = fminf(0.99f, b.z * __expf(power));
                    if (alpha >= 1.0f / 255.0f) {
                        const float test_T = T * (1.0f - alpha);
                        if (test_T < 0.0001f) {
                            done = true;
                        } else {
                            const float4 c = s_q2[j];
                            const float wgt = alpha * T;
                            Cr += c.x * wgt; Cg += c.y * wgt; Cb += c.z * wgt;
                            Dacc += b.w * wgt;
                            acc += wgt;
                            T = test_T;
                            last_contributor = (uint32_t)(base + j + 1);
                        }
                    }
                }
            }
            if (__ballot(!done) == 0) { wave_done = true; break; }     // this wave's 64 pixels are finished
        }
    }

    if (inside) {
        const size_t pix = (size_t)py * W + px;
        const size_t N = (size_t)W * H;
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
        out_color[pix] = Cr + T * bg[0];
        out_color[N + pix] = Cg + T * bg[1];
        out_color[2 * N + pix] = Cb + T * bg[2];
        out_depth[pix] = (acc > 0.5f) ? Dacc / acc : 0.0f;         // forward.cu:384-388
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
    hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(256), 0, s, W, H, gx, num_tiles, ranges, point_list, inst_gid,
                       rec, bg, final_T, n_contrib, out_color, out_depth);
}

}  // namespace lr
