// render_fwd.hip -- per-tile front-to-back alpha blending with depth output, for gfx950.
//
// Replaces FORWARD::render / renderCUDA (RAST/cuda_rasterizer/forward.cu:261-391).  Per-pixel
// semantics are the reference's exactly (same skip tests, same 0.99 clamp, same T < 1e-4 stop,
// same depth normalisation); the execution shape is CDNA4's:
//   * one 256-thread workgroup per 16x16 tile = 4 wave64, each wave owning an 8x8 pixel quadrant
//     (compact footprint -> more Gaussians can be rejected for the whole wave);
//   * every field the inner loop touches is staged in LDS as three float4 planes read with
//     broadcast ds_read_b128 (the reference re-reads colour and depth from global memory per pixel
//     per Gaussian, forward.cu:359, 364);
//   * wave-uniform rejection: a Gaussian whose exponent is below -log(255*opacity) (minus a safety
//     margin) for all 64 pixels cannot pass the alpha >= 1/255 test, so the wave skips the exp and
//     the blend for it with one v_cmp + s_cbranch; the exact per-pixel test still decides;
//   * per-wave early termination via 64-bit __ballot (the reference only stops per block).
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
             const uint32_t* __restrict__ point_list, const GaussRec* __restrict__ rec,
             const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             float* __restrict__ out_color, float* __restrict__ out_depth)
{
    __shared__ float4 s_q0[BATCH];      // x, y, conic a, conic b
    __shared__ float4 s_q1[BATCH];      // conic c, reject threshold, opacity, depth
    __shared__ float4 s_q2[BATCH];      // r, g, b, -

    const int tile = swizzled_tile(num_tiles);
    if (tile >= num_tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int px = tx * TILE_X + (w & 1) * 8 + (l & 7);
    const int py = ty * TILE_Y + (w >> 1) * 8 + (l >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f;
    float Cr = 0.f, Cg = 0.f, Cb = 0.f, Dacc = 0.f, acc = 0.000001f;
    uint32_t last_contributor = 0;
    bool done = !inside;

    for (int base = 0; base < total; base += BATCH) {
        if (__syncthreads_and(done)) break;
        const int cnt = min(BATCH, total - base);
        if (tid < cnt) {
            const uint32_t id = point_list[range.x + base + tid];
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            // alpha = min(0.99, o*exp(power)) >= 1/255 needs power >= -log(255*o); margin covers
            // the rounding of power/exp (|error| << 1e-3), so the test below is conservative.
            const float thr = -__logf(255.0f * b.y) - 0.01f;
            s_q0[tid] = a;
            s_q1[tid] = make_float4(b.x, thr, b.y, c.y);
            s_q2[tid] = make_float4(b.z, b.w, c.x, 0.f);
        }
        __syncthreads();

        for (int j = 0; j < cnt; j++) {
            if (__ballot(!done) == 0) break;                       // this wave's 64 pixels are finished
            const float4 a = s_q0[j];
            const float4 b = s_q1[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const bool cand = !done && power <= 0.0f && power >= b.y;
            if (__ballot(cand) == 0) continue;                     // nobody in the wave can reach 1/255
            if (cand) {
                const float alpha = fminf(0.99f, b.z * expf(power));
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
                       const GaussRec* rec, const float* bg, float* final_T,
                       uint32_t* n_contrib, float* out_color, float* out_depth, hipStream_t s)
{
    const int num_tiles = gx * gy;
    if (num_tiles <= 0) return;
    const int grid = ((num_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(256), 0, s, W, H, gx, num_tiles, ranges, point_list, rec, bg,
                       final_T, n_contrib, out_color, out_depth);
}

}  // namespace lr
