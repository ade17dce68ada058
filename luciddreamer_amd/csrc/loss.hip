// loss.hip -- fused L1 + DSSIM photometric loss and its gradient for gfx950.
//
// Replaces, for the training loop's loss (R/luciddreamer.py:301-304)
//     loss = (1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt))
// the reference's Python composition (R/utils/loss.py:18-69): five grouped 11x11 F.conv2d (121 taps each, the 2-D
// window is the outer product of a normalised 1-D Gaussian, sigma 1.5, zero padding 5), a dozen elementwise
// kernels, two reductions, and the autograd replay of all of it.  Here:
//   k_ssim_fwd : one pass over the image pair.  A 256-thread workgroup owns a 32x32 tile of one channel, stages
//                the 42x42 halo region of both images in LDS, runs the five window sums SEPARABLY (11 + 11 taps,
//                four adjacent outputs per thread per pass so that the sliding window re-uses LDS reads), forms the
//                SSIM value and the three partial derivatives dS/d(conv I), dS/d(conv I^2), dS/d(conv I*G) per
//                pixel, writes those three maps, and reduces sum(S) and sum|I-G| per workgroup (fixed order).
//   k_loss_final: sums the per-workgroup partials in a fixed order (double) -> {loss, l1, ssim}.
//   k_ssim_bwd : the adjoint of a symmetric zero-padded window sum is the same window sum, so
//                dL/dI(q) = -lambda/n * [ W*D1 + 2 I(q) W*D2 + G(q) W*D3 ](q) + (1-lambda)/n * sign(I-G)(q),
//                again separable through LDS, times the upstream scalar (device pointer, no host sync).
// HBM-bound by construction: ~60 B per pixel-channel over both passes; no atomics, deterministic.
#include "common.h"
#include <cmath>

namespace lr {

namespace {

constexpr int LT = 32;                 // tile edge (outputs)
constexpr int HALO = 5;                // window 11
constexpr int LR_IN = LT + 2 * HALO;   // 42
constexpr int LTHREADS = 256;
constexpr int NSTAGE = (LR_IN * LR_IN + LTHREADS - 1) / LTHREADS;   // halo elements staged per thread (7)

struct Win { float w[11]; };

// gaussian(11, 1.5) of R/utils/loss.py:26-28: exp in double, cast to float32, normalised in float32
Win make_window()
{
    Win g;
    float v[11], sum = 0.f;
    for (int x = 0; x < 11; x++) { v[x] = (float)std::exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); sum += v[x]; }
    for (int x = 0; x < 11; x++) g.w[x] = v[x] / sum;
    return g;
}

// Workgroups are dealt to the 8 XCDs round robin (blockIdx % 8) and each XCD has its own L2.  Neighbouring tiles share
// their halo rows, so every XCD gets a contiguous run of tiles (a band of the image): with the plain order the halo
// lines were fetched from HBM once per XCD that touched them (k_ssim_bwd: 339 MB fetched for 125 MB of maps).
// Returns the logical block (tile + channel * tiles) of this workgroup or -1 (grid padded to a multiple of 8).
__device__ __forceinline__ int xcd_band_block(int n_blocks)
{
    const int per_xcd = (n_blocks + 7) >> 3;
    const int lb = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
    return ((int)(blockIdx.x >> 3) < per_xcd && lb < n_blocks) ? lb : -1;
}

__device__ __forceinline__ float block_sum(float v, float* s_tmp)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    lds_barrier();
    if ((threadIdx.x & 63) == 0) s_tmp[w] = v;
    lds_barrier();
    return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

__global__ void __launch_bounds__(LTHREADS)
k_ssim_fwd(int H, int W, int tiles_x, int tiles_y, Win win, const float* __restrict__ img, const float* __restrict__ gt,
           float* __restrict__ D1, float* __restrict__ D2, float* __restrict__ D3, float2* __restrict__ partials,
           int n_blocks)
{
    // One LDS buffer, two tenants: the staged halo regions of I and G, then -- once every thread holds its horizontal
    // sums in registers -- the five horizontal maps.  28 KB instead of 42 KB per workgroup: 5 workgroups per CU, not 3
    // (the kernel is a chain of memory and LDS round trips; what it lacks is waves to hide them).
    __shared__ float s_raw[5 * LR_IN * (LT + 1)];
    float (*s_i)[LR_IN + 1] = reinterpret_cast<float (*)[LR_IN + 1]>(s_raw);
    float (*s_g)[LR_IN + 1] = reinterpret_cast<float (*)[LR_IN + 1]>(s_raw + LR_IN * (LR_IN + 1));
    float (*s_h)[LR_IN][LT + 1] = reinterpret_cast<float (*)[LR_IN][LT + 1]>(s_raw);     // I, G, I^2, G^2, I*G
    static_assert(2 * LR_IN * (LR_IN + 1) <= 5 * LR_IN * (LT + 1), "the halo regions must fit under the horizontal maps");
    __shared__ float s_tmp[4];

    const int lb = xcd_band_block(n_blocks);
    if (lb < 0) return;
    const int tile = lb % (tiles_x * tiles_y), ch = lb / (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * LT, y0 = (tile / tiles_x) * LT;
    const size_t plane = (size_t)ch * H * W;
    const int tid = threadIdx.x;

    // stage the halo region: ALL global loads of the thread are issued before the first LDS store (the workgroup's run
    // time is a chain of memory round trips at 3-5 waves per SIMD; a rolled loop pays one round trip per iteration).
    // Element p = tid + 256 i of the 42x42 region: (row, column) advance by (6, 4) per step with one carry -- one
    // integer division per thread instead of two per element (index arithmetic was a third of the kernel's instructions).
    float l1_part = 0.f;
    {
        const float* __restrict__ ip = img + plane;
        const float* __restrict__ gp = gt + plane;
        float ra[NSTAGE], rb[NSTAGE];
        int li[NSTAGE];                                       // LDS index, -1: nothing to store
        bool inner[NSTAGE];
        int ly = tid / LR_IN, lx = tid - ly * LR_IN;
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) {
            const int y = y0 + ly - HALO, x = x0 + lx - HALO;
            const bool in_region = tid + i * LTHREADS < LR_IN * LR_IN;
            const bool ok = in_region && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int q = ok ? y * W + x : 0;
            const float a = ip[q], b = gp[q];
            ra[i] = ok ? a : 0.f; rb[i] = ok ? b : 0.f;
            li[i] = in_region ? ly * (LR_IN + 1) + lx : -1;
            inner[i] = in_region && (unsigned)(ly - HALO) < (unsigned)LT && (unsigned)(lx - HALO) < (unsigned)LT;
            lx += LTHREADS % LR_IN; ly += LTHREADS / LR_IN;
            if (lx >= LR_IN) { lx -= LR_IN; ly += 1; }
        }
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) {
            if (li[i] >= 0) {
                (&s_i[0][0])[li[i]] = ra[i]; (&s_g[0][0])[li[i]] = rb[i];
                if (inner[i]) l1_part += fabsf(ra[i] - rb[i]);                          // outside the image = 0
            }
        }
    }
    lds_barrier();

    // horizontal pass: item = (row, group of 4 adjacent output columns); 336 items = up to two per thread, kept in
    // registers until every thread has read its inputs (the maps overwrite the halo regions)
    constexpr int HITEMS = LR_IN * (LT / 4), HROUNDS = (HITEMS + LTHREADS - 1) / LTHREADS;
    float hs[HROUNDS][5][4];
#pragma unroll
    for (int r = 0; r < HROUNDS; r++) {
        const int it = tid + r * LTHREADS;
        if (it < HITEMS) {
            const int row = it / (LT / 4), c0 = (it % (LT / 4)) * 4;
            float a[14], b[14];
#pragma unroll
            for (int k = 0; k < 14; k++) { a[k] = s_i[row][c0 + k]; b[k] = s_g[row][c0 + k]; }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const float w = win.w[k], u = a[o + k], v = b[o + k];
                    m1 += w * u; m2 += w * v; e11 += w * (u * u); e22 += w * (v * v); e12 += w * (u * v);
                }
                hs[r][0][o] = m1; hs[r][1][o] = m2; hs[r][2][o] = e11; hs[r][3][o] = e22; hs[r][4][o] = e12;
            }
        }
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < HROUNDS; r++) {
        const int it = tid + r * LTHREADS;
        if (it < HITEMS) {
            const int row = it / (LT / 4), c0 = (it % (LT / 4)) * 4;
#pragma unroll
            for (int m = 0; m < 5; m++)
#pragma unroll
                for (int o = 0; o < 4; o++) s_h[m][row][c0 + o] = hs[r][m][o];
        }
    }
    lds_barrier();

    // vertical pass: thread = (column, group of 4 adjacent output rows)
    const int col = tid % LT, r0 = (tid / LT) * 4;
    float acc[5][4];
#pragma unroll
    for (int m = 0; m < 5; m++) {
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; k++) v[k] = s_h[m][r0 + k][col];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) s += win.w[k] * v[o + k];
            acc[m][o] = s;
        }
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float ssim_part = 0.f;
    float* __restrict__ d1p = D1 + plane;
    float* __restrict__ d2p = D2 + plane;
    float* __restrict__ d3p = D3 + plane;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int y = y0 + r0 + o, x = x0 + col;
        if (y < H && x < W) {
            const float mu1 = acc[0][o], mu2 = acc[1][o];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = acc[2][o] - mu1_sq, s2 = acc[3][o] - mu2_sq, s12 = acc[4][o] - mu12;
            const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
            // two v_rcp_f32 (1 ulp) instead of three IEEE divisions: B1 >= C1, B2 >= C2 up to rounding, no special cases
            const float iB1 = __builtin_amdgcn_rcpf(B1), iB2 = __builtin_amdgcn_rcpf(B2);
            const float inv = iB1 * iB2;
            const float S = A1 * A2 * inv;
            ssim_part += S;
            // partial derivatives of S w.r.t. the window sums of I, I^2 and I*G (those of G, G^2 are not needed)
            const float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -S * iB1, dB2 = -S * iB2;
            const int q = y * W + x;
            d1p[q] = dA1 * 2.f * mu2 + dB1 * 2.f * mu1 - dB2 * 2.f * mu1 - dA2 * 2.f * mu2;
            d2p[q] = dB2;
            d3p[q] = 2.f * dA2;
        }
    }
    const float st = block_sum(ssim_part, s_tmp);
    const float lt = block_sum(l1_part, s_tmp);
    if (tid == 0) partials[lb] = make_float2(st, lt);
}

// {loss, l1, ssim} from the per-workgroup partials, fixed order, one workgroup of LTHREADS threads
__device__ __forceinline__ void loss_final(int n_blocks, double n_elems, float lambda, const float2* __restrict__ partials,
                                           float* __restrict__ out, double* s_a, double* s_b)
{
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += LTHREADS) { const float2 p = partials[i]; a += p.x; b += p.y; }
    s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    lds_barrier();
    for (int off = LTHREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { s_a[threadIdx.x] += s_a[threadIdx.x + off]; s_b[threadIdx.x] += s_b[threadIdx.x + off]; }
        lds_barrier();
    }
    if (threadIdx.x == 0) {
        const float ssim = (float)(s_a[0] / n_elems), l1 = (float)(s_b[0] / n_elems);
        out[0] = (1.0f - lambda) * l1 + lambda * (1.0f - ssim);
        out[1] = l1;
        out[2] = ssim;
    }
}

__global__ void __launch_bounds__(LTHREADS)
k_loss_final(int n_blocks, double n_elems, float lambda, const float2* __restrict__ partials, float* __restrict__ out)
{
    __shared__ double s_a[LTHREADS], s_b[LTHREADS];
    loss_final(n_blocks, n_elems, lambda, partials, out, s_a, s_b);
}

__global__ void __launch_bounds__(LTHREADS)
k_ssim_bwd(int H, int W, int tiles_x, int tiles_y, Win win, float lambda, float inv_n, const float* __restrict__ upstream,
           const float* __restrict__ w_ssim, const float* __restrict__ img, const float* __restrict__ gt, const float* __restrict__ D1,
           const float* __restrict__ D2, const float* __restrict__ D3, float* __restrict__ grad,
           const float2* __restrict__ partials, int n_blocks, double n_elems, float* __restrict__ out3)
{
    // as in k_ssim_fwd: the three horizontal maps take over the LDS of the staged halo regions (22 KB instead of 38 KB)
    __shared__ float s_raw[3 * LR_IN * (LR_IN + 1)];
    static_assert(sizeof(float) * 3 * LR_IN * (LR_IN + 1) >= sizeof(double) * 2 * LTHREADS, "scratch of the final sum");
    // fused step (views_core): the loss value's final sum rides in workgroup 0 instead of a launch of its own
    if (out3 != nullptr && blockIdx.x == 0) {
        double* s_a = reinterpret_cast<double*>(s_raw);
        loss_final(n_blocks, n_elems, lambda, partials, out3, s_a, s_a + LTHREADS);
        lds_barrier();
    }
    float (*s_d)[LR_IN][LR_IN + 1] = reinterpret_cast<float (*)[LR_IN][LR_IN + 1]>(s_raw);
    float (*s_h)[LR_IN][LT + 1] = reinterpret_cast<float (*)[LR_IN][LT + 1]>(s_raw);
    const int lb = xcd_band_block(n_blocks);
    if (lb < 0) return;
    const int tile = lb % (tiles_x * tiles_y), ch = lb / (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * LT, y0 = (tile / tiles_x) * LT;
    const size_t plane = (size_t)ch * H * W;
    const int tid = threadIdx.x;

    // the pixel's own I and G (needed only in the last lines) are requested first, together with the halo loads: as the
    // kernel's final dependent loads they cost every workgroup one more memory round trip
    const int col = tid % LT, r0 = (tid / LT) * 4;
    const float* __restrict__ ip = img + plane;
    const float* __restrict__ gp = gt + plane;
    float own_i[4], own_g[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int y = y0 + r0 + o, x = x0 + col;
        const int q = (y < H && x < W) ? y * W + x : 0;
        own_i[o] = ip[q]; own_g[o] = gp[q];
    }
    {
        const float* __restrict__ p1 = D1 + plane;
        const float* __restrict__ p2 = D2 + plane;
        const float* __restrict__ p3 = D3 + plane;
        float r1[NSTAGE], r2[NSTAGE], r3[NSTAGE];
        int li[NSTAGE];
        int ly = tid / LR_IN, lx = tid - ly * LR_IN;           // element tid + 256 i: (row, column) += (6, 4) with carry
#pragma unroll
        for (int i = 0; i < NSTAGE; i++) {                   // all loads first (see k_ssim_fwd)
            const int y = y0 + ly - HALO, x = x0 + lx - HALO;
            const bool in_region = tid + i * LTHREADS < LR_IN * LR_IN;
            const bool ok = in_region && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int q = ok ? y * W + x : 0;
            const float a = p1[q], b = p2[q], c = p3[q];
            r1[i] = ok ? a : 0.f; r2[i] = ok ? b : 0.f; r3[i] = ok ? c : 0.f;
            li[i] = in_region ? ly * (LR_IN + 1) + lx : -1;
            lx += LTHREADS % LR_IN; ly += LTHREADS / LR_IN;
            if (lx >= LR_IN) { lx -= LR_IN; ly += 1; }
        }
#pragma unroll
        for (int i = 0; i < NSTAGE; i++)
            if (li[i] >= 0) { (&s_d[0][0][0])[li[i]] = r1[i]; (&s_d[1][0][0])[li[i]] = r2[i]; (&s_d[2][0][0])[li[i]] = r3[i]; }
    }
    lds_barrier();
    constexpr int HITEMS = LR_IN * (LT / 4), HROUNDS = (HITEMS + LTHREADS - 1) / LTHREADS;
    float hs[HROUNDS][3][4];
#pragma unroll
    for (int r = 0; r < HROUNDS; r++) {
        const int it = tid + r * LTHREADS;
        if (it < HITEMS) {
            const int row = it / (LT / 4), c0 = (it % (LT / 4)) * 4;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                float v[14];
#pragma unroll
                for (int k = 0; k < 14; k++) v[k] = s_d[m][row][c0 + k];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; k++) sum += win.w[k] * v[o + k];
                    hs[r][m][o] = sum;
                }
            }
        }
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < HROUNDS; r++) {
        const int it = tid + r * LTHREADS;
        if (it < HITEMS) {
            const int row = it / (LT / 4), c0 = (it % (LT / 4)) * 4;
#pragma unroll
            for (int m = 0; m < 3; m++)
#pragma unroll
                for (int o = 0; o < 4; o++) s_h[m][row][c0 + o] = hs[r][m][o];
        }
    }
    lds_barrier();
    float acc[3][4];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        float v[14];
#pragma unroll
        for (int k = 0; k < 14; k++) v[k] = s_h[m][r0 + k][col];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) s += win.w[k] * v[o + k];
            acc[m][o] = s;
        }
    }
    const float up = upstream != nullptr ? upstream[0] : 1.0f;
    // two-weight form (lr_l1_dssim_backward_weights): upstream = dL/d l1, w_ssim = dL/d ssim, both device scalars -- the
    // caller composed the two means itself, with whatever weights
    const float k_ssim = w_ssim != nullptr ? inv_n * w_ssim[0] : -lambda * inv_n * up;
    const float k_l1 = w_ssim != nullptr ? inv_n * up : (1.0f - lambda) * inv_n * up;
    float* __restrict__ gradp = grad + plane;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int y = y0 + r0 + o, x = x0 + col;
        if (y < H && x < W) {
            const int q = y * W + x;
            const float a = own_i[o], b = own_g[o];
            const float d = a - b;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);          // torch.abs backward: sign, 0 at 0
            gradp[q] = k_ssim * (acc[0][o] + 2.f * a * acc[1][o] + b * acc[2][o]) + k_l1 * sgn;
        }
    }
}

}  // namespace

size_t loss_workspace_bytes(int C, int H, int W)
{
    const size_t n = (size_t)C * H * W;
    const size_t blocks = (size_t)C * ((W + LT - 1) / LT) * ((H + LT - 1) / LT);
    return align_up(3 * n * sizeof(float)) + align_up(blocks * sizeof(float2));
}

void launch_loss_forward(int C, int H, int W, const float* img, const float* gt, float lambda, float* out3, char* ws,
                         hipStream_t s, bool defer_final)
{
    static const Win win = make_window();
    const size_t n = (size_t)C * H * W;
    const int tx = (W + LT - 1) / LT, ty = (H + LT - 1) / LT;
    const int blocks = C * tx * ty;
    float* D = reinterpret_cast<float*>(ws);
    float2* partials = reinterpret_cast<float2*>(ws + align_up(3 * n * sizeof(float)));
    hipLaunchKernelGGL(k_ssim_fwd, dim3((blocks + 7) / 8 * 8), dim3(LTHREADS), 0, s, H, W, tx, ty, win, img, gt, D, D + n, D + 2 * n,
                       partials, blocks);
    if (!defer_final)      // otherwise launch_loss_backward(..., final_out3) follows on the same stream and forms the value
        hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(LTHREADS), 0, s, blocks, (double)n, lambda, partials, out3);
}

void launch_loss_backward(int C, int H, int W, const float* img, const float* gt, float lambda, const float* upstream,
                          const char* ws, float* grad, hipStream_t s, float* final_out3, const float* w_ssim)
{
    static const Win win = make_window();
    const size_t n = (size_t)C * H * W;
    const int tx = (W + LT - 1) / LT, ty = (H + LT - 1) / LT;
    const float* D = reinterpret_cast<const float*>(ws);
    const float2* partials = reinterpret_cast<const float2*>(ws + align_up(3 * n * sizeof(float)));
    hipLaunchKernelGGL(k_ssim_bwd, dim3((C * tx * ty + 7) / 8 * 8), dim3(LTHREADS), 0, s, H, W, tx, ty, win, lambda, (float)(1.0 / (double)n),
                       upstream, w_ssim, img, gt, D, D + n, D + 2 * n, grad, partials, C * tx * ty, (double)n, final_out3);
}

}  // namespace lr
