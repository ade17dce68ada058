// render_fwd.hip -- per-tile front-to-back alpha blending with depth output, for gfx950.
//
// Replaces FORWARD::render / renderCUDA (RAST/cuda_rasterizer/forward.cu:261-391).  Per-pixel
// semantics are the reference's exactly (same skip tests, same 0.99 clamp, same T < 1e-4 stop,
// same depth normalisation).  Two kernels with the same results to the bit (launch_render_fwd picks): k_render_fwd below --
// four waves per tile, what a lone view and every small image gets -- and k_render_fwd_tile further down -- one wave per tile, a
// lane owns four pixels, for large images while other views' kernels share the GPU.  k_render_fwd is VALU-issue bound
// (measured: 88 % VALU busy at C3, HBM idle), so its shape is chosen to minimise wave-instructions per pixel x Gaussian pair:
//   * one 256-thread workgroup per 16x16 tile = 4 wave64; a wave owns ONE 8x8 quadrant, a lane one pixel.  (An earlier
//     shape -- 2 waves per tile, two pixels per lane, a candidate stepping only the pixels whose quadrant it hit -- shared
//     the candidate walk and the LDS reads between two quadrants; the quadrant-per-wave shape measured 6 % (C3) to 11 %
//     (dense clouds) faster: a wave walks only its own quadrant's candidates, stops as soon as its own 64 pixels are done,
//     and twice as many waves hide each other's latencies.  A packed FP32 instruction costs twice the issue cycles of a
//     scalar one on MI355X (tools/valu_microbench.hip: 453 vs 912 G wave-instr/s), so packing two pixels buys nothing.)
//     The blend is branch-free: a pixel that skips a Gaussian blends it with weight 0, which is arithmetically identical
//     to the reference's `continue`.
//   * every field the inner loop touches is staged in LDS (the reference re-reads colour and depth
//     from global memory per pixel per Gaussian, forward.cu:359, 364);
//   * two-level loop per wave.  CULL: 64 staged Gaussians at a time, one per LANE, are tested against the
//     wave's 8x8 pixel box with the exact box-minimum of the conic quadratic (common.h box_hit);
//     __ballot turns the result into a 64-bit candidate mask.  BLEND: the wave walks only the set bits
//     (s_ff1), in list order, all lanes evaluating the same Gaussian from broadcast ds_reads.  A
//     non-candidate cannot reach alpha >= 1/255 on any pixel of the box (margin in common.h), so skipping
//     it is exactly the reference's `continue` (forward.cu:338-339).  The outcome per list position and quadrant is left in
//     the binning buffer for the backward, which used to repeat the test;
//   * per-wave early termination via 64-bit __ballot (the reference only stops per block);
//   * tiles are assigned to workgroups so that each XCD (its own 4 MiB L2) renders a contiguous
//     band of the image and re-uses the GaussRecs of Gaussians that straddle neighbouring tiles.
#include "common.h"

namespace lr {

namespace {

constexpr int THREADS = 256;        // 4 wave64 per tile
constexpr int NWAVES = THREADS / 64;
constexpr int BATCH = 256;          // staged Gaussians per round = threads per workgroup (14 KB of LDS)

// Per-pixel state of the blend.  `last` is what the backward needs from the reference's n_contrib (forward.cu:349, 379:
// the 1-based list position of the last blended Gaussian): every list position below it is evaluated, everything from it
// on is skipped.  Here it is the 0-based position of the Gaussian that STOPPED the pixel (T would fall below 1e-4), or
// the list length for a pixel that never stopped.  The positions between the reference's value and this one hold only
// Gaussians this pixel skipped (alpha < 1/255 or power > 0), which the backward skips again by the same test on the same
// arithmetic, so the gradients are identical -- and the common step loses the two instructions that tracked it.
struct PixState { float T, Cr, Cg, Cb, D, acc; uint32_t last; };

// One candidate for the 64 pixels of the wave.  The per-pixel predicates live in wave-uniform 64-bit lane masks
// (v_cmp writes them to an SGPR pair; they are combined on the scalar unit and fed back to v_cndmask through
// inverse_ballot at no VALU cost); `done` is the mask of finished pixels.
template <bool STRICT>
__device__ __forceinline__ void fwd_pixel(PixState& p, uint64_t& done, const float qA, const float qB, const float qC,
                                          const float r0, const float r1,
                                          const float dx, const float op, const float cr, const float cg, const float cb,
                                          const float depth, const uint32_t pos0)
{
    float power;                                                      // log2(e) x the reference's power (staged coefficients)
    const float alpha = fminf(0.99f, op * gauss_weight<STRICT>(qA, qB, qC, r0, r1, dx, power));
    const float test_T = p.T * (1.0f - alpha);
    // reference order of tests (forward.cu:331-347): power > 0 -> skip; alpha < 1/255 -> skip;
    // T*(1-alpha) < 1e-4 -> pixel done (this Gaussian is NOT blended)
    const uint64_t pass = ~done & __builtin_amdgcn_ballot_w64(power <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
    const uint64_t low_T = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
    const uint64_t stop = pass & low_T;
    const bool use = __builtin_amdgcn_inverse_ballot_w64(pass & ~low_T);
    done |= stop;
    const float wgt = use ? alpha * p.T : 0.f;
    p.Cr += cr * wgt; p.Cg += cg * wgt; p.Cb += cb * wgt; p.D += depth * wgt; p.acc += wgt;
    p.T = use ? test_T : p.T;
    if (stop != 0ull) {                                                 // rare: a pixel stops at most once
        asm volatile("");                                               // keep it a scalar branch (no selects in the common path)
        p.last = __builtin_amdgcn_inverse_ballot_w64(stop) ? pos0 : p.last;
    }
}

// The same step with the candidate's alpha already formed, and branch-free: for the PAIR loop below, where two candidates'
// alphas are evaluated side by side and only these few dependent instructions per candidate remain in sequence.  `enable`:
// all ones, or zero for the second half of an odd pair (the step is then the identity).
__device__ __forceinline__ void fwd_step(PixState& p, uint64_t& done, const float alpha, const float power, const uint64_t enable,
                                         const float cr, const float cg, const float cb, const float depth, const uint32_t pos0)
{
    const float test_T = p.T * (1.0f - alpha);
    const uint64_t pass = enable & ~done & __builtin_amdgcn_ballot_w64(power <= 0.0f) & __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
    const uint64_t low_T = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
    const uint64_t stop = pass & low_T;
    const bool use = __builtin_amdgcn_inverse_ballot_w64(pass & ~low_T);
    done |= stop;
    const float wgt = use ? alpha * p.T : 0.f;
    p.Cr += cr * wgt; p.Cg += cg * wgt; p.Cb += cb * wgt; p.D += depth * wgt; p.acc += wgt;
    p.T = use ? test_T : p.T;
    p.last = __builtin_amdgcn_inverse_ballot_w64(stop) ? pos0 : p.last;
}

// PAIR: the candidate loop takes two candidates per round (small images: few waves per SIMD, the kernel time is the longest
// wave's dependent chain -- LDS read, exponent, v_exp, the T recursion -- and two candidates' chains overlap except for the
// recursion itself; profiles/r05a_pmc_c5shape.json: VALU busy 49 %, 2.2 waves resident per SIMD on average at 512^2).
// Same operations per pixel and candidate in the same order: bit-identical images.  Per tile: only lists of PAIR_MIN_LIST and more
// (measured, forward alone: dense 512^2, 1 700 per tile, 180 -> 173 us; one layer of pixel-sized splats, 320 per tile, 53.7 -> 55.1).
constexpr int PAIR_MIN_LIST = 1024;
template <bool STRICT, bool PAIR>
__global__ void __launch_bounds__(THREADS)
k_render_fwd(int W, int H, int gx, int num_tiles, int tile_map, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ list_gid,
             const GaussRec* __restrict__ rec,
             const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             float* __restrict__ out_color, float* __restrict__ out_depth, uint8_t* __restrict__ quad_hits,
             GeomHeader* __restrict__ hdr, uint2* __restrict__ seg_list, float4* __restrict__ ckpt,
             uint32_t* __restrict__ tile_seg0, float4* __restrict__ c_final)
{
    __shared__ float4 s_q0[BATCH];      // x, y, Ap = -0.5 conic a, Bp = -conic b      (common.h gauss_power; x log2 e)
    __shared__ float4 s_q1[BATCH];      // Cp = -0.5 conic c (x log2 e), opacity, depth, qmax (cull threshold, x log2 e)
    __shared__ float4 s_q2[BATCH];      // r, g, b, -
    __shared__ float2 s_q3[BATCH];      // -b/c, -b/a (edge minimiser slopes for box_hit)
    __shared__ int s_wdone[NWAVES];
    __shared__ uint32_t s_seg0;

    const int tile = blend_tile(tile_map, num_tiles);
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int x0 = tx * TILE_X + (w & 1) * 8, y0 = ty * TILE_Y + (w >> 1) * 8;      // this wave's 8x8 quadrant
    const int px = x0 + (l & 7), py = y0 + (l >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)x0, bx1 = (float)(x0 + 7), by0 = (float)y0, by1 = (float)(y0 + 7);

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    PixState A = { 1.0f, 0.f, 0.f, 0.f, 0.f, 0.000001f, 0u };
    uint64_t done = __builtin_amdgcn_ballot_w64(!inside);             // lane mask of finished pixels (all 64 lanes are live)
    bool wave_done = done == ~0ull;

    // A list longer than BWD_SEG is cut into segments for the backward (common.h BinLayout::seg_list / ckpt): one atomic
    // reserves the tile's slots, the (tile, segment) pairs are listed, and at the top of every later staging round the
    // waves still blending leave {T, colour so far} of their pixels; the final colour goes to c_final at the end
    static_assert(BATCH == BWD_SEG, "a staging round of the forward is one segment of the backward");
    const int n_seg = total > 0 ? (total - 1) / BWD_SEG : 0;
    uint32_t seg0 = 0;
    int n_ck = 0;                                                     // checkpoints this wave has written (segments 1 .. n_ck)
    if (n_seg > 0) {
        if (tid == 0) { s_seg0 = atomicAdd(&hdr->n_seg, (uint32_t)n_seg); tile_seg0[tile] = s_seg0; }
        lds_barrier();
        seg0 = s_seg0;
        for (int g = 1 + tid; g <= n_seg; g += THREADS) seg_list[seg0 + g - 1] = make_uint2((uint32_t)tile, (uint32_t)g);
    }

    for (int base = 0; base < total; base += BATCH) {
        // all quadrants finished?  (also the barrier that protects the LDS planes of the previous batch)
        if (l == 0) s_wdone[w] = wave_done ? 1 : 0;
        lds_barrier();
        if (s_wdone[0] + s_wdone[1] + s_wdone[2] + s_wdone[3] == NWAVES) break;
        const int cnt = min(BATCH, total - base);
        if (base > 0 && !wave_done) {
            // every pixel of a wave that is done stopped in front of this position: the backward starts those from final_T
            ckpt[(size_t)(seg0 + n_ck) * TILE_PIX + tid] = make_float4(A.T, A.Cr, A.Cg, A.Cb);
            n_ck++;
        }
        if (tid < cnt) {
            const uint32_t id = list_gid[range.x + base + tid];               // the Gaussian of the list position (BinLayout::list_gid)
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            // default: Ap, Bp, Cp and the cull threshold scaled by log2(e); strict: the conic and the threshold as they are
            s_q0[tid] = STRICT ? make_float4(a.x, a.y, a.z, a.w) : make_float4(a.x, a.y, (-0.5f * LOG2E) * a.z, -LOG2E * a.w);
            s_q1[tid] = STRICT ? make_float4(b.x, b.y, c.y, c.z) : make_float4((-0.5f * LOG2E) * b.x, b.y, c.y, LOG2E * c.z);
            s_q2[tid] = make_float4(b.z, b.w, c.x, 0.f);
            s_q3[tid] = make_float2(-a.w / b.x, -a.w / a.z);
        }
        lds_barrier();
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
                    const float ca = STRICT ? a.z : -2.0f * a.z, cb = STRICT ? a.w : -a.w, cc = STRICT ? b.x : -2.0f * b.x;   // conic (x log2 e, like qmax)
                    hit = box_hit(a.x, a.y, ca, cb, cc, r.x, r.y, b.w, bx0, bx1, by0, by1);
                    // the outcome is kept for the backward (common.h BinLayout::quad_hits): every position a pixel of this
                    // quadrant can have blended lies in a chunk this wave culled
                    quad_hits[4 * ((size_t)range.x + (size_t)(base + j)) + w] = hit ? 1 : 0;
                }
            }
            uint64_t mask = __ballot(hit);
            // BLEND: walk the candidates in list order (a non-candidate is exactly the reference's `continue`: no pixel of
            // the quadrant can reach alpha >= 1/255)
            if (PAIR && total >= PAIR_MIN_LIST) {
                while (mask) {
                    const int k0 = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const bool two = mask != 0ull;
                    const int k1 = two ? __ffsll((long long)mask) - 1 : k0;
                    mask &= mask - 1;                                                  // (0 & anything = 0)
                    const int j0 = sb + k0, j1 = sb + k1;
                    float4 a0 = s_q0[j0], a1 = s_q0[j1];
                    float4 b0 = s_q1[j0], b1 = s_q1[j1];
                    const float4 c0 = s_q2[j0], c1 = s_q2[j1];
                    // Both candidates' reads are in flight before any arithmetic, and both alphas exist before the first step of
                    // the recursion: the empty asm statements tie the two candidates' values together (left to itself the compiler
                    // finishes candidate 0 -- reads, exponent, step -- before it even issues the reads of candidate 1)
                    asm volatile("" : "+v"(a0.x), "+v"(a0.y), "+v"(a0.z), "+v"(a0.w), "+v"(b0.x), "+v"(a1.x), "+v"(a1.y), "+v"(a1.z), "+v"(a1.w), "+v"(b1.x));
                    float r00, r01, r10, r11, power0, power1;
                    gauss_row<STRICT>(a0.w, b0.x, a0.y - pyf, r00, r01);                        // common.h gauss_power
                    gauss_row<STRICT>(a1.w, b1.x, a1.y - pyf, r10, r11);
                    float alpha0 = fminf(0.99f, b0.y * gauss_weight<STRICT>(a0.z, a0.w, b0.x, r00, r01, a0.x - pxf, power0));
                    float alpha1 = fminf(0.99f, b1.y * gauss_weight<STRICT>(a1.z, a1.w, b1.x, r10, r11, a1.x - pxf, power1));
                    asm volatile("" : "+v"(alpha0), "+v"(alpha1), "+v"(power0), "+v"(power1));
                    fwd_step(A, done, alpha0, power0, ~0ull, c0.x, c0.y, c0.z, b0.z, (uint32_t)(base + j0));
                    fwd_step(A, done, alpha1, power1, two ? ~0ull : 0ull, c1.x, c1.y, c1.z, b1.z, (uint32_t)(base + j1));
                }
            } else
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int j = sb + k;
                const float4 a = s_q0[j];
                const float4 b = s_q1[j];
                const float4 c = s_q2[j];
                const uint32_t pos0 = (uint32_t)(base + j);
                const float dy = a.y - pyf;
                float r0, r1;
                gauss_row<STRICT>(a.w, b.x, dy, r0, r1);                                        // common.h gauss_power
                fwd_pixel<STRICT>(A, done, a.z, a.w, b.x, r0, r1, a.x - pxf, b.y, c.x, c.y, c.z, b.z, pos0);
            }
            if (done == ~0ull) { wave_done = true; break; }             // all 64 pixels are finished
        }
    }

    if (inside) {
        const size_t N = (size_t)W * H;
        const size_t pix = (size_t)py * W + px;
        final_T[pix] = A.T;
        n_contrib[pix] = __builtin_amdgcn_inverse_ballot_w64(done) ? A.last : (uint32_t)total;
        const float fr = A.Cr + A.T * bg[0], fg = A.Cg + A.T * bg[1], fb = A.Cb + A.T * bg[2];
        out_color[pix] = fr;
        out_color[N + pix] = fg;
        out_color[2 * N + pix] = fb;
        // a segmented tile: the final colour once more, where the backward's later segments find it (they subtract a
        // checkpoint's colour-so-far from it: the colour still to come behind that position, background included)
        if (n_seg > 0) c_final[pix] = make_float4(fr, fg, fb, 0.f);
        out_depth[pix] = (A.acc > 0.5f) ? A.D / A.acc : 0.0f;         // forward.cu:384-388
    }
}


// (Round 6 built the producer / consumer split DESIGN.md 8.2 had proposed for small images -- eight waves per tile, per quadrant
//  one wave evaluating alpha into an LDS ring and one carrying the recursion, counters polled with s_sleep -- and removed it again:
//  bit-identical to the quadrant kernel on every test, and 2.2-2.4 x SLOWER (dense 512^2 168 -> 402 us, pixel-sized splats 53 -> 119
//  us): the hand-over costs more instructions than it moves -- VALU 48.6 M -> 131 M, SALU 24.9 M -> 116 M wave-instructions per view,
//  waves parked 140 M -> 554 M cycles (profiles/r06s_ab_fwd_split_dead_end.json, r06t_pmc_c5shape_fwd_*.json; code at commit
//  "Forward split for small images").)

// The TILE shape of the forward: ONE wave64 per 16x16 tile, a lane owns FOUR pixels -- the same position (l & 7, l >> 3) in
// each of the tile's four 8x8 quadrants (the layout of render_bwd.hip's TILE shape).  A round stages 64 list entries, one per
// lane: the lane culls its own entry against the four quadrants from registers (the four outcomes leave as ONE 4-byte store
// for the backward), the wave walks the union of the four candidate masks, reads a candidate's fields once, and steps only the
// quadrants it hit (scalar branches on the mask bits).  What it is for: everything the 4-wave shape pays per candidate and
// QUADRANT that is not a pixel step -- three broadcast LDS reads (8 + 6 + 6 LDS cycles), the mask bookkeeping on the scalar
// unit, the row terms of the exponent -- is paid once per candidate and TILE (an instance reaches 2.33 of its tile's 4
// quadrants at C3); no barriers, no partner waves to wait for, 3 KB of LDS.  Same operations per pixel and candidate in the
// same order as the quadrant kernel: the same bits (tests/test_gpu_variants.py).
constexpr int TSTAGE = 64;
template <bool STRICT>
__device__ __forceinline__ void
render_fwd_tile(int W, int H, int gx, int num_tiles, int tile_map, const uint2* __restrict__ ranges,
                  const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ list_gid,
                  const GaussRec* __restrict__ rec,
                  const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                  float* __restrict__ out_color, float* __restrict__ out_depth, uint8_t* __restrict__ quad_hits,
                  GeomHeader* __restrict__ hdr, uint2* __restrict__ seg_list, float4* __restrict__ ckpt,
                  uint32_t* __restrict__ tile_seg0, float4* __restrict__ c_final)
{
    __shared__ float4 s_q0[TSTAGE];     // as in k_render_fwd
    __shared__ float4 s_q1[TSTAGE];
    __shared__ float4 s_q2[TSTAGE];

    const int tile = blend_tile(tile_map, num_tiles);
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int l = threadIdx.x;
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y;
    const int pxl = x0 + (l & 7), pyt = y0 + (l >> 3);                // the lane's pixel in quadrant 0; + 8 for the right / lower ones
    const float pxf[2] = { (float)pxl, (float)(pxl + 8) }, pyf[2] = { (float)pyt, (float)(pyt + 8) };

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);

    PixState A[4];
    uint64_t done[4];
    bool inside[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        A[q] = PixState{ 1.0f, 0.f, 0.f, 0.f, 0.f, 0.000001f, 0u };
        inside[q] = pxl + (q & 1) * 8 < W && pyt + (q >> 1) * 8 < H;
        done[q] = __builtin_amdgcn_ballot_w64(!inside[q]);
    }

    // list segments for the backward: as in k_render_fwd (one atomic per tile; a checkpoint per pixel every BWD_SEG positions)
    static_assert(BWD_SEG % TSTAGE == 0, "checkpoints fall on round boundaries");
    const int n_seg = total > 0 ? (total - 1) / BWD_SEG : 0;
    uint32_t seg0 = 0;
    if (n_seg > 0) {
        if (l == 0) { seg0 = atomicAdd(&hdr->n_seg, (uint32_t)n_seg); tile_seg0[tile] = seg0; }
        seg0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg0);
        for (int g = 1 + l; g <= n_seg; g += 64) seg_list[seg0 + g - 1] = make_uint2((uint32_t)tile, (uint32_t)g);
    }

    // A round's records arrive through three dependent loads (list -> Gaussian id -> record), and a single-wave workgroup has no
    // partner to wait with: the first two are issued a round ahead and are in flight while this round's candidates are walked
    // (positions beyond the list's end re-read its last entry, never staged).  Fetching the records ahead as well -- into
    // registers: the compiler waits for them where they are issued; straight into a second set of LDS planes with
    // global_load_lds_dwordx4: correct, 6 instead of 7 waves per SIMD -- measured no faster (profiles/r05w_ab_fwd_tile.json).
    const uint32_t* __restrict__ lgid = list_gid + range.x;            // the Gaussian of every list position (BinLayout::list_gid)
    uint32_t id_next = total > 0 ? lgid[min(l, total - 1)] : 0u;
    for (int base = 0; base < total; base += TSTAGE) {
        if ((done[0] & done[1] & done[2] & done[3]) == ~0ull) break;
        const int cnt = min(TSTAGE, total - base);
        const uint32_t id = id_next;
        if (base + TSTAGE < total) id_next = lgid[min(base + TSTAGE + l, total - 1)];
        if (base > 0 && base % BWD_SEG == 0) {
            // (a quadrant whose pixels are all done stopped in front of this position: the backward starts those from final_T)
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (done[q] != ~0ull)
                    ckpt[(size_t)(seg0 + base / BWD_SEG - 1) * TILE_PIX + q * 64 + l] = make_float4(A[q].T, A[q].Cr, A[q].Cg, A[q].Cb);
        }
        bool hit[4] = { false, false, false, false };
        lds_barrier();                                                // (the previous round's reads are done: one wave, no waiting)
        if (l < cnt) {
            // the quadrants' boxes are formed here, per round, from tile coordinates the optimiser cannot see through: kept
            // across the candidate loop they are eight more live registers (the tile's corner is uniform, but float conversions
            // live in vector registers)
            int x0v = x0, y0v = y0;
            asm volatile("" : "+s"(x0v), "+s"(y0v));
            const float bx[2][2] = { { (float)x0v, (float)(x0v + 7) }, { (float)(x0v + 8), (float)(x0v + 15) } };
            const float by[2][2] = { { (float)y0v, (float)(y0v + 7) }, { (float)(y0v + 8), (float)(y0v + 15) } };
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            const float4 q0 = STRICT ? make_float4(a.x, a.y, a.z, a.w) : make_float4(a.x, a.y, (-0.5f * LOG2E) * a.z, -LOG2E * a.w);
            const float4 q1 = STRICT ? make_float4(b.x, b.y, c.y, c.z) : make_float4((-0.5f * LOG2E) * b.x, b.y, c.y, LOG2E * c.z);
            s_q0[l] = q0; s_q1[l] = q1; s_q2[l] = make_float4(b.z, b.w, c.x, 0.f);
            const float r_c = -a.w / b.x, r_a = -a.w / a.z;
            const float ca = STRICT ? q0.z : -2.0f * q0.z, cb = STRICT ? q0.w : -q0.w, cc = STRICT ? q1.x : -2.0f * q1.x;
#pragma unroll
            for (int q = 0; q < 4; q++)
                hit[q] = box_hit(q0.x, q0.y, ca, cb, cc, r_c, r_a, q1.w, bx[q & 1][0], bx[q & 1][1], by[q >> 1][0], by[q >> 1][1]);
            // the outcome per quadrant, kept for the backward (common.h BinLayout::quad_hits)
            reinterpret_cast<uint32_t*>(quad_hits)[(size_t)range.x + (size_t)(base + l)] =
                (hit[0] ? 1u : 0u) | (hit[1] ? 0x100u : 0u) | (hit[2] ? 0x10000u : 0u) | (hit[3] ? 0x1000000u : 0u);
        }
        uint64_t m[4];
#pragma unroll
        for (int q = 0; q < 4; q++) m[q] = done[q] == ~0ull ? 0ull : __ballot(hit[q]);
        uint64_t mask = (m[0] | m[1]) | (m[2] | m[3]);
        lds_barrier();
        while (mask) {
            const int k = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float4 fa = s_q0[k];
            const float4 fb = s_q1[k];
            const float4 fc = s_q2[k];
            const uint32_t pos0 = (uint32_t)(base + k);
            const uint64_t bit = 1ull << k;
#pragma unroll
            for (int row = 0; row < 2; row++) {
                if (((m[2 * row] | m[2 * row + 1]) & bit) == 0ull) continue;
                float r0, r1;
                gauss_row<STRICT>(fa.w, fb.x, fa.y - pyf[row], r0, r1);                         // common.h gauss_power
#pragma unroll
                for (int col = 0; col < 2; col++) {
                    const int q = 2 * row + col;
                    if ((m[q] & bit) == 0ull) continue;
                    fwd_pixel<STRICT>(A[q], done[q], fa.z, fa.w, fb.x, r0, r1, fa.x - pxf[col], fb.y, fc.x, fc.y, fc.z, fb.z, pos0);
                    if (done[q] == ~0ull) {                            // the quadrant is finished: its later candidates are nobody's
                        m[q] = 0ull;
                        mask &= (m[0] | m[1]) | (m[2] | m[3]);
                    }
                }
            }
        }
    }

    const size_t N = (size_t)W * H;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (!inside[q]) continue;
        const int px = pxl + (q & 1) * 8, py = pyt + (q >> 1) * 8;
        const size_t pix = (size_t)py * W + px;
        final_T[pix] = A[q].T;
        n_contrib[pix] = __builtin_amdgcn_inverse_ballot_w64(done[q]) ? A[q].last : (uint32_t)total;
        const float fr = A[q].Cr + A[q].T * bg[0], fg = A[q].Cg + A[q].T * bg[1], fb = A[q].Cb + A[q].T * bg[2];
        out_color[pix] = fr;
        out_color[N + pix] = fg;
        out_color[2 * N + pix] = fb;
        if (n_seg > 0) c_final[pix] = make_float4(fr, fg, fb, 0.f);
        out_depth[pix] = (A[q].acc > 0.5f) ? A[q].D / A[q].acc : 0.0f;         // forward.cu:384-388
    }
}
#define LR_FWD_TILE_PARAMS int W, int H, int gx, int num_tiles, int tile_map, const uint2* __restrict__ ranges,                 \
                  const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ list_gid,                              \
                  const GaussRec* __restrict__ rec,                                                                            \
                  const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,                 \
                  float* __restrict__ out_color, float* __restrict__ out_depth, uint8_t* __restrict__ quad_hits,               \
                  GeomHeader* __restrict__ hdr, uint2* __restrict__ seg_list, float4* __restrict__ ckpt,                       \
                  uint32_t* __restrict__ tile_seg0, float4* __restrict__ c_final
#define LR_FWD_TILE_PASS W, H, gx, num_tiles, tile_map, ranges, point_list, list_gid, rec, bg, final_T, n_contrib, out_color,  \
                  out_depth, quad_hits, hdr, seg_list, ckpt, tile_seg0, c_final
// 64 registers -> 8 waves per SIMD: the 8160 tiles of a 1080p view are all resident at once, as in the backward's TILE shape
// (round 6; with the compiler's own budget, 66-72 registers and 7 waves per SIMD, the last 992 waves waited for the first 7168:
// lone view C3 62.3 -> 59.0 us, C2 101 -> 92 us, dense 1080p 366 -> 342 us, three views in flight equal;
// profiles/r06h_ab_fwdtile8.json).  The quadrant boxes formed per round are what made room (render_fwd_tile).
template <bool STRICT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_vgpr(64)))
k_render_fwd_tile(LR_FWD_TILE_PARAMS)
{
    render_fwd_tile<STRICT>(LR_FWD_TILE_PASS);
}

}  // namespace

void launch_render_fwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const uint32_t* list_gid, const GaussRec* rec, const float* bg, float* final_T,
                       uint32_t* n_contrib, float* out_color, float* out_depth, uint8_t* quad_hits,
                       GeomHeader* hdr, uint2* seg_list, float4* ckpt, uint32_t* tile_seg0, float4* c_final,
                       long long inst_hint, hipStream_t s)
{
    const int num_tiles = gx * gy;
    if (num_tiles <= 0) return;
    const int tile_map = blend_tile_map(num_tiles);
    const int grid = ((num_tiles + 7) / 8) * 8;
    // Shape (same images, depth, checkpoints and gradients to the bit from all of them):
    //   quadrant kernel (4 waves per tile) -- small images, a lone large view, and large images of long lists;
    //   tile kernel (1 wave per tile) -- large images while other views' kernels are in flight: 10 % fewer VALU instructions and
    //   59 % fewer LDS reads per C3 view (28.7 M -> 25.7 M, 3.10 M -> 1.26 M; profiles/r05x_pmc_fwdtile_*.json).  Three views in
    //   flight, quadrant -> tile kernel (profiles/r06q_shape_sweep.json): C2 +2.5 %, C3 +1 %, 3 M / 1440p (700 instances per
    //   tile) +0.5 %, dense 1 M cloud at 1080p (460 per tile) -1.8 %; a lone view: -2 ... -8 %.  So: large images with other views
    //   in flight, and not on scenes whose earlier views told the host that the lists are longer than a staging round of the
    //   quadrant kernel on average (inst_hint: api.hip view_hint_instances; -1 = unknown).
    //   [Round 5's candidate-PAIR variant of the quadrant kernel (small images, lists of 1024 and more) no longer pays on any
    //    workload of the sweep (-0.5 ... -1.6 %): diagnostics build only.]
    // lr_tune_set("fwd_pair", 0 / 2) forces quadrant / tile (tests, A/B runs); 1 = the pair variant (diagnostics build).
    const int knob = tune_get(TUNE_FWD_PAIR);
    const bool long_lists = inst_hint > (long long)BATCH * num_tiles;
    const bool tile_shape = knob >= 0 ? knob == 2 : (num_tiles > 3072 && views_in_flight() >= 2 && !long_lists);
#ifdef LR_DIAGNOSTICS
    const bool pair = knob == 1;
#else
    constexpr bool pair = false;
#endif
    const bool strict = tune_get(TUNE_STRICT) > 0;
    note_fwd_shape(tile_shape ? 2 : pair ? 1 : 0);
#define LR_FWD_ARGS W, H, gx, num_tiles, tile_map, ranges, point_list, list_gid, rec, bg, final_T, n_contrib, out_color, out_depth, \
                    quad_hits, hdr, seg_list, ckpt, tile_seg0, c_final
    if (tile_shape) {
        if (strict) hipLaunchKernelGGL((k_render_fwd_tile<true>), dim3(grid), dim3(64), 0, s, LR_FWD_ARGS);
        else hipLaunchKernelGGL((k_render_fwd_tile<false>), dim3(grid), dim3(64), 0, s, LR_FWD_ARGS);
        return;
    }
#define LR_FWD(S, PR) hipLaunchKernelGGL((k_render_fwd<S, PR>), dim3(grid), dim3(THREADS), 0, s, LR_FWD_ARGS)
#ifdef LR_DIAGNOSTICS
    if (pair) { if (strict) LR_FWD(true, true); else LR_FWD(false, true); return; }
#endif
    if (strict) LR_FWD(true, false); else LR_FWD(false, false);
#undef LR_FWD_ARGS
#undef LR_FWD
}

}  // namespace lr
