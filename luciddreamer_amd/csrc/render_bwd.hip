// render_bwd.hip -- backward of the per-tile alpha blend, for gfx950.
//
// Replaces BACKWARD::render / renderCUDA (RAST/cuda_rasterizer/backward.cu:399-586).  The
// per-pixel recursion is the reference's (back-to-front, T un-blended by division, colour
// recursion through accum_rec, background term, 0.99 clamp not differentiated, depth gradient
// ignored).  The kernel is VALU-issue bound; the design removes wave-instructions:
//
//   * same two-level loop as the forward: per 64 staged Gaussians one lane each looks up the outcome of the forward's
//     exact quadrant test (box_hit, kept per list position in the binning buffer) -> 64-bit candidate mask; only
//     candidates (walked back to front with s_flbit) are evaluated per pixel.  Positions at or behind the deepest stop position of the wave's pixels
//     (render_fwd.hip PixState::last) are masked out up front (backward.cu:500-502, made wave-uniform).
//   * the nine per-(pixel,Gaussian) gradient terms reach memory as
//       reference: 9 global float atomicAdd per contributing pixel x Gaussian pair (:537-583)
//       here:      wave64 reduction of 8 terms with v_permlane32_swap / v_permlane16_swap (each swap+add
//                  halves TWO terms at once, all six swaps in one asm block) down to 16-lane rows, then 7 DPP adds
//                  that keep the terms TRANSPOSED inside the rows (row_merge3: a DPP add under a bank mask hands
//                  half of the lanes to another term at every level) instead of three row sums of 4 DPP adds each;
//                  -> ONE plain 12-lane ds_write into the wave's own copy of the per-batch accumulator (round 2:
//                     three LDS float atomics -- ds_add_f32 retires ~3 cycles per active lane on MI355X)
//                  -> ONE plain 48-byte store per tile instance into that instance's own slot
//                     (inst_grad[emission index]); the slots of a Gaussian are contiguous and are summed,
//                     in a fixed order, by k_gauss_bwd.  No global atomics at all, nothing to pre-zero.
#include <cstdlib>
#include "common.h"

namespace lr {

namespace {

#ifndef LR_QBATCH_BWD
#define LR_QBATCH_BWD 128           // staging round of the QUAD shape (per-wave accumulator columns: 24 KB of LDS at 128)
#endif
constexpr int BATCH2 = 64;          // staged Gaussians per round of the 2-wave shape (7 KB of LDS per workgroup, so registers --
                                    // 7 waves per SIMD -- and not LDS limit the occupancy; 3 % faster than 128, 32 is slower)
// threads per workgroup: 128 (2 wave64), or 256 with QUAD (one 8x8 quadrant per wave, one pixel per lane, as in the
// forward; chosen for small images by blend_quad below)

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum within each 16-lane row; every lane of the row ends with the row total
__device__ __forceinline__ float row_sum(float v)
{
    v += dpp<0xB1>(v);
    v += dpp<0x4E>(v);
    v += dpp<0x141>(v);
    v += dpp<0x140>(v);
    return v;
}
// (inline asm below: with ROCm 7.2 hipcc, __builtin_amdgcn_permlane32_swap followed by r[0] + r[1] returned 2 * r[0] --
//  verified on hardware)
// Reduce eight per-lane terms over the wave down to 16-lane rows: on return row r of `a0` holds the row partials of term
// {a0, a2, a1, a3}[r] and row r of `b0` those of {b0, b2, b1, b3}[r]; a row_sum of each finishes the job.  One asm block so
// that the six swaps share two hazard waits instead of paying one each (VALU write -> v_permlane*_swap read needs two
// wait states, swap -> VALU read one; every dependent pair below has that many instructions in between):
//   4 x [x <- (lanes<32: x[l] + x[l+32]) | (lanes>=32: y[l-32] + y[l])]   then   2 x the same over 16-lane rows
__device__ __forceinline__ void reduce8(float& a0, float a1, float a2, float a3, float& b0, float b1, float b2, float b3)
{
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\t"
                 "v_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\t"
                 "v_permlane32_swap_b32 %6, %7\n\t"
                 "v_add_f32 %0, %0, %1\n\t"
                 "v_add_f32 %2, %2, %3\n\t"
                 "v_add_f32 %4, %4, %5\n\t"
                 "v_add_f32 %6, %6, %7\n\t"
                 "v_permlane16_swap_b32 %0, %2\n\t"
                 "s_nop 0\n\t"
                 "v_permlane16_swap_b32 %4, %6\n\t"
                 "v_add_f32 %0, %0, %2\n\t"
                 "v_add_f32 %4, %4, %6\n\t"
                 "s_nop 1"                                  // whatever follows may be a DPP read of %0 / %4 (two wait states)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
}
// Finish the wave reduction inside the 16-lane rows with the terms kept TRANSPOSED: instead of three independent row sums
// (4 DPP adds each: every level adds a register to a shifted copy of itself and all sixteen lanes end up with the same
// total), each level hands one half of the lanes to another term -- a DPP add under a bank mask writes only the lanes it
// is enabled for, so `x[0..7] = a[l] + a[l+8]` and `x[8..15] = b[l] + b[l-8]` are two instructions that leave ONE register
// to carry on with.  In: ra / rb = the row partials reduce8 left (one term per row each), sB = the lane's raw db term.
// Out: lane 0 of each row = that row's `ra` term, lane 8 = its `rb` term, lane 4 = the row's partial of db.
// 7 DPP adds instead of 12, and the three accumulator updates become one.  Wait states by hand (inline asm is opaque to
// the hazard recogniser): a DPP source written by a VALU instruction needs two other instructions or nops in between.
// (All DPP controls cost the same here -- 4.3 to 4.9 cycles per wave64 add against 2.8 for a plain one,
// tools/valu_microbench.hip -- so only the COUNT matters: row sums by rotation instead of the compiler's quad_perm /
// row_mirror idiom measured no different, profiles/r03b_ab_bwd_red_dpp_store.json.)
__device__ __forceinline__ float row_merge3(float ra, float rb, float sB)
{
    asm volatile("v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"        // db: l + (l ^ 8); reduce8 ends with the wait
                 "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"        // lanes 8-15: rb[l] + rb[l-8]
                 "v_add_f32_dpp %1, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"        // lanes 0-7:  ra[l] + ra[l+8]
                 "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"        // lanes 4-7, 12-15: db[l] + db[l-4]
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %2, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"        // lanes 0-3, 8-11: m[l] + m[l+4]
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "+v"(ra), "+v"(rb), "+v"(sB));
    return sB;
}
// The same reduction THROUGH LDS (round 6; the TILE shape): the wave's eight terms are written as eight 256-byte planes
// [term][lane] with ds_write_addtid_b32 (address = M0 + offset + 4 * lane: no address register, 2 LDS cycles per plane), lane
// L = 8 t + s reads eight floats of term t with two ds_read_b128 -- 16-byte chunks chosen so that the sixteen lanes of each of
// the instruction's lane groups cover all 64 banks once (MI355X_MICROARCH.md, LDS: groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...):
// chunk block b1 = s >> 2 for terms 0, 1 (mod 4), 3 - (s >> 2) for terms 2, 3; the second read takes block b1 ^ 2 -- sums them
// in-lane (7 plain adds) and finishes over the 8 lanes of its term with DPP adds; db (the ninth term) stays on DPP and shares the
// levels below 8 lanes under bank masks.  VALU: 7 adds + 5 DPP adds instead of 6 swaps + 6 adds + 7 DPP adds.
// Out: lane 0 of row r = term 2r, lane 8 = term 2r + 1, lane 4 = row r's partial of db.
typedef float lr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lds_reduce9(float t0, float t1, float t2, float t3, float t4, float t5, float t6, float t7,
                                             float d, uint32_t plane_base, uint32_t a1, uint32_t a2)
{
    lr_f4 q0, q1;
    uint32_t m0_saved;
    // two blocks: the planes' sources are dead once the stores are issued, so the loads may land in the same registers
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %10\n\t"
                 "s_nop 0\n\t"                                                             // SALU write of M0 -> LDS add-TID: one wait state
                 "ds_write_addtid_b32 %2 offset:0\n\t"
                 "ds_write_addtid_b32 %3 offset:256\n\t"
                 "ds_write_addtid_b32 %4 offset:512\n\t"
                 "ds_write_addtid_b32 %5 offset:768\n\t"
                 "ds_write_addtid_b32 %6 offset:1024\n\t"
                 "ds_write_addtid_b32 %7 offset:1280\n\t"
                 "ds_write_addtid_b32 %8 offset:1536\n\t"
                 "ds_write_addtid_b32 %9 offset:1792\n\t"
                 "s_mov_b32 m0, %0\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf"            // db: l + (l ^ 8), under the LDS latency
                 : "=&s"(m0_saved), "+v"(d)
                 : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "s"(plane_base)
                 : "memory");
    // no wait between the stores and the loads: the LDS executes one wave's instructions in issue order
    asm volatile("ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(a1), "v"(a2) : "memory");
    float x = (q0.x + q0.y) + (q0.z + q0.w);
    float y = (q1.x + q1.y) + (q1.z + q1.w);
    asm volatile("v_add_f32 %0, %0, %2\n\t"                                                // the in-lane sum of the lane's eight floats
                 "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"       // lanes 4-7, 12-15: db[l] + db[l-4]
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %1, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"       // lanes 0-3, 8-11: v[l] + v[l+4]
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "+v"(x), "+v"(d) : "v"(y));
    return d;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}
template <bool B> struct BoolTag { static constexpr bool value = B; };

// s_acc column of each term: 0 sum D dx, 1 sum D dy, 2 sum D dx^2, 3 sum D dx dy, 4 sum D dy^2, 5 sum D (= dop), 6 dr, 7 dg,
// 8-11 db (one column per 16-lane row); the flush turns columns 0-4 into GradRec's dmx, dmy, dca, dcb, dcc (same float order)
//
// 128 threads = 2 wave64 per 16x16 tile; a wave owns a 16x8 half tile, a lane owns TWO pixels (same row, 8 columns
// apart: pixel A in the left 8x8 quadrant of the wave's box, pixel B in the right one).  Per-pixel state lives in
// SCALAR registers: a packed FP32 instruction costs the issue cycles of two scalar ones on MI355X
// (tools/valu_microbench.hip), so nothing is lost by not packing, and a candidate that failed the exact box test of one
// quadrant (48 % of them) steps only the other pixel -- half the per-pixel work.  The recursion is branch-free per
// pixel: a pixel for which the Gaussian is skipped processes it as a layer with alpha = 0 and G = 0, which leaves T,
// the colour recursion and every gradient term exactly unchanged (T * rcp(1-0) = T; A + 0 * d = A).
struct BwdPix {
    float T;            // transmittance in front of the current layer
    float A;            // (colour seen BEHIND the current layer, background included) . dL/dpixel
    float dLr, dLg, dLb;
    float pxf;
    uint32_t last;      // list positions below this one were blended by the forward (render_fwd.hip PixState::last)
};

// One layer for one pixel; adds the pixel's terms to the lane sums.  Only what varies per pixel is formed here: the
// colour recursion of backward.cu:517-533 enters only through its dot product with the pixel's dL/dpixel (constant along
// the list), so ONE scalar A replaces three accumulators, and the background term -T_final/(1-alpha) * bg.dL
// (backward.cu:556-560) is the last layer of the same recursion (A starts at bg.dL); of the geometric gradients only the
// moments of D = G * dL/dalpha (D, D dx, D dx^2 here; the dy factors and opacity, conic, -0.5, NDC scale later).
// The layer is folded into A as soon as its gradient is formed: accum_rec' = alpha * colour + (1 - alpha) * accum_rec =
// A + alpha * (colour.dL - A), and the difference in the bracket is the one dL/dalpha needs anyway (rounds 1-3 carried
// last_alpha / last colour to the next layer as the reference does and formed the same difference twice: one instruction and
// two registers per pixel more, same bits).
// FIRSTM / FIRSTC: the lane's moment / colour sums are ASSIGNED (first pixel of the lane for this candidate) instead of
// accumulated -- `0 + a * b` is not foldable under IEEE rules and would cost an extra v_fma per term next to the product
// that is needed anyway.
// CHECK_LAST: the `pos < last` test is compiled in.  A pixel the forward never stopped (T stayed above 1e-4) carries last = the
// list length, so the test can only fail for pixels that DID stop; a wave none of whose pixels stopped (99 % of the waves of a
// C3 view; most of a dense one do have stopped pixels) walks its candidates through the copy of the loop without it.
template <bool FIRSTM, bool FIRSTC, bool CHECK_LAST, bool STRICT = false>
__device__ __forceinline__ void bwd_pixel(BwdPix& p, const float qA, const float qB, const float qC, const float r0, const float r1,
                                          const float gx,
                                          const float op, const float cr, const float cg, const float cb, const uint32_t pos,
                                          float& sD, float& sMx, float& sMxx, float& sR, float& sG, float& sB)
{
    const float dx = gx - p.pxf;
    float power;                                                           // (log2(e) x) the reference's power
    const float t = op * gauss_weight<STRICT>(qA, qB, qC, r0, r1, dx, power);   // opacity x G: alpha before the 0.99 clamp
    // reference tests (backward.cu:500-515): behind the pixel's last contributor, power > 0, alpha < 1/255 -> skipped (the
    // clamp at 0.99 cannot change the outcome of the 1/255 test, so it is taken on the unclamped product).  ONE select: the
    // masked product serves as alpha (clamped) and, unclamped, as the weight of the geometric gradients -- what the lane sums
    // accumulate is opacity x G x dL/dalpha (the reference's dL/dG chain rule carries the opacity factor anyway,
    // backward.cu:563-570); the flush divides the one sum that wants G x dL/dalpha alone (dL/dopacity) by the opacity
    const bool v = (!CHECK_LAST || pos < p.last) && power <= 0.0f && t >= 1.0f / 255.0f;
    const float tm = v ? t : 0.f;
    const float alpha = fminf(0.99f, tm);
    const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);                 // one v_rcp per pixel
    p.T = p.T * rinv;
    const float dchan = alpha * p.T;
    // colour of this Gaussian . dL/dpixel, minus the colour behind: one fma chain started at -A (three instructions, not a
    // dot product and a subtraction)
    const float d = __builtin_fmaf(cb, p.dLb, __builtin_fmaf(cg, p.dLg, __builtin_fmaf(cr, p.dLr, -p.A)));
    const float dop = tm * (d * p.T);                                      // opacity x G x dL/dalpha
    p.A = __builtin_fmaf(alpha, d, p.A);                                   // a skipped layer (alpha = 0) leaves A as it is
    const float mx = dop * dx;
    if (FIRSTM) { sD = dop; sMx = mx; sMxx = mx * dx; }
    else { sD += dop; sMx += mx; sMxx += mx * dx; }
    if (FIRSTC) { sR = dchan * p.dLr; sG = dchan * p.dLg; sB = dchan * p.dLb; }
    else { sR += dchan * p.dLr; sG += dchan * p.dLg; sB += dchan * p.dLb; }
}

// MERGE: row_merge3 + one plain store per candidate into the wave's own accumulator copy (the default); false = the round-2
// reduction (three row sums, three LDS float atomics into a shared copy), kept selectable for A/B runs (tools/ab_bench.py).
// Measured on MI355X, k_render_bwd single stream (profiles/r03b_ab_bwd_red_dpp_store.json): C3 100.6 -> 94.2 us, dense
// 1 M cloud 526 -> 485 us, C5 shape (QUAD) 362 -> 324 us.
// One work item of the backward: segment `seg` of tile `tile` -- list positions [seg * BWD_SEG, (seg + 1) * BWD_SEG) -- or,
// with seg < 0, the whole list.  ck_slot: the checkpoint the forward left at the segment's deep end (common.h BinLayout::ckpt).
template <bool QUAD, bool MERGE, bool STRICT>
__device__ __forceinline__ void render_bwd_item(const int tile, const int seg, const uint32_t ck_slot,
             const float4* __restrict__ c_final, int W, int H, int gx, const uint2* __restrict__ ranges,
             const uint32_t* __restrict__ point_list, const GaussRec* __restrict__ rec,
             const float* __restrict__ bg, const float* __restrict__ final_Ts,
             const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
             char* __restrict__ bin_base, const GeomHeader* __restrict__ hdr, int force_check)
{
    constexpr int BATCH = QUAD ? LR_QBATCH_BWD : BATCH2;
    __shared__ float4 s_q0[BATCH];      // x, y, Ap = -0.5 conic a, Bp = -conic b      (common.h gauss_power; x log2 e)
    __shared__ float4 s_q1[BATCH];      // Cp = -0.5 conic c (x log2 e), opacity, -, -   (16-byte stride like s_q0 / s_q2: the
                                        // three broadcast reads of a candidate share ONE address register)
    __shared__ float4 s_q2[BATCH];      // r, g, b, -
    __shared__ uint32_t s_id[BATCH];    // emission index (instance slot) of each staged element
    __shared__ uint32_t s_hit[BATCH];   // the forward's quadrant tests of each staged element (common.h BinLayout::quad_hits)
    // per-batch gradient accumulator, columns below.  Every wave owns a copy: a wave meets a staged candidate at most once
    // per batch, so its contribution is a plain store (nothing is ever added twice to one word), and the flush sums the
    // copies in a fixed order: the backward is bit-repeatable at every image size.  (!MERGE, 2-wave shape: both waves add
    // into ONE copy with LDS float atomics -- two operands, so the sum does not depend on which wave comes first.)
    constexpr int NWAVES = QUAD ? 4 : 2;
    constexpr int NACC = (QUAD || MERGE) ? NWAVES : 1;
    __shared__ float s_acc[BATCH][NACC][12];
    __shared__ uint32_t s_wlast[NWAVES];

    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    // this wave's box: 16x8 (pixel A left, pixel B right quadrant), or with QUAD the 8x8 quadrant (w&1, w>>1), pixel A only
    const int x0 = tx * TILE_X + (QUAD ? (w & 1) * 8 : 0), y0 = ty * TILE_Y + (QUAD ? (w >> 1) : w) * 8;
    const int pxA = x0 + (l & 7), pxB = pxA + 8, py = y0 + (l >> 3);
    const bool insA = pxA < W && py < H, insB = !QUAD && pxB < W && py < H;
    const float pyf = (float)py;
    const size_t pixA = (size_t)py * W + pxA, pixB = pixA + 8;
    const size_t N = (size_t)W * H;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    if (total == 0) return;
    const int seg_lo = seg < 0 ? 0 : seg * BWD_SEG;                    // this item's stretch of the list
    const int seg_hi = seg < 0 ? total : min(total, seg_lo + BWD_SEG);
    if (seg_lo >= seg_hi) return;
    // R-dependent parts of the binning buffer, resolved on the device (the host does not know R here)
    const BinLayout BL = bin_layout((long long)hdr->bin_bound);
    const uint32_t* __restrict__ list_gid = reinterpret_cast<const uint32_t*>(bin_base + BL.list_gid);
    float4* __restrict__ inst_grad = reinterpret_cast<float4*>(bin_base + BL.inst_grad);
    const uint32_t* __restrict__ quad_hits = reinterpret_cast<const uint32_t*>(bin_base + BL.quad_hits);
    const float4* __restrict__ ck = reinterpret_cast<const float4*>(bin_base + BL.ckpt) + (size_t)ck_slot * TILE_PIX;

    BwdPix PA, PB;
    PA.pxf = (float)pxA; PB.pxf = (float)pxB;
    PA.T = insA ? final_Ts[pixA] : 0.f; PB.T = insB ? final_Ts[pixB] : 0.f;
    PA.last = insA ? n_contrib[pixA] : 0u; PB.last = insB ? n_contrib[pixB] : 0u;
    PA.dLr = PA.dLg = PA.dLb = 0.f; PB.dLr = PB.dLg = PB.dLb = 0.f;
    if (insA) { PA.dLr = dL_dpix[pixA]; PA.dLg = dL_dpix[N + pixA]; PA.dLb = dL_dpix[2 * N + pixA]; }
    if (insB) { PB.dLr = dL_dpix[pixB]; PB.dLg = dL_dpix[N + pixB]; PB.dLb = dL_dpix[2 * N + pixB]; }
    PA.A = bg[0] * PA.dLr + bg[1] * PA.dLg + bg[2] * PA.dLb;       // background . dL/dpixel: the deepest layer
    PB.A = bg[0] * PB.dLr + bg[1] * PB.dLg + bg[2] * PB.dLb;
    if (seg_hi < total) {
        // The recursion starts in the middle of the list.  A pixel the forward was still blending at position seg_hi
        // (last > seg_hi: it stopped later, or never) takes the forward's own T there and the colour still to come behind it --
        // its final colour minus the checkpoint's colour so far (render_fwd.hip); a pixel that had stopped by then starts as at the list's end (final T,
        // background behind it): every position of this segment at or behind its `last` is skipped anyway.
        const int qA = QUAD ? w : 2 * w;                         // quadrant of pixel A (pixel index of the checkpoint: quadrant * 64 + lane)
        if (PA.last > (uint32_t)seg_hi) {
            const float4 c = ck[qA * 64 + l], f = c_final[pixA];       // {T, colour so far} there; the pixel's final colour
            PA.T = c.x;
            PA.A = ((f.x - c.y) * PA.dLr + (f.y - c.z) * PA.dLg + (f.z - c.w) * PA.dLb) * __builtin_amdgcn_rcpf(c.x);
        }
        if (!QUAD && PB.last > (uint32_t)seg_hi) {
            const float4 c = ck[(qA + 1) * 64 + l], f = c_final[pixB];
            PB.T = c.x;
            PB.A = ((f.x - c.y) * PB.dLr + (f.y - c.z) * PB.dLg + (f.z - c.w) * PB.dLb) * __builtin_amdgcn_rcpf(c.x);
        }
    }
    const uint32_t lastL = wave_max_u32(PA.last), lastR = QUAD ? 0u : wave_max_u32(PB.last);   // per quadrant
    const uint32_t wave_last = max(lastL, lastR);                   // nothing at or behind this matters to the wave
    // did the forward stop any pixel of this wave early?  (pixels outside the image carry T = 0 and dL = 0: whatever they step
    // through contributes exact zeros, so they do not count)
    const bool any_stopped = force_check != 0 ||
                             __ballot((insA && PA.last < (uint32_t)total) || (insB && PB.last < (uint32_t)total)) != 0ull;
    const float ddelx_dx = (float)(0.5 * W);     // backward.cu:473-474 (double product, rounded once)
    const float ddely_dy = (float)(0.5 * H);

    // block-uniform: the deepest contributor of any pixel in the tile; batches entirely behind it are skipped
    if (l == 0) s_wlast[w] = wave_last;
    lds_barrier();
    uint32_t tile_last = 0;
#pragma unroll
    for (int i = 0; i < NWAVES; i++) tile_last = max(tile_last, s_wlast[i]);

    // LDS column written by this lane after the reductions: rows of reduce8 hold terms {v0, v2, v1, v3}
    const int row = l >> 4;
    const int col_a = (row == 0) ? 0 : (row == 1) ? 2 : (row == 2) ? 1 : 3;     // reduce8 a: (dmx, dmy, dca, dcb)
                                                                                // reduce8 b: (dcc, dop, dr, dg) at col_a + 4
    const bool row_leader = (l & 15) == 0;
    // MERGE: lanes 0 / 8 / 4 of every row end up with the row's a term, b term and db partial (row_merge3)
    const int l16 = l & 15;
    const bool merge_writer = (l16 & 3) == 0 && l16 < 12;
    const uint32_t merge_off = (uint32_t)(w * 12 + (l16 == 0 ? col_a : l16 == 8 ? col_a + 4 : 8 + row));

    for (int base = 0; base < seg_hi - seg_lo; base += BATCH) {
        // staged element i <-> list position pos = seg_hi-1-base-i (back to front)
        const int cnt = min(BATCH, seg_hi - seg_lo - base);
        const int pos_hi = seg_hi - 1 - base;            // position of staged element 0
        const int pos_lo = pos_hi - (cnt - 1);
        if ((uint32_t)pos_lo >= tile_last) {             // whole batch lies behind every last contributor:
            if (tid < cnt) {                             // its instance slots still have to read as zero
                float4* slot = inst_grad + 3 * (size_t)point_list[range.x + (pos_hi - tid)];
                slot[0] = make_float4(0.f, 0.f, 0.f, 0.f); slot[1] = slot[0]; slot[2] = slot[0];
            }
            continue;
        }
        lds_barrier();                                  // previous batch fully consumed / flushed
        if (tid < cnt) {
            const uint32_t e = point_list[range.x + (pos_hi - tid)];     // emission index of this instance
            s_hit[tid] = quad_hits[range.x + (pos_hi - tid)];
            const uint32_t id = list_gid[range.x + (pos_hi - tid)];      // its Gaussian (BinLayout::list_gid: no gather through e)
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            // as render_fwd.hip: scaled exponent coefficients, or the raw conic in strict mode
            s_q0[tid] = STRICT ? make_float4(a.x, a.y, a.z, a.w) : make_float4(a.x, a.y, (-0.5f * LOG2E) * a.z, -LOG2E * a.w);
            *reinterpret_cast<float2*>(&s_q1[tid]) = STRICT ? make_float2(b.x, b.y) : make_float2((-0.5f * LOG2E) * b.x, b.y);
            s_q2[tid] = make_float4(b.z, b.w, c.x, 0.f);
            s_id[tid] = e;
        }
        if (tid < BATCH) {
#pragma unroll
            for (int a = 0; a < NACC; a++)
#pragma unroll
                for (int k = 0; k < 12; k++) s_acc[tid][a][k] = 0.f;
        }
        lds_barrier();

        for (int sb = 0; sb < cnt; sb += 64) {
            // CULL: lane l takes staged Gaussian sb+l (list position pos_hi-(sb+l)): the forward's quadrant tests (byte q of
            // the word: quadrant q = x half + 2 * y half, as the forward's wave q saw it; measured against repeating box_hit
            // here: profiles/r03t_ab_bwd_cull.json), each quadrant with its own deepest last contributor.  Positions at or
            // behind a quadrant's last contributor may never have been tested by the forward: excluded by pos < last
            bool hitL = false, hitR = false;
            {
                const int j = sb + l;
                const uint32_t pos = (uint32_t)(pos_hi - j);
                if (j < cnt && pos < wave_last) {
                    const uint32_t h = s_hit[j] >> (QUAD ? 8 * w : 16 * w);
                    hitL = pos < lastL && (h & 0xffu) != 0u;
                    if (!QUAD) hitR = pos < lastR && (h & 0xff00u) != 0u;
                }
            }
            const uint64_t maskL = __ballot(hitL), maskR = QUAD ? 0ull : __ballot(hitR);
            uint64_t mask = maskL | maskR;
            auto walk = [&](auto chk) {
            constexpr bool CHECK = decltype(chk)::value;
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;       // staged order is already back to front
                mask &= mask - 1;
                const int j = sb + k;
                const uint32_t pos = (uint32_t)(pos_hi - j);
                const float4 a = s_q0[j];
                const float2 b = *reinterpret_cast<const float2*>(&s_q1[j]);   // Cp, opacity
                const float4 c = s_q2[j];
                const float dys = a.y - pyf;                      // both pixels of a lane share the row
                float r0, r1;
                gauss_row<STRICT>(a.w, b.x, dys, r0, r1);                                       // common.h gauss_power
                float sD = 0.f, sMx = 0.f, sMxx = 0.f, sR = 0.f, sG = 0.f, sB = 0.f;
                if (QUAD || ((maskL >> k) & 1ull)) bwd_pixel<true, true, CHECK, STRICT>(PA, a.z, a.w, b.x, r0, r1, a.x, b.y, c.x, c.y, c.z, pos, sD, sMx, sMxx, sR, sG, sB);
                if (!QUAD && ((maskR >> k) & 1ull)) bwd_pixel<false, false, CHECK, STRICT>(PB, a.z, a.w, b.x, r0, r1, a.x, b.y, c.x, c.y, c.z, pos, sD, sMx, sMxx, sR, sG, sB);
                // both pixels of a lane share dy, so the dy factors are applied to the lane's sums
                const float sMy = dys * sD, sMxy = dys * sMx;
                const float sMyy = dys * sMy;
                float ra = sMx, rb = sMyy;
                reduce8(ra, sMy, sMxx, sMxy, rb, sD, sR, sG);
                if (MERGE) {
                    const float rc = row_merge3(ra, rb, sB);
                    int jo = j * (12 * NACC);
                    asm volatile("" : "+s"(jo));              // scalar product, one v_add for the address (not a v_mad_u64_u32)
                    if (merge_writer) (&s_acc[0][0][0])[jo + (int)merge_off] = rc;
                } else {
                    ra = row_sum(ra); rb = row_sum(rb);
                    float rc = row_sum(sB);                           // every row: its partial of db
                    // keep the last DPP add of each row sum in front of the leader branch (otherwise the compiler sinks the
                    // add into the branch and leaves a v_mov_dpp + v_mov 0 pair behind: 3 instructions instead of 1)
                    asm volatile("" : "+v"(ra), "+v"(rb), "+v"(rc));
                    // one address per lane: columns col_a, col_a + 4, col_a + 8 (the db partial of each row has its own column)
                    float* dst = s_acc[j][QUAD ? w : 0] + col_a;
                    if (row_leader) {
                        atomicAdd(dst, ra);
                        atomicAdd(dst + 4, rb);
                        atomicAdd(dst + 8, rc);
                    }
                }
            }
            };
            if (any_stopped) walk(BoolTag<true>{}); else walk(BoolTag<false>{});
        }
        lds_barrier();
        if (tid < cnt) {
            // every instance owns one 48-byte slot: plain stores, no atomics, and the per-Gaussian sum in
            // k_gauss_bwd runs in a fixed order (the slot is written even when nothing contributed).  The factors that
            // are constant per Gaussian (opacity, conic entries, -0.5, the NDC scale of backward.cu:473-474) go in here.
            float a9[12];                              // sums of D dx, D dy, D dx^2, D dx dy, D dy^2, D, dr, dg, db (x4)
#pragma unroll
            for (int k = 0; k < 12; k++) {
                float v = s_acc[tid][0][k];
#pragma unroll
                for (int a = 1; a < NACC; a++) v += s_acc[tid][a][k];      // fixed order over the waves
                a9[k] = v;
            }
            const float4 q0 = s_q0[tid]; const float2 q1 = *reinterpret_cast<const float2*>(&s_q1[tid]);
            const float db = (a9[8] + a9[9]) + (a9[10] + a9[11]);
            const float ca = STRICT ? q0.z : (-2.0f * LN2) * q0.z, cb = STRICT ? q0.w : -LN2 * q0.w,
                        cc = STRICT ? q1.x : (-2.0f * LN2) * q1.x, o = q1.y;                   // conic (back from the scaled staging)
            // the moment sums carry the opacity factor already (bwd_pixel); dL/dopacity = sum G dL/dalpha = a9[5] / opacity
            // (an instance with opacity <= 0 never passes the 1/255 test: all its sums are zero)
            const float sx = a9[0], sy = a9[1], h = -0.5f;
            const float dopac = o > 0.f ? a9[5] * __builtin_amdgcn_rcpf(o) : 0.f;      // 1 ulp: the sum itself carries more
            float4* slot = inst_grad + 3 * (size_t)s_id[tid];
            slot[0] = make_float4((-ca * sx - cb * sy) * ddelx_dx, (-cc * sy - cb * sx) * ddely_dy, h * a9[2], h * a9[3]);
            slot[1] = make_float4(h * a9[4], dopac, a9[6], a9[7]);
            slot[2] = make_float4(db, 0.f, 0.f, 0.f);
        }
    }
}


// The TILE shape: ONE wave64 per 16x16 tile, a lane owns FOUR pixels -- the same position (l & 7, l >> 3) in each of the
// tile's four 8x8 quadrants.  What it is for: everything a wave pays per candidate that is not a pixel step -- the LDS reads of
// the staged element, the loop bookkeeping and above all the cross-lane reduction of the nine sums (29 of ~78 VALU
// instructions per candidate and wave in the 2-wave shape) -- is paid once per tile instance instead of once per half tile
// the instance reaches (1.55 at C3, 1.9 on dense clouds); the pixel steps stay per quadrant that passed the forward's box
// test (2.33 of 4 at C3).  The two rows of quadrants have their own dy, so the lane keeps the moments of the upper and the
// lower pixel pair apart (D, D dx, D dx^2 each) and applies the dy factors to each pair before the reduction.  Single-wave
// workgroups: 8160 of them at 1080p, no partner wave to wait for at the batch barriers.
template <int BATCH, bool STRICT, bool LDSRED = false>
__device__ __forceinline__ void render_bwd_tile(const int tile, const int seg, const uint32_t ck_slot,
                  const float4* __restrict__ c_final, int W, int H, int gx, const uint2* __restrict__ ranges,
                  const uint32_t* __restrict__ point_list, const GaussRec* __restrict__ rec,
                  const float* __restrict__ bg, const float* __restrict__ final_Ts,
                  const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                  char* __restrict__ bin_base, const GeomHeader* __restrict__ hdr, int force_check)
{
    __shared__ float4 s_q0[BATCH];      // as k_render_bwd
    __shared__ float4 s_q1[BATCH];
    __shared__ float4 s_q2[BATCH];
    __shared__ uint32_t s_id[BATCH];
    __shared__ float s_acc[BATCH * 12]; // one wave, one copy: a candidate is met once per batch, so its sums are plain stores
    __shared__ __attribute__((aligned(256))) float s_tr[LDSRED ? 8 * 64 : 1];      // lds_reduce9's planes

    const int tx = tile % gx, ty = tile / gx;
    const int l = threadIdx.x;
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y;
    const int pxL = x0 + (l & 7), pxR = pxL + 8, pyT = y0 + (l >> 3), pyB = pyT + 8;
    const float pyTf = (float)pyT, pyBf = (float)pyB;
    const size_t N = (size_t)W * H;

    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    if (total == 0) return;
    const int seg_lo = seg < 0 ? 0 : seg * BWD_SEG;                    // as render_bwd_item
    const int seg_hi = seg < 0 ? total : min(total, seg_lo + BWD_SEG);
    if (seg_lo >= seg_hi) return;
    const BinLayout BL = bin_layout((long long)hdr->bin_bound);
    const uint32_t* __restrict__ list_gid = reinterpret_cast<const uint32_t*>(bin_base + BL.list_gid);
    float4* __restrict__ inst_grad = reinterpret_cast<float4*>(bin_base + BL.inst_grad);
    const uint32_t* __restrict__ quad_hits = reinterpret_cast<const uint32_t*>(bin_base + BL.quad_hits);
    const float4* __restrict__ ck = reinterpret_cast<const float4*>(bin_base + BL.ckpt) + (size_t)ck_slot * TILE_PIX;

    // quadrant q = x half + 2 * y half (the forward's wave q): P0 upper left, P1 upper right, P2 lower left, P3 lower right
    BwdPix P0, P1, P2, P3;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    auto load_pixel = [&](BwdPix& p, int px, int py, int q) {
        const bool ins = px < W && py < H;
        const size_t pix = (size_t)py * W + px;
        p.pxf = (float)px;
        p.T = ins ? final_Ts[pix] : 0.f;
        p.last = ins ? n_contrib[pix] : 0u;
        p.dLr = ins ? dL_dpix[pix] : 0.f;
        p.dLg = ins ? dL_dpix[N + pix] : 0.f;
        p.dLb = ins ? dL_dpix[2 * N + pix] : 0.f;
        p.A = bg0 * p.dLr + bg1 * p.dLg + bg2 * p.dLb;
        if (seg_hi < total && p.last > (uint32_t)seg_hi) {          // as render_bwd_item: the forward's state at the segment's deep end
            const float4 c = ck[q * 64 + l], f = c_final[pix];
            p.T = c.x;
            p.A = ((f.x - c.y) * p.dLr + (f.y - c.z) * p.dLg + (f.z - c.w) * p.dLb) * __builtin_amdgcn_rcpf(c.x);
        }
    };
    load_pixel(P0, pxL, pyT, 0); load_pixel(P1, pxR, pyT, 1); load_pixel(P2, pxL, pyB, 2); load_pixel(P3, pxR, pyB, 3);
    const uint32_t last0 = wave_max_u32(P0.last), last1 = wave_max_u32(P1.last);
    const uint32_t last2 = wave_max_u32(P2.last), last3 = wave_max_u32(P3.last);
    const uint32_t tile_last = max(max(last0, last1), max(last2, last3));

    const int row = l >> 4;
    const int col_a = (row == 0) ? 0 : (row == 1) ? 2 : (row == 2) ? 1 : 3;     // as k_render_bwd (reduce8 / row_merge3 layout)
    const int l16 = l & 15;
    const bool merge_writer = (l16 & 3) == 0 && l16 < 12;
    // lds_reduce9 leaves terms 2 row / 2 row + 1 in lanes 0 / 8 (term t IS column t) and the row's db partial in lane 4
    const int merge_off = LDSRED ? (l16 == 0 ? 2 * row : l16 == 8 ? 2 * row + 1 : 8 + row)
                                 : (l16 == 0 ? col_a : l16 == 8 ? col_a + 4 : 8 + row);
    uint32_t tr_base = 0u, tr_a1 = 0u;
    if (LDSRED) {
        tr_base = (uint32_t)(uintptr_t)s_tr;
        const int t = l >> 3, sh = (l >> 2) & 1;
        tr_a1 = tr_base + (uint32_t)(t * 256 + (((t & 3) < 2) ? sh : 3 - sh) * 64 + (l & 3) * 16);
    }

    for (int base = 0; base < seg_hi - seg_lo; base += BATCH) {
        const int cnt = min(BATCH, seg_hi - seg_lo - base);
        const int pos_hi = seg_hi - 1 - base;            // position of staged element 0 (back to front)
        const int pos_lo = pos_hi - (cnt - 1);
        if ((uint32_t)pos_lo >= tile_last) {             // whole batch behind every last contributor: slots read as zero
            if (l < cnt) {
                float4* slot = inst_grad + 3 * (size_t)point_list[range.x + (pos_hi - l)];
                slot[0] = make_float4(0.f, 0.f, 0.f, 0.f); slot[1] = slot[0]; slot[2] = slot[0];
            }
            continue;
        }
        lds_barrier();
        // the lane's staging / flush addresses are formed HERE, per round, from a lane index the optimiser cannot see through:
        // hoisted out of the loop they are seven more live registers in a kernel that has 64, i.e. seven spilled dwords per
        // lane (15 MB of scratch stores per 1080p view, round 6) for a handful of integer operations per round
        int lv = l;
        asm volatile("" : "+v"(lv));
        uint32_t my_hit = 0u;
        if (lv < cnt) {
            const uint32_t e = point_list[range.x + (pos_hi - lv)];
            my_hit = quad_hits[range.x + (pos_hi - lv)];
            const uint32_t id = list_gid[range.x + (pos_hi - lv)];       // (BinLayout::list_gid: no gather through e)
            const float4* g = reinterpret_cast<const float4*>(rec + id);
            const float4 a = g[0], b = g[1], c = g[2];
            s_q0[lv] = STRICT ? make_float4(a.x, a.y, a.z, a.w) : make_float4(a.x, a.y, (-0.5f * LOG2E) * a.z, -LOG2E * a.w);
            *reinterpret_cast<float2*>(&s_q1[lv]) = STRICT ? make_float2(b.x, b.y) : make_float2((-0.5f * LOG2E) * b.x, b.y);
            s_q2[lv] = make_float4(b.z, b.w, c.x, 0.f);
            s_id[lv] = e;
        }
        if (lv < BATCH) {
            float4* z = reinterpret_cast<float4*>(&s_acc[lv * 12]);
            z[0] = make_float4(0.f, 0.f, 0.f, 0.f); z[1] = z[0]; z[2] = z[0];
        }
        lds_barrier();

        // CULL: the staging lane IS the lane that holds the element's quadrant tests (one batch = one wave-wide chunk)
        const uint32_t pos_l = (uint32_t)(pos_hi - l);
        const bool inb = l < cnt;
        const uint64_t m0 = __ballot(inb && pos_l < last0 && (my_hit & 0x000000ffu) != 0u);
        const uint64_t m1 = __ballot(inb && pos_l < last1 && (my_hit & 0x0000ff00u) != 0u);
        const uint64_t m2 = __ballot(inb && pos_l < last2 && (my_hit & 0x00ff0000u) != 0u);
        const uint64_t m3 = __ballot(inb && pos_l < last3 && (my_hit & 0xff000000u) != 0u);
        uint64_t mask = (m0 | m1) | (m2 | m3);
        // (always the copy WITH the `pos < last` test: a second copy of this loop costs the 64-VGPR budget 28 more spilled bytes
        //  and the kernel 5 % -- measured 96.0 -> 101.0 us at C3, profiles/r04i_ab_bwd_nocheck_tile.json)
        constexpr bool CHECK = true;
        while (mask) {
            const int j = __ffsll((long long)mask) - 1;       // staged order is already back to front
            mask &= mask - 1;
            const uint32_t pos = (uint32_t)(pos_hi - j);
            const float4 a = s_q0[j];
            const float2 b = *reinterpret_cast<const float2*>(&s_q1[j]);   // Cp, opacity
            const float4 c = s_q2[j];
            const float dysT = a.y - pyTf, dysB = a.y - pyBf;
            float tD = 0.f, tMx = 0.f, tMxx = 0.f, bD = 0.f, bMx = 0.f, bMxx = 0.f, sR = 0.f, sG = 0.f, sB = 0.f;
            const bool h0 = (m0 >> j) & 1ull, h1 = (m1 >> j) & 1ull, h2 = (m2 >> j) & 1ull, h3 = (m3 >> j) & 1ull;
            if (h0 || h1) {
                float Bd, Cdd;
                gauss_row<STRICT>(a.w, b.x, dysT, Bd, Cdd);                                              // common.h gauss_power
                if (h0) bwd_pixel<true, true, CHECK, STRICT>(P0, a.z, a.w, b.x, Bd, Cdd, a.x, b.y, c.x, c.y, c.z, pos, tD, tMx, tMxx, sR, sG, sB);
                if (h1) bwd_pixel<false, false, CHECK, STRICT>(P1, a.z, a.w, b.x, Bd, Cdd, a.x, b.y, c.x, c.y, c.z, pos, tD, tMx, tMxx, sR, sG, sB);
            }
            if (h2 || h3) {
                float Bd, Cdd;
                gauss_row<STRICT>(a.w, b.x, dysB, Bd, Cdd);
                if (h2) bwd_pixel<true, false, CHECK, STRICT>(P2, a.z, a.w, b.x, Bd, Cdd, a.x, b.y, c.x, c.y, c.z, pos, bD, bMx, bMxx, sR, sG, sB);
                if (h3) bwd_pixel<false, false, CHECK, STRICT>(P3, a.z, a.w, b.x, Bd, Cdd, a.x, b.y, c.x, c.y, c.z, pos, bD, bMx, bMxx, sR, sG, sB);
            }
            // the dy factors, per pixel pair (a pair shares its row)
            const float t1 = dysT * tD, t2 = dysB * bD;
            const float sD = tD + bD, sMxx = tMxx + bMxx;
            const float sMy = t1 + t2;
            const float sMxy = dysT * tMx + dysB * bMx;
            const float sMyy = dysT * t1 + dysB * t2;
            float rc;
            if (LDSRED) {
                rc = lds_reduce9(tMx + bMx, sMy, sMxx, sMxy, sMyy, sD, sR, sG, sB, tr_base, tr_a1, tr_a1 ^ 128u);
            } else {
                float ra = tMx + bMx;
                float rb = sMyy;
                reduce8(ra, sMy, sMxx, sMxy, rb, sD, sR, sG);
                rc = row_merge3(ra, rb, sB);
            }
            int jo = j * 12;
            asm volatile("" : "+s"(jo));                      // scalar product, one v_add for the address (not a v_mad_u64_u32)
            if (merge_writer) s_acc[jo + merge_off] = rc;
        }
        lds_barrier();
        if (lv < cnt) {
            int Wv = W, Hv = H;
            asm volatile("" : "+s"(Wv), "+s"(Hv));               // (as lv: formed per round, not kept in two registers)
            const float ddelx_dx = (float)(0.5 * Wv);            // backward.cu:473-474 (double product, rounded once)
            const float ddely_dy = (float)(0.5 * Hv);
            float a9[12];
#pragma unroll
            for (int k = 0; k < 12; k++) a9[k] = s_acc[lv * 12 + k];
            const float4 q0 = s_q0[lv]; const float2 q1 = *reinterpret_cast<const float2*>(&s_q1[lv]);
            const float db = (a9[8] + a9[9]) + (a9[10] + a9[11]);
            const float ca = STRICT ? q0.z : (-2.0f * LN2) * q0.z, cb = STRICT ? q0.w : -LN2 * q0.w,
                        cc = STRICT ? q1.x : (-2.0f * LN2) * q1.x, o = q1.y;
            const float sx = a9[0], sy = a9[1], h = -0.5f;          // as k_render_bwd: the sums carry the opacity factor
            const float dopac = o > 0.f ? a9[5] * __builtin_amdgcn_rcpf(o) : 0.f;      // 1 ulp: the sum itself carries more
            float4* slot = inst_grad + 3 * (size_t)s_id[lv];
            slot[0] = make_float4((-ca * sx - cb * sy) * ddelx_dx, (-cc * sy - cb * sx) * ddely_dy, h * a9[2], h * a9[3]);
            slot[1] = make_float4(h * a9[4], dopac, a9[6], a9[7]);
            slot[2] = make_float4(db, 0.f, 0.f, 0.f);
        }
    }
}

#define LR_BWD_PARAMS int W, int H, int gx, int num_tiles, int tile_map, const uint2* __restrict__ ranges,                   \
                      const uint32_t* __restrict__ point_list, const GaussRec* __restrict__ rec,                        \
                      const float* __restrict__ bg, const float* __restrict__ final_Ts,                                 \
                      const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,                        \
                      char* __restrict__ bin_base, const GeomHeader* __restrict__ hdr, int force_check
#define LR_BWD_PASS W, H, gx, num_tiles, tile_map, ranges, point_list, rec, bg, final_Ts, n_contrib, dL_dpix, bin_base, hdr, force_check
// Work item of a workgroup.  The first `grid_tiles` workgroups take the FIRST segment of their tile (blend_tile: XCD-aware
// tile map) -- or, with seg_on == 0, its whole list; the others walk the segments the forward listed (BinLayout::seg_list;
// their number is on the device, so the launch covers an upper bound and a workgroup strides over what is there: any launch
// size is correct).  Async mode: a view that needed more tile instances than its binning buffer held (flag set by the
// forward's scan) is NOT differentiated -- its instance list is truncated.  k_gauss_bwd skips it as well.
#define LR_BWD_KERNEL_BODY(ITEM)                                                                                            \
    if (hdr->overflow != 0u) return;                                                                                        \
    const bool listed = (int)blockIdx.x >= grid_tiles;                                                                      \
    int tile = 0, seg = seg_on ? 0 : -1;                                                                                    \
    uint32_t slot = 0u, e = 0u, n = 0u;                                                                                     \
    const uint2* __restrict__ seg_list = nullptr;                                                                           \
    if (!listed) {                                                                                                          \
        tile = blend_tile(tile_map, num_tiles);                                                                             \
        if (tile < 0) return;                                                                                               \
        if (seg_on) slot = tile_seg0[tile];             /* read only by a tile longer than one segment, which wrote it */   \
    } else {                                                                                                                \
        seg_list = reinterpret_cast<const uint2*>(bin_base + bin_layout((long long)hdr->bin_bound).seg_list);               \
        n = hdr->n_seg;                                                                                                     \
        e = blockIdx.x - (uint32_t)grid_tiles;                                                                              \
    }                                                                                                                       \
    for (;;) {                                          /* ONE inlined copy of the item for both kinds of workgroup */      \
        if (listed) {                                                                                                       \
            if (e >= n) break;                                                                                              \
            const uint2 ts = seg_list[e];               /* slot e holds segment ts.y; the checkpoint at its deep end: e + 1 */ \
            tile = (int)ts.x; seg = (int)ts.y; slot = e + 1u;                                                               \
        }                                                                                                                   \
        ITEM(tile, seg, slot, c_final, W, H, gx, ranges, point_list, rec, bg, final_Ts, n_contrib, dL_dpix, bin_base, hdr, force_check); \
        if (!listed) break;                                                                                                 \
        e += gridDim.x - (uint32_t)grid_tiles;                                                                              \
        lds_barrier();                                  /* the next item stages into the same LDS */                        \
    }
// The same without the stride loop: workgroup grid_tiles + e takes listed segment e and nothing else.  For the one-wave-per-tile
// shape, whose 64-register budget the loop costs three more spilled dwords per lane (+6 % at C3, profiles/r05e_pmc_c3.json).  The
// launch must then cover every listed segment: it does whenever the caller hands lr_backward the R / binning_capacity of its
// forward, as the reference's own backward requires of R (rasterizer_impl.cu:364-366: the binning state is laid out from it).
#define LR_BWD_KERNEL_BODY_ONE(ITEM)                                                                                        \
    if (hdr->overflow != 0u) return;                                                                                        \
    int tile, seg = seg_on ? 0 : -1;                                                                                        \
    uint32_t slot = 0u;                                                                                                     \
    /* a launch that does not reach every listed segment (the caller's R / binning_capacity was smaller than its forward's): \
       the gradients of the uncovered instances are missing -- said out loud (GeomHeader::bwd_uncovered: lr_check, debug) */ \
    if (blockIdx.x == 0 && threadIdx.x == 0)                                                                                \
        const_cast<GeomHeader*>(hdr)->bwd_uncovered = (seg_on && hdr->n_seg > gridDim.x - (uint32_t)grid_tiles) ? 1u : 0u;  \
    if ((int)blockIdx.x < grid_tiles) {                                                                                     \
        tile = blend_tile(tile_map, num_tiles);                                                                             \
        if (tile < 0) return;                                                                                               \
        if (seg_on) slot = tile_seg0[tile];                                                                                 \
    } else {                                                                                                                \
        const uint32_t e = blockIdx.x - (uint32_t)grid_tiles;                                                               \
        if (e >= hdr->n_seg) return;                                                                                        \
        const uint2 ts = reinterpret_cast<const uint2*>(bin_base + bin_layout((long long)hdr->bin_bound).seg_list)[e];      \
        tile = (int)ts.x; seg = (int)ts.y; slot = e + 1u;                                                                   \
    }                                                                                                                       \
    ITEM(tile, seg, slot, c_final, W, H, gx, ranges, point_list, rec, bg, final_Ts, n_contrib, dL_dpix, bin_base, hdr, force_check);
#define LR_BWD_SEG_PARAMS LR_BWD_PARAMS, const uint32_t* __restrict__ tile_seg0, const float4* __restrict__ c_final, int grid_tiles, int seg_on

// 7 waves per SIMD (<= 72 VGPRs); the QUAD shape's 32 KB of LDS per workgroup allow 4 workgroups = 4 waves per SIMD
template <bool QUAD, bool MERGE, bool STRICT = false>
__global__ void __launch_bounds__(QUAD ? 256 : 128) __attribute__((amdgpu_waves_per_eu(QUAD ? 4 : 7, 8)))
k_render_bwd(LR_BWD_SEG_PARAMS)
{
#define LR_ITEM render_bwd_item<QUAD, MERGE, STRICT>
    LR_BWD_KERNEL_BODY(LR_ITEM)
#undef LR_ITEM
}
// 64 VGPRs and 4.9 KB of LDS: 8 waves per SIMD, so the 8160 waves of a 1080p view are all resident at once (8192 slots).  The
// tiles of a view carry nearly the same load (C3: 70 instances on average, 102 at most): with 7 per SIMD the last 992 waves
// start when the first 7168 finish together and then run alone on their SIMDs, one instruction per ~5 cycles.
// The reduction goes through LDS (lds_reduce9, round 6): 2 KB of planes per wave, so 30 staged Gaussians per round keep the
// workgroup within the 5 KB that 8 waves per SIMD allow (160 KB / 32; the round size itself is worth nothing either way: the
// swap reduction at 30 and at 44 per round measured the same, profiles/r06c_ab_ldsred_matrix.json).
template <bool STRICT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_vgpr(64)))
k_render_bwd_tile(LR_BWD_SEG_PARAMS)
{
#define LR_ITEM render_bwd_tile<30, STRICT, true>
    LR_BWD_KERNEL_BODY_ONE(LR_ITEM)
#undef LR_ITEM
}
#ifdef LR_DIAGNOSTICS
// rounds 4-5: the lane-swap reduction (reduce8 + row_merge3), 44 staged Gaussians per round: A/B partner (bwd_red = 2)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_vgpr(64)))
k_render_bwd_tile_swap(LR_BWD_SEG_PARAMS)
{
#define LR_ITEM render_bwd_tile<44, false, false>
    LR_BWD_KERNEL_BODY_ONE(LR_ITEM)
#undef LR_ITEM
}
// the compiler's own register budget (69 VGPRs, 7 waves per SIMD), 64 staged Gaussians per round: A/B partner (bwd_red = 3)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_render_bwd_tile7(LR_BWD_SEG_PARAMS)
{
#define LR_ITEM render_bwd_tile<64, false>
    LR_BWD_KERNEL_BODY(LR_ITEM)
#undef LR_ITEM
}
#endif

#undef LR_BWD_PARAMS
#undef LR_BWD_PASS

}  // namespace

// Diagnostics builds only (-DLR_DIAGNOSTICS: `python -m luciddreamer_amd.build --diagnostics`, tools/ab_bench.py): an
// environment variable read once per process forces a variant without a call.  The product library reads none.
static int env_knob(const char* name)
{
#ifdef LR_DIAGNOSTICS
    const char* e = getenv(name);
    return e ? atoi(e) : -1;
#else
    (void)name;
    return -1;
#endif
}

int blend_tile_map(int num_tiles)
{
    static const int forced = env_knob("LR_TILE_MAP");
    if (tune_get(TUNE_TILE_MAP) >= 0) return tune_get(TUNE_TILE_MAP);
    if (forced >= 0) return forced;
    return num_tiles <= 4096 ? TILE_MAP_PLAIN : TILE_MAP_BANDS;
}

static thread_local int g_views_in_flight = 1;
ViewsInFlight::ViewsInFlight(int n) : prev(g_views_in_flight) { g_views_in_flight = n; }
ViewsInFlight::~ViewsInFlight() { g_views_in_flight = prev; }
int views_in_flight() { return max(g_views_in_flight, tune_get(TUNE_VIEWS_IN_FLIGHT)); }

int blend_shape(int num_tiles, long long inst_bound)
{
    static const int forced = env_knob("LR_BLEND_QUAD_BWD");
    if (tune_get(TUNE_BLEND_QUAD) >= 0) return tune_get(TUNE_BLEND_QUAD);
    if (forced >= 0) return forced;
    // Round 6 rule, from tools/shape_sweep.py on the final kernels (profiles/r06q_shape_sweep.json: every forced pair of
    // forward / backward shapes on six workloads, three views in flight and one; tests/test_gpu_zz_heuristics.py keeps it honest):
    //   * large images (> 3072 tiles): one wave per tile.  Since its reduction goes through LDS and it no longer spills (round 6)
    //     it beats the 2-wave shape also for a lone view: C2 +5 %, C3 +1.5 %, dense 1080p +6.8 %, 3 M / 1440p +3.3 % views/s
    //     (three views in flight: +3 ... +8 %).  [Rounds 4-5 picked the 2-wave shape for a lone view.]
    //   * small images: every workgroup is resident from the start and the kernel lasts as long as its longest per-wave chain --
    //     the 4-wave shape shortens the chain of a LONE view (pixel-sized splats at 512^2: 4130 against 3890 / 3340 views/s for
    //     2 / 1 waves; dense 1 M cloud: the three within 2 %), but with other views in flight the chains overlap anyway and the
    //     shape that issues the fewest instructions wins: dense 1 M cloud at 512^2 1967 (1 wave) / 1917 (2) / 1787 (4) views/s,
    //     pixel-sized splats 5960 / 6020 / 5800.
    //     A lone view of LONG lists (more than 1024 instances per tile by the bound the caller passes: the dense cloud) is the
    //     one case left to the 2-wave shape: 1410-1430 against 1380-1410 (4 waves) / 1400 (1 wave) views/s.
    if (num_tiles > 3072) return BLEND_TILE;
    if (views_in_flight() >= 2) return BLEND_TILE;
    return inst_bound > 1024ll * num_tiles ? BLEND_HALF : BLEND_QUAD;
}

// shapes of the process's last blend launches (lr_last_launch_shapes: tests assert which kernels a configuration ran; process-wide,
// not per thread: the autograd engine issues a backward from its own thread)
static volatile int g_last_fwd_shape = -1, g_last_bwd_shape = -1;
void note_fwd_shape(int shape) { g_last_fwd_shape = shape; }
int last_fwd_shape() { return g_last_fwd_shape; }
int last_bwd_shape() { return g_last_bwd_shape; }

void launch_render_bwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const GaussRec* rec, const float* bg, const float* final_T,
                       const uint32_t* n_contrib, const float* dL_dpix, char* bin_base, const GeomHeader* hdr,
                       const uint32_t* tile_seg0, const float4* c_final, long long seg_bound, hipStream_t s)
{
    const int num_tiles = gx * gy;
    if (num_tiles <= 0) return;
    const int tile_map = blend_tile_map(num_tiles);
    const int grid = ((num_tiles + 7) / 8) * 8;
    // lr_tune_set("bwd_red", 4): every wave walks the copy of the loop WITH the `pos < last` test (the copy a wave with a stopped
    // pixel takes: tests put whole scenes through it).  Diagnostics builds: 0 = the round-2 reduction, 3 = the 7-waves-per-SIMD
    // tile kernel, LR_BWD_LDS_PAD=<bytes> of unused dynamic LDS lowers the occupancy of the 2-wave shape.
    static const int forced_red = env_knob("LR_BWD_RED");
    const int red = tune_get(TUNE_BWD_RED) >= 0 ? tune_get(TUNE_BWD_RED) : forced_red;
    const int force_check = red == 4 ? 1 : 0;
#ifdef LR_DIAGNOSTICS
    static const int forced_pad = env_knob("LR_BWD_LDS_PAD");
    const int pad = forced_pad >= 0 ? forced_pad : 0;
    const bool merge = red != 0;
#else
    constexpr int pad = 0;
#endif
#define LR_BWD_ARGS W, H, gx, num_tiles, tile_map, ranges, point_list, rec, bg, final_T, n_contrib, dL_dpix, bin_base, hdr, force_check
    // instances the caller's bound allows (bin_seg_capacity(bound) = bound / BWD_SEG + 2): an upper estimate of the lists' lengths
    int shape = blend_shape(num_tiles, seg_bound > 2 ? (seg_bound - 2) * (long long)BWD_SEG : -1);
    const bool strict = tune_get(TUNE_STRICT) > 0;
    // Segments (common.h BWD_SEG): on by default; lr_tune_set("bwd_seg", 0) = one workgroup walks a tile's whole list (rounds
    // 1-4: what the segment tests compare with).  The launch adds workgroups for the listed segments up to the bound the host
    // knows, capped (a workgroup strides over the list, so the cap costs nothing but balance on absurdly long lists).
    static const int forced_seg = env_knob("LR_BWD_SEG");
    const int seg_on = (tune_get(TUNE_BWD_SEG) >= 0 ? tune_get(TUNE_BWD_SEG) : (forced_seg >= 0 ? forced_seg : 1)) != 0 ? 1 : 0;
    // (the one-wave-per-tile shape takes exactly one listed segment per workgroup: no cap there, and it is not used when the
    // caller gave no usable bound -- seg_bound < 0, see lr_backward -- or when segments are off and its workgroup walks the whole list)
    if (shape == BLEND_TILE && seg_on && seg_bound < 0) shape = BLEND_HALF;
    g_last_bwd_shape = shape;
    const long long cap = shape == BLEND_TILE ? 0x3fffffffll : 262144ll;
    const int extra = seg_on ? (int)(seg_bound < 1 ? 1024 : seg_bound > cap ? cap : seg_bound) : 0;
    const dim3 g(grid + extra);
#define LR_SEG_ARGS LR_BWD_ARGS, tile_seg0, c_final, grid, seg_on
    if (shape == BLEND_TILE) {
        if (strict) hipLaunchKernelGGL(k_render_bwd_tile<true>, g, dim3(64), 0, s, LR_SEG_ARGS);
#ifdef LR_DIAGNOSTICS
        else if (red == 2) hipLaunchKernelGGL(k_render_bwd_tile_swap, g, dim3(64), 0, s, LR_SEG_ARGS);
        else if (red == 3) hipLaunchKernelGGL(k_render_bwd_tile7, g, dim3(64), 0, s, LR_SEG_ARGS);
#endif
        else hipLaunchKernelGGL(k_render_bwd_tile<false>, g, dim3(64), 0, s, LR_SEG_ARGS);
    } else if (strict) {
        if (shape == BLEND_QUAD) hipLaunchKernelGGL((k_render_bwd<true, true, true>), g, dim3(256), 0, s, LR_SEG_ARGS);
        else hipLaunchKernelGGL((k_render_bwd<false, true, true>), g, dim3(128), pad, s, LR_SEG_ARGS);
#ifdef LR_DIAGNOSTICS
    } else if (!merge) {                    // the round-2 reduction (three row sums, three LDS float atomics into a shared copy)
        if (shape == BLEND_QUAD) hipLaunchKernelGGL((k_render_bwd<true, false>), g, dim3(256), 0, s, LR_SEG_ARGS);
        else hipLaunchKernelGGL((k_render_bwd<false, false>), g, dim3(128), pad, s, LR_SEG_ARGS);
#endif
    } else if (shape == BLEND_QUAD) {
        hipLaunchKernelGGL((k_render_bwd<true, true>), g, dim3(256), 0, s, LR_SEG_ARGS);
    } else {
        hipLaunchKernelGGL((k_render_bwd<false, true>), g, dim3(128), pad, s, LR_SEG_ARGS);
    }
#undef LR_SEG_ARGS
#undef LR_BWD_ARGS
}

}  // namespace lr
