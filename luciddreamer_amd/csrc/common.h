// common.h -- shared layouts and launcher declarations of the gfx950 rasterizer library.
// Written for MI355X (gfx950, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace lr {

constexpr int TILE_X = 16;          // RAST/cuda_rasterizer/config.h:16-17 (tile geometry is part of the
constexpr int TILE_Y = 16;          // observable behaviour: it decides which Gaussians reach a pixel)
constexpr int TILE_PIX = TILE_X * TILE_Y;
constexpr int WAVE = 64;

// Per-Gaussian screen-space record produced by preprocess and gathered by the blend kernels.
// One 48-byte record = three 16-byte loads from ONE place instead of gathers from four arrays
// (means2D / conic_opacity / rgb / depths in the reference's GeometryState).
struct __attribute__((aligned(16))) GaussRec {
    float x, y, ca, cb;             // pixel-space mean, conic a, conic b
    float cc, opacity, r, g;        // conic c, opacity, colour r, g
    float b, depth, qmax, pad1;     // colour b, view-space depth, log(255*opacity)+CULL_MARGIN (cull_qmax)
};
static_assert(sizeof(GaussRec) == 48, "GaussRec must be 48 bytes");

// Per-Gaussian gradient accumulator filled by the blend backward (one record, nine atomics).
struct __attribute__((aligned(16))) GradRec {
    float dmx, dmy, dca, dcb;       // dL/dmean2D.xy (NDC-scaled), dL/dconic a, b
    float dcc, dop, dr, dg;         // dL/dconic c, dL/dopacity, dL/dcolour r, g
    float db, pad0, pad1, pad2;
};
static_assert(sizeof(GradRec) == 48, "GradRec must be 48 bytes");

// What the binning needs of an emitting Gaussian, in one 16-byte gather: the result of preprocess's exact tile test as a
// bit mask over its tile rectangle (bit j = tile (miny + j / rw, minx + j % rw) passes; rectangles of up to HIT_MASK_TILES
// tiles), the rectangle's origin and width, and the depth bits of the sort word.  The count and scatter passes of the binning
// walk the set bits instead of repeating the ~45-instruction test per tile (it used to run three times per pair).
//   x, y : mask bits 0-31, 32-63      z : minx | miny << 12 | rw << 24, or 0 = no mask (larger rectangle, or a tile grid
//   w    : view depth as uint32                                               beyond 4096: the walk tests / emits from the record)
constexpr uint32_t HIT_MASK_TILES = 64;
constexpr int HIT_ORIGIN_LIMIT = 4096;       // rectangle origins the record can hold (12 bits each)
__host__ __device__ inline uint32_t hit_geo(int minx, int miny, int rw, uint32_t area, int origin_limit)
{
    return (area <= HIT_MASK_TILES && minx < origin_limit && miny < origin_limit)
               ? ((uint32_t)minx | ((uint32_t)miny << 12) | ((uint32_t)rw << 24)) : 0u;
}

// Header at offset 0 of the geom buffer (device-resident view state).
struct GeomHeader {
    uint32_t num_rendered;          // the reference's num_rendered: sum of tile-rectangle areas (reported to callers)
    uint32_t overflow;              // 1 if num_instances exceeded the binning capacity (async mode)
    uint32_t prefilter_trap;        // 1 if a culled point was seen with prefiltered=true
    uint32_t capacity;              // binning capacity (instances)
    uint32_t P;
    uint32_t num_sorted;            // min(num_instances, capacity): length of the instance list actually built
    uint32_t num_instances;         // tile instances after exact tile culling (what is emitted and sorted)
    uint32_t bin_bound;             // instance count the binning buffer was laid out for (R or capacity)
    uint32_t num_compact;           // Gaussians that emit at least one instance (length of the compacted arrays)
    uint32_t n_seg;                 // list segments beyond the first of every tile (BWD_SEG, below): entries of BinLayout::seg_list
    uint32_t bwd_uncovered;         // set by the one-wave-per-tile blend backward when its launch did not reach every listed segment
                                    // (lr_backward given a smaller R / binning_capacity than the forward): lr_check and debug mode report it
    uint32_t reserved[49];
    uint32_t sticky_overflow;       // set (never cleared by lr_forward) when a view overflowed: lr_views_accumulate / lr_views_check
    uint32_t reserved_tail[3];
};
static_assert(sizeof(GeomHeader) == 256, "GeomHeader must be 256 bytes");

// Workgroup barrier with the wait for this wave's own LDS operations spelled out.  __syncthreads() implies it (workgroup-scope
// release), but ROCm 7.2's hipcc dropped the `s_waitcnt lgkmcnt(0)` in front of the s_barrier that closes a batch of
// k_render_bwd's two-wave shape (a divergent ds_write / ds_add_f32 at the end of the candidate loop, then the loop exit): the
// other wave's flush could read the accumulator before the last store had landed.  Found as a timing-dependent gradient of
// ONE Gaussian in ~5 % of the runs of the LDS-atomic reduction (whose ds_add_f32 takes ~3 cycles per lane: a wide window);
// the default reduction's plain store never showed it in 2000 runs, but the instruction was missing there too.  Inline asm is
// opaque to the waitcnt insertion, so this wait cannot be optimised away; every barrier of the library goes through here.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

constexpr size_t ALIGN = 256;
__host__ __device__ inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

// ---- radix sort geometry -------------------------------------------------------------------
constexpr int RADIX_BITS = 8;
constexpr int RADIX_SIZE = 1 << RADIX_BITS;
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;                               // keys per thread per sub-tile
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;        // 2048 keys per sub-tile
constexpr int SORT_MAX_BLOCKS = 1024;

struct SortPlan { int nblocks; int chunk; };                // chunk: keys per block (multiple of SORT_TILE)
inline SortPlan sort_plan(long long n_bound) {
    SortPlan p;
    long long tiles = (n_bound + SORT_TILE - 1) / SORT_TILE;
    if (tiles < 1) tiles = 1;
    long long nb = tiles < SORT_MAX_BLOCKS ? tiles : SORT_MAX_BLOCKS;
    long long tiles_per_block = (tiles + nb - 1) / nb;
    p.chunk = (int)(tiles_per_block * SORT_TILE);
    p.nblocks = (int)((tiles + tiles_per_block - 1) / tiles_per_block);
    return p;
}
__host__ __device__ inline size_t sort_hist_bytes(long long n_bound) {
    return align_up((size_t)SORT_MAX_BLOCKS * RADIX_SIZE * 4 + RADIX_SIZE * 4);
}

// ---- scan geometry ---------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// ---- tile binning geometry (tilebin.hip) -----------------------------------------------------
constexpr int PART_BINS_MAX = 16384;        // partition bins: one 4-byte LDS counter each (64 KB)
constexpr int PART_BLOCKS_MAX = 256;        // workgroups of the count / scatter kernels (rows of the histogram)
constexpr int PART_THREADS = 1024;
constexpr int PART_MIN_GAUSS = 1024;        // emitting Gaussians per workgroup before another workgroup is used
constexpr int TSORT_LDS = 256;              // bin entries sorted by one wave (2 KB of LDS)
constexpr int TSORT_GROUP_LDS = 1024;        // bin entries the four waves of a k_tile_sort_small workgroup sort together (8 KB)
constexpr int TSORT_THREADS = 512;          // threads of k_tile_sort_large (bins of more than 1024 entries)
constexpr int TSORT_MID_LDS = 4096;         // ... its bin entries (32 KB of words + 16 KB of bucket counters)
constexpr int TSORT_BIG_LDS = 16384;        // ... entries sorted in LDS when the launch asks for 128 KB; beyond: in place in global memory
constexpr int TSORT_BIG_BLOCKS = 256;
constexpr int TSORT_CLASS_BLOCKS = 2048;     // grid cap of part B of k_tile_sort_small (bins of 257..1024 entries)
constexpr int TSORT_LARGE_BLOCKS = 768;      // grid cap of k_tile_sort_large (queue-fed; 48 KB of LDS each: three per CU are resident)
struct PartPlan { int bins; int sub_shift; };   // bin = tile >> sub_shift (0 up to 16384 tiles)
PartPlan part_plan(int num_tiles);

// ---- buffer layouts ----------------------------------------------------------------------------
struct GeomLayout {
    size_t header, rec, clamped, tiles_touched, offsets, vis_list, hitrec, total;
};
inline GeomLayout geom_layout(int P) {
    GeomLayout L; size_t o = 0; size_t Pz = P > 0 ? (size_t)P : 1;
    L.header = o;        o += align_up(sizeof(GeomHeader));
    L.rec = o;           o += align_up(Pz * sizeof(GaussRec));
    L.clamped = o;       o += align_up(Pz);
    L.tiles_touched = o; o += align_up(Pz * 4);      // instances each Gaussian emits (after exact tile culling)
    L.vis_list = o;      o += align_up(Pz * 4);      // ids of the emitting Gaussians, index order (num_compact entries)
    L.offsets = o;       o += align_up(Pz * 4);      // first instance slot of vis_list[k]
    L.hitrec = o;        o += align_up(Pz * 16);     // HitRec per Gaussian (written for the emitting ones only)
    L.total = o;
    return L;
}
struct ImgLayout { size_t final_T, n_contrib, ranges, part_hist, bin_total, bin_start, big_queue, tile_seg0, c_final, total; };
inline ImgLayout img_layout(int W, int H) {
    ImgLayout L; size_t o = 0; size_t N = (size_t)W * H; if (N == 0) N = 1;
    size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y); if (T == 0) T = 1;
    // partition bins (tilebin.hip part_plan, restated here so that the header-only layout needs no library call)
    size_t shift = 0;
    while (((T - 1) >> shift) + 1 > (size_t)PART_BINS_MAX) shift++;
    const size_t bins = ((T - 1) >> shift) + 1;
    L.final_T = o;   o += align_up(N * 4);
    L.n_contrib = o; o += align_up(N * 4);
    L.ranges = o;    o += align_up(T * 8);
    L.part_hist = o; o += align_up((size_t)PART_BLOCKS_MAX * bins * 4);   // per-workgroup bin counts -> bases
    L.bin_total = o; o += align_up(bins * 4);
    L.bin_start = o; o += align_up(bins * 4);
    L.big_queue = o; o += align_up((bins + 1) * 4);
    L.tile_seg0 = o; o += align_up(T * 4);           // first seg_list slot of each tile (written by the blend forward for tiles longer than BWD_SEG)
    L.c_final = o;   o += align_up(N * 16);          // final colour (background included) per pixel, written by the blend forward for the
                                                     // pixels of tiles longer than BWD_SEG only: what the backward's later segments subtract the
                                                     // checkpoint's colour-so-far from (allocation only for every other tile)
    L.total = o;
    return L;
}
// Segments of a tile's list.  The blend backward is a recursion along the list, but each stretch of BWD_SEG positions can
// start on its own once it knows, per pixel, the transmittance in front of its deepest layer and the colour behind it: the
// blend forward leaves {T, colour so far} at every BWD_SEG-th position it reaches (`ckpt`: one float4 per pixel of the tile;
// pixel index = quadrant * 64 + lane of the forward) and the pixel's final colour in ImgLayout::c_final -- the colour still to
// come behind the position is their difference -- and lists the (tile, segment) pairs beyond each
// tile's first segment in `seg_list` (slots reserved with one atomic per tile on GeomHeader::n_seg).  The backward runs one
// workgroup per tile for the first segment and one per listed pair: long lists (hundreds to thousands of instances per
// tile on small images and dense clouds) no longer make the kernel as slow as its longest tile.
constexpr int BWD_SEG = 256;                 // = the staging round of the blend forward
struct BinLayout { size_t point_list, words, quad_hits, inst_gid, inst_grad, seg_list, ckpt, list_gid, total; };
__host__ __device__ inline BinLayout bin_layout(long long R) {
    // point_list sits at offset 0: the tile-sorted list of EMISSION slots e; inst_gid[e] is the Gaussian and
    // inst_grad[e] the slot the blend backward writes that instance's nine partial sums to (one plain 48-byte
    // store per tile instance instead of nine global atomics).  `words` holds the 64-bit [sub-tile | depth | slot]
    // sort words between the scatter and the per-bin sort; once the list is sorted the same bytes hold `quad_hits`:
    // four bytes per LIST POSITION, byte q = "this instance can reach alpha >= 1/255 somewhere in 8x8 quadrant q of its
    // tile" (q = x half + 2 * y half), written by the blend forward's cull and read back by the blend backward, which
    // used to repeat the test.  The backward resolves the R-dependent offsets on the device from GeomHeader::bin_bound.
    BinLayout L; size_t o = 0; size_t Rz = R > 0 ? (size_t)R : 1;
    L.point_list = o; o += align_up(Rz * 4);
    L.words = o;      L.quad_hits = o; o += align_up(Rz * 8);
    L.inst_gid = o;   o += align_up(Rz * 4);
    L.inst_grad = o;  o += align_up(Rz * sizeof(GradRec));
    const size_t nseg = Rz / BWD_SEG + 2;                       // sum over the tiles of (length - 1) / BWD_SEG <= R / BWD_SEG
    L.seg_list = o;   o += align_up(nseg * 8);                  // uint2 {tile, segment}
    L.ckpt = o;       o += align_up(nseg * TILE_PIX * 16);       // float4 per pixel of the tile and listed segment
    // the Gaussian of every LIST POSITION (= inst_gid[point_list[pos]], written by the per-bin sort as it writes the list): the
    // blend kernels stage a round through list -> record instead of list -> slot -> Gaussian -> record -- one dependent gather
    // less per round, and a 4-byte gather from a 64-byte sector less per instance and kernel (round 6)
    L.list_gid = o;   o += align_up(Rz * 4);
    L.total = o;
    return L;
}
inline long long bin_seg_capacity(long long R) { return (R > 0 ? R : 1) / BWD_SEG + 2; }

// ---- exact tile culling ------------------------------------------------------------------------
// A (Gaussian, tile) pair can only matter if some pixel of the tile reaches alpha >= 1/255, i.e. if
// q(d) = 0.5*(a dx^2 + c dy^2) + b dx dy <= log(255*opacity) somewhere on the tile's 16x16 pixel centres.
// tile_hit() minimises the convex quadratic q over the tile's box of offsets exactly (interior or one of
// the four edges) and keeps the pair unless the minimum exceeds log(255*o) + CULL_MARGIN.  The margin is
// orders of magnitude above the float rounding of q / exp, and the blend kernels reject per pixel with the
// smaller margin 0.01, so dropping the pair can never change a pixel.  The reference emits one instance
// per tile of the 3-sigma bounding SQUARE (auxiliary.h:46-56, rasterizer_impl.cu:85-109); the instances
// dropped here are exactly ones its per-pixel `alpha < 1/255 -> continue` (forward.cu:338-339) skips.
constexpr float CULL_MARGIN = 0.02f;
constexpr uint32_t CULL_MAX_TILES = 96;       // larger rectangles are emitted unculled (bounded per-lane loop)

// Minimum-of-quadratic test on an axis-aligned box of pixel centres [x_lo,x_hi] x [y_lo,y_hi].
// r_c = -cb/cc and r_a = -cb/ca are the slopes of the edge-constrained minimisers (one division each per
// Gaussian, not per box).  Evaluated by the preprocess kernels (count + tile mask) and by the binning walk for
// rectangles without a mask -- which must agree bit for bit whatever the contraction setting of the translation
// unit -- and by the blend forward per 8x8 quadrant (the backward reads the forward's outcome back).
__device__ __forceinline__ bool box_hit(float mx, float my, float ca, float cb, float cc, float r_c, float r_a,
                                        float qmax, float x_lo, float x_hi, float y_lo, float y_hi)
{
#pragma clang fp contract(off)
    const float dx_lo = mx - x_hi, dx_hi = mx - x_lo;
    const float dy_lo = my - y_hi, dy_hi = my - y_lo;
    if (dx_lo <= 0.f && dx_hi >= 0.f && dy_lo <= 0.f && dy_hi >= 0.f) return true;
    if (!(ca > 0.f) || !(cc > 0.f)) return true;              // not a proper ellipse: keep (conservative)
    float qmin = 3.0e38f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float ex = k ? dx_hi : dx_lo;
        const float ys = fminf(dy_hi, fmaxf(dy_lo, r_c * ex));
        qmin = fminf(qmin, 0.5f * (ca * ex * ex + cc * ys * ys) + cb * ex * ys);
        const float ey = k ? dy_hi : dy_lo;
        const float xs = fminf(dx_hi, fmaxf(dx_lo, r_a * ey));
        qmin = fminf(qmin, 0.5f * (ca * xs * xs + cc * ey * ey) + cb * xs * ey);
    }
    return !(qmin > qmax);                                     // NaN anywhere -> keep
}
__device__ __forceinline__ bool tile_hit(float mx, float my, float ca, float cb, float cc, float r_c, float r_a,
                                         float qmax, int tx, int ty)
{
    return box_hit(mx, my, ca, cb, cc, r_c, r_a, qmax, (float)(tx * TILE_X), (float)(tx * TILE_X + TILE_X - 1),
                   (float)(ty * TILE_Y), (float)(ty * TILE_Y + TILE_Y - 1));
}
// log(255*o) + margin; opacity <= 0 can never reach 1/255: every tile is dropped (-inf).
__device__ __forceinline__ float cull_qmax(float opacity)
{
    return (opacity > 0.f) ? (logf(255.0f * opacity) + CULL_MARGIN) : -3.0e38f;
}

// ---- launchers (one per translation unit) ------------------------------------------------------
// Exponent of a splat at a pixel (offsets dx, dy from the centre):
//   power = -0.5 (a dx^2 + c dy^2) - b dx dy  (forward.cu:332-334)  =  (Ap dx + Bp dy) dx + Cp dy dy
// with Ap = -0.5 a, Bp = -b, Cp = -0.5 c formed once per Gaussian when it is staged, and Bd = Bp dy, Cdd = (Cp dy) dy
// formed once per candidate and lane (the pixels of a lane that share a row share them).  ONE definition for the blend
// forward and every shape of the blend backward -- and the ROUNDING is part of the definition: the backward re-decides
// `alpha >= 1/255` for every (pixel, candidate) pair, and a layer the forward blended must be a layer the backward
// differentiates.  Left to -ffp-contract=fast the compiler contracted the same source differently per call site (the
// forward and the one-pixel-per-lane backward fused the last product of Cdd into the final add, the two- and four-pixel
// shapes rounded Cdd first): alphas one ulp apart, and on the handful of pixels per view where alpha sits within an ulp of
// 1/255 the backward differentiated a layer the forward had skipped (or the reverse) -- 1-5 gradient rows per view moved by
// up to 1.5e-3 of their tensor's maximum and the shapes disagreed with each other on them (tools/shape_vs_oracle.py).
// So: two rounded products, two explicit FMAs, no contraction across the helpers.
__device__ __forceinline__ float gauss_bd(float Bp, float dy)
{
#pragma clang fp contract(off)
    return Bp * dy;
}
__device__ __forceinline__ float gauss_cdd(float Cp, float dy)
{
#pragma clang fp contract(off)
    const float t = Cp * dy;
    return t * dy;
}
__device__ __forceinline__ float gauss_power1(float Ap, float Bd, float Cdd, float dx)
{
    return __builtin_fmaf(__builtin_fmaf(Ap, dx, Bd), dx, Cdd);
}
// STRICT evaluation (lr_tune_set("strict", 1); luciddreamer_amd.config.set_strict_parity): the reference's own expression in
// the reference's operand order, every operation rounded on its own, and expf --
//     power = -0.5f * (a dx dx + c dy dy) - b dx dy;   alpha = min(0.99f, opacity * exp(power))      (forward.cu:332-337)
// -- i.e. exactly the float operations of the reference compiled without contraction (oracle/_ref, oracle/raster_oracle.c), so
// that every discrete decision of the blend (power > 0, alpha < 1/255, T (1 - alpha) < 1e-4) falls as it does there: images
// within 1e-5 on EVERY pixel instead of "outside the 0-3 threshold pixels of a view".  ~15 more instructions per pixel step
// (nine roundings instead of two FMAs, a range-reduced exp instead of v_exp_f32): a parity instrument, not the default.
// q = {a, b, c} raw conic in strict mode, {Ap, Bp, Cp} scaled otherwise; r0 / r1 = what is formed once per candidate and lane.
template <bool STRICT>
__device__ __forceinline__ void gauss_row(float qB, float qC, float dy, float& r0, float& r1)
{
    if (STRICT) { r0 = dy; r1 = 0.f; }
    else { r0 = gauss_bd(qB, dy); r1 = gauss_cdd(qC, dy); }
}
__device__ __forceinline__ float gauss_power_strict(float ca, float cb, float cc, float dx, float dy)
{
#pragma clang fp contract(off)
    return -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
}
// expf as the device library evaluates it (ocml: x log2(e) as a compensated product, split into integer and fraction,
// v_exp_f32 of the fraction, ldexp: <= 1 ulp), operation for operation -- without its two range checks (x < -104 -> 0,
// x > 88.7 -> inf: a compare and a select each).  The same bits wherever the result is a normal number; beyond, v_ldexp_f32
// under- and overflows to the same 0 / inf up to denormals, and an alpha that small is below 1/255 either way.  The strict-mode
// tests (tests/test_gpu_ref_selfcal.py, test_gpu_variants.py: no pixel beyond 1e-5, no gradient row beyond 1e-4, nothing masked)
// hold unchanged.
__device__ __forceinline__ float strict_expf(float x)
{
#pragma clang fp contract(off)
    const float c = 0x1.715476p+0f, cc = 0x1.4ae0bep-26f;       // log2(e) = c + cc
    const float ph = x * c;
    const float pl = __builtin_fmaf(x, cc, __builtin_fmaf(x, c, -ph));
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
// returns G = exp(power) and the value whose sign decides `power > 0` (log2(e) x power in the default mode)
template <bool STRICT>
__device__ __forceinline__ float gauss_weight(float qA, float qB, float qC, float r0, float r1, float dx, float& power)
{
    if (STRICT) {
        power = gauss_power_strict(qA, qB, qC, dx, r0);
        return strict_expf(power);
    }
    power = gauss_power1(qA, r0, r1, dx);
    return __builtin_amdgcn_exp2f(power);
}

// The blend kernels stage Ap, Bp, Cp (and the cull threshold qmax, which box_hit compares with the same quadratic form)
// multiplied by log2(e), so that G = exp(power) is one v_exp_f32 of the Horner value: the multiply of __expf leaves the
// per-pixel step of both kernels.  `power <= 0` reads the same on the scaled value.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
__device__ __forceinline__ float gauss_exp2(float power2) { return __builtin_amdgcn_exp2f(power2); }

struct ViewParams {
    const float* view;      // device, 16 floats, flat index m[4*col+row] (auxiliary.h:58-77)
    const float* proj;      // device, 16 floats
    const float* campos;    // device, 3 floats
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy, P, D, M;
    int hit_origin_limit;         // HIT_ORIGIN_LIMIT; 0 (lr_tune_set("hit_mask", 0), tests) = no masks: every rectangle walked from its record
    // raw-parameter mode (lr_forward_raw / lr_backward_raw): scales, rotations and opacities are the STORED
    // GaussianModel parameters (pre exp / normalize / sigmoid, R/scene/gaussian_model.py:97-117) and the SH
    // coefficients come as two arrays, `shs` = features_dc [P,1,3] and `sh_rest` = features_rest [P,M-1,3]
    // (no torch.cat copy).  The activations and their derivatives are applied inside the per-Gaussian kernels.
    int raw;
    const float* sh_rest;
    const float* opacity_raw;     // backward only (sigmoid derivative)
    float* dL_dsh_rest;           // backward only
};

// activations of the raw mode; one definition so that forward and backward recompute identical values
__device__ __forceinline__ float act_scale(float s) { return expf(s); }
__device__ __forceinline__ float act_opacity(float o) { return 1.0f / (1.0f + expf(-o)); }
// torch.nn.functional.normalize: q / max(||q||_2, 1e-12)
__device__ __forceinline__ float act_quat_inv_norm(float r, float x, float y, float z)
{
    return 1.0f / fmaxf(sqrtf(r * r + x * x + y * y + z * z), 1e-12f);
}

void launch_preprocess(const ViewParams& vp, const float* means3D, const float* scales, const float* rotations,
                       const float* opacities, const float* shs, const float* cov3D_precomp,
                       const float* colors_precomp, bool prefiltered, int* radii, GaussRec* rec,
                       uint8_t* clamped, uint32_t* tiles_touched, uint4* hitrec, uint32_t* depth_key,
                       GeomHeader* hdr, uint32_t binning_capacity, uint32_t* chunk_sums, bool sparse_view_hint, hipStream_t s);
// chunk_sums: [ceil(P / SCAN_TILE)] x {emitting Gaussians, instances, rectangle areas, -} + one word (the prefilter trap), ZERO on
// entry -- the library's own per-stream scratch (api.hip StreamScratch), cleared again by the first binning kernel once
// k_compact_write has consumed it: no launch that only zeroes.
inline size_t chunk_sum_words(int P) { return 4 * (((size_t)(P > 0 ? P : 1) + SCAN_TILE - 1) / SCAN_TILE) + 1; }
void launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present,
                         hipStream_t s);

// Stable LSD radix sort of (key,val) u32 pairs on bits [0,end_bit).  n_dev: device count (may be
// smaller than n_bound; blocks beyond it idle).  vals_in == nullptr means val = index (iota).
// Returns which buffers hold the result via *keys_out / *vals_out.
void radix_sort_pairs(uint32_t* key_a, uint32_t* key_b, uint32_t* val_a, uint32_t* val_b, bool vals_iota,
                      const uint32_t* n_dev, long long n_bound, int end_bit, uint32_t* hist,
                      uint32_t** keys_out, uint32_t** vals_out, hipStream_t s);

// Order-preserving compaction of the Gaussians with tiles_touched != 0 (vis_list) fused with the exclusive scan of their
// instance counts (offsets, by rank: the per-Gaussian backward reads first slot and count from it); fills every count of the header: num_compact, num_rendered
// (the reference's: sum of the tile-rectangle areas over all Gaussians), num_instances, num_sorted, overflow, bin_bound.
// block_sums: per SCAN_TILE chunk {emitting Gaussians, instances, rectangle areas, -}, accumulated by k_preprocess.
// capacity: the binning capacity of the call (0 = exact mode); log_slot: 12 host-visible words of the library's forward log or
// nullptr -- the kernel leaves the header there behind `log_tag` (lr_header_poll on a log ticket spins on the tag).
void launch_compact(int P, const uint32_t* tiles_touched, const uint4* block_sums,
                    uint32_t* vis_list, uint32_t* offsets, GeomHeader* hdr, uint32_t capacity, uint32_t* log_slot,
                    uint32_t log_tag, uint32_t* zero_words, uint32_t n_zero, bool reset_sticky, hipStream_t s);
// optional per-stage timing hook of launch_tile_binning (api.hip ProfScope events)
struct TileBinTimes { virtual void mark(int boundary, hipStream_t s) = 0; virtual ~TileBinTimes() {} };
// count -> scan -> scatter -> per-bin sort: point_list, inst_gid and ranges from the compacted list.  Returns 0, -1
// (LDS attribute) or -2 (sub-tile + depth + slot bits exceed 64).
int launch_tile_binning(int P, int gx, int gy, int slot_bits, const uint32_t* vis_list, const uint32_t* offsets,
                        const uint4* hitrec, const GaussRec* rec, const int* radii, GeomHeader* hdr,
                        uint32_t* part_hist, uint32_t* bin_total, uint32_t* bin_start, uint32_t* big_queue,
                        uint32_t* inst_gid, unsigned long long* words, uint32_t* point_list, uint32_t* list_gid, uint2* ranges,
                        long long bin_bound_hint, TileBinTimes* t, uint32_t* clear_words, uint32_t n_clear, hipStream_t s);

// Diagnostic tuning knobs (lr_tune_set in api.hip): kernel variants that can be switched at run time so that two of them
// are measured alternately in ONE process on ONE box (tools/ab_bench.py).  -1 = not set (the launcher's own rule).
enum TuneKey { TUNE_BWD_RED = 0, TUNE_BLEND_QUAD, TUNE_TILE_MAP, TUNE_PREPROCESS, TUNE_GAUSS_BWD, TUNE_TSORT, TUNE_WALK_OWN, TUNE_HIT_MASK, TUNE_VIEWS_IN_FLIGHT, TUNE_STRICT, TUNE_PART_SCAN, TUNE_BWD_SEG, TUNE_FWD_PAIR, TUNE_COUNT };
int tune_get(int key);

// Shape of the backward blend kernel (render_bwd.hip, where the rule and its measurements are): BLEND_QUAD = 4 waves per
// tile, one 8x8 quadrant per wave (small images, latency bound), BLEND_HALF = 2 waves per tile, two pixels per lane,
// BLEND_TILE = one wave per tile, four pixels per lane (issue bound: the per-candidate reduction is paid once per tile).
// LR_BLEND_QUAD_BWD=0/1/2 or lr_tune_set("blend_quad", .) forces one (diagnostics).
enum { BLEND_HALF = 0, BLEND_QUAD = 1, BLEND_TILE = 2 };
int blend_shape(int num_tiles, long long inst_bound = -1);
// Host-side hint: how many views' kernels the caller keeps in flight on different streams (lr_views_accumulate sets it for
// its own duration; a caller that pipelines the per-view entry points itself -- parallel.ViewStreams -- passes it through
// lr_tune_set("views_in_flight", n)).  Never a correctness input: it picks between kernel shapes with identical results.
struct ViewsInFlight { explicit ViewsInFlight(int n); ~ViewsInFlight(); int prev; };
int views_in_flight();
// shapes of the process's last blend launches (lr_last_launch_shapes): forward 0 quadrant / 1 quadrant with candidate
// pairs / 2 one wave per tile; backward BLEND_HALF / BLEND_QUAD / BLEND_TILE; -1 = none yet
void note_fwd_shape(int shape);
int last_fwd_shape();
int last_bwd_shape();
// Tile -> workgroup map of the blend kernels.  Workgroup b runs on XCD b % 8 (each XCD has its own L2):
//   TILE_MAP_BANDS : XCD x renders the contiguous tile band [x T/8, (x+1) T/8): Gaussians that straddle neighbouring tiles
//                    are re-read from the same L2.
//   TILE_MAP_PLAIN : tile t is workgroup t (XCD t % 8).  When splats crowd one image region the bands are unevenly loaded
//                    and the busiest XCD sets the kernel time; spreading every neighbourhood over all XCDs balances it.
// Measured (dense 1 M cloud, single view, blend forward / backward ms): 512^2 0.178/0.408 -> 0.164/0.334, 800^2
// 0.211/0.512 -> 0.149/0.378, 1280x720 0.177/0.406 -> 0.170/0.384; at 1080p and 1440p the two maps are within 3 % of
// each other either way (C3: bands 2 % ahead in the backward), so the large images keep the bands.
enum { TILE_MAP_BANDS = 0, TILE_MAP_PLAIN = 1 };
int blend_tile_map(int num_tiles);        // the rule (render_bwd.hip); LR_TILE_MAP=0/1 forces one (diagnostics)
__device__ __forceinline__ int blend_tile(int map, int num_tiles)
{
    int t = (int)blockIdx.x;
    if (map == TILE_MAP_BANDS) {
        const int per = (num_tiles + 7) >> 3;
        t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    }
    return t < num_tiles ? t : -1;
}
void launch_render_fwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const uint32_t* list_gid, const GaussRec* rec, const float* bg, float* final_T,
                       uint32_t* n_contrib, float* out_color, float* out_depth, uint8_t* quad_hits,
                       GeomHeader* hdr, uint2* seg_list, float4* ckpt, uint32_t* tile_seg0, float4* c_final,
                       long long inst_hint, hipStream_t s);
// seg_bound: upper bound of GeomHeader::n_seg known to the host (bin_seg_capacity of the instance bound of the call)
void launch_render_bwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const GaussRec* rec, const float* bg, const float* final_T,
                       const uint32_t* n_contrib, const float* dL_dpix, char* bin_base, const GeomHeader* hdr,
                       const uint32_t* tile_seg0, const float4* c_final, long long seg_bound, hipStream_t s);
// ---- Adam, element-wise (adam.hip and the step fused into the per-Gaussian backward, gauss_bwd.hip) -----------------------
// torch's single-tensor Adam (torch/optim/adam.py _single_tensor_adam; no weight decay, amsgrad or maximize) with the roundings
// of torch's own kernels: lerp = fma(w, b - a, a), addcmul = fma(value * t1, t2, self), addcdiv = fma(value, t1 / t2, self);
// mul_, sqrt, div, add are separate.  ONE definition: the fused step must give the bits of k_adam on the same gradient.
//   w1 = 1 - beta1, w2 = 1 - beta2, bc2_sqrt = sqrt(1 - beta2^t), step_size = lr / (1 - beta1^t)
__device__ __forceinline__ void adam_one(float gi, float& mi, float& vi, float& pi, float w1, float beta2, float w2,
                                         float bc2_sqrt, float eps, float step_size)
{
    mi = __builtin_fmaf(w1, gi - mi, mi);
    vi = vi * beta2;
    vi = __builtin_fmaf(w2 * gi, gi, vi);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = __builtin_fmaf(-step_size, mi / denom, pi);
}
void launch_gauss_bwd(const ViewParams& vp, const float* means3D, const float* scales, const float* rotations,
                      const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                      const uint32_t* vis_list, const uint8_t* clamped, const uint32_t* offsets,
                      const char* bin_base, const GeomHeader* hdr,
                      float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                      uint32_t accum_mask, float* acc16, hipStream_t s);
// acc16 [P][16]: per-step interleaved accumulator of the five small gradient rows (gauss_bwd.hip); added to the caller's
// tensors once per step
void launch_uninterleave_add(int P, const float* acc16, float* mean2D, float* opacity, float* mean3D, float* scale, float* rot,
                             hipStream_t s);
// row surgery of the parameter set (rows.hip)
size_t select_workspace_bytes(int P);
int launch_select_rows(int P, const uint8_t* mask, int n_tensors, const void* const* src, void* const* dst,
                       const unsigned* row_bytes, long long dst_row_offset, int* out_count, char* ws, hipStream_t s);
void launch_pack_ply(int P, int n_rest, const float* xyz, const float* f_dc, const float* f_rest, const float* opacity,
                     const float* scaling, const float* rotation, float* out, hipStream_t s);
void launch_densify_stats(int P, const int* radii, const float* dL_dmean2D, float* accum, float* denom, float* max_radii,
                          hipStream_t s);
// one-launch Adam step over several tensors (adam.hip)
int launch_adam(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                float* const* exp_avg_sq, const unsigned long long* numel, const double* lr, double beta1, double beta2,
                double eps, int step, hipStream_t s);
// ... where the gradient of a Gaussian's rows is only VALID if the view visited it (tiles_touched != 0: lr_backward_raw with
// LR_ACC_NO_ZERO_FILL left the other rows unwritten) and is taken as zero otherwise -- or for every row, when the view
// overflowed its binning buffer and the backward skipped it.  row_len[t]: floats per Gaussian of tensor t.
int launch_adam_masked(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const unsigned long long* numel, const unsigned int* row_len, const double* lr,
                       double beta1, double beta2, double eps, int step, const uint32_t* tiles_touched, const GeomHeader* hdr,
                       hipStream_t s);
// fused L1 + DSSIM loss (loss.hip)
size_t loss_workspace_bytes(int C, int H, int W);
// defer_final: leave {loss, l1, ssim} to the launch_loss_backward(..., final_out3) that follows on the same stream
// (workgroup 0 of the backward kernel sums the partials: one launch less per view in the fused step)
void launch_loss_forward(int C, int H, int W, const float* img, const float* gt, float lambda, float* out3, char* ws,
                         hipStream_t s, bool defer_final = false);
void launch_loss_backward(int C, int H, int W, const float* img, const float* gt, float lambda, const float* upstream,
                          const char* ws, float* grad, hipStream_t s, float* final_out3 = nullptr, const float* w_ssim = nullptr);
// bit positions of lr_backward's accumulate_mask (LR_ACC_* in lucid_raster.h)
enum { ACC_MEAN2D = 0, ACC_CONIC = 1, ACC_OPACITY = 2, ACC_COLOR = 3, ACC_MEAN3D = 4, ACC_COV3D = 5, ACC_SH = 6,
       ACC_SCALE = 7, ACC_ROT = 8 };
void launch_zero_outputs(float* const* ptrs, const unsigned long long* nfloats, int count, hipStream_t s);

void launch_dist2(int P, const float* points, float* out, char* workspace, hipStream_t s);
size_t dist2_workspace_bytes(int P);

}  // namespace lr
