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
    float b, depth, pad0, pad1;     // colour b, view-space depth
};
static_assert(sizeof(GaussRec) == 48, "GaussRec must be 48 bytes");

// Per-Gaussian gradient accumulator filled by the blend backward (one record, nine atomics).
struct __attribute__((aligned(16))) GradRec {
    float dmx, dmy, dca, dcb;       // dL/dmean2D.xy (NDC-scaled), dL/dconic a, b
    float dcc, dop, dr, dg;         // dL/dconic c, dL/dopacity, dL/dcolour r, g
    float db, pad0, pad1, pad2;
};
static_assert(sizeof(GradRec) == 48, "GradRec must be 48 bytes");

// Header at offset 0 of the geom buffer (device-resident view state).
struct GeomHeader {
    uint32_t num_rendered;          // total tile instances R of this view (device-side truth)
    uint32_t overflow;              // 1 if R exceeded the binning capacity (async mode)
    uint32_t prefilter_trap;        // 1 if a culled point was seen with prefiltered=true
    uint32_t capacity;              // binning capacity (instances)
    uint32_t P;
    uint32_t num_sorted;            // min(num_rendered, capacity): length of the instance list actually built
    uint32_t reserved[58];
};
static_assert(sizeof(GeomHeader) == 256, "GeomHeader must be 256 bytes");

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
inline size_t sort_hist_bytes(long long n_bound) {
    return align_up((size_t)SORT_MAX_BLOCKS * RADIX_SIZE * 4 + RADIX_SIZE * 4);
}

// ---- scan geometry ---------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// ---- buffer layouts ----------------------------------------------------------------------------
struct GeomLayout {
    size_t header, rec, clamped, tiles_touched, key_a, key_b, val_a, val_b, offsets, scan_sums, hist, grad, total;
};
inline GeomLayout geom_layout(int P) {
    GeomLayout L; size_t o = 0; size_t Pz = P > 0 ? (size_t)P : 1;
    L.header = o;        o += align_up(sizeof(GeomHeader));
    L.rec = o;           o += align_up(Pz * sizeof(GaussRec));
    L.clamped = o;       o += align_up(Pz);
    L.tiles_touched = o; o += align_up(Pz * 4);
    L.key_a = o;         o += align_up(Pz * 4);
    L.key_b = o;         o += align_up(Pz * 4);
    L.val_a = o;         o += align_up(Pz * 4);
    L.val_b = o;         o += align_up(Pz * 4);
    L.offsets = o;       o += align_up(Pz * 4);
    L.scan_sums = o;     o += align_up(((Pz + SCAN_TILE - 1) / SCAN_TILE + 1) * 4);
    L.hist = o;          o += sort_hist_bytes((long long)Pz);
    L.grad = o;          o += align_up(Pz * sizeof(GradRec));
    L.total = o;
    return L;
}
struct ImgLayout { size_t final_T, n_contrib, ranges, total; };
inline ImgLayout img_layout(int W, int H) {
    ImgLayout L; size_t o = 0; size_t N = (size_t)W * H; if (N == 0) N = 1;
    size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y); if (T == 0) T = 1;
    L.final_T = o;   o += align_up(N * 4);
    L.n_contrib = o; o += align_up(N * 4);
    L.ranges = o;    o += align_up(T * 8);
    L.total = o;
    return L;
}
struct BinLayout { size_t key_a, key_b, val_a, val_b, hist, total; };
inline BinLayout bin_layout(long long R) {
    BinLayout L; size_t o = 0; size_t Rz = R > 0 ? (size_t)R : 1;
    L.key_a = o; o += align_up(Rz * 4);
    L.key_b = o; o += align_up(Rz * 4);
    L.val_a = o; o += align_up(Rz * 4);
    L.val_b = o; o += align_up(Rz * 4);
    L.hist = o;  o += sort_hist_bytes((long long)Rz);
    L.total = o;
    return L;
}

// ---- launchers (one per translation unit) ------------------------------------------------------
struct ViewParams {
    const float* view;      // device, 16 floats, flat index m[4*col+row] (auxiliary.h:58-77)
    const float* proj;      // device, 16 floats
    const float* campos;    // device, 3 floats
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy, P, D, M;
};

void launch_preprocess(const ViewParams& vp, const float* means3D, const float* scales, const float* rotations,
                       const float* opacities, const float* shs, const float* cov3D_precomp,
                       const float* colors_precomp, bool prefiltered, int* radii, GaussRec* rec,
                       uint8_t* clamped, uint32_t* tiles_touched, uint32_t* depth_key, GeomHeader* hdr,
                       uint32_t binning_capacity, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present,
                         hipStream_t s);

// Stable LSD radix sort of (key,val) u32 pairs on bits [0,end_bit).  n_dev: device count (may be
// smaller than n_bound; blocks beyond it idle).  vals_in == nullptr means val = index (iota).
// Returns which buffers hold the result via *keys_out / *vals_out.
void radix_sort_pairs(uint32_t* key_a, uint32_t* key_b, uint32_t* val_a, uint32_t* val_b, bool vals_iota,
                      const uint32_t* n_dev, long long n_bound, int end_bit, uint32_t* hist,
                      uint32_t** keys_out, uint32_t** vals_out, hipStream_t s);

// offsets[k] = exclusive prefix of tiles_touched[order[k]], k in depth order; total -> hdr->num_rendered
// (and the overflow flag against hdr->capacity).
void launch_scan_tiles(int P, const uint32_t* order, const uint32_t* tiles_touched, uint32_t* offsets,
                       uint32_t* block_sums, GeomHeader* hdr, hipStream_t s);
void launch_emit(int P, int gx, int gy, const uint32_t* order, const uint32_t* offsets,
                 const uint32_t* tiles_touched, const GaussRec* rec, const int* radii, GeomHeader* hdr,
                 uint32_t* inst_keys, uint32_t* inst_vals, hipStream_t s);
void launch_ranges(const uint32_t* sorted_keys, const GeomHeader* hdr, long long n_bound, int num_tiles,
                   uint2* ranges, hipStream_t s);

void launch_render_fwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const GaussRec* rec, const float* bg, float* final_T,
                       uint32_t* n_contrib, float* out_color, float* out_depth, hipStream_t s);
void launch_render_bwd(int W, int H, int gx, int gy, const uint2* ranges, const uint32_t* point_list,
                       const GaussRec* rec, const float* bg, const float* final_T,
                       const uint32_t* n_contrib, const float* dL_dpix, GradRec* grad, hipStream_t s);
void launch_gauss_bwd(const ViewParams& vp, const float* means3D, const float* scales, const float* rotations,
                      const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                      const int* radii, const uint8_t* clamped, const GradRec* grad,
                      float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                      float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                      hipStream_t s);

void launch_dist2(int P, const float* points, float* out, char* workspace, hipStream_t s);
size_t dist2_workspace_bytes(int P);

}  // namespace lr
