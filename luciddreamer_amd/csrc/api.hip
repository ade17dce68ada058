// api.hip -- extern "C" entry points declared in include/lucid_raster.h (host orchestration).
//
// lr_forward  follows CudaRasterizer::Rasterizer::forward  (RAST/cuda_rasterizer/rasterizer_impl.cu:198-339)
// lr_backward follows CudaRasterizer::Rasterizer::backward (RAST/cuda_rasterizer/rasterizer_impl.cu:343-444)
// lr_mark_visible follows Rasterizer::markVisible (RAST/cuda_rasterizer/rasterizer_impl.cu:141-153)
#include "common.h"
#include "../../include/lucid_raster.h"

#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define LR_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail(LR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));          \
    } while (0)

// debug => synchronise and surface asynchronous errors (CHECK_CUDA, auxiliary.h:166-173)
#define LR_DEBUG_SYNC(debug, stream, what)                                                        \
    do {                                                                                          \
        if (debug) {                                                                              \
            hipError_t e__ = hipStreamSynchronize(stream);                                        \
            if (e__ == hipSuccess) e__ = hipGetLastError();                                       \
            if (e__ != hipSuccess)                                                                \
                return fail(LR_ERR_HIP, std::string("[HIP ERROR] after ") + what + ": " + hipGetErrorString(e__)); \
        }                                                                                         \
    } while (0)

// ---- optional per-stage HIP-event timing (used by bench.py for the roofline figures) ----------
enum Stage { ST_PREPROCESS = 0, ST_COMPACT, ST_BIN_COUNT, ST_BIN_SCAN, ST_BIN_SCATTER, ST_TILE_SORT, ST_RENDER_FWD,
             ST_GRAD_ZERO, ST_RENDER_BWD, ST_OUT_ZERO, ST_GAUSS_BWD, ST_COUNT };
const char* const kStageNames[ST_COUNT] = { "preprocess", "compact", "bin_count", "bin_scan", "bin_scatter", "tile_sort",
                                            "render_fwd", "grad_zero", "render_bwd", "out_zero", "gauss_bwd" };
struct ProfRec { int stage; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::mutex g_prof_mu;                    // the profiler is a single-device diagnostic; the lock only keeps it memory-safe
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;

hipEvent_t prof_event()
{
    {
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    hipStream_t s; ProfRec r; bool on;
    ProfScope(int stage, hipStream_t s_) : s(s_), on(g_prof_on)
    {
        if (on) { r.stage = stage; r.e0 = prof_event(); r.e1 = prof_event(); (void)hipEventRecord(r.e0, s); }
    }
    ~ProfScope()
    {
        if (on) { (void)hipEventRecord(r.e1, s); std::lock_guard<std::mutex> lock(g_prof_mu); g_prof_recs.push_back(r); }
    }
};

// stage boundaries inside launch_tile_binning: boundary i closes stage i-1 and opens stage i of
// {bin_count, bin_scan, bin_scatter, tile_sort}
struct BinProf : lr::TileBinTimes {
    ProfRec cur; bool open = false;
    void mark(int boundary, hipStream_t s) override
    {
        if (!g_prof_on) return;
        if (open) { (void)hipEventRecord(cur.e1, s); std::lock_guard<std::mutex> lock(g_prof_mu); g_prof_recs.push_back(cur); open = false; }
        if (boundary < 4) {
            cur.stage = ST_BIN_COUNT + boundary; cur.e0 = prof_event(); cur.e1 = prof_event();
            (void)hipEventRecord(cur.e0, s); open = true;
        }
    }
};

int g_tune[lr::TUNE_COUNT] = { -1, -1, -1, -1, -1, -1 };
const char* const kTuneNames[lr::TUNE_COUNT] = { "bwd_red", "blend_quad", "tile_map", "preprocess", "gauss_bwd", "tsort" };

int bits_for(uint32_t max_value)
{
    int b = 0;
    while (b < 32 && (max_value >> b) != 0) b++;
    return b < 1 ? 1 : b;
}

}  // namespace

namespace lr {
int tune_get(int key) { return (key >= 0 && key < TUNE_COUNT) ? g_tune[key] : -1; }
}  // namespace lr

extern "C" {

const char* lr_last_error(void) { return g_last_error.c_str(); }
const char* lr_version(void) { return "luciddreamer_amd-raster 0.1 (gfx950)"; }

size_t lr_geom_bytes(int P) { return lr::geom_layout(P).total; }
size_t lr_img_bytes(int width, int height) { return lr::img_layout(width, height).total; }
size_t lr_binning_bytes(long long R) { return lr::bin_layout(R).total; }

// raw-parameter mode (lr_forward_raw / lr_backward_raw): see ViewParams in common.h
struct RawArgs { const float* sh_rest; const float* opacity_raw; float* dL_dsh_rest; };

static int forward_core(lr_alloc_fn geom_alloc, void* geom_user, lr_alloc_fn binning_alloc, void* binning_user,
               lr_alloc_fn img_alloc, void* img_user, int P, int D, int M, const float* background,
               int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
               const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
               const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
               float* out_depth, int* radii, int debug, long long binning_capacity, void* stream_, const RawArgs* raw)
{
    using namespace lr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0 || width <= 0 || height <= 0) return fail(LR_ERR_INVALID_ARG, "P, width, height must be positive");
    if (!geom_alloc || !binning_alloc || !img_alloc) return fail(LR_ERR_INVALID_ARG, "allocator callbacks are required");
    if (!background || !viewmatrix || !projmatrix || !cam_pos || !out_color || !out_depth)
        return fail(LR_ERR_INVALID_ARG, "background/viewmatrix/projmatrix/cam_pos/out_color/out_depth are required");
    if (P > 0 && (!means3D || !opacities || !radii)) return fail(LR_ERR_INVALID_ARG, "means3D/opacities/radii are required");
    if (P > 0 && shs == nullptr && colors_precomp == nullptr)
        return fail(LR_ERR_INVALID_ARG, "For non-RGB, provide precomputed Gaussian colors!");      // rasterizer_impl.cu:243-246
    if (P > 0 && cov3D_precomp == nullptr && (scales == nullptr || rotations == nullptr))
        return fail(LR_ERR_INVALID_ARG, "provide scales+rotations or cov3D_precomp");
    if (shs != nullptr && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
        return fail(LR_ERR_INVALID_ARG, "SH degree must be 0..3 and (D+1)^2 <= M");
    if (binning_capacity < 0 || binning_capacity > 0xFFFFFFF0ll) return fail(LR_ERR_INVALID_ARG, "bad binning_capacity");

    const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
    const int num_tiles = gx * gy;

    const GeomLayout GL = geom_layout(P);
    const ImgLayout IL = img_layout(width, height);
    char* geom = geom_alloc(GL.total, geom_user);
    char* img = img_alloc(IL.total, img_user);
    if (!geom || !img) return fail(LR_ERR_ALLOC, "geom/img allocator returned NULL");

    GeomHeader* hdr = reinterpret_cast<GeomHeader*>(geom + GL.header);
    GaussRec* rec = reinterpret_cast<GaussRec*>(geom + GL.rec);
    uint8_t* clamped = reinterpret_cast<uint8_t*>(geom + GL.clamped);
    uint32_t* tiles_touched = reinterpret_cast<uint32_t*>(geom + GL.tiles_touched);
    uint32_t* vis_list = reinterpret_cast<uint32_t*>(geom + GL.vis_list);
    uint32_t* offsets = reinterpret_cast<uint32_t*>(geom + GL.offsets);
    uint32_t* goff = reinterpret_cast<uint32_t*>(geom + GL.goff);
    uint4* scan_sums = reinterpret_cast<uint4*>(geom + GL.scan_sums);
    float* final_T = reinterpret_cast<float*>(img + IL.final_T);
    uint32_t* n_contrib = reinterpret_cast<uint32_t*>(img + IL.n_contrib);
    uint2* ranges = reinterpret_cast<uint2*>(img + IL.ranges);
    uint32_t* part_hist = reinterpret_cast<uint32_t*>(img + IL.part_hist);
    uint32_t* bin_total = reinterpret_cast<uint32_t*>(img + IL.bin_total);
    uint32_t* bin_start = reinterpret_cast<uint32_t*>(img + IL.bin_start);
    uint32_t* big_queue = reinterpret_cast<uint32_t*>(img + IL.big_queue);

    // header and the compaction's chunk sums start zeroed (one launch); the preprocess kernel fills in {capacity, P}
    launch_forward_begin(hdr, scan_sums, P, s);

    ViewParams vp;
    vp.view = viewmatrix; vp.proj = projmatrix; vp.campos = cam_pos;
    vp.tan_fovx = tan_fovx; vp.tan_fovy = tan_fovy;
    vp.focal_y = height / (2.0f * tan_fovy);                   // rasterizer_impl.cu:223-224
    vp.focal_x = width / (2.0f * tan_fovx);
    vp.scale_modifier = scale_modifier;
    vp.W = width; vp.H = height; vp.gx = gx; vp.gy = gy; vp.P = P; vp.D = D; vp.M = M;
    vp.raw = raw != nullptr; vp.sh_rest = raw ? raw->sh_rest : nullptr; vp.opacity_raw = nullptr; vp.dL_dsh_rest = nullptr;

    long long R_bound = 0;
    int num_rendered = 0;
    uint32_t* point_list = nullptr;
    uint32_t* inst_gid = nullptr;
    bool ranges_written = false;

    if (P > 0) {
        // K1: cull / project / conic / colour -> GaussRec, radii, tile counts
        { ProfScope ps(ST_PREPROCESS, s);
        launch_preprocess(vp, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                          prefiltered != 0, radii, rec, clamped, tiles_touched, nullptr, hdr,
                          (uint32_t)binning_capacity, reinterpret_cast<uint32_t*>(scan_sums), s); }
        LR_DEBUG_SYNC(debug, s, "preprocess");

        // one scan in index order: the emitting Gaussians, their first instance slots, every count of the header
        { ProfScope ps(ST_COMPACT, s);
        launch_compact(P, tiles_touched, scan_sums, vis_list, offsets, goff, hdr, s); }
        LR_DEBUG_SYNC(debug, s, "compact");

        if (binning_capacity == 0) {
            // exact mode: one small read-back, like rasterizer_impl.cu:281-282: the reference's num_rendered
            // (returned to the caller) and the number of instances that survive exact tile culling
            // (sizes the binning buffer)
            uint32_t meta[8];
            LR_HIP_CHECK(hipMemcpyAsync(meta, hdr, sizeof(meta), hipMemcpyDeviceToHost, s));
            LR_HIP_CHECK(hipStreamSynchronize(s));
            R_bound = meta[6];
            // the return value is the reference's `int num_rendered` (rasterizer_impl.cu:281); negative values are error
            // codes here, so a count that does not fit is an explicit error instead of a wrapped-around one
            if (meta[0] > 0x7FFFFFFFu)
                return fail(LR_ERR_INVALID_ARG, "num_rendered exceeds INT_MAX: use async mode (binning_capacity > 0) and read the count with lr_check");
            num_rendered = (int)meta[0];
        } else {
            R_bound = binning_capacity;
            num_rendered = LR_NUM_RENDERED_ON_DEVICE;
        }

        const BinLayout BL = bin_layout(R_bound);
        char* bin = binning_alloc(BL.total, binning_user);
        if (!bin) return fail(LR_ERR_ALLOC, "binning allocator returned NULL");
        point_list = reinterpret_cast<uint32_t*>(bin + BL.point_list);
        inst_gid = reinterpret_cast<uint32_t*>(bin + BL.inst_gid);
        if (R_bound > 0) {
            // count -> scan -> scatter -> per-tile sort (tilebin.hip): the reference's (tile | depth) order
            // (rasterizer_impl.cu:301-309) and the tile ranges (:116-138)
            BinProf bp;
            const int rc = launch_tile_binning(P, gx, gy, bits_for((uint32_t)(R_bound - 1)), vis_list, offsets, tiles_touched, rec,
                                               radii, hdr, part_hist, bin_total, bin_start, big_queue, inst_gid,
                                               reinterpret_cast<unsigned long long*>(bin + BL.words), point_list, ranges,
                                               &bp, s);
            if (rc == -2) return fail(LR_ERR_INVALID_ARG, "image / instance count too large: tile and slot bits exceed the 64-bit sort word");
            if (rc != 0) return fail(LR_ERR_HIP, "could not reserve LDS for the binning kernels");
            ranges_written = true;
            LR_DEBUG_SYNC(debug, s, "tile binning");
        }
    } else {
        (void)binning_alloc(bin_layout(0).total, binning_user);
    }
    if (!ranges_written) LR_HIP_CHECK(hipMemsetAsync(ranges, 0, (size_t)num_tiles * sizeof(uint2), s));

    // K6: blend
    { ProfScope ps(ST_RENDER_FWD, s);
    launch_render_fwd(width, height, gx, gy, ranges, point_list, inst_gid, rec, background, final_T, n_contrib, out_color,
                      out_depth, s); }
    LR_DEBUG_SYNC(debug, s, "render");
    LR_HIP_CHECK(hipGetLastError());

    if (binning_capacity == 0 && prefiltered && P > 0) {
        uint32_t trap = 0;
        LR_HIP_CHECK(hipMemcpyAsync(&trap, &hdr->prefilter_trap, 4, hipMemcpyDeviceToHost, s));
        LR_HIP_CHECK(hipStreamSynchronize(s));
        if (trap) return fail(LR_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
    }
    return num_rendered;
}

int lr_forward(lr_alloc_fn geom_alloc, void* geom_user, lr_alloc_fn binning_alloc, void* binning_user,
               lr_alloc_fn img_alloc, void* img_user, int P, int D, int M, const float* background,
               int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
               const float* opacities, const float* scales, float scale_modifier, const float* rotations,
               const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
               const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
               float* out_depth, int* radii, int debug, long long binning_capacity, void* stream_)
{
    return forward_core(geom_alloc, geom_user, binning_alloc, binning_user, img_alloc, img_user, P, D, M, background, width,
                        height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth, radii, debug,
                        binning_capacity, stream_, nullptr);
}

int lr_forward_raw(lr_alloc_fn geom_alloc, void* geom_user, lr_alloc_fn binning_alloc, void* binning_user,
                   lr_alloc_fn img_alloc, void* img_user, int P, int D, int M, const float* background,
                   int width, int height, const float* xyz, const float* features_dc, const float* features_rest,
                   const float* opacity_raw, const float* scaling_raw, float scale_modifier, const float* rotation_raw,
                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                   float tan_fovy, float* out_color, float* out_depth, int* radii, int debug,
                   long long binning_capacity, void* stream_)
{
    if (P > 0 && (!features_dc || !opacity_raw || !scaling_raw || !rotation_raw || (M > 1 && !features_rest)))
        return fail(LR_ERR_INVALID_ARG, "raw mode needs features_dc, features_rest (M > 1), opacity, scaling and rotation");
    if (M < 1) return fail(LR_ERR_INVALID_ARG, "raw mode: M = 1 + number of features_rest coefficients must be >= 1");
    const RawArgs raw = { features_rest, opacity_raw, nullptr };
    return forward_core(geom_alloc, geom_user, binning_alloc, binning_user, img_alloc, img_user, P, D, M, background, width,
                        height, xyz, features_dc, nullptr, opacity_raw, scaling_raw, scale_modifier, rotation_raw, nullptr,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, 0, out_color, out_depth, radii, debug,
                        binning_capacity, stream_, &raw);
}

static int backward_core(int P, int D, int M, int R, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                const float* dL_dpix, const float* dL_depths, float* dL_dmean2D, float* dL_dconic,
                float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                float* dL_dscale, float* dL_drot, int debug, long long binning_capacity,
                unsigned int accumulate_mask, void* stream_, hipEvent_t wait_before_accumulate,
                const RawArgs* raw = nullptr)
{
    using namespace lr;
    (void)dL_depths;   // ignored, as in the reference (backward.cu:457-464, 539-554 commented out)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (P <= 0) return 0;
    if (!geom_buffer || !binning_buffer || !image_buffer) return fail(LR_ERR_INVALID_ARG, "scratch buffers are required");
    if (!dL_dpix || !dL_dmean2D || !dL_dopacity || !dL_dmean3D)
        return fail(LR_ERR_INVALID_ARG, "gradient outputs dL_dmean2D/dL_dopacity/dL_dmean3D are required");
    if (colors_precomp != nullptr && !dL_dcolor) return fail(LR_ERR_INVALID_ARG, "dL_dcolor is required with colors_precomp");
    if (cov3D_precomp != nullptr && !dL_dcov3D) return fail(LR_ERR_INVALID_ARG, "dL_dcov3D is required with cov3D_precomp");
    if (scales != nullptr && (!dL_dscale || !dL_drot)) return fail(LR_ERR_INVALID_ARG, "dL_dscale/dL_drot are required with scales");
    if (shs != nullptr && dL_dsh == nullptr) return fail(LR_ERR_INVALID_ARG, "dL_dsh is required when shs is given");
    {   // gradient tensors are read / written with 16-byte vector accesses (write mode and accumulate mode alike)
        const void* g16[10] = { dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                                raw ? raw->dL_dsh_rest : nullptr };
        for (const void* p : g16)
            if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0)
                return fail(LR_ERR_INVALID_ARG, "gradient outputs must be 16-byte aligned");
    }

    const int gx = (width + TILE_X - 1) / TILE_X, gy = (height + TILE_Y - 1) / TILE_Y;
    const GeomLayout GL = geom_layout(P);
    const ImgLayout IL = img_layout(width, height);
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom_buffer + GL.rec);
    const uint8_t* clamped = reinterpret_cast<const uint8_t*>(geom_buffer + GL.clamped);
    const uint32_t* tiles_touched = reinterpret_cast<const uint32_t*>(geom_buffer + GL.tiles_touched);
    const uint32_t* goff = reinterpret_cast<const uint32_t*>(geom_buffer + GL.goff);
    const uint32_t* vis_list = reinterpret_cast<const uint32_t*>(geom_buffer + GL.vis_list);
    const GeomHeader* hdr = reinterpret_cast<const GeomHeader*>(geom_buffer + GL.header);
    const float* final_T = reinterpret_cast<const float*>(image_buffer + IL.final_T);
    const uint32_t* n_contrib = reinterpret_cast<const uint32_t*>(image_buffer + IL.n_contrib);
    const uint2* ranges = reinterpret_cast<const uint2*>(image_buffer + IL.ranges);

    // The tile-sorted instance list always ends at offset 0 of the binning buffer (val_a), whatever
    // R / capacity the forward used, so nothing has to be read back here (R and binning_capacity are
    // accepted for symmetry with lr_forward).
    (void)R; (void)binning_capacity;
    const uint32_t* point_list = reinterpret_cast<const uint32_t*>(binning_buffer);

    ViewParams vp;
    vp.view = viewmatrix; vp.proj = projmatrix; vp.campos = campos;
    vp.tan_fovx = tan_fovx; vp.tan_fovy = tan_fovy;
    vp.focal_y = height / (2.0f * tan_fovy);
    vp.focal_x = width / (2.0f * tan_fovx);
    vp.scale_modifier = scale_modifier;
    vp.W = width; vp.H = height; vp.gx = gx; vp.gy = gy; vp.P = P; vp.D = D; vp.M = M;
    vp.raw = raw != nullptr;
    vp.sh_rest = raw ? raw->sh_rest : nullptr;
    vp.opacity_raw = raw ? raw->opacity_raw : nullptr;
    vp.dL_dsh_rest = raw ? raw->dL_dsh_rest : nullptr;

    { ProfScope ps(ST_RENDER_BWD, s);
    launch_render_bwd(width, height, gx, gy, ranges, point_list, rec, background, final_T, n_contrib, dL_dpix,
                      binning_buffer, hdr, s); }
    LR_DEBUG_SYNC(debug, s, "render backward");
    // another stream may still be accumulating into the same gradient tensors (lr_views_accumulate)
    if (wait_before_accumulate != nullptr) LR_HIP_CHECK(hipStreamWaitEvent(s, wait_before_accumulate, 0));
    {   // tensors in write mode are zero-filled here (one launch); accumulate-mode tensors are left alone
        ProfScope ps(ST_OUT_ZERO, s);
        const unsigned long long Pn = (unsigned long long)P;
        float* ptrs[9] = { dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot };
        // raw mode: dL_dsh is the features_dc gradient [P,3]; the features_rest gradient is a second tensor
        const unsigned long long nsh = raw ? 3 * Pn : (unsigned long long)M * 3 * Pn;
        const unsigned long long nf[9] = { 3 * Pn, 4 * Pn, Pn, 3 * Pn, 3 * Pn, 6 * Pn, nsh, 3 * Pn, 4 * Pn };
        float* zp[10]; unsigned long long zn[10]; int zc = 0;
        for (int k = 0; k < 9; k++)
            if (ptrs[k] != nullptr && !((accumulate_mask >> k) & 1u)) { zp[zc] = ptrs[k]; zn[zc] = nf[k]; zc++; }
        if (raw && raw->dL_dsh_rest != nullptr && M > 1 && !((accumulate_mask >> ACC_SH) & 1u) && zc < 9) {
            zp[zc] = raw->dL_dsh_rest; zn[zc] = (unsigned long long)(M - 1) * 3 * Pn; zc++;
        }
        launch_zero_outputs(zp, zn, zc, s);
    }
    { ProfScope ps(ST_GAUSS_BWD, s);
    launch_gauss_bwd(vp, means3D, scales, rotations, shs, cov3D_precomp, colors_precomp, vis_list, clamped, tiles_touched, goff,
                     binning_buffer, hdr,
                     dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                     dL_drot, accumulate_mask, s); }
    LR_DEBUG_SYNC(debug, s, "preprocess backward");
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

int lr_backward(int P, int D, int M, int R, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                const float* dL_dpix, const float* dL_depths, float* dL_dmean2D, float* dL_dconic,
                float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                float* dL_dscale, float* dL_drot, int debug, long long binning_capacity,
                unsigned int accumulate_mask, void* stream_)
{
    return backward_core(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier,
                         rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii,
                         geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_depths, dL_dmean2D, dL_dconic,
                         dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug,
                         binning_capacity, accumulate_mask, stream_, nullptr);
}

int lr_backward_raw(int P, int D, int M, int R, const float* background, int width, int height,
                    const float* xyz, const float* features_dc, const float* features_rest, const float* opacity_raw,
                    const float* scaling_raw, float scale_modifier, const float* rotation_raw,
                    const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                    float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                    const float* dL_dpix, float* dL_dmean2D, float* dL_dopacity_raw, float* dL_dxyz,
                    float* dL_dfeatures_dc, float* dL_dfeatures_rest, float* dL_dscaling_raw, float* dL_drotation_raw,
                    int debug, long long binning_capacity, unsigned int accumulate_mask, void* stream_)
{
    if (P > 0 && (!features_dc || !opacity_raw || !scaling_raw || !rotation_raw || (M > 1 && !features_rest)))
        return fail(LR_ERR_INVALID_ARG, "raw mode needs features_dc, features_rest (M > 1), opacity, scaling and rotation");
    if (P > 0 && (!dL_dfeatures_dc || (M > 1 && !dL_dfeatures_rest)))
        return fail(LR_ERR_INVALID_ARG, "raw mode needs dL_dfeatures_dc and dL_dfeatures_rest");
    const RawArgs raw = { features_rest, opacity_raw, dL_dfeatures_rest };
    return backward_core(P, D, M, R, background, width, height, xyz, features_dc, nullptr, scaling_raw, scale_modifier,
                         rotation_raw, nullptr, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer,
                         binning_buffer, image_buffer, dL_dpix, nullptr, dL_dmean2D, nullptr, dL_dopacity_raw, nullptr,
                         dL_dxyz, nullptr, dL_dfeatures_dc, dL_dscaling_raw, dL_drotation_raw, debug, binning_capacity,
                         accumulate_mask, stream_, nullptr, &raw);
}

// ---------------------------------------------------------------------------------------------------
// Multi-view step: forward + backward of n_views views of one parameter set, gradients ACCUMULATED into the
// caller's buffers, entirely enqueued from C (one call per optimisation step instead of 2 calls per view).
// Views alternate over internal HIP streams so that the launch-latency-bound forward of view i+1 runs under
// the two long backward kernels of view i; the accumulating kernel of view i+1 waits (event) for view i's.
// The caller's stream is forked at entry and joined at exit with events -- no host synchronisation.
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxViewStreams = 4;
struct ViewSlot { size_t geom, img, bin, color, depth, radii, loss_ws, grad_img, total; };
ViewSlot view_slot_layout(int P, int W, int H, long long capacity, bool with_loss = false)
{
    ViewSlot L; size_t o = 0;
    const size_t N = (size_t)W * H;
    L.geom = o;  o += lr::geom_layout(P).total;
    L.img = o;   o += lr::img_layout(W, H).total;
    L.bin = o;   o += lr::bin_layout(capacity).total;
    L.color = o; o += lr::align_up(3 * N * 4);
    L.depth = o; o += lr::align_up(N * 4);
    L.radii = o; o += lr::align_up((size_t)(P > 0 ? P : 1) * 4);
    L.loss_ws = o;  if (with_loss) o += lr::align_up(lr::loss_workspace_bytes(3, H, W));
    L.grad_img = o; if (with_loss) o += lr::align_up(3 * N * 4);
    L.total = o;
    return L;
}
struct SliceCookie { char* ptr; size_t bytes; };
char* slice_alloc(size_t bytes, void* user)
{
    SliceCookie* c = static_cast<SliceCookie*>(user);
    return bytes <= c->bytes ? c->ptr : nullptr;
}
// internal streams / events of the multi-view entry points: one set per device, created on first use
struct ViewStreamSet {
    hipStream_t streams[kMaxViewStreams] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t events[2 * kMaxViewStreams + 2] = {};
};
std::mutex g_view_mu;
std::map<int, ViewStreamSet> g_view_sets;
}  // namespace

size_t lr_views_workspace_bytes(int P, int width, int height, long long binning_capacity, int n_streams)
{
    if (n_streams < 1) n_streams = 1;
    if (n_streams > kMaxViewStreams) n_streams = kMaxViewStreams;
    return view_slot_layout(P, width, height, binning_capacity).total * (size_t)n_streams;
}

static int views_core(int n_views, const float* const* viewmatrices, const float* const* projmatrices,
                        const float* const* cam_positions, const float* tan_fovx, const float* tan_fovy,
                        int P, int D, int M, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* const* dL_dpix, float* const* out_color, int* const* out_radii,
                        float* acc_mean2D, float* acc_opacity, float* acc_color, float* acc_mean3D, float* acc_cov3D,
                        float* acc_sh, float* acc_scale, float* acc_rot,
                        char* workspace, size_t workspace_bytes, long long binning_capacity, int n_streams,
                        void* stream_, const float* const* targets, float lambda_dssim, float* out_losses)
{
    using namespace lr;
    const bool with_loss = targets != nullptr;
    hipStream_t caller = reinterpret_cast<hipStream_t>(stream_);
    if (n_views <= 0 || P <= 0) return 0;
    if (binning_capacity <= 0) return fail(LR_ERR_INVALID_ARG, "lr_views_accumulate runs in async mode: binning_capacity > 0 is required");
    if (!viewmatrices || !projmatrices || !cam_positions || !tan_fovx || !tan_fovy || (!dL_dpix && !with_loss) || !workspace)
        return fail(LR_ERR_INVALID_ARG, "per-view arrays and workspace are required");
    if (with_loss && !out_losses) return fail(LR_ERR_INVALID_ARG, "out_losses is required with targets");
    if (!acc_mean2D || !acc_opacity || !acc_mean3D) return fail(LR_ERR_INVALID_ARG, "acc_mean2D/acc_opacity/acc_mean3D are required");
    if (n_streams < 1) n_streams = 1;
    if (n_streams > kMaxViewStreams) n_streams = kMaxViewStreams;
    const int n_slots = n_streams;                 // slots the caller's workspace was sized for (lr_views_check reads all)
    if (n_streams > n_views) n_streams = n_views;
    const ViewSlot SL = view_slot_layout(P, width, height, binning_capacity, with_loss);
    if (workspace_bytes < SL.total * (size_t)n_streams) return fail(LR_ERR_INVALID_ARG, "workspace too small (lr_views_workspace_bytes)");

    int device = 0;
    LR_HIP_CHECK(hipGetDevice(&device));
    ViewStreamSet* vs;
    {
        std::lock_guard<std::mutex> lock(g_view_mu);
        vs = &g_view_sets[device];                              // std::map: the address stays valid
        for (int i = 0; i < n_streams; i++)
            if (!vs->streams[i]) LR_HIP_CHECK(hipStreamCreateWithFlags(&vs->streams[i], hipStreamNonBlocking));
        for (int i = 0; i < 2 * kMaxViewStreams + 2; i++)
            if (!vs->events[i]) LR_HIP_CHECK(hipEventCreateWithFlags(&vs->events[i], hipEventDisableTiming));
    }
    hipStream_t* const g_view_streams = vs->streams;
    hipEvent_t ev_fork = vs->events[0];
    hipEvent_t* ev_bwd = &vs->events[1];                        // ring of n_streams + 1 "accumulation done" events
    hipEvent_t* ev_join = &vs->events[2 + kMaxViewStreams];

    // sticky per-slot overflow words (see GeomHeader::sticky_overflow) start at zero -- in EVERY slot of the workspace,
    // also the ones a step with fewer views than streams leaves unused (lr_views_check reads them all); fork the streams
    for (int i = 0; i < n_slots && (size_t)(i + 1) * SL.total <= workspace_bytes; i++) {
        GeomHeader* hdr = reinterpret_cast<GeomHeader*>(workspace + (size_t)i * SL.total + SL.geom);
        LR_HIP_CHECK(hipMemsetAsync(&hdr->sticky_overflow, 0, 4, caller));
    }
    LR_HIP_CHECK(hipEventRecord(ev_fork, caller));
    for (int i = 0; i < n_streams; i++) LR_HIP_CHECK(hipStreamWaitEvent(g_view_streams[i], ev_fork, 0));

    unsigned int mask = LR_ACC_MEAN2D | LR_ACC_OPACITY | LR_ACC_MEAN3D;
    if (acc_color) mask |= LR_ACC_COLOR;
    if (acc_cov3D) mask |= LR_ACC_COV3D;
    if (acc_sh) mask |= LR_ACC_SH;
    if (acc_scale) mask |= LR_ACC_SCALE;
    if (acc_rot) mask |= LR_ACC_ROT;

    hipEvent_t prev = nullptr;
    for (int v = 0; v < n_views; v++) {
        const int si = v % n_streams;
        hipStream_t s = g_view_streams[si];
        char* slot = workspace + (size_t)si * SL.total;
        SliceCookie cg = { slot + SL.geom, SL.img - SL.geom }, ci = { slot + SL.img, SL.bin - SL.img },
                    cb = { slot + SL.bin, SL.color - SL.bin };
        float* color = (out_color && out_color[v]) ? out_color[v] : reinterpret_cast<float*>(slot + SL.color);
        float* depth = reinterpret_cast<float*>(slot + SL.depth);
        int* radii = (out_radii && out_radii[v]) ? out_radii[v] : reinterpret_cast<int*>(slot + SL.radii);
        int rc = lr_forward(slice_alloc, &cg, slice_alloc, &cb, slice_alloc, &ci, P, D, M, background, width, height,
                            means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                            viewmatrices[v], projmatrices[v], cam_positions[v], tan_fovx[v], tan_fovy[v], 0, color, depth,
                            radii, 0, binning_capacity, s);
        if (rc < 0 && rc != LR_NUM_RENDERED_ON_DEVICE) return rc;
        const float* view_grad = with_loss ? nullptr : dL_dpix[v];
        if (with_loss) {
            // the photometric loss of this view against its target and dL/dcolor, on the same stream
            // (R/luciddreamer.py:301-304); out_losses[3 v .. 3 v + 2] = {loss, l1, ssim}
            float* gimg = reinterpret_cast<float*>(slot + SL.grad_img);
            launch_loss_forward(3, height, width, color, targets[v], lambda_dssim, out_losses + 3 * (size_t)v, slot + SL.loss_ws, s, true);
            launch_loss_backward(3, height, width, color, targets[v], lambda_dssim, nullptr, slot + SL.loss_ws, gimg, s,
                                 out_losses + 3 * (size_t)v);
            view_grad = gimg;
        }
        rc = backward_core(P, D, M, LR_NUM_RENDERED_ON_DEVICE, background, width, height, means3D, shs, colors_precomp,
                           scales, scale_modifier, rotations, cov3D_precomp, viewmatrices[v], projmatrices[v],
                           cam_positions[v], tan_fovx[v], tan_fovy[v], radii, slot + SL.geom, slot + SL.bin, slot + SL.img,
                           view_grad, nullptr, acc_mean2D, nullptr, acc_opacity, acc_color, acc_mean3D, acc_cov3D, acc_sh,
                           acc_scale, acc_rot, 0, binning_capacity, mask, s, prev);
        if (rc < 0) return rc;
        hipEvent_t done = ev_bwd[v % (n_streams + 1)];
        LR_HIP_CHECK(hipEventRecord(done, s));
        prev = done;
    }
    for (int i = 0; i < n_streams; i++) {
        LR_HIP_CHECK(hipEventRecord(ev_join[i], g_view_streams[i]));
        LR_HIP_CHECK(hipStreamWaitEvent(caller, ev_join[i], 0));
    }
    return 0;
}

int lr_views_accumulate(int n_views, const float* const* viewmatrices, const float* const* projmatrices,
                        const float* const* cam_positions, const float* tan_fovx, const float* tan_fovy,
                        int P, int D, int M, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* const* dL_dpix, float* const* out_color, int* const* out_radii,
                        float* acc_mean2D, float* acc_opacity, float* acc_color, float* acc_mean3D, float* acc_cov3D,
                        float* acc_sh, float* acc_scale, float* acc_rot,
                        char* workspace, size_t workspace_bytes, long long binning_capacity, int n_streams,
                        void* stream_)
{
    return views_core(n_views, viewmatrices, projmatrices, cam_positions, tan_fovx, tan_fovy, P, D, M, background, width,
                      height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                      dL_dpix, out_color, out_radii, acc_mean2D, acc_opacity, acc_color, acc_mean3D, acc_cov3D, acc_sh,
                      acc_scale, acc_rot, workspace, workspace_bytes, binning_capacity, n_streams, stream_, nullptr, 0.f,
                      nullptr);
}

size_t lr_views_train_workspace_bytes(int P, int width, int height, long long binning_capacity, int n_streams)
{
    if (n_streams < 1) n_streams = 1;
    if (n_streams > kMaxViewStreams) n_streams = kMaxViewStreams;
    return view_slot_layout(P, width, height, binning_capacity, true).total * (size_t)n_streams;
}

int lr_views_train_accumulate(int n_views, const float* const* viewmatrices, const float* const* projmatrices,
                              const float* const* cam_positions, const float* tan_fovx, const float* tan_fovy,
                              int P, int D, int M, const float* background, int width, int height,
                              const float* means3D, const float* shs, const float* opacities, const float* scales,
                              float scale_modifier, const float* rotations, const float* const* targets,
                              float lambda_dssim, float* out_losses, float* const* out_color, int* const* out_radii,
                              float* acc_mean2D, float* acc_opacity, float* acc_mean3D, float* acc_sh, float* acc_scale,
                              float* acc_rot, char* workspace, size_t workspace_bytes, long long binning_capacity,
                              int n_streams, void* stream_)
{
    if (!targets) return fail(LR_ERR_INVALID_ARG, "targets are required");
    return views_core(n_views, viewmatrices, projmatrices, cam_positions, tan_fovx, tan_fovy, P, D, M, background, width,
                      height, means3D, shs, nullptr, opacities, scales, scale_modifier, rotations, nullptr, nullptr,
                      out_color, out_radii, acc_mean2D, acc_opacity, nullptr, acc_mean3D, nullptr, acc_sh, acc_scale,
                      acc_rot, workspace, workspace_bytes, binning_capacity, n_streams, stream_, targets, lambda_dssim,
                      out_losses);
}

static int views_check_core(const char* workspace, int P, int width, int height, long long binning_capacity, int n_streams,
                            void* stream_, bool with_loss)
{
    using namespace lr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (!workspace) return fail(LR_ERR_INVALID_ARG, "NULL workspace");
    if (n_streams < 1) n_streams = 1;
    if (n_streams > kMaxViewStreams) n_streams = kMaxViewStreams;
    const ViewSlot SL = view_slot_layout(P, width, height, binning_capacity, with_loss);
    uint32_t flags[kMaxViewStreams] = { 0, 0, 0, 0 };
    for (int i = 0; i < n_streams; i++) {
        const GeomHeader* hdr = reinterpret_cast<const GeomHeader*>(workspace + (size_t)i * SL.total + SL.geom);
        LR_HIP_CHECK(hipMemcpyAsync(&flags[i], &hdr->sticky_overflow, 4, hipMemcpyDeviceToHost, s));
    }
    LR_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < n_streams; i++)
        if (flags[i]) return fail(LR_ERR_OVERFLOW, "binning capacity exceeded by at least one view of the step");
    return 0;
}

int lr_views_check(const char* workspace, int P, int width, int height, long long binning_capacity, int n_streams,
                   void* stream_)
{
    return views_check_core(workspace, P, width, height, binning_capacity, n_streams, stream_, false);
}

int lr_views_train_check(const char* workspace, int P, int width, int height, long long binning_capacity, int n_streams,
                         void* stream_)
{
    return views_check_core(workspace, P, width, height, binning_capacity, n_streams, stream_, true);
}

size_t lr_select_workspace_bytes(int P) { return lr::select_workspace_bytes(P); }

int lr_select_rows(int P, const unsigned char* mask, int n_tensors, const void* const* src, void* const* dst,
                   const unsigned* row_bytes, long long dst_row_offset, int* out_count, void* workspace,
                   size_t workspace_bytes, void* stream_)
{
    if (P < 0 || n_tensors < 0 || dst_row_offset < 0) return fail(LR_ERR_INVALID_ARG, "P, n_tensors, dst_row_offset must be >= 0");
    if (!out_count || !workspace) return fail(LR_ERR_INVALID_ARG, "out_count and workspace are required");
    if (workspace_bytes < lr::select_workspace_bytes(P)) return fail(LR_ERR_INVALID_ARG, "select workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (P == 0) { LR_HIP_CHECK(hipMemsetAsync(out_count, 0, sizeof(int), s)); return 0; }
    if (!mask || (n_tensors > 0 && (!src || !dst || !row_bytes))) return fail(LR_ERR_INVALID_ARG, "mask/src/dst/row_bytes are required");
    for (int t = 0; t < n_tensors; t++) {
        if (!src[t] || !dst[t]) return fail(LR_ERR_INVALID_ARG, "NULL tensor in lr_select_rows");
        // appending behind the live rows of the same buffer is fine; compaction in place is not (rows would be
        // overwritten before other workgroups have read them)
        if (src[t] == dst[t] && dst_row_offset < P)
            return fail(LR_ERR_INVALID_ARG, "lr_select_rows: dst may alias src only when dst_row_offset >= P");
    }
    const int rc = lr::launch_select_rows(P, mask, n_tensors, src, dst, row_bytes, dst_row_offset, out_count,
                                          static_cast<char*>(workspace), s);
    if (rc == -1) return fail(LR_ERR_INVALID_ARG, "at most 32 tensors per lr_select_rows call");
    if (rc == -2) return fail(LR_ERR_INVALID_ARG, "row_bytes must be a positive multiple of 4");
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

int lr_densify_stats(int P, const int* radii, const float* dL_dmean2D, float* xyz_gradient_accum, float* denom,
                     float* max_radii2D, void* stream_)
{
    if (P < 0) return fail(LR_ERR_INVALID_ARG, "P must be >= 0");
    if (P == 0) return 0;
    if (!radii || !dL_dmean2D || !xyz_gradient_accum || !denom || !max_radii2D)
        return fail(LR_ERR_INVALID_ARG, "radii, dL_dmean2D and the three statistics are required");
    lr::launch_densify_stats(P, radii, dL_dmean2D, xyz_gradient_accum, denom, max_radii2D, reinterpret_cast<hipStream_t>(stream_));
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

int lr_pack_ply_rows(int P, int M, const float* xyz, const float* features_dc, const float* features_rest,
                     const float* opacity, const float* scaling, const float* rotation, float* out_rows, void* stream_)
{
    if (P < 0 || M < 1) return fail(LR_ERR_INVALID_ARG, "P >= 0 and M >= 1 required");
    if (P == 0) return 0;
    if (!xyz || !features_dc || !opacity || !scaling || !rotation || !out_rows || (M > 1 && !features_rest))
        return fail(LR_ERR_INVALID_ARG, "all parameter tensors and out_rows are required");
    lr::launch_pack_ply(P, M - 1, xyz, features_dc, features_rest, opacity, scaling, rotation, out_rows,
                        reinterpret_cast<hipStream_t>(stream_));
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

int lr_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const unsigned long long* numel, const double* lr, double beta1, double beta2,
                 double eps, int step, void* stream_)
{
    if (n_tensors < 0 || step < 1) return fail(LR_ERR_INVALID_ARG, "n_tensors >= 0 and step >= 1 required");
    if (n_tensors == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr) return fail(LR_ERR_INVALID_ARG, "NULL array in lr_adam_step");
    for (int t = 0; t < n_tensors; t++)
        if (numel[t] != 0 && (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t]))
            return fail(LR_ERR_INVALID_ARG, "NULL tensor in lr_adam_step");
    if (lr::launch_adam(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, step,
                        reinterpret_cast<hipStream_t>(stream_)) < 0)
        return fail(LR_ERR_INVALID_ARG, "at most 16 tensors per lr_adam_step call");
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t lr_loss_workspace_bytes(int channels, int height, int width)
{
    if (channels <= 0 || height <= 0 || width <= 0) return 0;
    return lr::loss_workspace_bytes(channels, height, width);
}

int lr_l1_dssim_forward(int channels, int height, int width, const float* image, const float* gt, float lambda_dssim,
                        float* out_loss3, void* workspace, size_t workspace_bytes, void* stream_)
{
    if (channels <= 0 || height <= 0 || width <= 0) return fail(LR_ERR_INVALID_ARG, "channels, height, width must be positive");
    if (!image || !gt || !out_loss3 || !workspace) return fail(LR_ERR_INVALID_ARG, "image/gt/out_loss3/workspace are required");
    if (workspace_bytes < lr::loss_workspace_bytes(channels, height, width))
        return fail(LR_ERR_INVALID_ARG, "loss workspace too small (lr_loss_workspace_bytes)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    lr::launch_loss_forward(channels, height, width, image, gt, lambda_dssim, out_loss3, static_cast<char*>(workspace), s);
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

int lr_l1_dssim_backward(int channels, int height, int width, const float* image, const float* gt, float lambda_dssim,
                         const float* upstream, const void* workspace, float* dL_dimage, void* stream_)
{
    if (channels <= 0 || height <= 0 || width <= 0) return fail(LR_ERR_INVALID_ARG, "channels, height, width must be positive");
    if (!image || !gt || !workspace || !dL_dimage) return fail(LR_ERR_INVALID_ARG, "image/gt/workspace/dL_dimage are required");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    lr::launch_loss_backward(channels, height, width, image, gt, lambda_dssim, upstream, static_cast<const char*>(workspace),
                             dL_dimage, s);
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

int lr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    unsigned char* present, void* stream_)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0) return fail(LR_ERR_INVALID_ARG, "P must be >= 0");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !projmatrix || !present) return fail(LR_ERR_INVALID_ARG, "NULL argument");
    lr::launch_mark_visible(P, means3D, viewmatrix, projmatrix, present, s);
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

int lr_check(const char* geom_buffer, long long* num_rendered, void* stream_)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (!geom_buffer) return fail(LR_ERR_INVALID_ARG, "NULL geom buffer");
    uint32_t meta[8];
    LR_HIP_CHECK(hipMemcpyAsync(meta, geom_buffer, sizeof(meta), hipMemcpyDeviceToHost, s));
    LR_HIP_CHECK(hipStreamSynchronize(s));
    if (num_rendered) *num_rendered = meta[0];
    if (meta[2]) return fail(LR_ERR_PREFILTERED, "Point is filtered although prefiltered is set. This shouldn't happen!");
    if (meta[1]) return fail(LR_ERR_OVERFLOW, "binning capacity exceeded");
    return 0;
}

namespace {
// A ticket's event belongs to the device that was current when it was created: tickets are only ever re-used on that
// device (one process may drive several devices; recording an event on another device's stream is an error).
struct HeaderTicket { hipEvent_t ev = nullptr; uint32_t* host = nullptr; bool busy = false; int device = -1; };
std::mutex g_hdr_mu;
std::vector<HeaderTicket> g_hdr_tickets;
void header_ticket_release(long long t)
{
    std::lock_guard<std::mutex> lock(g_hdr_mu);
    if (t >= 0 && (size_t)t < g_hdr_tickets.size()) g_hdr_tickets[(size_t)t].busy = false;
}
}  // namespace

long long lr_header_post(const char* geom_buffer, void* stream_)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (!geom_buffer) return fail(LR_ERR_INVALID_ARG, "NULL geom buffer");
    int device = 0;
    LR_HIP_CHECK(hipGetDevice(&device));
    long long t = -1;
    HeaderTicket h;
    {
        std::lock_guard<std::mutex> lock(g_hdr_mu);
        for (size_t i = 0; i < g_hdr_tickets.size(); i++)
            if (!g_hdr_tickets[i].busy && g_hdr_tickets[i].device == device) { t = (long long)i; break; }
        if (t < 0) {
            HeaderTicket n;
            n.device = device;
            LR_HIP_CHECK(hipEventCreateWithFlags(&n.ev, hipEventDisableTiming));
            const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&n.host), 8 * sizeof(uint32_t), hipHostMallocDefault);
            if (e != hipSuccess) { (void)hipEventDestroy(n.ev); return fail(LR_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
            g_hdr_tickets.push_back(n);
            t = (long long)g_hdr_tickets.size() - 1;
        }
        g_hdr_tickets[(size_t)t].busy = true;
        h = g_hdr_tickets[(size_t)t];
    }
    hipError_t e = hipMemcpyAsync(h.host, geom_buffer, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipEventRecord(h.ev, s);
    if (e != hipSuccess) {                                   // the ticket must not stay busy for ever
        header_ticket_release(t);
        return fail(LR_ERR_HIP, std::string("lr_header_post: ") + hipGetErrorString(e));
    }
    return t;
}

int lr_header_poll(long long ticket, int block, unsigned int* out8)
{
    HeaderTicket h;
    {
        std::lock_guard<std::mutex> lock(g_hdr_mu);
        if (ticket < 0 || (size_t)ticket >= g_hdr_tickets.size() || !g_hdr_tickets[(size_t)ticket].busy)
            return fail(LR_ERR_INVALID_ARG, "lr_header_poll: unknown ticket");
        h = g_hdr_tickets[(size_t)ticket];
    }
    hipError_t q = block ? hipEventSynchronize(h.ev) : hipEventQuery(h.ev);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) {
        header_ticket_release(ticket);
        return fail(LR_ERR_HIP, std::string("lr_header_poll: ") + hipGetErrorString(q));
    }
    if (out8) for (int i = 0; i < 8; i++) out8[i] = h.host[i];
    header_ticket_release(ticket);
    return 1;
}

int lr_tune_set(const char* name, int value)
{
    if (!name) return fail(LR_ERR_INVALID_ARG, "lr_tune_set: NULL name");
    for (int i = 0; i < lr::TUNE_COUNT; i++)
        if (std::strcmp(name, kTuneNames[i]) == 0) { g_tune[i] = value; return 0; }
    return fail(LR_ERR_INVALID_ARG, std::string("lr_tune_set: unknown knob ") + name);
}

int lr_profile_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
    g_prof_recs.clear();
    g_prof_on = on != 0;
    return ST_COUNT;
}

const char* lr_profile_stage_name(int stage) { return (stage >= 0 && stage < ST_COUNT) ? kStageNames[stage] : ""; }

int lr_profile_read(double* ms_per_stage, long long* calls_per_stage, int n_stages)
{
    if (!ms_per_stage || !calls_per_stage || n_stages < ST_COUNT) return fail(LR_ERR_INVALID_ARG, "need ST_COUNT slots");
    for (int i = 0; i < n_stages; i++) { ms_per_stage[i] = 0.0; calls_per_stage[i] = 0; }
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof_recs) {
        LR_HIP_CHECK(hipEventSynchronize(r.e1));
        float ms = 0.f;
        LR_HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
        ms_per_stage[r.stage] += ms;
        calls_per_stage[r.stage] += 1;
    }
    return ST_COUNT;
}

size_t lr_dist2_workspace_bytes(int P) { return lr::dist2_workspace_bytes(P); }

int lr_dist2(int P, const float* points, float* out, char* workspace, void* stream_)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0) return fail(LR_ERR_INVALID_ARG, "P must be >= 0");
    if (P == 0) return 0;
    if (!points || !out || !workspace) return fail(LR_ERR_INVALID_ARG, "NULL argument");
    lr::launch_dist2(P, points, out, workspace, s);
    LR_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
