/*
 * ref_capi.cpp -- extern "C" face of oracle/_ref: calls the reference's own
 * CudaRasterizer::Rasterizer::{forward,backward,markVisible} (RAST/cuda_rasterizer/rasterizer.h:24-86,
 * compiled from /root/reference by oracle/build_ref.py over the host-side CUDA stand-in) and
 * SimpleKNN::knn (KNN/simple_knn.h).  TEST INFRASTRUCTURE ONLY.
 *
 * The signatures are those of oracle/raster_oracle.c (oracle_forward / oracle_backward / ...), so that
 * oracle/oracle.py drives both through one front-end.  The three scratch buffers are carved by the
 * reference's own GeometryState/ImageState/BinningState::fromChunk (rasterizer_impl.cu:155-194); the
 * accessors below read them through the reference's struct definitions (rasterizer_impl.h).
 */
#include "cuda_runtime.h"
#include "rasterizer_impl.h"          /* reference header: pulls in rasterizer.h */
#include "simple_knn.h"               /* reference header */

namespace shim { bool g_trapped = false; int g_threads = 0; }

using namespace CudaRasterizer;

struct RefState {
    std::vector<char> geom, binning, img;
    std::vector<int> radii;               /* the forward's radii output, which the binding hands to the backward */
    GeometryState g;
    BinningState b;
    ImageState i;
    int P, W, H, R;
};

extern "C" {

void ref_set_threads(int n) { shim::g_threads = n; }

int ref_forward(int P, int D, int M, const float* background, int W, int H, const float* means3D, const float* shs,
                const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                float* out_depth, int* radii, void** state_out)
{
    RefState* st = new RefState();
    st->P = P; st->W = W; st->H = H; st->R = 0;
    /* RAST/rasterize_points.cu:27-33: resize a byte tensor, hand back its data pointer (zero-filled here so that the
     * never-written entries of culled Gaussians read as zeros instead of torch::empty garbage) */
    auto geomFunc = [st](size_t n) { st->geom.assign(n, 0); return st->geom.data(); };
    auto binFunc = [st](size_t n) { st->binning.assign(n, 0); return st->binning.data(); };
    auto imgFunc = [st](size_t n) { st->img.assign(n, 0); return st->img.data(); };
    shim::g_trapped = false;
    int rendered = Rasterizer::forward(geomFunc, binFunc, imgFunc, P, D, M, background, W, H, means3D, shs, colors_precomp,
                                       opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                       cam_pos, tan_fovx, tan_fovy, prefiltered != 0, out_color, out_depth, radii, false);
    st->R = rendered;
    st->radii.assign(radii, radii + P);
    char* p = st->geom.data();
    st->g = GeometryState::fromChunk(p, P);
    p = st->binning.data();
    st->b = BinningState::fromChunk(p, rendered);
    p = st->img.data();
    st->i = ImageState::fromChunk(p, (size_t)W * H);
    *state_out = st;
    return shim::g_trapped ? -2 : rendered;
}

void ref_backward(const void* state, int P, int D, int M, const float* background, int W, int H, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* campos, float tan_fovx, float tan_fovy, const float* dL_dpix, const float* dL_depths,
                  float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                  float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    RefState* st = (RefState*)state;
    /* radii: the binding passes the radii tensor of the forward (rasterize_points.cu:177) */
    Rasterizer::backward(P, D, M, st->R, background, W, H, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, st->radii.data(),
                         st->geom.data(), st->binning.data(), st->img.data(), dL_dpix, dL_depths, dL_dmean2D, dL_dconic,
                         dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, false);
}

void ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present)
{
    static_assert(sizeof(bool) == 1, "bool[P] is returned as bytes");
    Rasterizer::markVisible(P, (float*)means3D, (float*)viewmatrix, (float*)projmatrix, (bool*)present);
}

void ref_state_free(void* s) { delete (RefState*)s; }
int ref_state_R(const void* s) { return ((const RefState*)s)->R; }
#define ACCESSOR(name, expr) const void* ref_state_##name(const void* s) { const RefState* st = (const RefState*)s; return expr; }
ACCESSOR(depths, st->g.depths)
ACCESSOR(clamped, st->g.clamped)
ACCESSOR(means2D, st->g.means2D)
ACCESSOR(cov3D, st->g.cov3D)
ACCESSOR(conic_opacity, st->g.conic_opacity)
ACCESSOR(rgb, st->g.rgb)
ACCESSOR(tiles_touched, st->g.tiles_touched)
ACCESSOR(point_offsets, st->g.point_offsets)
ACCESSOR(point_list, st->b.point_list)
ACCESSOR(point_list_keys, st->b.point_list_keys)
ACCESSOR(ranges, st->i.ranges)
ACCESSOR(final_T, st->i.accum_alpha)
ACCESSOR(n_contrib, st->i.n_contrib)

/* simple_knn.distCUDA2 (KNN/spatial.cu:15-26): means pre-filled with 0.0 by the binding */
void ref_dist2(int P, const float* points, float* out)
{
    SimpleKNN::knn(P, (float3*)points, out);
}

}
