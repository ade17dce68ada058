/*
 * lucid_raster.h -- C-ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary under LucidDreamer's `depth_diff_gaussian_rasterization_min`
 * extension.  Each entry point replaces one static method of the reference's
 * CudaRasterizer::Rasterizer (RAST/ = /root/reference/submodules/depth-diff-gaussian-rasterization-min/):
 *
 *   lr_forward       <->  Rasterizer::forward      RAST/cuda_rasterizer/rasterizer.h:31-55,
 *                                                  RAST/cuda_rasterizer/rasterizer_impl.cu:198-339
 *   lr_backward      <->  Rasterizer::backward     RAST/cuda_rasterizer/rasterizer.h:57-86,
 *                                                  RAST/cuda_rasterizer/rasterizer_impl.cu:343-444
 *   lr_mark_visible  <->  Rasterizer::markVisible  RAST/cuda_rasterizer/rasterizer.h:24-29,
 *                                                  RAST/cuda_rasterizer/rasterizer_impl.cu:141-153
 *   lr_dist2         <->  SimpleKNN::knn           /root/reference/submodules/simple-knn/simple_knn.h,
 *                                                  simple_knn.cu:186-221 (distCUDA2, spatial.cu:15-26)
 *
 * Plain pointers and sizes only (no torch types).  All pointers are DEVICE pointers (HBM) unless
 * stated otherwise; all arrays are dense row-major float32/int32 exactly as the reference lays
 * them out.  Differences from the reference signature, all additive:
 *   - the three std::function<char*(size_t)> allocators become C callbacks + a user cookie;
 *   - every call takes the HIP stream to enqueue on (the reference uses the legacy default stream);
 *   - lr_forward takes `binning_capacity` (see below) to run WITHOUT the per-view host sync;
 *   - errors are returned as negative codes (lr_last_error() gives the text) instead of C++
 *     exceptions / device traps;
 *   - gradient outputs of lr_backward need NOT be zero-filled by the caller.
 *
 * Optional inputs (shs / colors_precomp, scales+rotations / cov3D_precomp) are NULL when absent
 * (the reference tests the pointer for nullptr: forward.cu:205,241; backward.cu:390,394).
 */
#ifndef LUCID_RASTER_H_INCLUDED
#define LUCID_RASTER_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Allocator callback: must return a device pointer to at least `bytes` bytes, 256-byte aligned,
 * that stays valid until the matching lr_backward has completed
 * (replaces std::function<char*(size_t)>, RAST/rasterize_points.cu:27-33). */
typedef char* (*lr_alloc_fn)(size_t bytes, void* user);

/* Error codes (negative return values). */
#define LR_ERR_INVALID_ARG   (-10)  /* bad shapes / NULL required pointer / non-RGB (rasterizer_impl.cu:243-246) */
#define LR_ERR_HIP           (-11)  /* a HIP runtime call failed (debug => after a stream sync, auxiliary.h:166-173) */
#define LR_ERR_PREFILTERED   (-12)  /* "Point is filtered although prefiltered is set" (auxiliary.h:156-160) */
#define LR_ERR_OVERFLOW      (-13)  /* async mode: num_rendered exceeded binning_capacity */
#define LR_ERR_ALLOC         (-14)  /* allocator callback returned NULL */
#define LR_NUM_RENDERED_ON_DEVICE (-1) /* lr_forward return value in async mode */

/* lr_backward accumulate_mask bits (one per gradient output, in argument order) */
#define LR_ACC_MEAN2D  (1u << 0)
#define LR_ACC_CONIC   (1u << 1)
#define LR_ACC_OPACITY (1u << 2)
#define LR_ACC_COLOR   (1u << 3)
#define LR_ACC_MEAN3D  (1u << 4)
#define LR_ACC_COV3D   (1u << 5)
#define LR_ACC_SH      (1u << 6)
#define LR_ACC_SCALE   (1u << 7)
#define LR_ACC_ROT     (1u << 8)
/* write-mode outputs: rows of Gaussians without a tile instance stay UNWRITTEN instead of zero-filled (see lr_adam_step_masked) */
#define LR_ACC_NO_ZERO_FILL (1u << 31)

const char* lr_last_error(void);
const char* lr_version(void);

/* Scratch sizes (bytes).  geom depends on P only, img on W,H only, binning on the number of
 * tile instances R (num_rendered) -- cf. required<GeometryState/ImageState/BinningState>,
 * RAST/cuda_rasterizer/rasterizer_impl.cu:226, 239, 284. */
size_t lr_geom_bytes(int P);
size_t lr_img_bytes(int width, int height);
size_t lr_binning_bytes(long long R);

/*
 * Forward.  Returns num_rendered (>= 0) in exact mode, LR_NUM_RENDERED_ON_DEVICE in async mode,
 * or a negative LR_ERR_*.
 *
 * binning_capacity == 0  (exact mode, reference behaviour): after the tile-count scan the host
 *     reads num_rendered back (one 4-byte D2H + stream sync, as rasterizer_impl.cu:281-282) and
 *     sizes the binning buffer exactly.
 * binning_capacity  > 0  (async mode): no host synchronisation at all.  The binning buffer is
 *     sized for `binning_capacity` tile instances; num_rendered stays in the geom buffer header.
 *     If the view needs more, nothing is written out of bounds, the overflow flag in the header
 *     is set and lr_check (or lr_views_check for the multi-view entry points) returns LR_ERR_OVERFLOW.  The HOST side
 *     of lr_backward does not look at the flag (it would cost a synchronisation), its KERNELS do: the backward of an
 *     overflowed view writes nothing -- its gradients are zero (write mode) or it adds nothing (accumulate mode); a
 *     truncated instance list is never differentiated.  Callers that need that view's gradients call lr_check after
 *     the step and run the view again with binning_capacity = 0 (the Python operator does this by itself).
 *
 * out_color [3,H,W], out_depth [1,H,W], radii [P] are fully written (no pre-fill needed).
 */
int lr_forward(lr_alloc_fn geom_alloc, void* geom_user,
               lr_alloc_fn binning_alloc, void* binning_user,
               lr_alloc_fn img_alloc, void* img_user,
               int P, int D, int M,
               const float* background,
               int width, int height,
               const float* means3D,
               const float* shs,
               const float* colors_precomp,
               const float* opacities,
               const float* scales,
               float scale_modifier,
               const float* rotations,
               const float* cov3D_precomp,
               const float* viewmatrix,
               const float* projmatrix,
               const float* cam_pos,
               float tan_fovx, float tan_fovy,
               int prefiltered,
               float* out_color,
               float* out_depth,
               int* radii,
               int debug,
               long long binning_capacity,
               void* stream);

/*
 * Backward.  R is the value lr_forward returned and binning_capacity the value it was given
 * (exact mode: R >= 0, capacity 0; async mode: R = LR_NUM_RENDERED_ON_DEVICE, capacity > 0; no host
 * synchronisation happens in either mode -- use lr_check to learn about an overflow).  As in the reference, whose backward lays the
 * binning state out from R (rasterizer_impl.cu:364-366), these two values ARE used: they bound the number of list segments the
 * blend backward launches workgroups for (one per 256 instances of a tile's list beyond its first 256).  (Library versions up
 * to 0.3 accepted and ignored them.)  A call whose values are SMALLER than its forward's may leave listed segments without a
 * workgroup: the kernel notices, the view's gradients are incomplete, and the condition is reported as LR_ERR_INVALID_ARG by
 * lr_check on the view's geom buffer and, with debug != 0, by lr_backward itself.  Larger values only cost idle workgroups;
 * R = LR_NUM_RENDERED_ON_DEVICE with capacity 0 ("unknown") selects kernels that are correct for any launch size.
 * dL_depths is accepted and ignored, exactly as the reference does
 * (RAST/cuda_rasterizer/backward.cu:457-464, 539-554 are commented out).
 * accumulate_mask: bit k set (LR_ACC_*) => that output is ACCUMULATED into (rows of visible Gaussians are
 * added to the existing contents, rows of culled Gaussians are not touched); bit clear => the output is
 * fully written (zero rows for culled Gaussians), no pre-fill needed.  0 reproduces the reference contract.
 * Every gradient output pointer (in either mode) must be 16-byte aligned: the kernels use 16-byte vector
 * accesses on them; a misaligned pointer is rejected with LR_ERR_INVALID_ARG.
 * Outputs:
 *   dL_dmean2D [P,3] (z = 0), dL_dconic [P,4] (slots x,y,w; may be NULL), dL_dopacity [P],
 *   dL_dcolor [P,3], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3] (NULL iff shs NULL),
 *   dL_dscale [P,3], dL_drot [P,4] (written as zeros when cov3D_precomp is used).
 * Intermediate outputs the caller does not need may be NULL: dL_dconic always; dL_dcolor unless
 * colors_precomp is given; dL_dcov3D unless cov3D_precomp is given; dL_dscale/dL_drot unless scales is given.
 * Returns 0 or a negative LR_ERR_*.
 */
int lr_backward(int P, int D, int M, int R,
                const float* background,
                int width, int height,
                const float* means3D,
                const float* shs,
                const float* colors_precomp,
                const float* scales,
                float scale_modifier,
                const float* rotations,
                const float* cov3D_precomp,
                const float* viewmatrix,
                const float* projmatrix,
                const float* campos,
                float tan_fovx, float tan_fovy,
                const int* radii,
                char* geom_buffer,
                char* binning_buffer,
                char* image_buffer,
                const float* dL_dpix,
                const float* dL_depths,
                float* dL_dmean2D,
                float* dL_dconic,
                float* dL_dopacity,
                float* dL_dcolor,
                float* dL_dmean3D,
                float* dL_dcov3D,
                float* dL_dsh,
                float* dL_dscale,
                float* dL_drot,
                int debug,
                long long binning_capacity,
                unsigned int accumulate_mask,
                void* stream);

/*
 * Raw-parameter fast path (no counterpart in the reference's native code).
 * The reference's Python caller materialises the activated parameters for every view before calling the
 * rasterizer -- exp(scaling), normalize(rotation), sigmoid(opacity) and torch.cat(features_dc, features_rest)
 * (R/scene/gaussian_model.py:97-117, R/gaussian_renderer/__init__.py:53-80): a 192 B/Gaussian copy plus four
 * elementwise kernels forward, and their autograd counterparts backward.  These two entry points take the STORED
 * GaussianModel tensors instead and apply the activations (forward) and their derivatives (backward) inside the
 * per-Gaussian kernels:
 *   xyz [P,3]; features_dc [P,1,3]; features_rest [P,M-1,3] (NULL iff M == 1); opacity_raw [P,1] (pre-sigmoid);
 *   scaling_raw [P,3] (pre-exp); rotation_raw [P,4] (pre-normalisation, torch.nn.functional.normalize eps 1e-12).
 * M = 1 + number of rest coefficients; D as in lr_forward.  colors_precomp / cov3D_precomp / prefiltered do not
 * exist in this mode.  Everything else (scratch buffers, binning_capacity, stream, return value, errors) is as in
 * lr_forward / lr_backward; the scratch buffers of a raw forward must be used with lr_backward_raw.
 * Gradients are with respect to the stored tensors: dL_dopacity_raw [P], dL_dxyz [P,3], dL_dfeatures_dc [P,3],
 * dL_dfeatures_rest [P,M-1,3], dL_dscaling_raw [P,3], dL_drotation_raw [P,4]; dL_dmean2D [P,3] as in lr_backward.
 * accumulate_mask uses the LR_ACC_* bits of the corresponding lr_backward outputs (LR_ACC_SH covers both feature
 * tensors).
 */
int lr_forward_raw(lr_alloc_fn geom_alloc, void* geom_user,
                   lr_alloc_fn binning_alloc, void* binning_user,
                   lr_alloc_fn img_alloc, void* img_user,
                   int P, int D, int M,
                   const float* background,
                   int width, int height,
                   const float* xyz,
                   const float* features_dc,
                   const float* features_rest,
                   const float* opacity_raw,
                   const float* scaling_raw,
                   float scale_modifier,
                   const float* rotation_raw,
                   const float* viewmatrix,
                   const float* projmatrix,
                   const float* cam_pos,
                   float tan_fovx, float tan_fovy,
                   float* out_color,
                   float* out_depth,
                   int* radii,
                   int debug,
                   long long binning_capacity,
                   void* stream);

int lr_backward_raw(int P, int D, int M, int R,
                    const float* background,
                    int width, int height,
                    const float* xyz,
                    const float* features_dc,
                    const float* features_rest,
                    const float* opacity_raw,
                    const float* scaling_raw,
                    float scale_modifier,
                    const float* rotation_raw,
                    const float* viewmatrix,
                    const float* projmatrix,
                    const float* campos,
                    float tan_fovx, float tan_fovy,
                    const int* radii,
                    char* geom_buffer,
                    char* binning_buffer,
                    char* image_buffer,
                    const float* dL_dpix,
                    float* dL_dmean2D,
                    float* dL_dopacity_raw,
                    float* dL_dxyz,
                    float* dL_dfeatures_dc,
                    float* dL_dfeatures_rest,
                    float* dL_dscaling_raw,
                    float* dL_drotation_raw,
                    int debug,
                    long long binning_capacity,
                    unsigned int accumulate_mask,
                    void* stream);

/*
 * Multi-view step (new; the reference renders one view per Python iteration, luciddreamer.py:291-304).
 * Runs lr_forward + lr_backward for n_views views of ONE parameter set and ACCUMULATES the gradients into the
 * acc_* buffers (same shapes as lr_backward's outputs; acc_color / acc_cov3D / acc_sh / acc_scale / acc_rot may be
 * NULL when the corresponding input is absent).  Everything is enqueued from C in one call: views alternate over
 * up to 4 chains (forward of view i+1 overlaps the backward of view i; the accumulating kernels are chained by
 * events): the chain that ends with the last view runs on `stream` itself, the others on streams of the library
 * which are forked from / joined to `stream` with events -- no host synchronisation; everything the call
 * enqueued is ordered before whatever is enqueued on `stream` after it.
 * Async mode only (binning_capacity > 0); an overflow of any view is latched per slot and reported by
 * lr_views_check (which synchronises).  Per-view arrays are HOST arrays of length n_views holding DEVICE
 * pointers (viewmatrices, projmatrices, cam_positions, dL_dpix [3,H,W], optional out_color [3,H,W] and
 * out_radii [P], entries or whole arrays may be NULL) or floats (tan_fovx, tan_fovy).
 * workspace: lr_views_workspace_bytes(P, W, H, binning_capacity, n_streams) device bytes, 256-byte aligned.
 */
size_t lr_views_workspace_bytes(int P, int width, int height, long long binning_capacity, int n_streams);
int lr_views_accumulate(int n_views, const float* const* viewmatrices, const float* const* projmatrices,
                        const float* const* cam_positions, const float* tan_fovx, const float* tan_fovy,
                        int P, int D, int M, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* const* dL_dpix, float* const* out_color, int* const* out_radii,
                        float* acc_mean2D, float* acc_opacity, float* acc_color, float* acc_mean3D, float* acc_cov3D,
                        float* acc_sh, float* acc_scale, float* acc_rot,
                        char* workspace, size_t workspace_bytes, long long binning_capacity, int n_streams,
                        void* stream);
int lr_views_check(const char* workspace, int P, int width, int height, long long binning_capacity, int n_streams,
                   void* stream);

/*
 * The same multi-view step with the photometric loss inside (render -> L1 + DSSIM against a target -> backward, per
 * view, on the view's stream): the per-iteration body of the training loop, R/luciddreamer.py:296-304, for n_views
 * views of one parameter set.  targets[v]: [3,H,W] device images; out_losses: device float[3 * n_views] receiving
 * {loss, l1, ssim} per view; gradients of sum_v loss_v are accumulated into the acc_* buffers (SH colours and
 * scale/rotation covariances only).  Workspace from lr_views_train_workspace_bytes; overflow check with
 * lr_views_train_check.  No host synchronisation.
 */
size_t lr_views_train_workspace_bytes(int P, int width, int height, long long binning_capacity, int n_streams);
int lr_views_train_accumulate(int n_views, const float* const* viewmatrices, const float* const* projmatrices,
                              const float* const* cam_positions, const float* tan_fovx, const float* tan_fovy,
                              int P, int D, int M, const float* background, int width, int height,
                              const float* means3D, const float* shs, const float* opacities, const float* scales,
                              float scale_modifier, const float* rotations, const float* const* targets,
                              float lambda_dssim, float* out_losses, float* const* out_color, int* const* out_radii,
                              float* acc_mean2D, float* acc_opacity, float* acc_mean3D, float* acc_sh, float* acc_scale,
                              float* acc_rot, char* workspace, size_t workspace_bytes, long long binning_capacity,
                              int n_streams, void* stream);
int lr_views_train_check(const char* workspace, int P, int width, int height, long long binning_capacity, int n_streams,
                         void* stream);

/*
 * Row surgery of the Gaussian parameter set.  The reference changes the number of Gaussians
 * with boolean-mask indexing / torch.cat applied tensor by tensor to the six parameters and both Adam moments of
 * each (R/scene/gaussian_model.py:273-340 prune_points, _prune_optimizer, cat_tensors_to_optimizer; :342-403
 * densify_and_clone / densify_and_split / densify_and_prune).
 *   lr_select_rows: for every i in [0,P) with mask[i] != 0, in increasing i, row i of each of the n_tensors source
 *     tensors is copied to row (dst_row_offset + rank(i)) of its destination; rank(i) = number of selected rows
 *     before i.  row_bytes[t] (multiple of 4) is the row size of tensor t; src/dst/row_bytes are HOST arrays of
 *     n_tensors (<= 32) entries holding DEVICE pointers; dst may alias src only when dst_row_offset >= P (appending
 *     behind the live rows of a capacity buffer); a compaction goes to the other half of a ping-pong buffer.  *out_count (device int) receives the number
 *     of selected rows.  n_tensors == 0 only counts.  No host synchronisation.
 *   lr_pack_ply_rows: builds the 17 + 3*(M-1) float vertex records written by GaussianModel.save_ply (:193-208:
 *     x y z nx ny nz, f_dc_*, f_rest_* channel-major, opacity, scale_*, rot_*) in out_rows [P, 17+3(M-1)] (device).
 */
/* Densification statistics of one rendered view in one pass (R/luciddreamer.py:310-311 and R/scene/gaussian_model.py:405-407):
 * for every Gaussian with radii > 0: max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |dL_dmean2D[:, :2]|;
 * denom += 1.  radii [P] int32, dL_dmean2D [P,3] (the gradient of `means2D`), statistics [P,1], [P,1], [P] float32. */
int lr_densify_stats(int P, const int* radii, const float* dL_dmean2D, float* xyz_gradient_accum, float* denom,
                     float* max_radii2D, void* stream);
size_t lr_select_workspace_bytes(int P);
int lr_select_rows(int P, const unsigned char* mask, int n_tensors, const void* const* src, void* const* dst,
                   const unsigned* row_bytes, long long dst_row_offset, int* out_count, void* workspace,
                   size_t workspace_bytes, void* stream);
int lr_pack_ply_rows(int P, int M, const float* xyz, const float* features_dc, const float* features_rest,
                     const float* opacity, const float* scaling, const float* rotation, float* out_rows, void* stream);

/*
 * One-launch Adam step over the parameter tensors of a GaussianModel (the optimiser step after the gradient
 * all-reduce of the data-parallel step; the reference uses torch.optim.Adam(l, lr=0.0, eps=1e-15) with one param group
 * -- one learning rate -- per tensor, R/scene/gaussian_model.py:152-165).  Formula and operation order of torch's
 * single-tensor Adam without weight decay / amsgrad / maximize.  params, grads, exp_avg, exp_avg_sq, numel, lr are
 * HOST arrays of n_tensors (<= 16) entries (device pointers / element counts / learning rates); step is the 1-based
 * step count of these tensors.  Hyper-parameters are doubles (as in Python) and rounded to float once, after
 * 1 - beta and lr / (1 - beta1^t) have been formed.  Updates params, exp_avg, exp_avg_sq in place.
 */
int lr_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const unsigned long long* numel, const double* lr, double beta1, double beta2,
                 double eps, int step, void* stream);

/*
 * lr_adam_step for gradients that are only VALID in the rows of the Gaussians one view visited (SURVEY.md section 8f-4;
 * R/luciddreamer.py:296-327 runs `loss.backward()` and `optimizer.step()` back to back, one view per iteration).  A backward
 * called with LR_ACC_NO_ZERO_FILL in its accumulate_mask writes the gradient rows of the Gaussians that own a tile instance and
 * leaves every other row of its write-mode outputs UNWRITTEN (dL_dmean2D excepted, which is zero-filled as always: its readers
 * go by radii > 0); this step takes the gradient of a Gaussian without a tile instance as zero without reading it -- or of every
 * Gaussian, when the view overflowed its binning buffer and the backward skipped it.  Saved against lr_backward* +
 * lr_adam_step: the zero-fill pass over the gradient tensors (the reference memsets 300 B per Gaussian per backward,
 * rasterize_points.cu:154-162) and the read of those zeros.  Same arithmetic element for element: parameters and moments are
 * bit-identical to the unmasked pair.  geom_buffer: the scratch of the view's forward (its per-Gaussian instance counts are the
 * mask); every tensor must have P rows, row_len[t] floats each (numel[t] = P * row_len[t]), all arrays 16-byte aligned.
 */
int lr_adam_step_masked(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const unsigned long long* numel, const unsigned int* row_len, const double* lr,
                        double beta1, double beta2, double eps, int step, const char* geom_buffer, int P, void* stream);

/*
 * Fused photometric loss of the training loop:
 *     loss = (1 - lambda) * mean|image - gt| + lambda * (1 - mean(SSIM_map(image, gt)))
 * replacing l1_loss + ssim of R/utils/loss.py:18-69 as composed in R/luciddreamer.py:301-304 (11x11 window = outer
 * product of a normalised Gaussian, sigma 1.5; zero padding 5; C1 = 0.01^2, C2 = 0.03^2; mean over all C*H*W).
 * image, gt: [C,H,W] float32, contiguous (a batch [B,C,H,W] is passed as channels = B*C).
 *   lr_l1_dssim_forward : out_loss3 (device, 3 floats) = {loss, l1, ssim}; fills `workspace` (device,
 *                         lr_loss_workspace_bytes) with what the backward needs.  No host synchronisation.
 *   lr_l1_dssim_backward: dL_dimage [C,H,W] = upstream * d loss / d image, from the workspace of the forward on the
 *                         same inputs; `upstream` is a device scalar (autograd's grad_output) or NULL for 1.
 * Deterministic (no atomics).  Return 0 or a negative LR_ERR_*.
 */
size_t lr_loss_workspace_bytes(int channels, int height, int width);
int lr_l1_dssim_forward(int channels, int height, int width, const float* image, const float* gt, float lambda_dssim,
                        float* out_loss3, void* workspace, size_t workspace_bytes, void* stream);
int lr_l1_dssim_backward(int channels, int height, int width, const float* image, const float* gt, float lambda_dssim,
                         const float* upstream, const void* workspace, float* dL_dimage, void* stream);
/* The same backward for a caller that composes l1 and ssim ITSELF (R/luciddreamer.py:301-303 calls l1_loss and ssim
 * separately and weights them in Python): dL_dimage = w_l1[0] * d l1 / d image + w_ssim[0] * d ssim / d image, both weights
 * device scalars (autograd's grad_outputs of the two means) -- no host synchronisation to read them. */
int lr_l1_dssim_backward_weights(int channels, int height, int width, const float* image, const float* gt, const float* w_l1,
                                 const float* w_ssim, const void* workspace, float* dL_dimage, void* stream);

/* present[P] (1 byte each) = view-space z > 0.2.  Returns 0 or a negative LR_ERR_*. */
int lr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    unsigned char* present, void* stream);

/* Synchronises `stream` and reports the header state of a geom buffer written by lr_forward:
 * num_rendered via *num_rendered (may be NULL); returns 0, LR_ERR_OVERFLOW or LR_ERR_PREFILTERED. */
int lr_check(const char* geom_buffer, long long* num_rendered, void* stream);

/* The same without blocking (async mode's deferred overflow check): lr_header_post enqueues, on `stream`, a copy of the
 * first 8 header words {num_rendered, overflow, prefilter trap, capacity, P, num_sorted, num_instances, bin_bound} into
 * pinned memory owned by the library and returns a ticket >= 0 (or a negative LR_ERR_*).  lr_header_poll(ticket, block,
 * out8) returns 1 and fills out8 once the copy has completed (the ticket is then released), 0 if it has not and
 * block == 0, a negative LR_ERR_* on a bad ticket.  The device current at lr_header_post must be the buffer's. */
long long lr_header_post(const char* geom_buffer, void* stream);
/* "Verify in the forward": lr_request_early_header() makes the next async-mode lr_forward / lr_forward_raw on the calling
 * thread post such a ticket itself as soon as the view's counts are final -- after the compaction scan, with the binning and
 * blend kernels enqueued behind it -- and lr_take_early_ticket() hands it out (-1: none, e.g. exact mode or P == 0).  A caller
 * that polls it with block = 1 right after lr_forward returns waits only for the preprocess and scan kernels (the host has
 * enqueued the rest of the forward meanwhile, so the GPU does not idle as it does during exact mode's read-back) and knows
 * before it hands the image to anyone whether the view overflowed; if so it calls lr_forward again with binning_capacity = 0.
 * The Python operator's default policy. */
void lr_request_early_header(void);
long long lr_take_early_ticket(void);
int lr_header_poll(long long ticket, int block, unsigned int* out8);

/* Optional per-stage timing with HIP events recorded on the call's stream (bench.py roofline leg).
 * lr_profile_enable(1) clears and starts recording, (0) stops; returns the number of stages.
 * lr_profile_read waits for the recorded events and returns, per stage, the summed elapsed
 * milliseconds and the number of recorded calls.  Stage names: lr_profile_stage_name(i).
 * The stages "preprocess", "render_fwd", "render_bwd", "gauss_bwd" are exactly one kernel launch each. */
/* Diagnostics: switch a kernel variant at run time (benchmark tooling measures two variants alternately in one process).
 * Knobs: "bwd_red" (reduction variant of the blend backward), "blend_quad", "tile_map", "preprocess", "gauss_bwd", "tsort",
 * "walk_own" (instances of a Gaussian the binning walks on its own lane), "hit_mask" (0: binning without preprocess's tile masks);
 * value -1 restores the library's own rule.  Results are identical up to float summation order whatever the setting.
 * Not part of the reference interface (it has no equivalent). */
/* Ticket of the header of the LAST async-mode (binning_capacity > 0) lr_forward / lr_forward_raw on the calling thread, for
 * lr_header_poll -- or -1 (no such forward yet, exact mode, P == 0).  Costs nothing: async-mode forwards leave their header in
 * a ring of host-visible slots written by the scan kernel itself (no copy, no event; the poll spins on the slot's tag), which
 * is what a caller that keeps several views in flight checks its views with (luciddreamer_amd/config.py).  The slot of a
 * ticket is re-used after 4096 further forwards on the device. */
long long lr_forward_ticket(void);
/* A STEP of several views issued through the per-view entry points (what lr_views_accumulate is for callers that need the
 * rendered image between forward and backward).  Between lr_step_begin and lr_step_end on a device, every accumulate-mode
 * lr_backward / lr_backward_raw whose accumulate_mask covers mean2D, opacity, mean3D, scale and rotation (and that uses neither
 * colors_precomp nor cov3D_precomp) adds those five rows into ONE interleaved 64-byte row per Gaussian owned by the library
 * instead of five scattered 12-16 byte read-modify-writes; lr_step_end(stream) adds the touched rows into the five tensors
 * the step's calls named (calls that name other tensors, or another P, accumulate directly as before).  The calls of a step
 * must be ordered among themselves -- one stream, or lr_backward_wait_event chaining -- as accumulate-mode calls into shared
 * tensors must be anyway; `stream` of lr_step_end must be ordered after all of them.  Returns 0 or a negative LR_ERR_*. */
int lr_step_begin(void);
int lr_step_end(void* stream);
/* Close a step WITHOUT handing its rows over: for a step that was abandoned (error in the caller's loop, a begin whose end never
 * came).  The tensors named by its views may be gone by then; nothing is written through the remembered pointers. */
int lr_step_abort(void);
/* Accumulate-mode backward passes of different views on different streams add into the SAME gradient tensors and must not
 * overlap there.  `event` (a hipEvent_t, or NULL) is consumed by the next lr_backward / lr_backward_raw on the calling thread:
 * its stream waits for the event after the blend backward (which writes only the call's own scratch) and before the kernels
 * that touch the outputs -- so a caller that records an event after each backward and passes it to the next one chains the
 * accumulations while the blend backward of one view still overlaps the per-Gaussian backward of the previous one
 * (csrc/torch_ext.cpp does this for the autograd operator; lr_views_accumulate does it internally). */
void lr_backward_wait_event(void* event);
/* Test hook: force one of the SHIPPED code paths that the library otherwise picks by rule (value -1 = the rule again).
 * Results never depend on it beyond float rounding between kernel shapes; the parity suite runs every path through it.
 *   "strict" 1           the blend in the reference's own float operations (luciddreamer_amd.config.set_strict_parity)
 *   "views_in_flight" n  hint: the caller keeps n views' kernels in flight on different streams (parallel.ViewStreams)
 *   "blend_quad" 0/1/2   blend backward: 2 waves per tile / 4 waves per tile / 1 wave per tile
 *   "fwd_pair" 0/2       blend forward: quadrant kernel / 1 wave per tile
 *   "tile_map" 0/1       tile -> workgroup map: XCD bands / plain
 *   "bwd_seg" 0          the blend backward walks whole lists instead of 256-position segments
 *   "bwd_red" 4          every wave of the blend backward takes the loop copy with the `pos < last` test
 *   "preprocess" 0/1     plain / pooled preprocess kernel;  "hit_mask" 0: no tile masks;  "tsort" 0/1/2, "walk_own" n: binning
 *   "gauss_bwd" 0        no interleaved step accumulator
 * Values that select a RETIRED kernel ("part_scan", "bwd_red" 0 / 2 / 3, "fwd_pair" 1) and every LR_* environment override exist only in the
 * diagnostics build (-DLR_DIAGNOSTICS, `python -m luciddreamer_amd.build --diagnostics`; lr_version() then says "+diagnostics");
 * the product library rejects them with LR_ERR_INVALID_ARG and reads no environment variable. */
int lr_tune_set(const char* name, int value);
/* Kernel shapes of the process's last blend launches (either pointer may be NULL): forward 0 quadrant kernel, 1 with
 * candidate pairs, 2 one wave per tile; backward 0 two waves per tile, 1 four, 2 one; -1 = none yet.  For tests that must
 * know WHICH kernels a configuration ran (e.g. that the headline's step ran the one-wave-per-tile pair). */
int lr_last_launch_shapes(int* forward_shape, int* backward_shape);
int lr_profile_enable(int on);
const char* lr_profile_stage_name(int stage);
int lr_profile_read(double* ms_per_stage, long long* calls_per_stage, int n_stages);

/* Mean squared distance to the 3 nearest other points (simple-knn distCUDA2).
 * points [P,3] -> out [P].  workspace: lr_dist2_workspace_bytes(P) device bytes. */
size_t lr_dist2_workspace_bytes(int P);
int lr_dist2(int P, const float* points, float* out, char* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LUCID_RASTER_H_INCLUDED */
