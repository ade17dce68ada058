// torch_ext.cpp -- compiled binding `luciddreamer_amd._C_ext` between torch tensors and the C-ABI of liblucid_raster.so
// (include/lucid_raster.h).  Takes the place of the reference's pybind11 layer RAST/rasterize_points.cu:35-221 (tensor
// checks, output / gradient allocation, resizable scratch tensors handed to the rasterizer through allocator
// callbacks, pointer marshalling) for the per-view entry points:
//
//     rasterize_gaussians            RAST/rasterize_points.h:18-38
//     rasterize_gaussians_backward   RAST/rasterize_points.h:40-63
//     mark_visible                   RAST/rasterize_points.h:65-68
//     (+ the raw-parameter pair and check(), which have no reference counterpart)
//
// No device code here: torch is used for device memory and the current HIP stream only.  Round 1 did this in
// Python over ctypes (0.35 ms of host time per view against 0.25 ms of GPU time); here a forward costs three
// at::empty calls for the scratch tensors, three for the outputs and one C call.
#include <torch/extension.h>
// ROCm builds of torch present HIP devices under the device type "cuda": the guard / stream classes to use are the
// "MasqueradingAsCUDA" ones (the plain c10::hip guard rejects a "cuda" device)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "lucid_raster.h"

namespace {

using OptT = c10::optional<at::Tensor>;

void require_device(const at::Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), "luciddreamer_amd: ", name, " must be on a HIP device (got ", t.device(),
                "); this rasterizer has no CPU path (neither has the reference: RAST/rasterize_points.cu:72)");
}

// float32, contiguous, on `dev`; empty tensors (the reference's `torch.Tensor([])` placeholders,
// RAST/.../__init__.py:198-208) and None become "absent"
struct Arg {
    at::Tensor t;
    const float* p = nullptr;
};
Arg f32(const OptT& o, const c10::Device& dev, const char* name)
{
    Arg a;
    if (!o.has_value() || !o->defined() || o->numel() == 0) return a;
    TORCH_CHECK(o->scalar_type() == at::kFloat, name, " must be float32 (got ", o->scalar_type(), ")");
    a.t = *o;
    if (a.t.device() != dev) a.t = a.t.to(dev);
    if (!a.t.is_contiguous()) a.t = a.t.contiguous();
    a.p = a.t.data_ptr<float>();
    return a;
}

struct Scratch {
    at::Tensor t;
    c10::Device dev;
    explicit Scratch(c10::Device d) : dev(d) {}
};
char* scratch_alloc(size_t bytes, void* user)
{
    Scratch* s = static_cast<Scratch*>(user);
    s->t = at::empty({static_cast<int64_t>(bytes > 0 ? bytes : 1)}, at::TensorOptions().dtype(at::kByte).device(s->dev));
    return static_cast<char*>(s->t.data_ptr());
}

// Backward passes that ACCUMULATE into caller tensors (fused gradient accumulation: several views add into the same .grad)
// are chained per device: every such backward waits -- on the device, right before its accumulating kernels -- for the
// previous one's event and records its own.  The autograd engine runs the nodes of one device on one thread, so host order
// is accumulation order; views on different streams (parallel.ViewStreams) can then share ONE engine pass for a whole group
// of views instead of paying the engine's thread hand-off per view (profiles/r04d_host_breakdown.txt: 130 us of 180).
struct AccumulateChain {
    hipEvent_t ev[2] = { nullptr, nullptr };
    int cur = 0;
    bool recorded = false;
};
AccumulateChain& accumulate_chain(int device)
{
    static std::mutex mu;
    static std::map<int, AccumulateChain> chains;
    std::lock_guard<std::mutex> lock(mu);
    AccumulateChain& c = chains[device];
    if (!c.ev[0]) {
        // device-side ordering only (never synchronised with by the host): no system-scope fence at the record
        (void)hipEventCreateWithFlags(&c.ev[0], hipEventDisableTiming | hipEventDisableSystemFence);
        (void)hipEventCreateWithFlags(&c.ev[1], hipEventDisableTiming | hipEventDisableSystemFence);
    }
    return c;
}
// Set by the compiled autograd node for the duration of a backward that runs under fused gradient accumulation: such a
// backward joins the chain even when IT only writes (the first view of a step whose leaves have no .grad yet: its result
// becomes the .grad the next view -- possibly on another stream -- accumulates into).
thread_local bool t_fused_backward = false;
struct FusedBackwardScope {
    bool prev;
    explicit FusedBackwardScope(bool on) : prev(t_fused_backward) { t_fused_backward = on; }
    ~FusedBackwardScope() { t_fused_backward = prev; }
};
struct ChainScope {
    AccumulateChain* c = nullptr;
    hipStream_t s = nullptr;
    ChainScope(bool accumulating, int device, hipStream_t stream) : s(stream)
    {
        if (!accumulating) return;
        c = &accumulate_chain(device);
        if (c->recorded) lr_backward_wait_event(c->ev[c->cur]);
    }
    ~ChainScope()
    {
        if (!c) return;
        c->cur ^= 1;                                      // the event just waited on may still be pending on other streams
        (void)hipEventRecord(c->ev[c->cur], s);
        c->recorded = true;
    }
};

[[noreturn]] void raise_for(int rc, const char* what)
{
    const std::string msg = lr_last_error();
    // the reference raises std::runtime_error for all of these (rasterizer_impl.cu:243-246, auxiliary.h:166-173)
    TORCH_CHECK(false, msg.empty() ? std::string(what) + " failed with code " + std::to_string(rc) : msg);
}

int sh_coeffs(const OptT& sh)        // rasterize_points.cu:84-88
{
    return (sh.has_value() && sh->defined() && sh->numel() != 0 && sh->size(0) != 0) ? static_cast<int>(sh->size(1)) : 0;
}

using FwdResult = std::tuple<int64_t, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>;

FwdResult empty_forward(const c10::Device& dev, int64_t H, int64_t W)
{
    // rasterize_points.cu:68-82: zero images, empty scratch, nothing launched
    auto f = at::TensorOptions().dtype(at::kFloat).device(dev);
    auto b = at::TensorOptions().dtype(at::kByte).device(dev);
    return FwdResult(0, at::zeros({3, H, W}, f), at::zeros({1, H, W}, f), at::empty({0}, f.dtype(at::kInt)),
                     at::empty({0}, b), at::empty({0}, b), at::empty({0}, b));
}

FwdResult rasterize_gaussians(const at::Tensor& background, const at::Tensor& means3D, const OptT& colors, const at::Tensor& opacity,
                              const OptT& scales, const OptT& rotations, double scale_modifier, const OptT& cov3D_precomp,
                              const at::Tensor& viewmatrix, const at::Tensor& projmatrix, double tan_fovx, double tan_fovy,
                              int64_t image_height, int64_t image_width, const OptT& sh, int64_t degree, const at::Tensor& campos,
                              bool prefiltered, bool debug, int64_t binning_capacity)
{
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");   // :57-59
    require_device(means3D, "means3D");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0), H = image_height, W = image_width;
    if (P == 0) return empty_forward(dev, H, W);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    at::Tensor out_color = at::empty({3, H, W}, fopt);          // fully written by the library: no zero fill (:68-70)
    at::Tensor out_depth = at::empty({1, H, W}, fopt);
    at::Tensor radii = at::empty({P}, fopt.dtype(at::kInt));
    const Arg m = f32(means3D, dev, "means3D"), bg = f32(background, dev, "background"), col = f32(colors, dev, "colors_precomp"),
              op = f32(opacity, dev, "opacities"), sc = f32(scales, dev, "scales"), rot = f32(rotations, dev, "rotations"),
              cov = f32(cov3D_precomp, dev, "cov3D_precomp"), view = f32(viewmatrix, dev, "viewmatrix"),
              proj = f32(projmatrix, dev, "projmatrix"), cam = f32(campos, dev, "campos"), shc = f32(sh, dev, "sh");
    Scratch geom(dev), binning(dev), img(dev);
    const int rc = lr_forward(scratch_alloc, &geom, scratch_alloc, &binning, scratch_alloc, &img, static_cast<int>(P),
                              static_cast<int>(degree), sh_coeffs(sh), bg.p, static_cast<int>(W), static_cast<int>(H), m.p, shc.p,
                              col.p, op.p, sc.p, static_cast<float>(scale_modifier), rot.p, cov.p, view.p, proj.p, cam.p,
                              static_cast<float>(tan_fovx), static_cast<float>(tan_fovy), prefiltered ? 1 : 0,
                              out_color.data_ptr<float>(), out_depth.data_ptr<float>(), radii.data_ptr<int>(), debug ? 1 : 0,
                              static_cast<long long>(binning_capacity), c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
    if (rc < 0 && rc != LR_NUM_RENDERED_ON_DEVICE) raise_for(rc, "rasterize_gaussians");
    if (!binning.t.defined()) binning.t = at::empty({0}, at::TensorOptions().dtype(at::kByte).device(dev));
    return FwdResult(rc, out_color, out_depth, radii, geom.t, binning.t, img.t);
}

// accumulate: eight optional tensors in the order of the returned tuple (means2D, colors, opacity, means3D, cov3D, sh,
// scales, rotations); a given tensor receives `+=` in place (rows of culled Gaussians untouched) and its slot of the
// result is None.  skip_unused: gradients of absent input representations are not materialised (None).
std::vector<OptT> rasterize_gaussians_backward(
    const at::Tensor& background, const at::Tensor& means3D, const at::Tensor& radii, const OptT& colors, const OptT& scales,
    const OptT& rotations, double scale_modifier, const OptT& cov3D_precomp, const at::Tensor& viewmatrix,
    const at::Tensor& projmatrix, double tan_fovx, double tan_fovy, const at::Tensor& dL_dout_color, const OptT& dL_dout_depth,
    const OptT& sh, int64_t degree, const at::Tensor& campos, const at::Tensor& geomBuffer, int64_t R,
    const at::Tensor& binningBuffer, const at::Tensor& imageBuffer, bool debug, int64_t binning_capacity,
    const std::vector<OptT>& accumulate, bool skip_unused)
{
    require_device(means3D, "means3D");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0);
    const int64_t H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    const int M = sh_coeffs(sh);
    TORCH_CHECK(accumulate.empty() || accumulate.size() == 8, "accumulate: eight entries or none");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    // order of the result tuple (rasterize_points.cu:199) and the LR_ACC_* bit of each entry
    static const int kBit[8] = { 0, 3, 2, 4, 5, 6, 7, 8 };
    const std::vector<int64_t> shapes[8] = { { P, 3 }, { P, 3 }, { P, 1 }, { P, 3 }, { P, 6 }, { P, M, 3 }, { P, 3 }, { P, 4 } };
    const Arg col = f32(colors, dev, "colors_precomp"), sc = f32(scales, dev, "scales"), rot = f32(rotations, dev, "rotations"),
              cov = f32(cov3D_precomp, dev, "cov3D_precomp");
    const bool unused[8] = { false, skip_unused && !col.p, false, false, skip_unused && !cov.p, false, skip_unused && !sc.p,
                             skip_unused && !sc.p };
    at::Tensor out[8];
    float* ptr[8];
    unsigned int mask = 0;
    std::vector<OptT> result(8);
    for (int k = 0; k < 8; k++) {
        ptr[k] = nullptr;
        if (unused[k]) continue;
        if (!accumulate.empty() && accumulate[k].has_value() && accumulate[k]->defined()) {
            const at::Tensor& t = *accumulate[k];
            int64_t n = 1;
            for (int64_t d : shapes[k]) n *= d;
            TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous() && t.device() == dev && t.numel() == n,
                        "accumulate tensors must be contiguous float32 of the gradient's size on ", dev);
            mask |= 1u << kBit[k];
            out[k] = t;
        } else {
            out[k] = at::empty(shapes[k], fopt);     // fully written by the library (culled rows = 0), cf. :154-162
            result[k] = out[k];
        }
        ptr[k] = out[k].numel() ? out[k].data_ptr<float>() : nullptr;
    }
    if (P != 0) {
        const Arg m = f32(means3D, dev, "means3D"), bg = f32(background, dev, "background"), view = f32(viewmatrix, dev, "viewmatrix"),
                  proj = f32(projmatrix, dev, "projmatrix"), cam = f32(campos, dev, "campos"), shc = f32(sh, dev, "sh"),
                  gc = f32(dL_dout_color, dev, "dL_dout_color"), gd = f32(dL_dout_depth, dev, "dL_dout_depth");
        const at::Tensor radii_c = radii.contiguous();
        hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        ChainScope chain(mask != 0 || t_fused_backward, dev.index(), cur);
        const int rc = lr_backward(static_cast<int>(P), static_cast<int>(degree), M, static_cast<int>(R), bg.p, static_cast<int>(W),
                                   static_cast<int>(H), m.p, shc.p, col.p, sc.p, static_cast<float>(scale_modifier), rot.p, cov.p,
                                   view.p, proj.p, cam.p, static_cast<float>(tan_fovx), static_cast<float>(tan_fovy),
                                   radii_c.data_ptr<int>(), static_cast<char*>(geomBuffer.data_ptr()),
                                   static_cast<char*>(binningBuffer.data_ptr()), static_cast<char*>(imageBuffer.data_ptr()), gc.p, gd.p,
                                   ptr[0], nullptr, ptr[2], ptr[1], ptr[3], ptr[4], M ? ptr[5] : nullptr, ptr[6], ptr[7],
                                   debug ? 1 : 0, static_cast<long long>(binning_capacity), mask,
                                   c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
        if (rc < 0) raise_for(rc, "rasterize_gaussians_backward");
    }
    return result;
}

FwdResult rasterize_gaussians_raw(const at::Tensor& background, const at::Tensor& xyz, const at::Tensor& features_dc,
                                  const OptT& features_rest, const at::Tensor& opacity_raw, const at::Tensor& scaling_raw,
                                  const at::Tensor& rotation_raw, double scale_modifier, const at::Tensor& viewmatrix,
                                  const at::Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t image_height,
                                  int64_t image_width, int64_t degree, const at::Tensor& campos, bool debug,
                                  int64_t binning_capacity)
{
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "means3D must have dimensions (num_points, 3)");
    require_device(xyz, "xyz");
    const c10::Device dev = xyz.device();
    const int64_t P = xyz.size(0), H = image_height, W = image_width;
    if (P == 0) return empty_forward(dev, H, W);
    TORCH_CHECK(features_dc.numel() == 3 * P, "features_dc must have dimensions (num_points, 1, 3)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const Arg rest = f32(features_rest, dev, "features_rest");
    const int M = 1 + (rest.p ? static_cast<int>(features_rest->size(1)) : 0);
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    at::Tensor out_color = at::empty({3, H, W}, fopt), out_depth = at::empty({1, H, W}, fopt);
    at::Tensor radii = at::empty({P}, fopt.dtype(at::kInt));
    const Arg bg = f32(background, dev, "background"), x = f32(xyz, dev, "xyz"), dc = f32(features_dc, dev, "features_dc"),
              op = f32(opacity_raw, dev, "opacity"), sc = f32(scaling_raw, dev, "scaling"), rot = f32(rotation_raw, dev, "rotation"),
              view = f32(viewmatrix, dev, "viewmatrix"), proj = f32(projmatrix, dev, "projmatrix"), cam = f32(campos, dev, "campos");
    Scratch geom(dev), binning(dev), img(dev);
    const int rc = lr_forward_raw(scratch_alloc, &geom, scratch_alloc, &binning, scratch_alloc, &img, static_cast<int>(P),
                                  static_cast<int>(degree), M, bg.p, static_cast<int>(W), static_cast<int>(H), x.p, dc.p, rest.p, op.p,
                                  sc.p, static_cast<float>(scale_modifier), rot.p, view.p, proj.p, cam.p, static_cast<float>(tan_fovx),
                                  static_cast<float>(tan_fovy), out_color.data_ptr<float>(), out_depth.data_ptr<float>(),
                                  radii.data_ptr<int>(), debug ? 1 : 0, static_cast<long long>(binning_capacity),
                                  c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
    if (rc < 0 && rc != LR_NUM_RENDERED_ON_DEVICE) raise_for(rc, "rasterize_gaussians_raw");
    if (!binning.t.defined()) binning.t = at::empty({0}, at::TensorOptions().dtype(at::kByte).device(dev));
    return FwdResult(rc, out_color, out_depth, radii, geom.t, binning.t, img.t);
}

// result / accumulate order: (means2D, xyz, features_dc, features_rest, opacity, scaling, rotation)
std::vector<OptT> rasterize_gaussians_raw_backward(
    const at::Tensor& background, const at::Tensor& xyz, const at::Tensor& radii, const at::Tensor& features_dc,
    const OptT& features_rest, const at::Tensor& opacity_raw, const at::Tensor& scaling_raw, const at::Tensor& rotation_raw,
    double scale_modifier, const at::Tensor& viewmatrix, const at::Tensor& projmatrix, double tan_fovx, double tan_fovy,
    const at::Tensor& dL_dout_color, int64_t degree, const at::Tensor& campos, const at::Tensor& geomBuffer, int64_t R,
    const at::Tensor& binningBuffer, const at::Tensor& imageBuffer, bool debug, int64_t binning_capacity,
    const std::vector<OptT>& accumulate, bool no_zero_fill)
{
    require_device(xyz, "xyz");
    const c10::Device dev = xyz.device();
    const int64_t P = xyz.size(0);
    const int64_t H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const Arg rest = f32(features_rest, dev, "features_rest");
    const int64_t nrest = rest.p ? features_rest->size(1) : 0;
    const int M = 1 + static_cast<int>(nrest);
    TORCH_CHECK(accumulate.empty() || accumulate.size() == 7, "accumulate: seven entries or none");
    auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    static const int kBit[7] = { 0, 4, 6, 6, 2, 7, 8 };                 // features_dc and features_rest share LR_ACC_SH
    const std::vector<int64_t> shapes[7] = { { P, 3 }, { P, 3 }, { P, 1, 3 }, { P, nrest, 3 }, { P, 1 }, { P, 3 }, { P, 4 } };
    const bool have_acc = !accumulate.empty();
    const bool feat_acc = have_acc && accumulate[2].has_value() && accumulate[2]->defined() &&
                          (nrest == 0 || (accumulate[3].has_value() && accumulate[3]->defined()));
    at::Tensor out[7];
    float* ptr[7];
    unsigned int mask = 0;
    std::vector<OptT> result(7);
    for (int k = 0; k < 7; k++) {
        const bool is_feat = (k == 2 || k == 3);
        const bool acc = have_acc && (is_feat ? feat_acc : (accumulate[k].has_value() && accumulate[k]->defined())) &&
                         !(k == 3 && nrest == 0);
        if (acc) {
            const at::Tensor& t = *accumulate[k];
            int64_t n = 1;
            for (int64_t d : shapes[k]) n *= d;
            TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous() && t.device() == dev && t.numel() == n,
                        "accumulate tensors must be contiguous float32 of the gradient's size on ", dev);
            mask |= 1u << kBit[k];
            out[k] = t;
        } else {
            out[k] = at::empty(shapes[k], fopt);
            if (!(is_feat && feat_acc)) result[k] = out[k];
        }
        ptr[k] = out[k].numel() ? out[k].data_ptr<float>() : nullptr;
    }
    // no_zero_fill: rows of Gaussians the view did not visit stay unwritten in the write-mode outputs (means2D excepted): for the
    // masked optimizer step (adam_step_masked below), which does not read them
    if (no_zero_fill) mask |= LR_ACC_NO_ZERO_FILL;
    if (P != 0) {
        const Arg bg = f32(background, dev, "background"), x = f32(xyz, dev, "xyz"), dc = f32(features_dc, dev, "features_dc"),
                  op = f32(opacity_raw, dev, "opacity"), sc = f32(scaling_raw, dev, "scaling"), rot = f32(rotation_raw, dev, "rotation"),
                  view = f32(viewmatrix, dev, "viewmatrix"), proj = f32(projmatrix, dev, "projmatrix"), cam = f32(campos, dev, "campos"),
                  gc = f32(dL_dout_color, dev, "dL_dout_color");
        const at::Tensor radii_c = radii.contiguous();
        hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        ChainScope chain((mask & ~LR_ACC_NO_ZERO_FILL) != 0, dev.index(), cur);
        const int rc = lr_backward_raw(static_cast<int>(P), static_cast<int>(degree), M, static_cast<int>(R), bg.p, static_cast<int>(W),
                                       static_cast<int>(H), x.p, dc.p, rest.p, op.p, sc.p, static_cast<float>(scale_modifier), rot.p,
                                       view.p, proj.p, cam.p, static_cast<float>(tan_fovx), static_cast<float>(tan_fovy),
                                       radii_c.data_ptr<int>(), static_cast<char*>(geomBuffer.data_ptr()),
                                       static_cast<char*>(binningBuffer.data_ptr()), static_cast<char*>(imageBuffer.data_ptr()), gc.p,
                                       ptr[0], ptr[4], ptr[1], ptr[2], nrest ? ptr[3] : nullptr, ptr[5], ptr[6], debug ? 1 : 0,
                                       static_cast<long long>(binning_capacity), mask,
                                       c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
        if (rc < 0) raise_for(rc, "rasterize_gaussians_raw_backward");
    }
    return result;
}

// lr_adam_step_masked: one Adam step over tensors with P rows each whose gradients are only valid in the rows of the Gaussians
// the view of `geomBuffer` visited (a raw-mode backward with no_zero_fill); luciddreamer_amd/optim.py FusedAdam
void adam_step_masked(const std::vector<at::Tensor>& params, const std::vector<at::Tensor>& grads,
                      const std::vector<at::Tensor>& exp_avg, const std::vector<at::Tensor>& exp_avg_sq,
                      const std::vector<double>& lrs, double beta1, double beta2, double eps, int64_t step,
                      const at::Tensor& geomBuffer)
{
    const size_t n = params.size();
    TORCH_CHECK(n >= 1 && n <= 16 && grads.size() == n && exp_avg.size() == n && exp_avg_sq.size() == n && lrs.size() == n,
                "adam_step_masked: 1..16 tensors with one gradient, two moments and one learning rate each");
    require_device(params[0], "params");
    const c10::Device dev = params[0].device();
    const int64_t P = params[0].size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    at::NoGradGuard no_grad;
    std::vector<float*> p(n), m(n), v(n);
    std::vector<const float*> g(n);
    std::vector<unsigned long long> numel(n);
    std::vector<unsigned int> row(n);
    for (size_t t = 0; t < n; t++) {
        for (const at::Tensor* x : { &params[t], &grads[t], &exp_avg[t], &exp_avg_sq[t] })
            TORCH_CHECK(x->scalar_type() == at::kFloat && x->is_contiguous() && x->device() == dev && x->numel() == params[t].numel(),
                        "adam_step_masked: contiguous float32 tensors of the parameter's size on its device");
        TORCH_CHECK(params[t].dim() >= 1 && params[t].size(0) == P, "adam_step_masked: every tensor has one row per Gaussian");
        numel[t] = static_cast<unsigned long long>(params[t].numel());
        row[t] = P ? static_cast<unsigned int>(params[t].numel() / P) : 1u;
        p[t] = numel[t] ? params[t].data_ptr<float>() : nullptr;
        g[t] = numel[t] ? grads[t].data_ptr<float>() : nullptr;
        m[t] = numel[t] ? exp_avg[t].data_ptr<float>() : nullptr;
        v[t] = numel[t] ? exp_avg_sq[t].data_ptr<float>() : nullptr;
    }
    // tensors without elements (features_rest at M = 1) are left out
    std::vector<float*> p2, m2, v2; std::vector<const float*> g2; std::vector<unsigned long long> n2; std::vector<unsigned int> r2;
    std::vector<double> l2;
    for (size_t t = 0; t < n; t++)
        if (numel[t]) { p2.push_back(p[t]); g2.push_back(g[t]); m2.push_back(m[t]); v2.push_back(v[t]); n2.push_back(numel[t]);
                        r2.push_back(row[t]); l2.push_back(lrs[t]); }
    if (p2.empty() || P == 0) return;
    const int rc = lr_adam_step_masked(static_cast<int>(p2.size()), p2.data(), g2.data(), m2.data(), v2.data(), n2.data(), r2.data(),
                                       l2.data(), beta1, beta2, eps, static_cast<int>(step),
                                       static_cast<const char*>(geomBuffer.data_ptr()), static_cast<int>(P),
                                       c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
    if (rc < 0) raise_for(rc, "adam_step_masked");
}

// ------------------------------------------------------------------------------------------------------------------
// The autograd node of the drop-in operator in C++ (RAST/depth_diff_gaussian_rasterization_min/__init__.py:44-156 is a
// Python torch.autograd.Function).  With the Python node a 1080p view cost ~230 us of host time against ~190 us of GPU
// time (profiles/r03v_host_profile.txt): the drop-in path was host-bound.  Here forward and backward run without the
// interpreter: the engine's device thread calls lr_backward directly, no GIL, no argument tuples.  luciddreamer_amd/
// rasterizer.py keeps the Python node for settings.debug (it writes the reference's snapshot_*.dump files on failure).
//
// forward arguments: the eight differentiable inputs in the reference's order (means3D, means2D, sh, colors_precomp,
// opacities, scales, rotations, cov3Ds_precomp; absent ones as empty tensors), the four tensors of the settings tuple,
// then its scalars, the async-mode binning capacity and the fused-accumulation switch (config.py).
// returns (color, radii, depth, geom); num_rendered of the call is left in a thread-local (last_num_rendered()).
// ------------------------------------------------------------------------------------------------------------------
thread_local int64_t g_last_num_rendered = 0;

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
    static constexpr int kForwardArgs = 21;

    static variable_list forward(AutogradContext* ctx, const at::Tensor& means3D, const at::Tensor& means2D, const at::Tensor& sh,
                                 const at::Tensor& colors, const at::Tensor& opacities, const at::Tensor& scales,
                                 const at::Tensor& rotations, const at::Tensor& cov3D, const at::Tensor& bg,
                                 const at::Tensor& viewmatrix, const at::Tensor& projmatrix, const at::Tensor& campos,
                                 double scale_modifier, double tan_fovx, double tan_fovy, int64_t H, int64_t W, int64_t degree,
                                 bool prefiltered, int64_t binning_capacity, bool fused_accumulate)
    {
        FwdResult r = rasterize_gaussians(bg, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D, viewmatrix,
                                          projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, false,
                                          binning_capacity);
        g_last_num_rendered = std::get<0>(r);
        const at::Tensor &color = std::get<1>(r), &depth = std::get<2>(r), &radii = std::get<3>(r), &geom = std::get<4>(r);
        // the inputs the reference saves go through save_for_backward (__init__.py:93: an in-place change of one of them before
        // backward is an error there too); everything else is kept as plain data, without a version check
        ctx->save_for_backward({means3D, sh, colors, scales, rotations, cov3D});
        ctx->saved_data["means2D"] = means2D;
        ctx->saved_data["opacities"] = opacities;
        ctx->saved_data["bg"] = bg;
        ctx->saved_data["viewmatrix"] = viewmatrix;
        ctx->saved_data["projmatrix"] = projmatrix;
        ctx->saved_data["campos"] = campos;
        ctx->saved_data["radii"] = radii;
        ctx->saved_data["geom"] = geom;
        ctx->saved_data["binning"] = std::get<5>(r);
        ctx->saved_data["img"] = std::get<6>(r);
        ctx->saved_data["scale_modifier"] = scale_modifier;
        ctx->saved_data["tan_fovx"] = tan_fovx;
        ctx->saved_data["tan_fovy"] = tan_fovy;
        ctx->saved_data["H"] = H;
        ctx->saved_data["W"] = W;
        ctx->saved_data["degree"] = degree;
        ctx->saved_data["num_rendered"] = std::get<0>(r);
        ctx->saved_data["capacity"] = binning_capacity;
        ctx->saved_data["fused"] = fused_accumulate;
        ctx->mark_non_differentiable({radii, geom});
        return {color, radii, depth, geom};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grad_out)
    {
        const variable_list in = ctx->get_saved_variables();
        const at::Tensor &means3D = in[0], &sh = in[1], &colors = in[2], &scales = in[3], &rotations = in[4], &cov3D = in[5];
        auto& d = ctx->saved_data;
        const at::Tensor means2D = d["means2D"].toTensor(), opacities = d["opacities"].toTensor(), bg = d["bg"].toTensor(),
                         viewmatrix = d["viewmatrix"].toTensor(), projmatrix = d["projmatrix"].toTensor(),
                         campos = d["campos"].toTensor();
        const int64_t H = d["H"].toInt(), W = d["W"].toInt();
        const c10::Device dev = means3D.device();
        at::Tensor g_color = grad_out[0];
        if (!g_color.defined()) g_color = at::zeros({3, H, W}, at::TensorOptions().dtype(at::kFloat).device(dev));
        const OptT g_depth = grad_out[2].defined() ? OptT(grad_out[2]) : OptT();
        // config.set_fused_grad_accumulation: a leaf input whose .grad exists (contiguous float32, 16-byte aligned: the kernels
        // accumulate with 16-byte accesses) receives `+=` inside the kernel; its slot of the result stays undefined
        const bool fused = d["fused"].toBool();
        auto leaf_grad = [&](const at::Tensor& t) -> OptT {
            if (!fused || !t.defined() || t.numel() == 0 || !t.requires_grad() || !t.is_leaf()) return OptT();
            const at::Tensor& g = t.grad();
            if (!g.defined() || !g.is_contiguous() || g.scalar_type() != at::kFloat || g.device() != dev ||
                reinterpret_cast<uintptr_t>(g.data_ptr()) % 16 != 0)
                return OptT();
            return OptT(g);
        };
        // order of rasterize_gaussians_backward's result: means2D, colors, opacity, means3D, cov3D, sh, scales, rotations
        std::vector<OptT> acc;
        if (fused)
            acc = { leaf_grad(means2D), leaf_grad(colors), leaf_grad(opacities), leaf_grad(means3D), leaf_grad(cov3D), leaf_grad(sh),
                    leaf_grad(scales), leaf_grad(rotations) };
        FusedBackwardScope fused_scope(fused);
        const std::vector<OptT> g = rasterize_gaussians_backward(
            bg, means3D, d["radii"].toTensor(), colors, scales, rotations, d["scale_modifier"].toDouble(), cov3D, viewmatrix,
            projmatrix, d["tan_fovx"].toDouble(), d["tan_fovy"].toDouble(), g_color, g_depth, sh, d["degree"].toInt(), campos,
            d["geom"].toTensor(), d["num_rendered"].toInt(), d["binning"].toTensor(), d["img"].toTensor(), false,
            d["capacity"].toInt(), acc, true);
        auto slot = [&](int k) { return g[k].has_value() ? *g[k] : at::Tensor(); };
        variable_list out(kForwardArgs);                       // one per forward argument; undefined = no gradient
        out[0] = slot(3); out[1] = slot(0); out[2] = slot(5); out[3] = slot(1); out[4] = slot(2); out[5] = slot(6); out[6] = slot(7);
        out[7] = slot(4);
        return out;
    }
};

std::vector<at::Tensor> rasterize_autograd(const at::Tensor& means3D, const at::Tensor& means2D, const at::Tensor& sh,
                                           const at::Tensor& colors, const at::Tensor& opacities, const at::Tensor& scales,
                                           const at::Tensor& rotations, const at::Tensor& cov3D, const at::Tensor& bg,
                                           const at::Tensor& viewmatrix, const at::Tensor& projmatrix, const at::Tensor& campos,
                                           double scale_modifier, double tan_fovx, double tan_fovy, int64_t H, int64_t W,
                                           int64_t degree, bool prefiltered, int64_t binning_capacity, bool fused_accumulate)
{
    return RasterizeFn::apply(means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, bg, viewmatrix, projmatrix, campos,
                              scale_modifier, tan_fovx, tan_fovy, H, W, degree, prefiltered, binning_capacity, fused_accumulate);
}

// ------------------------------------------------------------------------------------------------------------------
// One view forward + backward in ONE call, for a caller that knows dL/dcolor before the forward (parallel.ViewStreams.run_view
// with grad_output: a fixed upstream gradient, a loss formed elsewhere): no autograd node is built and nothing returns to the
// interpreter between the two halves -- the node, its saved variables and the call of its backward were ~100 us of a view's
// ~135 us of host time at 1080p, against ~35 us for the ten launches themselves (profiles/r06i_host_profile_dropin.txt).
// Every differentiable input that requires a gradient must be a LEAF whose .grad exists (contiguous float32, 16-byte
// aligned): the kernels add into it in place, exactly what the compiled node does under fused gradient accumulation
// (RasterizeFn::backward above), in the same accumulate chain.  Returns an empty vector -- nothing done -- when an input does
// not qualify: the caller takes the autograd path.  Otherwise (color, radii, depth, geom) with num_rendered in
// last_num_rendered(); the images carry no grad_fn.
// ------------------------------------------------------------------------------------------------------------------
std::vector<at::Tensor> rasterize_view_step(const at::Tensor& means3D, const at::Tensor& means2D, const at::Tensor& sh,
                                            const at::Tensor& colors, const at::Tensor& opacities, const at::Tensor& scales,
                                            const at::Tensor& rotations, const at::Tensor& cov3D, const at::Tensor& bg,
                                            const at::Tensor& viewmatrix, const at::Tensor& projmatrix, const at::Tensor& campos,
                                            double scale_modifier, double tan_fovx, double tan_fovy, int64_t H, int64_t W,
                                            int64_t degree, bool prefiltered, int64_t binning_capacity, const at::Tensor& grad_color)
{
    require_device(means3D, "means3D");
    const c10::Device dev = means3D.device();
    // order of rasterize_gaussians_backward's accumulate list: means2D, colors, opacity, means3D, cov3D, sh, scales, rotations
    const at::Tensor* in[8] = { &means2D, &colors, &opacities, &means3D, &cov3D, &sh, &scales, &rotations };
    std::vector<OptT> acc(8);
    for (int k = 0; k < 8; k++) {
        const at::Tensor& t = *in[k];
        if (!t.defined() || t.numel() == 0 || !t.requires_grad()) continue;          // no gradient wanted: written to a scratch tensor
        if (!t.is_leaf()) return {};
        const at::Tensor& g = t.grad();
        if (!g.defined() || !g.is_contiguous() || g.scalar_type() != at::kFloat || g.device() != dev || g.numel() != t.numel() ||
            reinterpret_cast<uintptr_t>(g.data_ptr()) % 16 != 0)
            return {};
        acc[k] = g;
    }
    if (!grad_color.defined() || grad_color.dim() != 3 || grad_color.size(1) != H || grad_color.size(2) != W) return {};
    at::NoGradGuard no_grad;
    FwdResult r = rasterize_gaussians(bg, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D, viewmatrix,
                                      projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, false, binning_capacity);
    g_last_num_rendered = std::get<0>(r);
    if (means3D.size(0) != 0) {
        FusedBackwardScope fused_scope(true);
        (void)rasterize_gaussians_backward(bg, means3D, std::get<3>(r), colors, scales, rotations, scale_modifier, cov3D, viewmatrix,
                                           projmatrix, tan_fovx, tan_fovy, grad_color, OptT(), sh, degree, campos, std::get<4>(r),
                                           std::get<0>(r), std::get<5>(r), std::get<6>(r), false, binning_capacity, acc, true);
    }
    return { std::get<1>(r), std::get<3>(r), std::get<2>(r), std::get<4>(r) };
}

at::Tensor mark_visible(const at::Tensor& means3D, const at::Tensor& viewmatrix, const at::Tensor& projmatrix)
{
    require_device(means3D, "means3D");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0);
    at::Tensor present = at::empty({P}, at::TensorOptions().dtype(at::kBool).device(dev));
    if (P != 0) {
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const Arg m = f32(means3D, dev, "means3D"), v = f32(viewmatrix, dev, "viewmatrix"), p = f32(projmatrix, dev, "projmatrix");
        const int rc = lr_mark_visible(static_cast<int>(P), m.p, v.p, p.p, static_cast<unsigned char*>(present.data_ptr()),
                                       c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream());
        if (rc < 0) raise_for(rc, "mark_visible");
    }
    return present;
}

// synchronise and return num_rendered of a forward; raises on async-mode overflow / prefiltered trap
int64_t check(const at::Tensor& geomBuffer)
{
    c10::hip::HIPGuardMasqueradingAsCUDA guard(geomBuffer.device());
    long long n = 0;
    const int rc = lr_check(static_cast<const char*>(geomBuffer.data_ptr()), &n,
                            c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(geomBuffer.device().index()).stream());
    if (rc < 0) raise_for(rc, "check");
    return n;
}

// async mode's deferred check without torch objects on the way: ticket of a non-blocking header read-back on the
// current stream, and its poll (None while the copy is in flight, else the 8 header words)
int64_t header_post(const at::Tensor& geomBuffer)
{
    c10::hip::HIPGuardMasqueradingAsCUDA guard(geomBuffer.device());
    const long long t = lr_header_post(static_cast<const char*>(geomBuffer.data_ptr()),
                                       c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(geomBuffer.device().index()).stream());
    if (t < 0) raise_for((int)t, "header_post");
    return t;
}

py::object header_poll(int64_t ticket, bool block)
{
    unsigned int w[8];
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = lr_header_poll(ticket, block ? 1 : 0, w);
    }
    if (rc < 0) raise_for(rc, "header_poll");
    if (rc == 0) return py::none();
    return py::make_tuple(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "luciddreamer_amd: torch <-> liblucid_raster.so (C-ABI) binding of the per-view rasterizer entry points";
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
    m.def("rasterize_gaussians_raw", &rasterize_gaussians_raw);
    m.def("rasterize_gaussians_raw_backward", &rasterize_gaussians_raw_backward);
    m.def("rasterize_autograd", &rasterize_autograd);
    m.def("last_num_rendered", [] { return g_last_num_rendered; });
    m.def("mark_visible", &mark_visible);
    m.def("rasterize_view_step", &rasterize_view_step);
    m.def("adam_step_masked", &adam_step_masked);
    m.def("check", &check);
    m.def("header_post", &header_post);
    m.def("request_early_header", [] { lr_request_early_header(); });
    m.def("take_early_ticket", [] { return (int64_t)lr_take_early_ticket(); });
    m.def("last_forward_ticket", [] { return (int64_t)lr_forward_ticket(); });
    m.def("header_poll", &header_poll);
    m.def("version", [] { return std::string(lr_version()); });
}
