// adam.hip -- one-launch Adam step over all parameter tensors of a GaussianModel (the optimiser step that follows the
// gradient all-reduce of the data-parallel step; torch.optim.Adam(l, lr=0.0, eps=1e-15) in
// R/scene/gaussian_model.py:165 with one param group, hence one learning rate, per tensor).
//
// torch's default (foreach) Adam runs ~10 elementwise kernels per step over every tensor; this is one pass that reads
// param, grad, exp_avg, exp_avg_sq (16 B per element) and writes param, exp_avg, exp_avg_sq (12 B).  Same formula and
// operation order as torch's single-tensor implementation (torch/optim/adam.py _single_tensor_adam, no weight decay, no
// amsgrad, maximize=False):
//   exp_avg    <- exp_avg + (1 - beta1) * (grad - exp_avg)                      (lerp_)
//   exp_avg_sq <- beta2 * exp_avg_sq + (1 - beta2) * grad * grad              (mul_, addcmul_)
//   denom      <- sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps
//   param      <- param - (lr / (1 - beta1^t)) * exp_avg / denom                (addcdiv_)
// HBM-bound, 28 B per element -- 16 where gradient and both moments are zero: such an element's update is the identity in
// exact arithmetic AND in float (exp_avg = fma(w1, 0 - 0, 0) = 0, exp_avg_sq = 0 * beta2 + 0 = 0, param = fma(-step, 0 / eps,
// param) = param, signed zeros included), so its three stores are skipped.  That is every SH coefficient above the active
// degree: LucidDreamer trains 2990 iterations and raises the degree every 1000 (R/luciddreamer.py:287-288), i.e. 45, 36 and
// 24 of a Gaussian's 59 parameters never receive a gradient during its three thousands.
#include "common.h"
#include <cmath>
#include <cstdint>

namespace lr {

namespace {

constexpr int ADAM_MAX_TENSORS = 16;
struct AdamTensors {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    unsigned long long n[ADAM_MAX_TENSORS];
    float step_size[ADAM_MAX_TENSORS];          // lr / (1 - beta1^t), per tensor
    int count;
};

__global__ void __launch_bounds__(256)
k_adam(AdamTensors T, float w1, float beta2, float w2, float bc2_sqrt, float eps)
{
    const int t = blockIdx.y;
    if (t >= T.count) return;
    float* __restrict__ p = T.p[t];
    const float* __restrict__ g = T.g[t];
    float* __restrict__ m = T.m[t];
    float* __restrict__ v = T.v[t];
    const unsigned long long n = T.n[t];
    const float step_size = T.step_size[t];
    // the same roundings as torch's kernels (which hipcc compiles with FMA contraction): lerp = fma(w, b - a, a),
    // addcmul = fma(value * t1, t2, self), addcdiv = fma(value, t1 / t2, self); mul_, sqrt, div, add are separate
    auto one = [&](float gi, float& mi, float& vi, float& pi) { adam_one(gi, mi, vi, pi, w1, beta2, w2, bc2_sqrt, eps, step_size); };
    // 16 bytes per lane and access (the four arrays are whole allocations: 16-byte aligned); round 3 moved one float per
    // lane and reached 4.6-5.3 TB/s of the 6.29 TB/s this part streams
    const bool wide = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    const unsigned long long n4 = wide ? n / 4 : 0;
    float4* __restrict__ p4 = reinterpret_cast<float4*>(p);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
    float4* __restrict__ m4 = reinterpret_cast<float4*>(m);
    float4* __restrict__ v4 = reinterpret_cast<float4*>(v);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * 256) {
        const float4 gi = g4[i];
        float4 mi = m4[i], vi = v4[i], pi = p4[i];
        // all twelve words zero (either sign): nothing to store
        const uint32_t any = (__float_as_uint(gi.x) | __float_as_uint(gi.y) | __float_as_uint(gi.z) | __float_as_uint(gi.w) |
                              __float_as_uint(mi.x) | __float_as_uint(mi.y) | __float_as_uint(mi.z) | __float_as_uint(mi.w) |
                              __float_as_uint(vi.x) | __float_as_uint(vi.y) | __float_as_uint(vi.z) | __float_as_uint(vi.w)) & 0x7fffffffu;
        if (any == 0u) continue;
        one(gi.x, mi.x, vi.x, pi.x); one(gi.y, mi.y, vi.y, pi.y); one(gi.z, mi.z, vi.z, pi.z); one(gi.w, mi.w, vi.w, pi.w);
        p4[i] = pi; m4[i] = mi; v4[i] = vi;
    }
    for (unsigned long long i = 4 * n4 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        float mi = m[i], vi = v[i], pi = p[i];
        const float gi = g[i];
        if (((__float_as_uint(gi) | __float_as_uint(mi) | __float_as_uint(vi)) & 0x7fffffffu) == 0u) continue;
        one(gi, mi, vi, pi);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// The same step where a Gaussian's gradient rows are only VALID if the view visited it (launch_adam_masked): element e of tensor
// t belongs to Gaussian e / row_len[t]; tiles_touched == 0 -> gradient zero, the gradient array is not read (the backward left
// those rows unwritten: no zero-fill pass, no read of zeros -- lr_backward_raw with LR_ACC_NO_ZERO_FILL).  RL: the row length as
// a compile-time constant (1, 3, 4, 45 = the GaussianModel tensors at SH degree 3; 0 = the run-time value): element -> Gaussian
// is a division per float4 group, and by a run-time divisor it costs more than the group's memory traffic.
struct AdamMasked { AdamTensors T; unsigned int row_len[ADAM_MAX_TENSORS]; };
template <unsigned int RL>
__device__ __forceinline__ void adam_masked_tensor(const AdamMasked& A, const int t, const uint32_t* __restrict__ tiles_touched,
                                                   const bool none_valid, float w1, float beta2, float w2, float bc2_sqrt, float eps)
{
    float* __restrict__ p = A.T.p[t];
    const float* __restrict__ g = A.T.g[t];
    float* __restrict__ m = A.T.m[t];
    float* __restrict__ v = A.T.v[t];
    const unsigned long long n = A.T.n[t];
    const unsigned int rl = RL ? RL : A.row_len[t];
    const float step_size = A.T.step_size[t];
    float4* __restrict__ p4 = reinterpret_cast<float4*>(p);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
    float4* __restrict__ m4 = reinterpret_cast<float4*>(m);
    float4* __restrict__ v4 = reinterpret_cast<float4*>(v);
    const unsigned int n4 = (unsigned int)(n / 4);                 // < 2^32 elements per tensor (checked by the host)
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const unsigned int e0 = 4u * i;
        const unsigned int g0 = e0 / rl, g3 = (e0 + 3u) / rl;
        bool valid[4];
        bool any_valid;
        if (g0 == g3) {
            any_valid = !none_valid && tiles_touched[g0] != 0u;
            valid[0] = valid[1] = valid[2] = valid[3] = any_valid;
        } else {
            any_valid = false;
#pragma unroll
            for (int k = 0; k < 4; k++) { valid[k] = !none_valid && tiles_touched[(e0 + k) / rl] != 0u; any_valid |= valid[k]; }
        }
        float4 mi = m4[i], vi = v4[i];
        float4 gi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any_valid) {
            gi = g4[i];
            if (!valid[0]) gi.x = 0.f;
            if (!valid[1]) gi.y = 0.f;
            if (!valid[2]) gi.z = 0.f;
            if (!valid[3]) gi.w = 0.f;
        }
        // all twelve words zero (either sign): nothing to store (as k_adam)
        const uint32_t any = (__float_as_uint(gi.x) | __float_as_uint(gi.y) | __float_as_uint(gi.z) | __float_as_uint(gi.w) |
                              __float_as_uint(mi.x) | __float_as_uint(mi.y) | __float_as_uint(mi.z) | __float_as_uint(mi.w) |
                              __float_as_uint(vi.x) | __float_as_uint(vi.y) | __float_as_uint(vi.z) | __float_as_uint(vi.w)) & 0x7fffffffu;
        if (any == 0u) continue;
        float4 pi = p4[i];
        adam_one(gi.x, mi.x, vi.x, pi.x, w1, beta2, w2, bc2_sqrt, eps, step_size);
        adam_one(gi.y, mi.y, vi.y, pi.y, w1, beta2, w2, bc2_sqrt, eps, step_size);
        adam_one(gi.z, mi.z, vi.z, pi.z, w1, beta2, w2, bc2_sqrt, eps, step_size);
        adam_one(gi.w, mi.w, vi.w, pi.w, w1, beta2, w2, bc2_sqrt, eps, step_size);
        p4[i] = pi; m4[i] = mi; v4[i] = vi;
    }
    for (unsigned long long i = 4ull * n4 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const bool ok = !none_valid && tiles_touched[(unsigned int)(i / rl)] != 0u;
        float mi = m[i], vi = v[i];
        const float gi = ok ? g[i] : 0.f;
        if (((__float_as_uint(gi) | __float_as_uint(mi) | __float_as_uint(vi)) & 0x7fffffffu) == 0u) continue;
        float pi = p[i];
        adam_one(gi, mi, vi, pi, w1, beta2, w2, bc2_sqrt, eps, step_size);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}
__global__ void __launch_bounds__(256)
k_adam_masked(AdamMasked A, float w1, float beta2, float w2, float bc2_sqrt, float eps, const uint32_t* __restrict__ tiles_touched,
              const GeomHeader* __restrict__ hdr)
{
    const int t = blockIdx.y;
    if (t >= A.T.count || A.T.n[t] == 0) return;
    const bool none_valid = hdr->overflow != 0u;                // the backward skipped the whole view: no row was written
    switch (A.row_len[t]) {
        case 1: adam_masked_tensor<1>(A, t, tiles_touched, none_valid, w1, beta2, w2, bc2_sqrt, eps); break;
        case 3: adam_masked_tensor<3>(A, t, tiles_touched, none_valid, w1, beta2, w2, bc2_sqrt, eps); break;
        case 4: adam_masked_tensor<4>(A, t, tiles_touched, none_valid, w1, beta2, w2, bc2_sqrt, eps); break;
        case 45: adam_masked_tensor<45>(A, t, tiles_touched, none_valid, w1, beta2, w2, bc2_sqrt, eps); break;
        default: adam_masked_tensor<0>(A, t, tiles_touched, none_valid, w1, beta2, w2, bc2_sqrt, eps); break;
    }
}

}  // namespace

static unsigned long long fill_adam_tensors(AdamTensors& T, int n_tensors, float* const* params, const float* const* grads,
                                            float* const* exp_avg, float* const* exp_avg_sq, const unsigned long long* numel,
                                            const double* lr, double bc1)
{
    T.count = n_tensors;
    unsigned long long max_n = 0;
    for (int t = 0; t < ADAM_MAX_TENSORS; t++) {
        const bool on = t < n_tensors;
        T.p[t] = on ? params[t] : nullptr; T.g[t] = on ? grads[t] : nullptr;
        T.m[t] = on ? exp_avg[t] : nullptr; T.v[t] = on ? exp_avg_sq[t] : nullptr;
        T.n[t] = on ? numel[t] : 0;
        T.step_size[t] = on ? (float)(lr[t] / bc1) : 0.f;
        if (on && numel[t] > max_n) max_n = numel[t];
    }
    return max_n;
}

int launch_adam_masked(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const unsigned long long* numel, const unsigned int* row_len, const double* lr,
                       double beta1, double beta2, double eps, int step, const uint32_t* tiles_touched, const GeomHeader* hdr,
                       hipStream_t s)
{
    if (n_tensors > ADAM_MAX_TENSORS) return -1;
    AdamMasked A;
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    const unsigned long long max_n = fill_adam_tensors(A.T, n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, bc1);
    for (int t = 0; t < ADAM_MAX_TENSORS; t++) {
        A.row_len[t] = (t < n_tensors && row_len[t] > 0) ? row_len[t] : 1u;
        if (t < n_tensors) {
            if (numel[t] >= (1ull << 32)) return -2;
            // 16-byte accesses on all four arrays (whole torch allocations are; a view at an odd offset is refused)
            if ((reinterpret_cast<uintptr_t>(params[t]) | reinterpret_cast<uintptr_t>(grads[t]) | reinterpret_cast<uintptr_t>(exp_avg[t]) |
                 reinterpret_cast<uintptr_t>(exp_avg_sq[t])) & 15u)
                return -3;
        }
    }
    if (max_n == 0) return 0;
    unsigned long long blocks = (max_n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_adam_masked, dim3((unsigned)blocks, n_tensors), dim3(256), 0, s, A, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)std::sqrt(bc2), (float)eps, tiles_touched, hdr);
    return 0;
}

int launch_adam(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                float* const* exp_avg_sq, const unsigned long long* numel, const double* lr, double beta1, double beta2,
                double eps, int step, hipStream_t s)
{
    if (n_tensors > ADAM_MAX_TENSORS) return -1;
    AdamTensors T;
    T.count = n_tensors;
    // scalars are formed in double, as Python does for torch.optim.Adam, and rounded to float once
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    unsigned long long max_n = 0;
    for (int t = 0; t < ADAM_MAX_TENSORS; t++) {
        const bool on = t < n_tensors;
        T.p[t] = on ? params[t] : nullptr; T.g[t] = on ? grads[t] : nullptr;
        T.m[t] = on ? exp_avg[t] : nullptr; T.v[t] = on ? exp_avg_sq[t] : nullptr;
        T.n[t] = on ? numel[t] : 0;
        T.step_size[t] = on ? (float)(lr[t] / bc1) : 0.f;
        if (on && numel[t] > max_n) max_n = numel[t];
    }
    if (max_n == 0) return 0;
    unsigned long long blocks = (max_n / 4 + 255) / 256;            // a thread moves four elements per turn
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks, n_tensors), dim3(256), 0, s, T, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)std::sqrt(bc2), (float)eps);
    return 0;
}

}  // namespace lr
