// knn.hip -- mean squared distance to the 3 nearest neighbours (simple-knn distCUDA2) for gfx950.
//
// Replaces SimpleKNN::knn (/root/reference/submodules/simple-knn/simple_knn.cu:186-221):
// Morton ordering + per-box AABB pruning are acceleration structures; the result is the exact
// 3-NN mean (self excluded), written at the point's ORIGINAL index (simple_knn.cu:182).
//
// Pipeline (no host synchronisation; the reference does two blocking D2H copies for the AABB):
//   1. k_aabb       : scene bounding box by wave/LDS reduction + ordered-int atomics
//   2. k_morton     : 30-bit Morton code per point (simple_knn.cu:45-61)
//   3. radix sort   : the same stable 8-bit LSD passes as the rasterizer's binning (32-bit keys)
//   4. k_box_minmax : AABB of every BOX consecutive sorted points
//   5. k_box_knn    : per point, seed the 3 best from +-3 Morton neighbours, then visit only boxes
//                     whose AABB distance can still beat the current 3rd best; the box's points are
//                     staged through LDS so the inner loop is a broadcast ds_read, not a gather.
#include "common.h"
#include <float.h>

namespace lr {

namespace {

constexpr int BOX = 256;   // points per box (one workgroup); the reference uses 1024

struct KnnLayout { size_t aabb, codes_a, codes_b, idx_a, idx_b, hist, sorted_pts, boxes, nP, total; };
inline KnnLayout knn_layout(int P)
{
    KnnLayout L; size_t o = 0; size_t Pz = P > 0 ? (size_t)P : 1;
    size_t nbox = (Pz + BOX - 1) / BOX;
    L.aabb = o;       o += align_up(8 * 4);
    L.nP = o;         o += align_up(4);
    L.codes_a = o;    o += align_up(Pz * 4);
    L.codes_b = o;    o += align_up(Pz * 4);
    L.idx_a = o;      o += align_up(Pz * 4);
    L.idx_b = o;      o += align_up(Pz * 4);
    L.hist = o;       o += sort_hist_bytes((long long)Pz);
    L.sorted_pts = o; o += align_up(Pz * 16);
    L.boxes = o;      o += align_up(nbox * 32);
    L.total = o;
    return L;
}

// order-preserving float <-> uint mapping for atomicMin/atomicMax
__device__ __forceinline__ uint32_t f2o(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float o2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void k_knn_init(uint32_t* aabb, uint32_t* nP, int P)
{
    // The reference reduces with init {0,0,0} (simple_knn.cu:189-200), i.e. the box always contains the origin.
    if (threadIdx.x < 3) aabb[threadIdx.x] = f2o(0.0f);
    else if (threadIdx.x < 6) aabb[threadIdx.x] = f2o(0.0f);
    if (threadIdx.x == 6) *nP = (uint32_t)P;
}

__global__ void __launch_bounds__(256)
k_aabb(int P, const float* __restrict__ pts, uint32_t* aabb)
{
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) { float v = pts[3 * (size_t)i + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], off));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { atomicMin(&aabb[c], f2o(mn[c])); atomicMax(&aabb[3 + c], f2o(mx[c])); }
    }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ void __launch_bounds__(256)
k_morton(int P, const float* __restrict__ pts, const uint32_t* __restrict__ aabb, uint32_t* __restrict__ codes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float mn = o2f(aabb[c]), mx = o2f(aabb[3 + c]);
        const float v = pts[3 * (size_t)i + c];
        const float ext = mx - mn;
        float t = ext > 0.f ? (v - mn) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        code |= prep_morton((uint32_t)(t * ((1 << 10) - 1))) << c;
    }
    codes[i] = code;
}

// gather points into Morton order as float4 (x,y,z,original index bits) and build box AABBs
__global__ void __launch_bounds__(BOX)
k_box_minmax(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order, float4* __restrict__ sorted,
             float* __restrict__ boxes)
{
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int i = blockIdx.x * BOX + threadIdx.x;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    if (i < P) {
        const uint32_t id = order[i];
        const float x = pts[3 * (size_t)id], y = pts[3 * (size_t)id + 1], z = pts[3 * (size_t)id + 2];
        sorted[i] = make_float4(x, y, z, __uint_as_float(id));
        mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], off));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { s_mn[w][c] = mn[c]; s_mx[w][c] = mx[c]; }
    }
    lds_barrier();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        float a = s_mn[0][c], b = s_mx[0][c];
        for (int k = 1; k < BOX / 64; k++) { a = fminf(a, s_mn[k][c]); b = fmaxf(b, s_mx[k][c]); }
        boxes[8 * (size_t)blockIdx.x + c] = a;
        boxes[8 * (size_t)blockIdx.x + 4 + c] = b;
    }
}

__device__ __forceinline__ void update3(float d, float (&best)[3])
{
    // updateKBest<3> (simple_knn.cu:131-145): insertion into an ascending triple
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > d) { float t = best[j]; best[j] = d; d = t; }
    }
}

__device__ __forceinline__ float dist_box(const float* __restrict__ box, float x, float y, float z)
{
    // distBoxPoint (simple_knn.cu:119-129)
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (x < box[0] || x > box[4]) dx = fminf(fabsf(x - box[0]), fabsf(x - box[4]));
    if (y < box[1] || y > box[5]) dy = fminf(fabsf(y - box[1]), fabsf(y - box[5]));
    if (z < box[2] || z > box[6]) dz = fminf(fabsf(z - box[2]), fabsf(z - box[6]));
    return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(BOX)
k_box_knn(int P, const float4* __restrict__ sorted, const float* __restrict__ boxes, float* __restrict__ out)
{
    __shared__ float4 s_pts[BOX];
    const int nbox = (P + BOX - 1) / BOX;
    const int i = blockIdx.x * BOX + threadIdx.x;
    const bool live = i < P;
    float4 me = live ? sorted[i] : make_float4(0, 0, 0, 0);
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    if (live) {
        for (int k = max(0, i - 3); k <= min(P - 1, i + 3); k++) {
            if (k == i) continue;
            const float4 q = sorted[k];
            const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
            update3(dx * dx + dy * dy + dz * dz, best);
        }
    }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;

    for (int b = 0; b < nbox; b++) {
        const float* box = boxes + 8 * (size_t)b;
        // lane-level test as in the reference (simple_knn.cu:166-169); the box is staged only if some
        // point of this workgroup still needs it
        const float d = live ? dist_box(box, me.x, me.y, me.z) : FLT_MAX;
        const bool need = live && !(d > reject || d > best[2]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // as common.h lds_barrier()
        if (!__syncthreads_or(need)) continue;
        const int j = b * BOX + threadIdx.x;
        s_pts[threadIdx.x] = (j < P) ? sorted[j] : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0);
        lds_barrier();
        if (need) {
            const int cnt = min(BOX, P - b * BOX);
            for (int k = 0; k < cnt; k++) {
                if (b * BOX + k == i) continue;
                const float4 q = s_pts[k];
                const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
                update3(dx * dx + dy * dy + dz * dz, best);
            }
        }
        lds_barrier();
    }
    if (live) out[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

size_t dist2_workspace_bytes(int P) { return knn_layout(P).total; }

void launch_dist2(int P, const float* points, float* out, char* ws, hipStream_t s)
{
    const KnnLayout L = knn_layout(P);
    uint32_t* aabb = reinterpret_cast<uint32_t*>(ws + L.aabb);
    uint32_t* nP = reinterpret_cast<uint32_t*>(ws + L.nP);
    uint32_t* codes_a = reinterpret_cast<uint32_t*>(ws + L.codes_a);
    uint32_t* codes_b = reinterpret_cast<uint32_t*>(ws + L.codes_b);
    uint32_t* idx_a = reinterpret_cast<uint32_t*>(ws + L.idx_a);
    uint32_t* idx_b = reinterpret_cast<uint32_t*>(ws + L.idx_b);
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + L.hist);
    float4* sorted = reinterpret_cast<float4*>(ws + L.sorted_pts);
    float* boxes = reinterpret_cast<float*>(ws + L.boxes);
    const int nbox = (P + BOX - 1) / BOX;

    hipLaunchKernelGGL(k_knn_init, dim3(1), dim3(64), 0, s, aabb, nP, P);
    const int rb = min(1024, (P + 255) / 256);
    hipLaunchKernelGGL(k_aabb, dim3(rb), dim3(256), 0, s, P, points, aabb);
    hipLaunchKernelGGL(k_morton, dim3((P + 255) / 256), dim3(256), 0, s, P, points, aabb, codes_a);
    uint32_t *ks, *order;
    radix_sort_pairs(codes_a, codes_b, idx_a, idx_b, /*iota*/ true, nP, P, 30, hist, &ks, &order, s);
    hipLaunchKernelGGL(k_box_minmax, dim3(nbox), dim3(BOX), 0, s, P, points, order, sorted, boxes);
    hipLaunchKernelGGL(k_box_knn, dim3(nbox), dim3(BOX), 0, s, P, sorted, boxes, out);
}

}  // namespace lr
