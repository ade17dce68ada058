// rows.hip -- device-side row surgery for the Gaussian parameter set.
//
// The reference changes the number of Gaussians every 100 iterations with boolean-mask indexing and torch.cat over
// each of the six parameter tensors AND both Adam moments of each, one tensor at a time, re-creating nn.Parameter
// objects and optimizer state entries (R/scene/gaussian_model.py:273-340 _prune_optimizer / prune_points /
// cat_tensors_to_optimizer, :342-403 densify_*): ~40 kernel launches and as many allocations per call, followed by
// torch.cuda.empty_cache().  Here one order-preserving selection moves ALL tensors at once:
//   k_mask_count / k_mask_rank : two-kernel exclusive scan of the byte mask -> rank[i], total (device)
//   k_gather_rows              : blockIdx.y = tensor; selected rows of every tensor are copied to
//                                dst[(dst_row_offset + rank[i])] in source order (prune = compaction into the
//                                other half of a ping-pong buffer; clone/split = append behind the live rows)
//   k_pack_ply                 : the 62-float vertex record of save_ply (:193-208: x y z, zero normals, f_dc and
//                                f_rest channel-major, opacity, scale, rot) built on the device -> one D2H copy
//                                instead of seven copies + a Python list of P tuples.
// Pure byte movement: HBM-bound, coalesced in source order, no atomics.
#include "common.h"

namespace lr {

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;          // mask bytes per workgroup

__device__ __forceinline__ uint32_t rs_block_sum(uint32_t v, uint32_t* s_tmp)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    lds_barrier();
    return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

__global__ void __launch_bounds__(RS_THREADS)
k_mask_count(int P, const uint8_t* __restrict__ mask, uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t s_tmp[4];
    const int base = blockIdx.x * RS_TILE;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const int k = base + i * RS_THREADS + threadIdx.x;
        if (k < P && mask[k] != 0) c++;
    }
    c = rs_block_sum(c, s_tmp);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = c;
}

// rank[i] = number of selected rows before i (valid where mask[i] != 0); total -> *out_count
__global__ void __launch_bounds__(RS_THREADS)
k_mask_rank(int P, const uint8_t* __restrict__ mask, const uint32_t* __restrict__ block_counts,
            uint32_t* __restrict__ rank, int* __restrict__ out_count)
{
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_wave[4];
    uint32_t pre = 0, all = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += RS_THREADS) {
        const uint32_t v = block_counts[i];
        if (i < (int)blockIdx.x) pre += v;
        all += v;
    }
    pre = rs_block_sum(pre, s_tmp);
    if (blockIdx.x == 0) {
        all = rs_block_sum(all, s_tmp);
        if (threadIdx.x == 0) *out_count = (int)all;
    }
    // blocked arrangement: thread t owns RS_ITEMS consecutive rows
    const int base = blockIdx.x * RS_TILE + threadIdx.x * RS_ITEMS;
    uint32_t flag[RS_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) { flag[i] = (base + i < P && mask[base + i] != 0) ? 1u : 0u; sum += flag[i]; }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if ((int)(threadIdx.x & 63) >= off) inc += t;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_wave[w] = inc;
    lds_barrier();
    uint32_t wbase = 0;
    for (int i = 0; i < w; i++) wbase += s_wave[i];
    uint32_t run = pre + wbase + inc - sum;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++)
        if (base + i < P) { rank[base + i] = run; run += flag[i]; }
}

constexpr int MAX_ROW_TENSORS = 32;
struct RowTensors {
    const uint32_t* src[MAX_ROW_TENSORS];
    uint32_t* dst[MAX_ROW_TENSORS];
    uint32_t row_words[MAX_ROW_TENSORS];
    int count;
};

constexpr int GR_ROWS = 128;           // source rows per workgroup
__global__ void __launch_bounds__(RS_THREADS)
k_gather_rows(int P, const uint8_t* __restrict__ mask, const uint32_t* __restrict__ rank, RowTensors T,
              long long dst_row_offset)
{
    __shared__ uint32_t s_rank[GR_ROWS];          // 0xFFFFFFFF = not selected
    const int t = blockIdx.y;
    const int base = blockIdx.x * GR_ROWS;
    for (int r = threadIdx.x; r < GR_ROWS; r += RS_THREADS) {
        const int i = base + r;
        s_rank[r] = (i < P && mask[i] != 0) ? rank[i] : 0xFFFFFFFFu;
    }
    lds_barrier();
    const uint32_t w = T.row_words[t];
    const uint32_t* __restrict__ src = T.src[t] + (size_t)base * w;
    uint32_t* __restrict__ dst = T.dst[t];
    const uint32_t n_words = (uint32_t)min(GR_ROWS, P - base) * w;
    for (uint32_t q = threadIdx.x; q < n_words; q += RS_THREADS) {
        const uint32_t r = q / w, c = q - r * w;
        const uint32_t rk = s_rank[r];
        if (rk != 0xFFFFFFFFu) dst[((size_t)dst_row_offset + rk) * w + c] = src[q];
    }
}

// one thread per (vertex, property)
__global__ void __launch_bounds__(RS_THREADS)
k_pack_ply(int P, int n_rest, const float* __restrict__ xyz, const float* __restrict__ f_dc,
           const float* __restrict__ f_rest, const float* __restrict__ opacity, const float* __restrict__ scaling,
           const float* __restrict__ rotation, float* __restrict__ out)
{
    const int props = 6 + 3 + 3 * n_rest + 1 + 3 + 4;
    const size_t q = (size_t)blockIdx.x * RS_THREADS + threadIdx.x;
    if (q >= (size_t)P * props) return;
    const size_t i = q / props;
    int c = (int)(q - i * props);
    float v;
    if (c < 3) v = xyz[3 * i + c];
    else if (c < 6) v = 0.f;                                            // normals
    else if ((c -= 6) < 3) v = f_dc[3 * i + c];                        // [P,1,3] transposed = the 3 channels
    else if ((c -= 3) < 3 * n_rest) {                                   // channel-major: f_rest_{ch*n_rest + k}
        const int ch = c / n_rest, k = c - ch * n_rest;
        v = f_rest[(i * n_rest + k) * 3 + ch];
    }
    else if ((c -= 3 * n_rest) < 1) v = opacity[i];
    else if ((c -= 1) < 3) v = scaling[3 * i + c];
    else v = rotation[4 * i + (c - 3)];
    out[q] = v;
}

// Densification statistics of one training view in one pass (R/luciddreamer.py:310-311 and
// GaussianModel.add_densification_stats, R/scene/gaussian_model.py:405-407): for every visible Gaussian
//   max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |dL/dmean2D.xy|; denom += 1.
__global__ void __launch_bounds__(RS_THREADS)
k_densify_stats(int P, const int* __restrict__ radii, const float* __restrict__ dL_dmean2D,
                float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ max_radii)
{
    const int i = blockIdx.x * RS_THREADS + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = dL_dmean2D[3 * (size_t)i], gy = dL_dmean2D[3 * (size_t)i + 1];
    max_radii[i] = fmaxf(max_radii[i], (float)r);
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}

}  // namespace

void launch_densify_stats(int P, const int* radii, const float* dL_dmean2D, float* accum, float* denom, float* max_radii,
                          hipStream_t s)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_densify_stats, dim3((P + RS_THREADS - 1) / RS_THREADS), dim3(RS_THREADS), 0, s, P, radii, dL_dmean2D,
                       accum, denom, max_radii);
}

size_t select_workspace_bytes(int P)
{
    const size_t Pz = P > 0 ? (size_t)P : 1;
    const size_t blocks = (Pz + RS_TILE - 1) / RS_TILE;
    return align_up(Pz * 4) + align_up(blocks * 4);
}

int launch_select_rows(int P, const uint8_t* mask, int n_tensors, const void* const* src, void* const* dst,
                       const unsigned* row_bytes, long long dst_row_offset, int* out_count, char* ws, hipStream_t s)
{
    if (n_tensors > MAX_ROW_TENSORS) return -1;
    uint32_t* rank = reinterpret_cast<uint32_t*>(ws);
    uint32_t* block_counts = reinterpret_cast<uint32_t*>(ws + align_up((size_t)(P > 0 ? P : 1) * 4));
    const int nb = (P + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(k_mask_count, dim3(nb), dim3(RS_THREADS), 0, s, P, mask, block_counts);
    hipLaunchKernelGGL(k_mask_rank, dim3(nb), dim3(RS_THREADS), 0, s, P, mask, block_counts, rank, out_count);
    if (n_tensors > 0) {
        RowTensors T;
        T.count = n_tensors;
        for (int t = 0; t < n_tensors; t++) {
            if (row_bytes[t] == 0 || (row_bytes[t] & 3u)) return -2;
            T.src[t] = static_cast<const uint32_t*>(src[t]);
            T.dst[t] = static_cast<uint32_t*>(dst[t]);
            T.row_words[t] = row_bytes[t] / 4;
        }
        for (int t = n_tensors; t < MAX_ROW_TENSORS; t++) { T.src[t] = nullptr; T.dst[t] = nullptr; T.row_words[t] = 1; }
        hipLaunchKernelGGL(k_gather_rows, dim3((P + GR_ROWS - 1) / GR_ROWS, n_tensors), dim3(RS_THREADS), 0, s, P, mask, rank, T,
                           dst_row_offset);
    }
    return 0;
}

void launch_pack_ply(int P, int n_rest, const float* xyz, const float* f_dc, const float* f_rest, const float* opacity,
                     const float* scaling, const float* rotation, float* out, hipStream_t s)
{
    const size_t total = (size_t)P * (17 + 3 * n_rest);
    if (total == 0) return;
    hipLaunchKernelGGL(k_pack_ply, dim3((unsigned)((total + RS_THREADS - 1) / RS_THREADS)), dim3(RS_THREADS), 0, s, P, n_rest, xyz,
                       f_dc, f_rest, opacity, scaling, rotation, out);
}

}  // namespace lr
