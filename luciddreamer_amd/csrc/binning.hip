// binning.hip -- stable LSD radix sort of (key, value) u32 pairs for gfx950 (radix_sort_pairs).
//
// Used by the Morton ordering of simple-knn's replacement (knn.hip; cub::DeviceRadixSort::SortPairs in
// KNN/simple_knn.cu:210-213).  The rasterizer's tile binning no longer sorts globally (tilebin.hip).
//
// Radix pass = 3 kernels: per-block digit histogram -> per-digit scan over blocks -> stable
// scatter.  Ranking inside the scatter is wave-synchronous: 8 x 64-bit __ballot digit matching per
// key, no per-key LDS atomics, deterministic (stable) by construction.  All kernels take their element count
// from device memory.
#include "common.h"

namespace lr {

namespace {

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63;
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

// -------------------------------------------------------------------------------------------
// radix pass, kernel 1: per-block digit histogram.  hist[d * nblocks + b]
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SORT_THREADS)
k_radix_hist(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_dev, int chunk, int shift,
             uint32_t mask, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_hist[RADIX_SIZE];
    const uint32_t n = *n_dev;
    const uint32_t b = blockIdx.x, nb = gridDim.x;
    s_hist[threadIdx.x] = 0;
    lds_barrier();
    const uint64_t beg = (uint64_t)b * chunk;
    uint64_t end = beg + chunk; if (end > n) end = n;
    for (uint64_t i = beg + threadIdx.x; i < end; i += SORT_THREADS) {
        const uint32_t d = (keys[i] >> shift) & mask;
        atomicAdd(&s_hist[d], 1u);
    }
    lds_barrier();
    hist[(size_t)threadIdx.x * nb + b] = s_hist[threadIdx.x];
}

// -------------------------------------------------------------------------------------------
// radix pass, kernel 2: block d scans hist[d][0..nb) (exclusive, in place) and writes total[d].
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_radix_scan(uint32_t* __restrict__ hist, int nb, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_wave[4];
    const int d = blockIdx.x;
    uint32_t* row = hist + (size_t)d * nb;
    // nb <= SORT_MAX_BLOCKS = 1024 -> 4 consecutive entries per thread
    const int base = threadIdx.x * 4;
    uint32_t v[4];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { v[i] = (base + i < nb) ? row[base + i] : 0u; sum += v[i]; }
    // wave inclusive scan of `sum`
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(inc, off);
        if ((int)(threadIdx.x & 63) >= off) inc += t;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_wave[w] = inc;
    lds_barrier();
    uint32_t wbase = 0;
    for (int i = 0; i < w; i++) wbase += s_wave[i];
    uint32_t run = wbase + inc - sum;
#pragma unroll
    for (int i = 0; i < 4; i++) { if (base + i < nb) row[base + i] = run; run += v[i]; }
    if (threadIdx.x == 255) totals[d] = wbase + inc;
}

// -------------------------------------------------------------------------------------------
// radix pass, kernel 3: stable scatter.
// Arrangement inside a 2048-key sub-tile: wave w owns keys [w*512, w*512+512); its i-th step
// (i = 0..7) covers 64 consecutive keys, one per lane -> memory order == (wave, step, lane) order.
// -------------------------------------------------------------------------------------------
template <bool IOTA>
__global__ void __launch_bounds__(SORT_THREADS)
k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                const uint32_t* __restrict__ n_dev, int chunk, int shift, uint32_t mask,
                const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals)
{
    __shared__ uint32_t s_run[RADIX_SIZE];               // running global offset per digit for this block
    __shared__ uint32_t s_cnt[4][RADIX_SIZE];            // per-wave digit counts / running bases
    const uint32_t n = *n_dev;
    const uint32_t b = blockIdx.x, nb = gridDim.x;
    const uint64_t beg = (uint64_t)b * chunk;
    if (beg >= n) return;
    uint64_t end = beg + chunk; if (end > n) end = n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // digit base = exclusive prefix of totals over digits + this block's scanned histogram entry
    {
        uint32_t t = totals[tid];
        uint32_t inc = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == 63) s_cnt[0][w] = inc;
        lds_barrier();
        uint32_t wbase = 0;
        for (int i = 0; i < w; i++) wbase += s_cnt[0][i];
        s_run[tid] = wbase + inc - t + hist[(size_t)tid * nb + b];
        lds_barrier();
    }

    const uint64_t lt = lanemask_lt();
    for (uint64_t tile = beg; tile < end; tile += SORT_TILE) {
        // phase 1: load keys, per-wave digit histogram
#pragma unroll
        for (int i = 0; i < 4; i++) s_cnt[i][tid] = 0;
        lds_barrier();
        uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
        const uint64_t wbeg = tile + (uint64_t)w * (WAVE * SORT_ITEMS);
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; i++) {
            const uint64_t g = wbeg + (uint64_t)i * WAVE + lane;
            const bool ok = g < end;
            key[i] = ok ? keys_in[g] : 0xFFFFFFFFu;
            val[i] = ok ? (IOTA ? (uint32_t)g : vals_in[g]) : 0u;
            if (ok) atomicAdd(&s_cnt[w][(key[i] >> shift) & mask], 1u);
        }
        lds_barrier();
        // phase 2: thread d turns the 4 wave counts of digit d into running bases
        {
            uint32_t run = s_run[tid];
#pragma unroll
            for (int i = 0; i < 4; i++) { uint32_t c = s_cnt[i][tid]; s_cnt[i][tid] = run; run += c; }
            s_run[tid] = run;
        }
        lds_barrier();
        // phase 3: wave-synchronous stable ranking + scatter
        volatile uint32_t* cnt = s_cnt[w];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; i++) {
            const uint64_t g = wbeg + (uint64_t)i * WAVE + lane;
            const bool ok = g < end;
            const uint32_t d = (key[i] >> shift) & mask;
            uint64_t m = __ballot(ok);
#pragma unroll
            for (int bit = 0; bit < RADIX_BITS; bit++) {
                const uint64_t bb = __ballot((d >> bit) & 1u);
                m &= ((d >> bit) & 1u) ? bb : ~bb;
            }
            // m: lanes (valid) holding the same digit as this lane
            if (ok) {
                const uint32_t base = cnt[d];
                const uint32_t rank = __popcll(m & lt);
                const uint32_t pos = base + rank;
                keys_out[pos] = key[i];
                vals_out[pos] = val[i];
            }
            __builtin_amdgcn_wave_barrier();
            if (ok && (m & lt) == 0) cnt[d] += (uint32_t)__popcll(m);   // group leader advances the base
            __builtin_amdgcn_wave_barrier();
        }
        lds_barrier();
    }
}

}  // namespace

void radix_sort_pairs(uint32_t* key_a, uint32_t* key_b, uint32_t* val_a, uint32_t* val_b, bool vals_iota,
                      const uint32_t* n_dev, long long n_bound, int end_bit, uint32_t* hist,
                      uint32_t** keys_out, uint32_t** vals_out, hipStream_t s)
{
    uint32_t *kin = key_a, *kout = key_b, *vin = val_a, *vout = val_b;
    const SortPlan plan = sort_plan(n_bound);
    uint32_t* totals = hist + (size_t)SORT_MAX_BLOCKS * RADIX_SIZE;
    bool first = true;
    for (int shift = 0; shift < end_bit; shift += RADIX_BITS) {
        const int bits = (end_bit - shift) < RADIX_BITS ? (end_bit - shift) : RADIX_BITS;
        const uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(k_radix_hist, dim3(plan.nblocks), dim3(SORT_THREADS), 0, s, kin, n_dev, plan.chunk,
                           shift, mask, hist);
        hipLaunchKernelGGL(k_radix_scan, dim3(RADIX_SIZE), dim3(256), 0, s, hist, plan.nblocks, totals);
        if (first && vals_iota)
            hipLaunchKernelGGL(k_radix_scatter<true>, dim3(plan.nblocks), dim3(SORT_THREADS), 0, s, kin, vin, kout,
                               vout, n_dev, plan.chunk, shift, mask, hist, totals);
        else
            hipLaunchKernelGGL(k_radix_scatter<false>, dim3(plan.nblocks), dim3(SORT_THREADS), 0, s, kin, vin, kout,
                               vout, n_dev, plan.chunk, shift, mask, hist, totals);
        first = false;
        uint32_t* t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    *keys_out = kin;
    *vals_out = vin;
}

}  // namespace lr
