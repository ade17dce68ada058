// binning.hip -- tile binning for gfx950: depth sort of Gaussians, tile-count scan, instance
// emission, stable tile partition, per-tile ranges.
//
// Replaces cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:278), duplicateWithKeys (:70-111),
// cub::DeviceRadixSort::SortPairs on 64-bit keys (:304-309) and identifyTileRanges (:116-138).
//
// Required invariant (what the reference's stable 64-bit sort produces): inside every tile the
// instances are ordered by (float bits of view depth ascending, Gaussian index ascending).
// MI355X design: instead of sorting R tile instances on a 46-bit key (6 radix passes over
// 12 B/instance), sort the P Gaussians ONCE by their 32-bit depth key (stable, value = index),
// emit instances in that order, and stably partition the R instances by tile id only
// (ceil(log2(tiles)) bits -> 2 passes over 8 B/instance).  Stability of both sorts gives exactly
// the reference order.  All kernels take their element count from device memory, so the whole
// pipeline can run without a host round trip (async mode of lr_forward).
//
// Radix pass = 3 kernels: per-block digit histogram -> per-digit scan over blocks -> stable
// scatter.  Ranking inside the scatter is wave-synchronous: 8 x 64-bit __ballot digit matching per
// key, no per-key LDS atomics, deterministic (stable) by construction.
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
    __syncthreads();
    const uint64_t beg = (uint64_t)b * chunk;
    uint64_t end = beg + chunk; if (end > n) end = n;
    for (uint64_t i = beg + threadIdx.x; i < end; i += SORT_THREADS) {
        const uint32_t d = (keys[i] >> shift) & mask;
        atomicAdd(&s_hist[d], 1u);
    }
    __syncthreads();
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
    __syncthreads();
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
        __syncthreads();
        uint32_t wbase = 0;
        for (int i = 0; i < w; i++) wbase += s_cnt[0][i];
        s_run[tid] = wbase + inc - t + hist[(size_t)tid * nb + b];
        __syncthreads();
    }

    const uint64_t lt = lanemask_lt();
    for (uint64_t tile = beg; tile < end; tile += SORT_TILE) {
        // phase 1: load keys, per-wave digit histogram
#pragma unroll
        for (int i = 0; i < 4; i++) s_cnt[i][tid] = 0;
        __syncthreads();
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
        __syncthreads();
        // phase 2: thread d turns the 4 wave counts of digit d into running bases
        {
            uint32_t run = s_run[tid];
#pragma unroll
            for (int i = 0; i < 4; i++) { uint32_t c = s_cnt[i][tid]; s_cnt[i][tid] = run; run += c; }
            s_run[tid] = run;
        }
        __syncthreads();
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
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------
// tile-count scan in depth order (2 kernels)
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_reduce_sum(uint32_t v, uint32_t* s_tmp)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += s_tmp[i];
    __syncthreads();
    return t;
}

// -------------------------------------------------------------------------------------------
// order-preserving compaction of the Gaussians that emit at least one instance (2 kernels).  Camera paths see
// a small part of a scene (~10 % in the rotate360 bench), and culled Gaussians would otherwise ride through
// all four depth-sort passes, the tile-count scan and the emission kernel.  Index order is kept, so the
// stable depth sort still breaks ties by Gaussian index exactly like the reference.  The reference's
// num_rendered (sum of rectangle areas over ALL Gaussians) is totalled here as well.
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SCAN_THREADS)
k_compact_reduce(int P, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ tiles_ref,
                 uint2* __restrict__ block_sums)
{
    __shared__ uint32_t s_tmp[4];
    const int base = blockIdx.x * SCAN_TILE;
    uint32_t cnt = 0, sum_ref = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const int k = base + i * SCAN_THREADS + threadIdx.x;
        if (k < P) { cnt += tiles_touched[k] != 0 ? 1u : 0u; sum_ref += tiles_ref[k]; }
    }
    cnt = block_reduce_sum(cnt, s_tmp);
    sum_ref = block_reduce_sum(sum_ref, s_tmp);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = make_uint2(cnt, sum_ref);
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_compact_write(int P, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ depth_key,
                const uint2* __restrict__ block_sums, uint32_t* __restrict__ ckey, uint32_t* __restrict__ cidx,
                uint32_t* __restrict__ vis_list, GeomHeader* hdr)
{
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_wave[4];
    const bool last_block = blockIdx.x == gridDim.x - 1;
    uint32_t pre = 0, ref_total = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += SCAN_THREADS) {
        const uint2 v = block_sums[i];
        if (i < (int)blockIdx.x) pre += v.x;
        ref_total += v.y;
    }
    pre = block_reduce_sum(pre, s_tmp);
    if (last_block) ref_total = block_reduce_sum(ref_total, s_tmp);
    // blocked arrangement keeps index order: thread t owns SCAN_ITEMS consecutive Gaussians
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t flag[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const int k = base + i;
        flag[i] = (k < P && tiles_touched[k] != 0) ? 1u : 0u;
        sum += flag[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(inc, off);
        if ((int)(threadIdx.x & 63) >= off) inc += t;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int i = 0; i < w; i++) wbase += s_wave[i];
    uint32_t run = pre + wbase + inc - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (flag[i]) { ckey[run] = depth_key[base + i]; cidx[run] = (uint32_t)(base + i); vis_list[run] = (uint32_t)(base + i); run++; }
    }
    if (last_block && threadIdx.x == SCAN_THREADS - 1) {
        hdr->num_compact = run;
        hdr->num_rendered = ref_total;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_reduce(const GeomHeader* __restrict__ hdr, const uint32_t* __restrict__ order,
              const uint32_t* __restrict__ tiles_touched, uint2* __restrict__ block_sums)
{
    __shared__ uint32_t s_tmp[4];
    const int n = (int)hdr->num_compact;
    const int base = blockIdx.x * SCAN_TILE;
    uint32_t sum = 0;
    if (base < n) {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const int k = base + i * SCAN_THREADS + threadIdx.x;
            if (k < n) sum += tiles_touched[order[k]];
        }
    }
    sum = block_reduce_sum(sum, s_tmp);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = make_uint2(sum, 0u);
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_write(const GeomHeader* __restrict__ hdr_in, const uint32_t* __restrict__ order,
             const uint32_t* __restrict__ tiles_touched, const uint2* __restrict__ block_sums,
             uint32_t* __restrict__ offsets, GeomHeader* hdr)
{
    __shared__ uint32_t s_tmp[4];
    __shared__ uint32_t s_wave[4];
    const int P = (int)hdr_in->num_compact;          // ranks beyond the compacted count do not exist
    const bool last_block = blockIdx.x == gridDim.x - 1;
    if (!last_block && (int)(blockIdx.x * SCAN_TILE) >= P) return;
    // prefix of the preceding blocks' sums
    uint32_t pre = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_THREADS) pre += block_sums[i].x;
    pre = block_reduce_sum(pre, s_tmp);

    // blocked arrangement: thread t owns SCAN_ITEMS consecutive ranks
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const int k = base + i;
        v[i] = (k < P) ? tiles_touched[order[k]] : 0u;
        sum += v[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(inc, off);
        if ((int)(threadIdx.x & 63) >= off) inc += t;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int i = 0; i < w; i++) wbase += s_wave[i];
    uint32_t run = pre + wbase + inc - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const int k = base + i;
        if (k < P) offsets[k] = run;
        run += v[i];
    }
    if (last_block && threadIdx.x == SCAN_THREADS - 1) {
        const uint32_t total = run;
        hdr->num_instances = total;
        const bool over = (hdr->capacity != 0 && total > hdr->capacity);
        hdr->overflow = over ? 1u : 0u;
        if (over) hdr->sticky_overflow = 1u;
        hdr->num_sorted = over ? hdr->capacity : total;
    }
}

// -------------------------------------------------------------------------------------------
// instance emission in depth order.  One wave handles 64 consecutive depth ranks; rectangles of
// up to SMALL tiles are written by their own lane, larger ones by the whole wave (coalesced).
// Key = tile id (y * gx + x), value = Gaussian index  (rasterizer_impl.cu:85-109).
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_rect_dev(float px, float py, int radius, int gx, int gy,
                                              int& minx, int& miny, int& maxx, int& maxy)
{
    minx = min(gx, max(0, (int)((px - radius) / TILE_X)));
    miny = min(gy, max(0, (int)((py - radius) / TILE_Y)));
    maxx = min(gx, max(0, (int)((px + radius + TILE_X - 1) / TILE_X)));
    maxy = min(gy, max(0, (int)((py + radius + TILE_Y - 1) / TILE_Y)));
}

__global__ void __launch_bounds__(256)
k_emit(int P, int gx, int gy, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
       const uint32_t* __restrict__ tiles_touched, const GaussRec* __restrict__ rec,
       const int* __restrict__ radii, GeomHeader* __restrict__ hdr, uint32_t bin_bound,
       uint32_t* __restrict__ inst_keys, uint32_t* __restrict__ inst_vals, uint32_t* __restrict__ goff)
{
    constexpr uint32_t SMALL = 20;      // rectangles up to this many tiles are walked by their own lane
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt();
    const uint32_t cap = hdr->capacity != 0 ? hdr->capacity : 0xFFFFFFFFu;
    if (k == 0) hdr->bin_bound = bin_bound;
    P = (int)hdr->num_compact;                     // depth ranks that exist (compacted, every one emits)
    if ((int)(blockIdx.x * blockDim.x) >= P) return;
    uint32_t idx = 0, tt = 0, off = 0, area = 0;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, qmax = 0.f, r_c = 0.f, r_a = 0.f;
    if (k < P) {
        idx = order[k];
        tt = tiles_touched[idx];                   // instances to emit (after exact tile culling, preprocess.hip)
        if (tt != 0) {
            off = offsets[k];
            goff[idx] = off;                           // where this Gaussian's instance slots start (emission order)
            const float4* g = reinterpret_cast<const float4*>(rec + idx);
            const float4 q0 = g[0];
            const float4 q1 = g[1];
            const float4 q2 = g[2];
            mx = q0.x; my = q0.y; ca = q0.z; cb = q0.w; cc = q1.x;
            qmax = q2.z;                               // computed once in k_preprocess
            r_c = -cb / cc; r_a = -cb / ca;
            tile_rect_dev(mx, my, radii[idx], gx, gy, minx, miny, maxx, maxy);
            area = (uint32_t)(maxx - minx) * (uint32_t)(maxy - miny);
        }
    }
    const int rw = maxx - minx;
    const bool culled = area <= CULL_MAX_TILES;   // same rule as the count in k_preprocess
    if (tt != 0 && area <= SMALL) {
        uint32_t o = off;
        for (int y = miny; y < maxy; y++)
            for (int x = minx; x < maxx; x++) {
                if (tile_hit(mx, my, ca, cb, cc, r_c, r_a, qmax, x, y)) {
                    if (o < cap) { inst_keys[o] = (uint32_t)(y * gx + x); inst_vals[o] = idx; }
                    o++;
                }
            }
    }
    uint64_t big = __ballot(tt != 0 && area > SMALL);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t b_idx = __shfl(idx, src), b_area = __shfl(area, src);
        uint32_t b_off = __shfl(off, src);
        const int b_minx = __shfl(minx, src), b_miny = __shfl(miny, src), b_rw = __shfl(rw, src);
        const bool b_culled = __shfl((int)culled, src) != 0;
        const float b_mx = __shfl(mx, src), b_my = __shfl(my, src), b_ca = __shfl(ca, src), b_cb = __shfl(cb, src);
        const float b_cc = __shfl(cc, src), b_qmax = __shfl(qmax, src);
        const float b_rc = __shfl(r_c, src), b_ra = __shfl(r_a, src);
        for (uint32_t j0 = 0; j0 < b_area; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool hit = j < b_area;
            int y = 0, x = 0;
            if (hit) {
                y = b_miny + (int)(j / (uint32_t)b_rw); x = b_minx + (int)(j % (uint32_t)b_rw);
                if (b_culled) hit = tile_hit(b_mx, b_my, b_ca, b_cb, b_cc, b_rc, b_ra, b_qmax, x, y);
            }
            const uint64_t m = __ballot(hit);
            if (hit) {
                const uint32_t o = b_off + (uint32_t)__popcll(m & lt);
                if (o < cap) { inst_keys[o] = (uint32_t)(y * gx + x); inst_vals[o] = b_idx; }
            }
            b_off += (uint32_t)__popcll(m);
        }
    }
}

// -------------------------------------------------------------------------------------------
// per-tile [begin,end) in the tile-sorted instance list (rasterizer_impl.cu:116-138); `ranges`
// is zeroed beforehand (rasterizer_impl.cu:311).
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_ranges(const uint32_t* __restrict__ keys, const GeomHeader* __restrict__ hdr, uint2* __restrict__ ranges)
{
    const uint32_t n = hdr->num_sorted;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t cur = keys[i];
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = keys[i - 1];
        if (cur != prev) { ranges[prev].y = i; ranges[cur].x = i; }
    }
    if (i == n - 1) ranges[cur].y = n;
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

void launch_compact(int P, const uint32_t* tiles_touched, uint32_t* tiles_ref, const uint32_t* depth_key,
                    uint2* block_sums, uint32_t* ckey, uint32_t* cidx, GeomHeader* hdr, hipStream_t s)
{
    const int nb = (P + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_compact_reduce, dim3(nb), dim3(SCAN_THREADS), 0, s, P, tiles_touched, tiles_ref, block_sums);
    // tiles_ref is dead once k_compact_reduce has totalled it: the same array then receives the index-ordered
    // list of emitting Gaussians (the sort below destroys cidx), which the per-Gaussian backward walks
    hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(SCAN_THREADS), 0, s, P, tiles_touched, depth_key, block_sums,
                       ckey, cidx, tiles_ref, hdr);
}

void launch_scan_tiles(int P, const uint32_t* order, const uint32_t* tiles_touched, const uint32_t* tiles_ref,
                       uint32_t* offsets, uint2* block_sums, GeomHeader* hdr, hipStream_t s)
{
    (void)tiles_ref;
    const int nb = (P + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, s, hdr, order, tiles_touched, block_sums);
    hipLaunchKernelGGL(k_scan_write, dim3(nb), dim3(SCAN_THREADS), 0, s, hdr, order, tiles_touched, block_sums,
                       offsets, hdr);
}

void launch_emit(int P, int gx, int gy, const uint32_t* order, const uint32_t* offsets,
                 const uint32_t* tiles_touched, const GaussRec* rec, const int* radii, GeomHeader* hdr,
                 uint32_t bin_bound, uint32_t* inst_keys, uint32_t* inst_gid, uint32_t* goff, hipStream_t s)
{
    hipLaunchKernelGGL(k_emit, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, gy, order, offsets, tiles_touched,
                       rec, radii, hdr, bin_bound, inst_keys, inst_gid, goff);
}

void launch_ranges(const uint32_t* sorted_keys, const GeomHeader* hdr, long long n_bound, int num_tiles,
                   uint2* ranges, hipStream_t s)
{
    (void)hipMemsetAsync(ranges, 0, (size_t)num_tiles * sizeof(uint2), s);    // errors surface at the caller's hipGetLastError
    if (n_bound <= 0) return;
    hipLaunchKernelGGL(k_ranges, dim3((unsigned)((n_bound + 255) / 256)), dim3(256), 0, s, sorted_keys, hdr, ranges);
}

}  // namespace lr
