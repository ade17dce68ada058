// tilebin.hip -- tile binning for gfx950: from per-Gaussian tile counts to depth-ordered per-tile instance lists.
//
// Replaces cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:278), duplicateWithKeys (:70-111),
// cub::DeviceRadixSort::SortPairs on 64-bit [tile | depth] keys (:304-309) and identifyTileRanges (:116-138).
//
// Required invariant (what the reference's stable 64-bit sort produces): inside every tile the instances are
// ordered by (float bits of view depth ascending, Gaussian index ascending).
//
// MI355X design (round 2; the round-1 pipeline sorted the Gaussians globally by depth with 4 radix passes, emitted
// in depth order and stably partitioned by tile with 2 more: 21 launches, 0.19 ms per C3 view for ~47 MB of bytes):
//
//   k_compact_write (chunk sums: k_preprocess)   one scan over the P Gaussians gives, in INDEX order, the list of emitting
//                                        Gaussians (vis_list), each one's first instance slot (offsets, by rank) and
//                                        every count of the header.  Instance slots are therefore contiguous per
//                                        Gaussian and ascending with the Gaussian index.
//   k_part<COUNT>                        a few hundred large workgroups each walk a contiguous chunk of vis_list -- the
//                                        set bits of the tile mask preprocess left per Gaussian (common.h HitRec; round 3:
//                                        the walks used to repeat the exact tile test of every rectangle tile) -- and
//                                        histogram the hits over the partition bins in LDS (one bin per tile up to 16384
//                                        tiles; 2^s neighbouring tiles per bin beyond that).  Integer LDS atomics: counts
//                                        are order-free.
//                                        Each workgroup then reserves its range inside every bin it found entries for with
//                                        ONE returning atomic per (workgroup, non-empty bin) on the bin's cursor -- zeroed by
//                                        the scan kernel before it, holding the bin's total after it.  (Rounds 2-4 wrote
//                                        per-workgroup counts and ran a kernel, k_part_scan1, over them: a launch and 4.9 us
//                                        for what the count kernel's tail does in ~3; lr_tune_set("part_scan", 1) still
//                                        selects it.  Which workgroup gets which range now depends on arrival -- and does
//                                        not matter, see the sort below.)
//   k_part<SCATTER>                      every workgroup first scans the bin totals itself (bin start; workgroup 0 also
//                                        publishes the per-tile ranges: no identifyTileRanges pass, no memset); then
//                                        the same walk again; every hit takes the next free position of its bin from
//                                        an LDS cursor and stores ONE 64-bit word [sub-tile | depth bits | slot].
//                                        The order inside a bin at this point is arbitrary -- and irrelevant:
//   k_tile_sort_small / _large           every bin's words are sorted in LDS: <= 256 entries by one wave (bitonic network up
//                                        to 64, bucket sort above), <= 1024 by a workgroup and <= 4096 by a larger one
//                                        (bucket sort), beyond that the network (in LDS up to 16384 entries when the
//                                        launch has the room, else in global memory).
//                                        The word is a TOTAL order -- depth bits, then slot, and slots ascend with the
//                                        Gaussian index -- so the result is exactly the reference's list, bit-for-bit
//                                        repeatable, whatever order the scatter produced.
//
// 6 launches instead of 21 (7 for the whole forward), no global depth sort (the depth order is only ever needed inside a tile), no global
// atomics per instance (one per workgroup and bin), nothing that depends on a host round trip: every kernel takes its counts
// from the device-side header.
#include <mutex>
#include "common.h"

namespace lr {

namespace {

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63;
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

__device__ __forceinline__ uint32_t block_reduce_sum(uint32_t v, uint32_t* s_tmp)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    lds_barrier();
    uint32_t t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += s_tmp[i];
    lds_barrier();
    return t;
}

// -------------------------------------------------------------------------------------------
// compaction + instance offsets, index order.  block_sums[b] = {emitting Gaussians, instances, reference rectangle
// areas} of chunk b, accumulated by k_preprocess (preprocess.hip).
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SCAN_THREADS)
k_compact_write(int P, const uint32_t* __restrict__ tiles_touched, const uint4* __restrict__ block_sums,
                uint32_t* __restrict__ vis_list, uint32_t* __restrict__ offsets, GeomHeader* hdr, uint32_t capacity,
                uint32_t* __restrict__ log_slot, uint32_t log_tag, uint32_t* __restrict__ zero_words, uint32_t n_zero,
                int reset_sticky)
{
    __shared__ uint32_t s_tmp[4];
    // the per-bin cursors of the partition (k_part<0> reserves its ranges on them with atomics): zero before its launch
    for (uint32_t i = blockIdx.x * SCAN_THREADS + threadIdx.x; i < n_zero; i += gridDim.x * SCAN_THREADS) zero_words[i] = 0u;
    __shared__ uint32_t s_wc[4], s_wi[4];
    const bool last_block = blockIdx.x == gridDim.x - 1;
    uint32_t pre_c = 0, pre_i = 0, ref_total = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += SCAN_THREADS) {
        const uint4 v = block_sums[i];
        if (i < (int)blockIdx.x) { pre_c += v.x; pre_i += v.y; }
        ref_total += v.z;
    }
    pre_c = block_reduce_sum(pre_c, s_tmp);
    pre_i = block_reduce_sum(pre_i, s_tmp);
    if (last_block) ref_total = block_reduce_sum(ref_total, s_tmp);
    // blocked arrangement keeps index order: thread t owns SCAN_ITEMS consecutive Gaussians
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t tt[SCAN_ITEMS];
    uint32_t sum_c = 0, sum_i = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const int k = base + i;
        tt[i] = (k < P) ? tiles_touched[k] : 0u;
        sum_c += tt[i] != 0 ? 1u : 0u;
        sum_i += tt[i];
    }
    uint32_t inc_c = sum_c, inc_i = sum_i;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = __shfl_up(inc_c, off), b = __shfl_up(inc_i, off);
        if ((int)(threadIdx.x & 63) >= off) { inc_c += a; inc_i += b; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) { s_wc[w] = inc_c; s_wi[w] = inc_i; }
    lds_barrier();
    uint32_t wb_c = 0, wb_i = 0;
    for (int i = 0; i < w; i++) { wb_c += s_wc[i]; wb_i += s_wi[i]; }
    uint32_t run_c = pre_c + wb_c + inc_c - sum_c;
    uint32_t run_i = pre_i + wb_i + inc_i - sum_i;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (tt[i]) {
            vis_list[run_c] = (uint32_t)(base + i);
            offsets[run_c] = run_i;
            run_c++;
            run_i += tt[i];
        }
    }
    if (last_block && threadIdx.x == SCAN_THREADS - 1) {
        // EVERY per-call word of the header is written here (the buffer is the caller's, fresh memory per call: nothing is
        // zeroed up front); the prefilter trap was left behind the chunk sums by k_preprocess
        const uint32_t total = run_i;
        const bool over = (capacity != 0 && total > capacity);
        const uint32_t trap = reinterpret_cast<const uint32_t*>(block_sums)[4 * (size_t)gridDim.x];
        hdr->num_rendered = ref_total;          // the reference's count (sum of rectangle areas)
        hdr->overflow = over ? 1u : 0u;
        hdr->prefilter_trap = trap;
        hdr->capacity = capacity;
        hdr->P = (uint32_t)P;
        hdr->num_sorted = over ? capacity : total;
        hdr->num_instances = total;             // after exact tile culling: what is binned
        hdr->bin_bound = capacity != 0 ? capacity : total;   // what the binning buffer is laid out for
        hdr->num_compact = run_c;
        hdr->n_seg = 0u;                        // the blend forward reserves its list segments on it
        hdr->bwd_uncovered = 0u;
        // sticky across the views of a step on this buffer; the step's first view starts it (api.hip views_core)
        if (over || reset_sticky) hdr->sticky_overflow = over ? 1u : 0u;
        if (log_slot != nullptr) {
            // the library's forward log (host-visible memory, api.hip ForwardLog): the header words first, the tag last,
            // each with system scope -- the host spins on the tag instead of waiting for a copy and an event
            const uint32_t w[9] = { ref_total, over ? 1u : 0u, trap, capacity, (uint32_t)P, over ? capacity : total, total,
                                    capacity != 0 ? capacity : total, run_c };
#pragma unroll
            for (int i = 0; i < 9; i++) __hip_atomic_store(log_slot + 1 + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(log_slot, log_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// -------------------------------------------------------------------------------------------
// the walk over a chunk of emitting Gaussians: f(tile, slot, gid, depth_bits) for every (Gaussian, tile) pair that
// passes the exact tile test, slots consecutive per Gaussian in (y, x) order (rasterizer_impl.cu:85-109).
// Rectangles of up to SMALL tiles are walked by their own lane, larger ones by the whole wave.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_rect_dev(float px, float py, int radius, int gx, int gy,
                                              int& minx, int& miny, int& maxx, int& maxy)
{
    minx = min(gx, max(0, (int)((px - radius) / TILE_X)));
    miny = min(gy, max(0, (int)((py - radius) / TILE_Y)));
    maxx = min(gx, max(0, (int)((px + radius + TILE_X - 1) / TILE_X)));
    maxy = min(gy, max(0, (int)((py + radius + TILE_Y - 1) / TILE_Y)));
}

// workgroups of the partition kernels and the compact ranks [beg, end) of workgroup b: a pure function of the
// device-side count, evaluated identically by the count, scan and scatter kernels
__device__ __forceinline__ int part_active_blocks(uint32_t V)
{
    const uint32_t nb = (V + PART_MIN_GAUSS - 1) / PART_MIN_GAUSS;
    return (int)(nb < 1u ? 1u : (nb > (uint32_t)PART_BLOCKS_MAX ? (uint32_t)PART_BLOCKS_MAX : nb));
}
__device__ __forceinline__ void part_chunk(uint32_t V, int nb, int b, uint32_t& beg, uint32_t& end)
{
    uint32_t per = (V + (uint32_t)nb - 1) / (uint32_t)nb;
    per = (per + 63u) & ~63u;
    beg = (uint32_t)b * per;
    end = beg + per;
    if (beg > V) beg = V;
    if (end > V) end = V;
}

// LDS index of a bin's counter: one pad word per 16 bins.  bin_prefix_to_lds gives every thread 16 CONSECUTIVE bins; without
// the pad the lanes of a wave would hit only two of the 32 banks (16-word stride).
__device__ __forceinline__ int bin_slot(int bin) { return bin + (bin >> 4); }

template <class F>
__device__ __forceinline__ void walk_chunk(uint32_t beg, uint32_t end, int gx, int gy, int own_max,
                                           const uint32_t* __restrict__ vis_list, const uint32_t* __restrict__ offsets,
                                           const uint4* __restrict__ hitrec, const GaussRec* __restrict__ rec,
                                           const int* __restrict__ radii, F f)
{
    // Masked entries (common.h HitRec: rectangles of up to 64 tiles, the bulk): the set bits ARE the instances, in slot
    // order.  Up to own_max of them are walked by their own lane, more by the whole wave (one round: lane j takes bit j).
    // Unmasked entries (larger rectangles) are walked by the whole wave from the record, testing the culled ones again.
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt();
    for (uint32_t k0 = beg; k0 < end; k0 += PART_THREADS) {          // beg, end are multiples of 64 or the list end
        const uint32_t k = k0 + threadIdx.x;
        uint32_t idx = 0, off = 0, geo = 0, dbits = 0, n_own = 0;
        uint64_t mask = 0ull;
        bool valid = false;
        if (k < end) {
            idx = vis_list[k];
            off = offsets[k];
            const uint4 h = hitrec[idx];
            mask = (uint64_t)h.x | ((uint64_t)h.y << 32);
            geo = h.z; dbits = h.w;                                   // view depth > 0.2: bit 31 clear, bit order == float order
            valid = true;
        }
        const bool masked = valid && geo != 0u;
        const int minx = (int)(geo & 0xfffu), miny = (int)((geo >> 12) & 0xfffu);
        const uint32_t rw = geo >> 24;
        const uint32_t rcp = masked ? 65536u / rw + 1u : 0u;          // (j * rcp) >> 16 == j / rw for j < 64, rw <= 64
        if (masked) n_own = (uint32_t)__popcll(mask);
        if (masked && n_own <= (uint32_t)own_max) {
            uint64_t m = mask;
            uint32_t o = off;
            while (m) {
                const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
                m &= m - 1ull;
                const uint32_t ry = (j * rcp) >> 16, rx = j - ry * rw;
                f((uint32_t)((miny + (int)ry) * gx + minx + (int)rx), o, idx, dbits);
                o++;
            }
        }
        uint64_t coop = __ballot(masked && n_own > (uint32_t)own_max);
        while (coop) {
            const int src = __ffsll((long long)coop) - 1;
            coop &= coop - 1;
            const uint32_t b_lo = __shfl((uint32_t)mask, src), b_hi = __shfl((uint32_t)(mask >> 32), src);
            const uint64_t b_mask = (uint64_t)b_lo | ((uint64_t)b_hi << 32);
            const uint32_t b_geo = __shfl(geo, src), b_rcp = __shfl(rcp, src);
            const uint32_t b_idx = __shfl(idx, src), b_db = __shfl(dbits, src), b_off = __shfl(off, src);
            if ((b_mask >> lane) & 1ull) {
                const uint32_t b_rw = b_geo >> 24;
                const uint32_t ry = ((uint32_t)lane * b_rcp) >> 16, rx = (uint32_t)lane - ry * b_rw;
                f((uint32_t)(((int)((b_geo >> 12) & 0xfffu) + (int)ry) * gx + (int)(b_geo & 0xfffu) + (int)rx),
                  b_off + (uint32_t)__popcll(b_mask & lt), b_idx, b_db);
            }
        }
        uint64_t big = __ballot(valid && geo == 0u);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const uint32_t b_idx = __shfl(idx, src), b_db = __shfl(dbits, src);
            uint32_t b_off = __shfl(off, src);
            // wave-uniform loads of the one record
            const float4* g = reinterpret_cast<const float4*>(rec + b_idx);
            const float4 q0 = g[0], q1 = g[1], q2 = g[2];
            const float b_mx = q0.x, b_my = q0.y, b_ca = q0.z, b_cb = q0.w, b_cc = q1.x, b_qmax = q2.z;
            const float b_rc = -b_cb / b_cc, b_ra = -b_cb / b_ca;
            int b_minx, b_miny, b_maxx, b_maxy;
            tile_rect_dev(b_mx, b_my, radii[b_idx], gx, gy, b_minx, b_miny, b_maxx, b_maxy);
            const int b_rw = b_maxx - b_minx;
            const uint32_t b_area = (uint32_t)b_rw * (uint32_t)(b_maxy - b_miny);
            const bool b_culled = b_area <= CULL_MAX_TILES;           // same rule as the count in k_preprocess
            for (uint32_t j0 = 0; j0 < b_area; j0 += 64) {
                const uint32_t j = j0 + lane;
                bool hit = j < b_area;
                int y = 0, x = 0;
                if (hit) {
                    y = b_miny + (int)(j / (uint32_t)b_rw); x = b_minx + (int)(j % (uint32_t)b_rw);
                    if (b_culled) hit = tile_hit(b_mx, b_my, b_ca, b_cb, b_cc, b_rc, b_ra, b_qmax, x, y);
                }
                const uint64_t m = __ballot(hit);
                if (hit) f((uint32_t)(y * gx + x), b_off + (uint32_t)__popcll(m & lt), b_idx, b_db);
                b_off += (uint32_t)__popcll(m);
            }
        }
    }
}

// Exclusive prefix over the bins, evaluated by every scatter workgroup for itself, into its LDS cursor array
// (cursor = bin start + this workgroup's base inside the bin); workgroup 0 also publishes bin_start and the per-tile
// ranges (a bin is a tile when sub_shift == 0; otherwise the ranges are zeroed here and filled by the per-bin sort) and
// builds the queue of the bins k_tile_sort_large takes (more than 1024 entries).
// Global memory is touched with consecutive lanes on consecutive words only (the totals are staged through the cursor
// array; a thread then scans 16 CONSECUTIVE bins out of LDS, conflict-free thanks to bin_slot's padding): with each thread
// loading its own 64-byte run straight from global memory this prologue cost 12 us of the kernel's 32.
__device__ __forceinline__ void bin_prefix_to_lds(int bins, int num_tiles, int sub_shift, const uint32_t* __restrict__ bin_total,
                                                  const uint32_t* __restrict__ row, uint32_t* s_bin, bool publish,
                                                  uint32_t* __restrict__ bin_start, uint2* __restrict__ ranges,
                                                  uint32_t* __restrict__ big_queue)
{
    constexpr int PER = PART_BINS_MAX / PART_THREADS;     // 16 consecutive bins per thread in the scan phase
    __shared__ uint32_t s_wave[PART_THREADS / 64];
    __shared__ uint32_t s_total;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < bins; i += PART_THREADS) s_bin[bin_slot(i)] = bin_total[i];
    lds_barrier();
    const int base = threadIdx.x * PER;
    uint32_t v[PER];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) { v[i] = (base + i < bins) ? s_bin[bin_slot(base + i)] : 0u; sum += v[i]; }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    lds_barrier();
    uint32_t run = inc - sum;
    for (int j = 0; j < w; j++) run += s_wave[j];
    if (threadIdx.x == PART_THREADS - 1) s_total = run + sum;
    uint32_t q = 0, nbig = 0;
    if (publish) {
        // the queue of bins for k_tile_sort_large, in bin order, by a second block scan (no atomics: returning global
        // atomics on one word cost ~45 ns each on this part, and a dense 512^2 view queues every tile)
#pragma unroll
        for (int i = 0; i < PER; i++) nbig += v[i] > (uint32_t)TSORT_GROUP_LDS ? 1u : 0u;
        uint32_t binc = nbig;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(binc, off);
            if (lane >= off) binc += t;
        }
        lds_barrier();                                  // s_wave is reused
        if (lane == 63) s_wave[w] = binc;
        lds_barrier();
        q = binc - nbig;
        for (int j = 0; j < w; j++) q += s_wave[j];
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int bin = base + i;
        if (bin < bins) {
            s_bin[bin_slot(bin)] = run;                   // exclusive start of the bin
            if (publish && v[i] > (uint32_t)TSORT_GROUP_LDS) big_queue[1 + q++] = (uint32_t)bin;
        }
        run += v[i];
    }
    if (publish && threadIdx.x == PART_THREADS - 1) big_queue[0] = q;
    lds_barrier();
    // coalesced pass: starts (and the next bin's start = this bin's end) out of LDS, this workgroup's row from global
    for (int i0 = 0; i0 < bins; i0 += PART_THREADS) {
        const int i = i0 + threadIdx.x;
        uint32_t st = 0, en = 0, r = 0;
        if (i < bins) {
            st = s_bin[bin_slot(i)];
            en = (i + 1 < bins) ? s_bin[bin_slot(i + 1)] : s_total;
            r = row[i];
        }
        lds_barrier();                                  // every start of this sweep is read before any is advanced
        if (i < bins) {
            s_bin[bin_slot(i)] = st + r;
            if (publish) {
                bin_start[i] = st;
                if (sub_shift == 0) ranges[i] = en > st ? make_uint2(st, en) : make_uint2(0u, 0u);   // empty: (0,0), :311
            }
        }
    }
    if (publish && sub_shift != 0)
        for (int t = threadIdx.x; t < num_tiles; t += PART_THREADS) ranges[t] = make_uint2(0u, 0u);
}

// MODE 0: count (per-workgroup bin histogram -> part_hist[b][bin]; also records inst_gid[slot])
// MODE 1: scatter (64-bit words into their bins)
template <int MODE>
__global__ void __launch_bounds__(PART_THREADS)
k_part(int gx, int gy, int bins, int sub_shift, int slot_bits, int own_max, const uint32_t* __restrict__ vis_list,
       const uint32_t* __restrict__ offsets, const uint4* __restrict__ hitrec, const GaussRec* __restrict__ rec, const int* __restrict__ radii, const GeomHeader* __restrict__ hdr,
       uint32_t* __restrict__ part_hist, uint32_t* __restrict__ bin_total, uint32_t* __restrict__ bin_start,
       uint2* __restrict__ ranges, uint32_t* __restrict__ big_queue, uint32_t* __restrict__ inst_gid,
       unsigned long long* __restrict__ words, uint32_t* __restrict__ clear_words, uint32_t n_clear, int reserve)
{
    extern __shared__ uint32_t s_bin[];                  // [bins + bins / 16 + 1], indexed through bin_slot()
    // the forward's chunk sums (library scratch) have been consumed by k_compact_write: zero for the next forward on this stream
    if (MODE == 0 && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < n_clear; i += PART_THREADS) clear_words[i] = 0u;
    const uint32_t V = hdr->num_compact;
    const int nb = part_active_blocks(V);
    const int b = (int)blockIdx.x;
    if (b >= nb) return;
    const uint32_t cap = hdr->num_sorted;                // instances that fit the binning buffer (all, in exact mode)
    uint32_t* row = part_hist + (size_t)b * bins;
    if (MODE == 0) {
        for (int i = threadIdx.x; i < bins + (bins >> 4) + 1; i += PART_THREADS) s_bin[i] = 0u;
    } else {
        bin_prefix_to_lds(bins, gx * gy, sub_shift, bin_total, row, s_bin, b == 0, bin_start, ranges, big_queue);
    }
    lds_barrier();
    uint32_t beg, end;
    part_chunk(V, nb, b, beg, end);
    const uint32_t sub_mask = (1u << sub_shift) - 1u;
    if (MODE == 0) {
        walk_chunk(beg, end, gx, gy, own_max, vis_list, offsets, hitrec, rec, radii,
                   [&](uint32_t tile, uint32_t slot, uint32_t gid, uint32_t) {
                       if (slot < cap) { atomicAdd(&s_bin[bin_slot((int)(tile >> sub_shift))], 1u); inst_gid[slot] = gid; }
                   });
        lds_barrier();
        if (reserve) {
            // this workgroup's range inside every bin it found entries for: ONE returning atomic per (workgroup, non-empty
            // bin) on the bin's cursor (zeroed by the scan kernel), which ends up holding the bin's total.  Which workgroup
            // gets which range depends on arrival -- and does not matter: the order inside a bin is made by the per-bin sort,
            // a total order on the words.  No scan over the workgroups' rows, no launch for it (k_part_scan1: 4.9 us + a gap)
            for (int i = threadIdx.x; i < bins; i += PART_THREADS) {
                const uint32_t c = s_bin[bin_slot(i)];
                row[i] = c != 0u ? atomicAdd(&bin_total[i], c) : 0u;
            }
        } else {
            for (int i = threadIdx.x; i < bins; i += PART_THREADS) row[i] = s_bin[bin_slot(i)];
        }
    } else {
        walk_chunk(beg, end, gx, gy, own_max, vis_list, offsets, hitrec, rec, radii,
                   [&](uint32_t tile, uint32_t slot, uint32_t, uint32_t dbits) {
                       if (slot < cap) {
                           const uint32_t pos = atomicAdd(&s_bin[bin_slot((int)(tile >> sub_shift))], 1u);
                           words[pos] = ((unsigned long long)(tile & sub_mask) << (31 + slot_bits)) |
                                        ((unsigned long long)dbits << slot_bits) | (unsigned long long)slot;
                       }
                   });
    }
}

// per bin: exclusive prefix over the active workgroups (in place), total -> bin_total.  A workgroup takes 16 bins; its
// 256 threads are 16 bins x 16 groups of rows (= partition workgroups), so a thread loads at most 16 counts, all
// independent, and the groups are joined through LDS: two memory round trips for the kernel instead of one per eight rows
// of one thread walking a whole column (8.4 us at C3: 32 workgroups, twelve dependent rounds).  Lanes 0-15 of a quarter wave
// touch 16 consecutive words of one row.
#ifdef LR_DIAGNOSTICS        // retired in round 5 (the count kernel reserves its ranges itself); A/B partner of the diagnostics build
constexpr int SCAN1_BINS = 16, SCAN1_GROUPS = 16, SCAN1_ROWS = PART_BLOCKS_MAX / SCAN1_GROUPS;
__global__ void __launch_bounds__(SCAN1_BINS * SCAN1_GROUPS)
k_part_scan1(int bins, const GeomHeader* __restrict__ hdr, uint32_t* __restrict__ part_hist, uint32_t* __restrict__ bin_total)
{
    __shared__ uint32_t s_tot[SCAN1_GROUPS][SCAN1_BINS + 1];
    const int bl = threadIdx.x & (SCAN1_BINS - 1), g = threadIdx.x / SCAN1_BINS;
    const int bin = blockIdx.x * SCAN1_BINS + bl;
    const int nb = part_active_blocks(hdr->num_compact);
    const int per = (nb + SCAN1_GROUPS - 1) / SCAN1_GROUPS;            // rows per group (<= SCAN1_ROWS)
    const int r0 = g * per;
    uint32_t v[SCAN1_ROWS];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN1_ROWS; i++) {
        const bool on = i < per && r0 + i < nb && bin < bins;
        v[i] = on ? part_hist[(size_t)(r0 + i) * bins + bin] : 0u;
    }
#pragma unroll
    for (int i = 0; i < SCAN1_ROWS; i++) sum += v[i];
    s_tot[g][bl] = sum;
    lds_barrier();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int j = 0; j < SCAN1_GROUPS; j++) { const uint32_t t = s_tot[j][bl]; if (j < g) run += t; total += t; }
    if (bin >= bins) return;
#pragma unroll
    for (int i = 0; i < SCAN1_ROWS; i++) {
        if (i < per && r0 + i < nb) part_hist[(size_t)(r0 + i) * bins + bin] = run;
        run += v[i];
    }
    if (g == 0) bin_total[bin] = total;
}
#endif

// -------------------------------------------------------------------------------------------
// sorting one bin.  All comparators ascending (the first step of every merge mirrors the block), so positions
// >= n behave as +infinity padding without being stored.
// -------------------------------------------------------------------------------------------
template <class Sync>
__device__ __forceinline__ void bitonic_sort(unsigned long long* a, uint32_t n, uint32_t tid, uint32_t nthreads, Sync sync)
{
    // k = 2^lk, j = 2^lj: comparator indices by shifts and masks (`c / j`, `c % j` with run-time operands are ~40-instruction
    // divisions each: they were most of this kernel's arithmetic)
    uint32_t lm = 0;
    while ((1u << lm) < n) lm++;
    const uint32_t m = 1u << lm;
    for (uint32_t lk = 1; lk <= lm; lk++) {
        const uint32_t k = 1u << lk;
        for (uint32_t lj = lk; lj-- > 0;) {
            const uint32_t j = 1u << lj;
            const bool flip = (lj == lk - 1);
            for (uint32_t c = tid; c < (m >> 1); c += nthreads) {
                const uint32_t hi = c >> lj, lo = c & (j - 1u);
                uint32_t i, l;
                if (flip) { i = (hi << lk) + lo; l = i ^ (k - 1u); }           // mirror inside the block of k
                else { i = (hi << (lj + 1u)) + lo; l = i + j; }
                if (l < n) {
                    const unsigned long long x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            sync();
        }
    }
}

// after the sort: the list entries (slots) and, when a bin holds several tiles, the per-tile ranges
__device__ __forceinline__ void write_sorted(const unsigned long long* a, uint32_t n, uint32_t start, int bin, int sub_shift,
                                             int slot_bits, int num_tiles, uint32_t tid, uint32_t nthreads,
                                             uint32_t* __restrict__ point_list, uint2* __restrict__ ranges,
                                             const uint32_t* __restrict__ inst_gid, uint32_t* __restrict__ list_gid)
{
    const unsigned long long slot_mask = (1ull << slot_bits) - 1ull;
    for (uint32_t i = tid; i < n; i += nthreads) {
        const unsigned long long w = a[i];
        const uint32_t slot = (uint32_t)(w & slot_mask);
        point_list[start + i] = slot;
        list_gid[start + i] = inst_gid[slot];       // the Gaussian of the list position (common.h BinLayout::list_gid)
        if (sub_shift != 0) {
            const uint32_t sub = (uint32_t)(w >> (31 + slot_bits));
            const int tile = (bin << sub_shift) + (int)sub;
            if (tile < num_tiles) {
                if (i == 0 || (uint32_t)(a[i - 1] >> (31 + slot_bits)) != sub) ranges[tile].x = start + i;
                if (i == n - 1 || (uint32_t)(a[i + 1] >> (31 + slot_bits)) != sub) ranges[tile].y = start + i + 1;
            }
        }
    }
}

// One bin of up to THREADS * PER entries, sorted by THREADS threads into s_out: the bucket sort described above k_tile_sort_large.
// s_cnt: THREADS * PER counters; s_red: 2 words per wave; s_wave: one per wave; s_bad: one flag.  Ends with s_out complete
// (barrier included).  Any monotone map of the keys onto the buckets keeps the result exact -- the order inside a bucket is
// made by an insertion sort on the full 64-bit word, a bin whose keys pile up falls back to the bitonic network.
template <int THREADS, int PER, class Sync>
__device__ __forceinline__ void bucket_sort_bin(const unsigned long long* __restrict__ src, uint32_t n, int slot_bits,
                                                unsigned long long* s_out, uint32_t* s_cnt, unsigned long long* s_red,
                                                uint32_t* s_wave, uint32_t* s_bad_p, int tid, Sync sync)
{
    const int lane = tid & 63, w = tid >> 6;
    unsigned long long item[PER];
    unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * THREADS;
        item[k] = i < n ? src[i] : ~0ull;
        if (i < n) { const unsigned long long key = item[k] >> slot_bits; kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax; }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) s_cnt[tid + k * THREADS] = 0u;
    if (tid == 0) *s_bad_p = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(kmin, off), b = __shfl_xor(kmax, off);
        kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { s_red[2 * w] = kmin; s_red[2 * w + 1] = kmax; }
    sync();
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++) { kmin = s_red[2 * i] < kmin ? s_red[2 * i] : kmin; kmax = s_red[2 * i + 1] > kmax ? s_red[2 * i + 1] : kmax; }
    // bucket of a key: floor((key - kmin) * nbuckets / (span + 1)), evaluated in double (span < 2^34: exact enough to be
    // monotone, which is all that is needed)
    uint32_t nb = 1;
    while (nb < n) nb <<= 1;                                          // <= THREADS * PER
    const double scale = (double)nb / ((double)(kmax - kmin) + 1.0);
    uint32_t bucket[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * THREADS;
        uint32_t bk = (uint32_t)((double)((item[k] >> slot_bits) - kmin) * scale);
        bucket[k] = bk < nb ? bk : nb - 1;
        if (i < n) atomicAdd(&s_cnt[bucket[k]], 1u);
    }
    sync();
    // exclusive scan of the bucket counts: thread t owns buckets [t*PER, t*PER+PER)
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { c[k] = s_cnt[tid * PER + k]; sum += c[k]; if (c[k] > 32u) *s_bad_p = 1u; }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    sync();
    uint32_t run = inc - sum;
    for (int j = 0; j < w; j++) run += s_wave[j];
    const uint32_t my_first = run;
#pragma unroll
    for (int k = 0; k < PER; k++) { s_cnt[tid * PER + k] = run; run += c[k]; }
    sync();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * THREADS;
        if (i < n) s_out[atomicAdd(&s_cnt[bucket[k]], 1u)] = item[k];
    }
    sync();
    if (*s_bad_p) {
        bitonic_sort(s_out, n, (uint32_t)tid, (uint32_t)THREADS, sync);
    } else {
        // each thread orders its own PER consecutive buckets: the segment [my_first, run)
        uint32_t lo = my_first;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint32_t hi = lo + c[k];
            for (uint32_t i = lo + 1; i < hi; i++) {
                const unsigned long long x = s_out[i];
                uint32_t j = i;
                while (j > lo && s_out[j - 1] > x) { s_out[j] = s_out[j - 1]; j--; }
                s_out[j] = x;
            }
            lo = hi;
        }
        sync();
    }
}

// Two launches for all bin sizes (round 2 had four, three of which found nothing to do on a sparse view and still cost a
// launch each):
//   k_tile_sort_small  256 threads = 4 waves per workgroup.  Part A of the grid: one workgroup per 4 consecutive bins, a bin
//                      of up to 256 entries (a C3 tile holds ~70) sorted by ONE wave in its own 2 KB of LDS, no block
//                      barrier involved.  Part B (up to 2048 more workgroups, striding over the bins): bins of 257..1024
//                      entries by a whole workgroup with the four slices as one 8 KB array (the dense 1080p / 1440p
//                      clouds: 400..700 per tile).
//   k_tile_sort_large  fed by the queue the scatter kernel built (bins of more than 1024 entries), 512 threads: up to
//                      4096 entries the bucket sort, beyond that the bitonic network -- in LDS when the launch was given
//                      room for it (lds_entries), else in place in global memory.
__global__ void __launch_bounds__(256)
k_tile_sort_small(int bins, int groups4, int sub_shift, int slot_bits, int num_tiles, int bucket_b, const uint32_t* __restrict__ bin_start,
                  const uint32_t* __restrict__ bin_total, const unsigned long long* __restrict__ words,
                  uint32_t* __restrict__ point_list, uint2* __restrict__ ranges, const uint32_t* __restrict__ inst_gid,
                  uint32_t* __restrict__ list_gid)
{
    __shared__ unsigned long long s_a[4 * TSORT_LDS];         // 4 x 256 entries = TSORT_GROUP_LDS
    __shared__ uint32_t s_cnt[TSORT_GROUP_LDS];               // part B's bucket counters
    __shared__ unsigned long long s_red[8];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_bad[4];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if ((int)blockIdx.x < groups4) {
        // part A: workgroup g takes bins 4 g .. 4 g + 3, one per wave, if they hold at most 256 entries
        const int bin = (int)blockIdx.x * 4 + w;
        const uint32_t n = bin < bins ? bin_total[bin] : 0u;
        if (n == 0 || n > (uint32_t)TSORT_LDS) return;
        unsigned long long* a = s_a + w * TSORT_LDS;
        const uint32_t start = bin_start[bin];
        // one wave, its own slice: LDS operations of a wave execute in order, so a wave-level fence is all the exchange
        // between its lanes needs (no workgroup barrier anywhere in this part)
        auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
        if (bucket_b == 2 && n > 64) {
            // the bucket sort by one wave in its own slices (C3: sort stage 19.4 -> 18.5 us, dense 1 M cloud 59.4 -> 56.2;
            // profiles/r04d_ab_tsort_wave.json); up to 64 entries the network's 21 stages stay cheaper
            bucket_sort_bin<64, TSORT_LDS / 64>(words + start, n, slot_bits, a, s_cnt + w * TSORT_LDS, s_red + 2 * w, s_wave + w,
                                                &s_bad[w], l, wsync);
            write_sorted(a, n, start, bin, sub_shift, slot_bits, num_tiles, (uint32_t)l, 64u, point_list, ranges, inst_gid, list_gid);
            return;
        }
        for (uint32_t i = l; i < n; i += 64) a[i] = words[start + i];
        wsync();
        if (n > 1) bitonic_sort(a, n, (uint32_t)l, 64u, wsync);
        write_sorted(a, n, start, bin, sub_shift, slot_bits, num_tiles, (uint32_t)l, 64u, point_list, ranges, inst_gid, list_gid);
        return;
    }
    // part B: the remaining workgroups walk the bins with a stride and take those of 257..1024 entries, one at a time, with
    // the four slices as one array (a sparse view has none: these workgroups read their bins' counts and leave)
    const int first = (int)blockIdx.x - groups4, stride = (int)gridDim.x - groups4;
    for (int bin = first; bin < bins; bin += stride) {
        const uint32_t n = bin_total[bin];
        if (n <= (uint32_t)TSORT_LDS || n > (uint32_t)TSORT_GROUP_LDS) continue;
        const uint32_t start = bin_start[bin];
        lds_barrier();
        if (bucket_b) {
            // the bucket sort of k_tile_sort_large at this size: 4 entries and 4 buckets per thread, ~7 barriers instead of
            // the network's 45-55
            bucket_sort_bin<256, TSORT_GROUP_LDS / 256>(words + start, n, slot_bits, s_a, s_cnt, s_red, s_wave, &s_bad[0],
                                                        (int)threadIdx.x, [] { lds_barrier(); });
        } else {
            for (uint32_t i = threadIdx.x; i < n; i += 256) s_a[i] = words[start + i];
            lds_barrier();
            bitonic_sort(s_a, n, threadIdx.x, 256u, [] { lds_barrier(); });
        }
        write_sorted(s_a, n, start, bin, sub_shift, slot_bits, num_tiles, threadIdx.x, 256u, point_list, ranges, inst_gid, list_gid);
    }
}

// 1025..4096 entries: a bucket sort.  The sort key of a word is everything above its slot bits (sub-tile, depth bits); keys
// are mapped monotonically onto ~n buckets between the bin's smallest and largest key, counted, scanned and scattered
// with LDS atomics (order inside a bucket arbitrary), then every bucket -- one or two entries on average -- is put in
// order by an insertion sort on the full 64-bit word.  Any monotone map keeps the result exact; the map only decides how
// evenly the buckets fill.  A bin whose keys pile up (a bucket of more than 32 entries, e.g. many splats at one depth)
// falls back to the bitonic network.  ~8 barriers instead of the network's 66 at these sizes (dense 512^2 view: 0.145 ->
// 0.052 ms).
__global__ void __launch_bounds__(TSORT_THREADS)
k_tile_sort_large(int sub_shift, int slot_bits, int num_tiles, uint32_t lds_entries, const uint32_t* __restrict__ bin_start,
                  const uint32_t* __restrict__ bin_total, unsigned long long* __restrict__ words,
                  uint32_t* __restrict__ point_list, uint2* __restrict__ ranges, const uint32_t* __restrict__ big_queue,
                  const uint32_t* __restrict__ inst_gid, uint32_t* __restrict__ list_gid)
{
    constexpr int PER = TSORT_MID_LDS / TSORT_THREADS;                // 8 items / buckets per thread
    extern __shared__ unsigned long long s_out[];                     // [max(TSORT_MID_LDS, lds_entries)]
    __shared__ uint32_t s_cnt[TSORT_MID_LDS];
    __shared__ unsigned long long s_red[2 * (TSORT_THREADS / 64)];
    __shared__ uint32_t s_wave[TSORT_THREADS / 64];
    __shared__ uint32_t s_bad;
    const int tid = threadIdx.x;
    const uint32_t count = big_queue[0];
    for (uint32_t qi = blockIdx.x; qi < count; qi += gridDim.x) {
    const int bin = (int)big_queue[1 + qi];
    const uint32_t n = bin_total[bin];
    const uint32_t start = bin_start[bin];
    lds_barrier();
    if (n > (uint32_t)TSORT_MID_LDS) {
        if (n <= lds_entries) {
            for (uint32_t i = tid; i < n; i += TSORT_THREADS) s_out[i] = words[start + i];
            lds_barrier();
            bitonic_sort(s_out, n, (uint32_t)tid, (uint32_t)TSORT_THREADS, [] { lds_barrier(); });
            write_sorted(s_out, n, start, bin, sub_shift, slot_bits, num_tiles, (uint32_t)tid, (uint32_t)TSORT_THREADS, point_list, ranges, inst_gid, list_gid);
        } else {
            // larger than the LDS of this launch: the same network in place in global memory (one workgroup, L2-resident;
            // slow, but a single tile with that many splats is slow to blend anyway)
            unsigned long long* a = words + start;
            bitonic_sort(a, n, (uint32_t)tid, (uint32_t)TSORT_THREADS, [] { __threadfence(); lds_barrier(); });
            write_sorted(a, n, start, bin, sub_shift, slot_bits, num_tiles, (uint32_t)tid, (uint32_t)TSORT_THREADS, point_list, ranges, inst_gid, list_gid);
        }
        continue;
    }
    bucket_sort_bin<TSORT_THREADS, PER>(words + start, n, slot_bits, s_out, s_cnt, s_red, s_wave, &s_bad, tid, [] { lds_barrier(); });
    write_sorted(s_out, n, start, bin, sub_shift, slot_bits, num_tiles, (uint32_t)tid, (uint32_t)TSORT_THREADS, point_list, ranges, inst_gid, list_gid);
    }
}

}  // namespace

PartPlan part_plan(int num_tiles)
{
    PartPlan p;
    p.sub_shift = 0;
    while (((num_tiles - 1) >> p.sub_shift) + 1 > PART_BINS_MAX) p.sub_shift++;
    p.bins = num_tiles > 0 ? ((num_tiles - 1) >> p.sub_shift) + 1 : 1;
    return p;
}

void launch_compact(int P, const uint32_t* tiles_touched, const uint4* block_sums,
                    uint32_t* vis_list, uint32_t* offsets, GeomHeader* hdr, uint32_t capacity, uint32_t* log_slot,
                    uint32_t log_tag, uint32_t* zero_words, uint32_t n_zero, bool reset_sticky, hipStream_t s)
{
    const int nb = (P + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(SCAN_THREADS), 0, s, P, tiles_touched, block_sums, vis_list,
                       offsets, hdr, capacity, log_slot, log_tag, zero_words, n_zero, reset_sticky ? 1 : 0);
}

int launch_tile_binning(int P, int gx, int gy, int slot_bits, const uint32_t* vis_list, const uint32_t* offsets,
                        const uint4* hitrec, const GaussRec* rec, const int* radii, GeomHeader* hdr,
                        uint32_t* part_hist, uint32_t* bin_total, uint32_t* bin_start, uint32_t* big_queue,
                        uint32_t* inst_gid, unsigned long long* words, uint32_t* point_list, uint32_t* list_gid, uint2* ranges,
                        long long bin_bound_hint, TileBinTimes* t, uint32_t* clear_words, uint32_t n_clear, hipStream_t s)
{
    // the partition kernels keep one counter per bin in LDS (up to 64 KB), the large-bin sort up to 128 KB.  The attribute is
    // set once per DEVICE (a process may drive several; runtimes that keep it per device would otherwise refuse the
    // > 64 KB launches on the second one)
    {
        static std::mutex mu;
        static bool attr_done[64] = {};
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess) return -1;
        std::lock_guard<std::mutex> lock(mu);
        if (device < 0 || device >= 64 || !attr_done[device]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_part<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (PART_BINS_MAX + PART_BINS_MAX / 16 + 1) * 4) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(k_part<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (PART_BINS_MAX + PART_BINS_MAX / 16 + 1) * 4) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_sort_large), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    TSORT_BIG_LDS * 8) != hipSuccess)
                return -1;
            if (device >= 0 && device < 64) attr_done[device] = true;
        }
    }
    const int num_tiles = gx * gy;
    const PartPlan pp = part_plan(num_tiles);
    if (pp.sub_shift + 31 + slot_bits > 64) return -2;
    // workgroups launched: what the largest possible list needs; the kernels retire the surplus from the device count
    long long nb = ((long long)P + PART_MIN_GAUSS - 1) / PART_MIN_GAUSS;
    if (nb < 1) nb = 1;
    if (nb > PART_BLOCKS_MAX) nb = PART_BLOCKS_MAX;
    const size_t lds = ((size_t)pp.bins + pp.bins / 16 + 1) * 4;
    // instances of one Gaussian walked by its own lane before the wave shares them (lr_tune_set("walk_own", n))
    const int own_max = tune_get(TUNE_WALK_OWN) >= 0 ? tune_get(TUNE_WALK_OWN) : 12;
    // lr_tune_set("part_scan", 1): the count kernel leaves per-workgroup counts and k_part_scan1 turns them into bases (rounds
    // 2-4; A/B partner); default: the count kernel reserves its ranges itself (bin_total arrives zeroed from the scan kernel)
#ifdef LR_DIAGNOSTICS
    const int reserve = tune_get(TUNE_PART_SCAN) == 1 ? 0 : 1;
#else
    constexpr int reserve = 1;
#endif
    if (t) t->mark(0, s);
    hipLaunchKernelGGL(k_part<0>, dim3((unsigned)nb), dim3(PART_THREADS), lds, s, gx, gy, pp.bins, pp.sub_shift, slot_bits,
                       own_max, vis_list, offsets, hitrec, rec, radii, hdr, part_hist, bin_total, bin_start, ranges, big_queue,
                       inst_gid, words, clear_words, n_clear, reserve);
#ifdef LR_DIAGNOSTICS
    if (!reserve) {
        if (t) t->mark(1, s);
        hipLaunchKernelGGL(k_part_scan1, dim3((pp.bins + SCAN1_BINS - 1) / SCAN1_BINS), dim3(SCAN1_BINS * SCAN1_GROUPS), 0, s, pp.bins, hdr,
                           part_hist, bin_total);
    }
#endif
    if (t) t->mark(2, s);                                   // closes the stage that is open, opens the scatter's
    hipLaunchKernelGGL(k_part<1>, dim3((unsigned)nb), dim3(PART_THREADS), lds, s, gx, gy, pp.bins, pp.sub_shift, slot_bits,
                       own_max, vis_list, offsets, hitrec, rec, radii, hdr, part_hist, bin_total, bin_start, ranges, big_queue,
                       inst_gid, words, nullptr, 0u, reserve);
    if (t) t->mark(3, s);
    const int groups4 = (pp.bins + 3) / 4;
    const int part_b = pp.bins < TSORT_CLASS_BLOCKS ? pp.bins : TSORT_CLASS_BLOCKS;
    // per-bin algorithm: 2 (default) = bucket sort from 65 entries up, 1 = only for 257..1024 (the wave-sized bins through the
    // bitonic network), 0 = the network for everything up to 1024 (lr_tune_set("tsort", v): A/B runs)
    const int bucket_b = tune_get(TUNE_TSORT) >= 0 ? tune_get(TUNE_TSORT) : 2;
    hipLaunchKernelGGL(k_tile_sort_small, dim3(groups4 + part_b), dim3(256), 0, s, pp.bins, groups4, pp.sub_shift, slot_bits,
                       num_tiles, bucket_b, bin_start, bin_total, words, point_list, ranges, inst_gid, list_gid);
    // the large bins: LDS for the bucket sort (32 KB of words; three workgroups per CU) unless the AVERAGE bin is already
    // beyond it -- then 128 KB, so that bins of up to 16384 entries are sorted in LDS (a hint for speed only: a bin that
    // does not fit this launch's LDS is sorted in place in global memory).  Capped grid: a sparse view queues nothing
    const long long avg_bin = bin_bound_hint / (long long)pp.bins;
    const bool huge = avg_bin > (long long)TSORT_MID_LDS;
    const uint32_t lds_entries = huge ? (uint32_t)TSORT_BIG_LDS : (uint32_t)TSORT_MID_LDS;
    // what is resident at once (the queue is strided) -- or, where the AVERAGE bin is far below the queue's threshold (a C3
    // view: 70 entries per tile, the queue almost always empty), a small grid: what a launch costs that finds nothing to do
    // is its workgroups (768: 4.4 us, 64: the launch floor).  A hint for speed only, like `huge`
    const int large_cap = huge ? TSORT_BIG_BLOCKS : (avg_bin <= (long long)TSORT_LDS ? TSORT_LARGE_BLOCKS / 12 : TSORT_LARGE_BLOCKS);
    const int large_blocks = pp.bins < large_cap ? pp.bins : large_cap;
    hipLaunchKernelGGL(k_tile_sort_large, dim3(large_blocks), dim3(TSORT_THREADS), (size_t)lds_entries * 8, s, pp.sub_shift,
                       slot_bits, num_tiles, lds_entries, bin_start, bin_total, words, point_list, ranges, big_queue, inst_gid, list_gid);
    if (t) t->mark(4, s);
    return hipGetLastError() == hipSuccess ? 0 : -1;        // a refused launch (LDS attribute, grid) surfaces here, not at the blend
}

}  // namespace lr
