/*
 * cuda_host_shim.h -- just enough of the CUDA programming model, on the host, to compile the
 * reference's own .cu files (RAST/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu and
 * KNN/simple_knn.cu) with g++ and run them on CPU cores.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/_ref recipe, see oracle/build_ref.py).  Nothing in the product
 * path includes, links or loads this.  It exists so that oracle/raster_oracle.c (the restatement)
 * can be pinned against the reference's OWN source text: the kernels, their launch order, the
 * scratch-buffer carving and the host orchestration all come from /root/reference unchanged; this
 * header only supplies what the CUDA toolkit would have supplied:
 *
 *   - vector types (float2/3/4, uint2, dim3), the __global__/__device__/... qualifiers,
 *     CUDA's integer/float min/max overload set, atomicAdd(float*), __trap();
 *   - the execution model: a launch runs every thread block; the threads of a block are
 *     cooperative fibers (ucontext) scheduled round-robin in thread_rank order, so that
 *     __syncthreads(), block.sync() and __syncthreads_count() have their CUDA meaning and
 *     __shared__ variables are per block.  Blocks are distributed over OpenMP threads
 *     (ref_set_threads(1) gives a fixed, repeatable order of the float atomics);
 *   - cooperative_groups::this_grid()/this_thread_block() as used by the reference;
 *   - the two CUB device primitives the reference calls (InclusiveSum, stable SortPairs on a bit
 *     range) plus DeviceReduce::Reduce for simple-knn, implemented from their documented contract;
 *   - cudaMemcpy/cudaMemset/cudaMalloc/cudaFree/cudaDeviceSynchronize as host calls.
 *
 * Arithmetic differences from a real CUDA build that remain: the host libm's expf/sqrtf instead of
 * the device's, and no FMA contraction (the recipe compiles with -ffp-contract=off; nvcc contracts).
 */
#ifndef LUCID_REF_CUDA_HOST_SHIM_H
#define LUCID_REF_CUDA_HOST_SHIM_H

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <ucontext.h>
#include <omp.h>

#include <algorithm>
#include <iostream>
#include <numeric>
#include <stdexcept>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

/* ---- qualifiers ---------------------------------------------------------------------------- */
#define __global__ static
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#ifndef __restrict__
#define __restrict__ __restrict
#endif

/* ---- vector types -------------------------------------------------------------------------- */
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};

/* ---- CUDA's min/max overload set (math_functions.hpp): mixed signedness resolves to unsigned --- */
inline int min(int a, int b) { return a < b ? a : b; }
inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
inline float min(float a, float b) { return fminf(a, b); }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
inline float max(float a, float b) { return fmaxf(a, b); }

static_assert(std::is_same<decltype(exp(1.0f)), float>::value, "exp(float) must be the float overload, as in device code");
static_assert(std::is_same<decltype(sqrt(1.0f)), float>::value, "sqrt(float) must be the float overload");
static_assert(std::is_same<decltype(ceil(1.0f)), float>::value, "ceil(float) must be the float overload");

/* ---- runtime stubs ------------------------------------------------------------------------- */
enum cudaError_t { cudaSuccess = 0, cudaErrorLaunchFailure = 719 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
inline cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind) { memcpy(dst, src, n); return cudaSuccess; }
inline cudaError_t cudaMemset(void* dst, int v, size_t n) { memset(dst, v, n); return cudaSuccess; }
template <class T> inline cudaError_t cudaMalloc(T** p, size_t n) { *p = (T*)malloc(n ? n : 1); return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
inline const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "unspecified launch failure"; }

namespace shim {

/* One fiber per CUDA thread of the block being executed. */
struct Fiber {
    ucontext_t ctx;
    char* stack;
    bool done;
    uint3 tid;
    unsigned int rank;
};

struct Worker {                      /* one per OpenMP thread */
    ucontext_t sched;
    std::vector<Fiber> fibers;
    Fiber* cur = nullptr;
    dim3 bid, bdim, gdim;
    int count_accum = 0, count_result = 0;
    void (*body)(void*) = nullptr;
    void* body_arg = nullptr;
    ~Worker() { for (auto& f : fibers) free(f.stack); }
};

inline Worker& worker() { static thread_local Worker w; return w; }
extern bool g_trapped;               /* set by __trap(); defined in ref_capi.cpp */
extern int g_threads;                /* 0 = OpenMP default */
static const size_t kStack = 64 * 1024;

inline void yield() { Worker& w = worker(); swapcontext(&w.cur->ctx, &w.sched); }

static void fiber_entry()
{
    Worker& w = worker();
    w.body(w.body_arg);
    w.cur->done = true;              /* returning resumes uc_link = the scheduler */
}

/* Run one thread block to completion: passes over the live fibers in thread_rank order; every pass ends
 * with each live fiber either finished or parked at the same barrier. */
inline void run_block(Worker& w, unsigned int nthreads)
{
    if (w.fibers.size() < nthreads) {
        size_t old = w.fibers.size();
        w.fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) w.fibers[i].stack = (char*)malloc(kStack);
    }
    for (unsigned int t = 0; t < nthreads; t++) {
        Fiber& f = w.fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &w.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        f.done = false;
        f.rank = t;
        f.tid.x = t % w.bdim.x;
        f.tid.y = (t / w.bdim.x) % w.bdim.y;
        f.tid.z = t / (w.bdim.x * w.bdim.y);
    }
    w.count_accum = w.count_result = 0;
    unsigned int live = nthreads;
    while (live) {
        live = 0;
        for (unsigned int t = 0; t < nthreads; t++) {
            Fiber& f = w.fibers[t];
            if (f.done) continue;
            w.cur = &f;
            swapcontext(&w.sched, &f.ctx);
            if (!f.done) live++;
        }
        w.count_result = w.count_accum;
        w.count_accum = 0;
    }
    w.cur = nullptr;
}

template <class... A>
struct Launcher {
    void (*kernel)(A...);
    dim3 grid, block;
    template <class... B>
    void operator()(B&&... b) const
    {
        std::tuple<typename std::decay<A>::type...> args(std::forward<B>(b)...);
        struct Ctx { void (*k)(A...); std::tuple<typename std::decay<A>::type...>* a; } ctx = { kernel, &args };
        auto body = [](void* p) { Ctx* c = (Ctx*)p; std::apply(c->k, *c->a); };
        const long long nblocks = (long long)grid.x * grid.y * grid.z;
        const unsigned int nthreads = block.x * block.y * block.z;
        const dim3 g = grid, bd = block;
        const int nt = g_threads > 0 ? g_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
        for (long long b = 0; b < nblocks; b++) {
            Worker& w = worker();
            w.body = body;
            w.body_arg = &ctx;
            w.gdim = g;
            w.bdim = bd;
            w.bid.x = (unsigned int)(b % g.x);
            w.bid.y = (unsigned int)((b / g.x) % g.y);
            w.bid.z = (unsigned int)(b / ((long long)g.x * g.y));
            run_block(w, nthreads);
        }
    }
};

template <class... A>
inline Launcher<A...> launch(void (*k)(A...), dim3 grid, dim3 block) { return Launcher<A...>{ k, grid, block }; }

}  // namespace shim

#define threadIdx (shim::worker().cur->tid)
#define blockIdx (shim::worker().bid)
#define blockDim (shim::worker().bdim)
#define gridDim (shim::worker().gdim)

inline void __syncthreads() { shim::yield(); }
inline int __syncthreads_count(int pred)
{
    shim::Worker& w = shim::worker();
    w.count_accum += pred ? 1 : 0;
    shim::yield();
    return w.count_result;
}
/* device trap: remember it and retire the calling thread (auxiliary.h:156-160 prints first) */
inline void __trap()
{
    shim::g_trapped = true;
    shim::Worker& w = shim::worker();
    w.cur->done = true;
    swapcontext(&w.cur->ctx, &w.sched);
}
/* float atomicAdd: blocks may run on different OpenMP threads */
inline float atomicAdd(float* addr, float v)
{
    uint32_t* a = (uint32_t*)addr;
    uint32_t old = __atomic_load_n(a, __ATOMIC_RELAXED), neu;
    float f;
    do {
        memcpy(&f, &old, 4);
        f += v;
        memcpy(&neu, &f, 4);
    } while (!__atomic_compare_exchange_n(a, &old, neu, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}

/* device printf (auxiliary.h:158 prints once per culled-but-prefiltered Gaussian): keep the first message only */
namespace shim {
template <class... A> inline int device_printf(const char* fmt, A... a)
{
    static bool said = false;
    if (said) return 0;
    said = true;
    int n = fprintf(stderr, fmt, a...);
    fputc('\n', stderr);
    return n;
}
}  // namespace shim
#define printf(...) shim::device_printf(__VA_ARGS__)

/* ---- cooperative groups, as far as the reference uses them ---------------------------------- */
namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const
    {
        shim::Worker& w = shim::worker();
        unsigned long long blk = ((unsigned long long)w.bid.z * w.gdim.y + w.bid.y) * w.gdim.x + w.bid.x;
        return blk * ((unsigned long long)w.bdim.x * w.bdim.y * w.bdim.z) + w.cur->rank;
    }
};
struct thread_block {
    dim3 group_index() const { return shim::worker().bid; }
    dim3 thread_index() const { uint3 t = shim::worker().cur->tid; return dim3(t.x, t.y, t.z); }
    unsigned int thread_rank() const { return shim::worker().cur->rank; }
    void sync() const { shim::yield(); }
};
inline grid_group this_grid() { return grid_group(); }
inline thread_block this_thread_block() { return thread_block(); }
}  // namespace cooperative_groups

/* ---- CUB device primitives (documented contracts) ------------------------------------------- */
namespace cub {
struct DeviceScan {
    template <class In, class Out, class N>
    static cudaError_t InclusiveSum(void* tmp, size_t& bytes, In in, Out out, N n)
    {
        if (tmp == nullptr) { bytes = 1; return cudaSuccess; }
        typename std::remove_reference<decltype(out[0])>::type acc = 0;
        for (N i = 0; i < n; i++) { acc += in[i]; out[i] = acc; }
        return cudaSuccess;
    }
};
struct DeviceRadixSort {
    /* stable sort of (key, value) pairs comparing only key bits [begin_bit, end_bit) */
    template <class K, class V, class N>
    static cudaError_t SortPairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, N n,
                                 int begin_bit = 0, int end_bit = sizeof(K) * 8)
    {
        if (tmp == nullptr) { bytes = 1; return cudaSuccess; }
        std::vector<K> ka(kin, kin + n), kb((size_t)n);
        std::vector<V> va(vin, vin + n), vb((size_t)n);
        for (int bit = begin_bit; bit < end_bit; bit += 8) {
            const int width = std::min(8, end_bit - bit);
            const unsigned int mask = (1u << width) - 1;
            size_t hist[257] = { 0 };
            for (size_t i = 0; i < (size_t)n; i++) hist[((ka[i] >> bit) & mask) + 1]++;
            for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
            for (size_t i = 0; i < (size_t)n; i++) {
                size_t dst = hist[(ka[i] >> bit) & mask]++;
                kb[dst] = ka[i];
                vb[dst] = va[i];
            }
            ka.swap(kb);
            va.swap(vb);
        }
        std::copy(ka.begin(), ka.end(), kout);
        std::copy(va.begin(), va.end(), vout);
        return cudaSuccess;
    }
};
struct DeviceReduce {
    template <class In, class Out, class N, class Op, class T>
    static cudaError_t Reduce(void* tmp, size_t& bytes, In in, Out out, N n, Op op, T init)
    {
        if (tmp == nullptr) { bytes = 1; return cudaSuccess; }
        T acc = init;
        for (N i = 0; i < n; i++) acc = op(acc, in[i]);
        *out = acc;
        return cudaSuccess;
    }
};
}  // namespace cub

/* ---- thrust, as far as simple-knn uses it --------------------------------------------------- */
namespace thrust {
template <class T> struct device_ptr { T* p; T* get() const { return p; } };
template <class T>
struct device_vector {
    std::vector<T> v;
    device_vector() {}
    explicit device_vector(size_t n) : v(n) {}
    device_ptr<T> data() { return device_ptr<T>{ v.data() }; }
    typename std::vector<T>::iterator begin() { return v.begin(); }
    typename std::vector<T>::iterator end() { return v.end(); }
    void resize(size_t n) { v.resize(n); }
};
template <class It> inline void sequence(It a, It b) { std::iota(a, b, 0); }
}  // namespace thrust

#endif
