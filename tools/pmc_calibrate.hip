// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of THIS library
// (MI355X_MICROARCH.md, section HBM: FETCH_SIZE reports half of a wide coalesced streaming read; "other access widths
// and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o /tmp/pmc_calibrate
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -o fetch -- /tmp/pmc_calibrate
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -o write -- /tmp/pmc_calibrate
//   (tools/pmc_run.sh does both; tools/pmc_summary.py prints counter / known bytes per kernel)
//
// Every kernel moves a KNOWN number of bytes over buffers far larger than the 256 MB Infinity Cache (each launch
// touches fresh 1 GiB regions), in one of the patterns the rasterizer uses:
//   cal_read16      16 B per lane, coalesced          (SH rows, GaussRec streaming)        known: N * 16 read
//   cal_read4       4 B per lane, coalesced           (index / count arrays)               known: N * 4 read
//   cal_gather48    48-byte records at random indices (GaussRec gathers of the blend)      known: N * 48 read (+ N * 4 index)
//   cal_write16     16 B per lane, coalesced                                              known: N * 16 written
//   cal_write4      4 B per lane, coalesced                                               known: N * 4 written
//   cal_scatter8    8-byte words at random positions  (bin scatter of tilebin.hip)         known: N * 8 written (+ N * 4 index)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void cal_read16(const float4* __restrict__ in, float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = in[i];
    if (v.x + v.y + v.z + v.w == 123.456f) out[0] = 1.f;
}
__global__ void cal_read4(const float* __restrict__ in, float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (in[i] == 123.456f) out[0] = 1.f;
}
__global__ void cal_gather48(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* r = rec + 3 * (size_t)idx[i];
    const float4 a = r[0], b = r[1], c = r[2];
    if (a.x + b.y + c.z == 123.456f) out[0] = 1.f;
}
__global__ void cal_write16(float4* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void cal_write4(float* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 1.f;
}
__global__ void cal_scatter8(unsigned long long* out, const uint32_t* __restrict__ idx, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[idx[i]] = i;
}

int main()
{
    const size_t GiB = 1ull << 30;
    const size_t n16 = GiB / 16, n4 = GiB / 4, nrec = GiB / 48, n8 = GiB / 8;
    char* buf = nullptr;                       // 6 disjoint 1 GiB regions + index arrays
    if (hipMalloc(&buf, 6 * GiB) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, 6 * GiB);
    float* out = nullptr;
    (void)hipMalloc(&out, 256);
    const size_t ngather = 16u << 20, nscatter = 32u << 20;
    std::vector<uint32_t> h(ngather > nscatter ? ngather : nscatter);
    uint32_t *idx_g = nullptr, *idx_s = nullptr;
    (void)hipMalloc(&idx_g, ngather * 4);
    (void)hipMalloc(&idx_s, nscatter * 4);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < ngather; i++) h[i] = (uint32_t)(rnd() % nrec);
    (void)hipMemcpy(idx_g, h.data(), ngather * 4, hipMemcpyHostToDevice);
    // a random permutation-like target: distinct positions so that every word is written once
    for (size_t i = 0; i < nscatter; i++) h[i] = (uint32_t)((i * 2654435761ull) % n8);
    (void)hipMemcpy(idx_s, h.data(), nscatter * 4, hipMemcpyHostToDevice);
    (void)hipDeviceSynchronize();
    const int T = 256;
    hipLaunchKernelGGL(cal_read16, dim3((unsigned)((n16 + T - 1) / T)), dim3(T), 0, 0, (const float4*)(buf + 0 * GiB), out, n16);
    hipLaunchKernelGGL(cal_read4, dim3((unsigned)((n4 + T - 1) / T)), dim3(T), 0, 0, (const float*)(buf + 1 * GiB), out, n4);
    hipLaunchKernelGGL(cal_gather48, dim3((unsigned)((ngather + T - 1) / T)), dim3(T), 0, 0, (const float4*)(buf + 2 * GiB), idx_g, out, ngather);
    hipLaunchKernelGGL(cal_write16, dim3((unsigned)((n16 + T - 1) / T)), dim3(T), 0, 0, (float4*)(buf + 3 * GiB), n16);
    hipLaunchKernelGGL(cal_write4, dim3((unsigned)((n4 + T - 1) / T)), dim3(T), 0, 0, (float*)(buf + 4 * GiB), n4);
    hipLaunchKernelGGL(cal_scatter8, dim3((unsigned)((nscatter + T - 1) / T)), dim3(T), 0, 0, (unsigned long long*)(buf + 5 * GiB), idx_s, nscatter);
    (void)hipDeviceSynchronize();
    printf("{\"cal_read16\": {\"read\": %zu, \"written\": 0}, \"cal_read4\": {\"read\": %zu, \"written\": 0}, "
           "\"cal_gather48\": {\"read\": %zu, \"written\": 0, \"records\": %zu}, \"cal_write16\": {\"read\": 0, \"written\": %zu}, "
           "\"cal_write4\": {\"read\": 0, \"written\": %zu}, \"cal_scatter8\": {\"read\": %zu, \"written\": %zu}}\n",
           n16 * 16, n4 * 4, ngather * 48 + ngather * 4, ngather, n16 * 16, n4 * 4, nscatter * 4, nscatter * 8);
    return 0;
}
