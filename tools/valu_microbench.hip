// VALU issue-rate micro-benchmark quoted in DESIGN.md section 4 ("What the counters say"):
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o /tmp/valu_microbench && /tmp/valu_microbench
// Every wave runs a long dependent-free stream of one instruction type; the kernel fills all 256 CUs x 4 SIMDs with
// 8 waves each.  Reported: wave-instructions per second and cycles per instruction per SIMD at 2.4 GHz.
// Measured on MI355X: v_fma_f32 537 G/s (4.6 cycles), v_pk_fma_f32 530 G/s (4.6 cycles, i.e. twice the flops),
// v_exp_f32 + v_mul_f32 pair ~12 cycles per exp.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, UNROLL = 16;

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed, int iters)
{
    float a[UNROLL];
    v2f p[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + i + threadIdx.x; p[i] = v2f{ a[i], a[i] + 1.f }; }
    const float m = 1.0000001f, c = 1e-9f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], v2f{ m, m }, v2f{ c, c });
            if (MODE == 2) a[i] = __expf(a[i] * 1e-9f);
            if (MODE == 3) a[i] = __builtin_amdgcn_exp2f(a[i]);                          // v_exp_f32 alone
            if (MODE == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);                           // v_rcp_f32
            if (MODE == 5) a[i] = fminf(a[i], m);                                        // v_min_f32
            if (MODE == 6) a[i] = (a[i] < c) ? m : a[i];                                 // v_cmp + v_cndmask
            if (MODE == 7) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xf, 0xf, false));  // v_add_f32_dpp
            if (MODE == 9) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a[i]), 0x041F));   // xor 1 via the LDS crossbar + v_add
            if (MODE == 10) a[i] += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a[i]), 0xB1, 0xf, 0xf, false)) * m;   // v_mov_dpp + v_fma
            if (MODE == 8) { float y = a[(i + 1) % UNROLL]; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(y)); a[(i + 1) % UNROLL] = y; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;                       // 8 x 256 threads per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 16);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, ITERS);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 /*waves*/ * ITERS * UNROLL * ((MODE == 2 || MODE == 6 || MODE == 9 || MODE == 10) ? 2 : 1);
    const double gps = insts / (ms * 1e-3) / 1e9;
    printf("%s: %.3f ms, %.2f G wave-instr/s, per-SIMD cycles/instr at 2.4GHz = %.2f\n", name, ms, gps, 1024 * 2.4 / gps);
}

int main()
{
    float* d;
    hipMalloc(&d, 4);
    run<0>("v_fma_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_exp_f32(+mul)", d);
    run<3>("v_exp_f32", d);
    run<4>("v_rcp_f32", d);
    run<5>("v_min_f32", d);
    run<6>("v_cmp+v_cndmask", d);
    run<7>("v_add_f32_dpp", d);
    run<8>("v_permlane32_swap (+s_nop 1)", d);
    run<9>("ds_swizzle + v_add_f32", d);
    run<10>("v_mov_dpp + v_fma", d);
    return 0;
}
