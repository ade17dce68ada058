// VALU issue-rate micro-benchmark quoted in DESIGN.md section 4 ("What the counters say"):
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o /tmp/valu_microbench && /tmp/valu_microbench
// Every wave runs a long dependent-free stream of one instruction type; the kernel fills all 256 CUs x 4 SIMDs with
// 8 waves each.  Reported: wave-instructions per second and cycles per instruction per SIMD at 2.4 GHz.
// Measured on MI355X: v_fma_f32 537 G/s (4.6 cycles), v_pk_fma_f32 530 G/s (4.6 cycles, i.e. twice the flops),
// v_exp_f32 + v_mul_f32 pair ~12 cycles per exp.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096, UNROLL = 16;

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed, int iters)
{
    float a[UNROLL];
    v2f p[UNROLL];
    unsigned long long q[UNROLL];
    __shared__ float lds[4096];
    const unsigned lds_addr = (unsigned)(size_t)lds + 4u * (threadIdx.x & 1023), lds_addr4 = (unsigned)(size_t)lds + 16u * (threadIdx.x & 255);
    const unsigned lds_addr12 = lds_addr;
    float4v q4 = { seed, seed, seed, seed };
    const unsigned lds_bcast = (unsigned)(size_t)lds + 16u * (threadIdx.x >> 6);
    if ((MODE >= 16 && MODE <= 20) || MODE >= 37) { for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f; __syncthreads(); }
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + i + threadIdx.x; p[i] = v2f{ a[i], a[i] + 1.f }; q[i] = i; }
    const float m = 1.0000001f, c = 1e-9f;
    if (MODE == 37 && !((threadIdx.x & 3) == 0 && (threadIdx.x & 15) < 12)) iters = 0;     // 12 of 64 lanes run the loop
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
            if (MODE == 11) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 12) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a[i]));
            if (MODE == 13) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(a[i]));
            if (MODE == 22) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 23) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (MODE == 24) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (MODE == 25) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 26) asm volatile("v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 27) asm volatile("v_add_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 28) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 29) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 30) asm volatile("v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 31) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == 32) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" :: "v"(a[i]), "v"(m) : "s20", "s21");
            if (MODE == 33) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            if (MODE == 34) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 35) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 36) asm volatile("v_min_f32 %0, 0x3f7d70a4, %0" : "+v"(a[i]));
            if (MODE == 37) asm volatile("ds_write_b32 %1, %0" :: "v"(a[i]), "v"(lds_addr12) : "memory", "exec");     // issued by 12 lanes (exec set outside)
            if (MODE == 38) asm volatile("ds_write_b128 %1, %0" :: "v"(q4), "v"(lds_addr4) : "memory");
            if (MODE == 14) { float y = a[(i + 1) % UNROLL]; asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(y)); a[(i + 1) % UNROLL] = y; }
            if (MODE == 15) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
            if (MODE == 16) asm volatile("ds_write_b32 %1, %0" :: "v"(a[i]), "v"(lds_addr) : "memory");                     // 64 distinct dwords
            if (MODE == 17) { float4v t; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lds_bcast) : "memory"); a[i] += t.x; }   // broadcast read + add
            if (MODE == 18) { float4v t; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lds_addr4) : "memory"); a[i] += t.x; }   // distinct 16-byte reads + add
            if (MODE == 19) asm volatile("ds_add_f32 %1, %0" :: "v"(a[i]), "v"(lds_addr) : "memory");                       // 64 distinct addresses
            if (MODE == 20) { if ((threadIdx.x & 3) == 0 && (threadIdx.x & 15) < 12) asm volatile("ds_add_f32 %1, %0" :: "v"(a[i]), "v"(lds_addr) : "memory"); }  // 12 lanes of 64
            if (MODE == 21) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(threadIdx.x), "v"(48u) : "vcc");
            if (MODE == 8) { float y = a[(i + 1) % UNROLL]; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(y)); a[(i + 1) % UNROLL] = y; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s += a[i] + p[i].x + p[i].y + (float)q[i];
    if ((MODE >= 16 && MODE <= 20) || MODE >= 37) s += lds[threadIdx.x];
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
    const double insts = (double)blocks * 4 /*waves*/ * ITERS * UNROLL * ((MODE == 2 || MODE == 6 || MODE == 9 || MODE == 10 || MODE == 17 || MODE == 18 || MODE == 31) ? 2 : 1);
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
    run<15>("v_add_f32 (asm)", d);
    run<11>("v_add_f32_dpp row_ror:8", d);
    run<12>("v_add_f32_dpp row_ror:8 bank_mask:0xc", d);
    run<13>("v_add_f32_dpp row_shr:4 bank_mask:0xa", d);
    run<14>("v_permlane16_swap (+s_nop 1)", d);
    run<16>("ds_write_b32 (64 distinct dwords)", d);
    run<17>("ds_read_b128 broadcast + v_add (2 instr)", d);
    run<18>("ds_read_b128 distinct + v_add (2 instr)", d);
    run<19>("ds_add_f32 (64 distinct)", d);
    run<20>("ds_add_f32 (12 of 64 lanes)", d);
    run<21>("v_mad_u64_u32", d);
    run<22>("v_add_f32_dpp quad_perm (asm, no bound_ctrl)", d);
    run<23>("v_add_f32_dpp quad_perm bound_ctrl:1", d);
    run<24>("v_add_f32_dpp row_ror:8 bound_ctrl:1", d);
    run<25>("v_add_f32_dpp row_mirror", d);
    run<26>("v_add_f32_dpp row_half_mirror", d);
    run<27>("v_add_f32_dpp row_shl:1", d);
    run<28>("v_add_f32_dpp row_ror:1", d);
    run<29>("v_add_f32_dpp row_bcast:15 row_mask:0xa", d);
    run<30>("v_add_f32_dpp wave_shr:1", d);
    run<31>("v_cmp_lt_f32 vcc + v_cndmask (2 instr)", d);
    run<32>("v_cmp_lt_f32 -> sgpr pair (e64)", d);
    run<33>("v_cndmask_b32 vcc", d);
    run<34>("v_min_f32 (asm)", d);
    run<35>("v_mul_f32 (asm)", d);
    run<36>("v_min_f32 with literal", d);
    run<37>("ds_write_b32 by 12 of 64 lanes", d);
    run<38>("ds_write_b128 (64 x 16 B)", d);
    return 0;
}
