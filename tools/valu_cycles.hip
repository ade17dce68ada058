// Cycles per wave64 instruction per SIMD, measured in SHADER CYCLES (s_memtime brackets the instruction stream inside every
// wave), not derived from a kernel's wall time and an assumed clock:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_cycles.hip -o /tmp/valu_cycles && /tmp/valu_cycles
// tools/valu_microbench.hip (round 3) timed whole launches with HIP events and divided by 2.4 GHz: a bare v_fma_f32 read 2.8
// cycles against the guide's 2, and v_min_f32 8.5 / 5.0 depending on how it was written.  Here every wave runs ITER rounds of
// 8 instructions on 8 INDEPENDENT accumulators (no dependent issue stalls) between two s_memtime reads; W waves share a SIMD
// (W = 1, 2, 4, 8); cycles per instruction per SIMD = (longest wave's ticks) / (W x instructions per wave).  The launch's wall
// time is printed beside it: ticks / wall = the clock the shader ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

constexpr int ITER = 32768;      // ~0.6 ms per wave: every workgroup of a launch is resident at once (a short kernel's
                                 // later workgroups start after its first ones ended: fewer waves share a SIMD than launched)

#define OP8(STR) asm volatile(STR(0) STR(1) STR(2) STR(3) STR(4) STR(5) STR(6) STR(7) \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m), "v"(c) : "vcc", "s20", "s21")
#define S_FMA(i)   "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define S_ADD(i)   "v_add_f32 %" #i ", %" #i ", %8\n\t"
#define S_MUL(i)   "v_mul_f32 %" #i ", %" #i ", %8\n\t"
#define S_MIN(i)   "v_min_f32 %" #i ", %" #i ", %8\n\t"
#define S_MINL(i)  "v_min_f32 %" #i ", 0x3f7d70a4, %" #i "\n\t"
#define S_EXP(i)   "v_exp_f32 %" #i ", %" #i "\n\t"
#define S_RCP(i)   "v_rcp_f32 %" #i ", %" #i "\n\t"
#define S_CMPS(i)  "v_cmp_lt_f32 s[20:21], %" #i ", %8\n\t"
#define S_CMPV(i)  "v_cmp_lt_f32 vcc, %" #i ", %8\n\t"
#define S_CND(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define S_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define S_CND0(i)  "v_cndmask_b32_e64 %" #i ", 0, %" #i ", s[20:21]\n\t"
#define S_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define S_CNDE64V(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n\t"
#define S_CMP_MOV_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n\tv_mov_b32 %" #i ", %" #i "\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define S_CMP_CND_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\tv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define S_DPPQ(i)  "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define S_DPPR(i)  "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define S_DPPB(i)  "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
#define S_MOV(i)   "v_mov_b32 %" #i ", %8\n\t"
#define S_SUB(i)   "v_sub_f32 %" #i ", 1.0, %" #i "\n\t"

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* ticks, float* sink, float seed)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    const float m = 1.0000001f, c = 1e-9f;
    // a lane mask for the selects that read an SGPR pair: every other lane
    asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555\n\ts_mov_b64 vcc, s[20:21]" ::: "s20", "s21", "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
        if (MODE == 0) OP8(S_FMA);
        if (MODE == 1) OP8(S_ADD);
        if (MODE == 2) OP8(S_MUL);
        if (MODE == 3) OP8(S_MIN);
        if (MODE == 4) OP8(S_MINL);
        if (MODE == 5) OP8(S_EXP);
        if (MODE == 6) OP8(S_RCP);
        if (MODE == 7) OP8(S_CMPS);
        if (MODE == 8) OP8(S_CMPV);
        if (MODE == 9) OP8(S_CND);
        if (MODE == 10) OP8(S_DPPQ);
        if (MODE == 11) OP8(S_DPPR);
        if (MODE == 12) OP8(S_DPPB);
        if (MODE == 13) OP8(S_MOV);
        if (MODE == 14) OP8(S_SUB);
        if (MODE == 17) OP8(S_CND64);
        if (MODE == 18) OP8(S_CND0);
        if (MODE == 19) OP8(S_CMPCND);
        if (MODE == 20) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(m) : "vcc"); OP8(S_CND); }       // vcc from ONE VALU compare
        if (MODE == 21) OP8(S_CNDE64V);        // the same select in the VOP3 encoding, vcc named explicitly
        if (MODE == 22) OP8(S_CMP_MOV_CND);    // compare -> unrelated VALU instruction -> select
        if (MODE == 23) OP8(S_CMP_CND_CND);    // compare -> two selects on the same vcc
        if (MODE == 15) {      // lane swaps: four independent pairs
            asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        }
        if (MODE == 16) {
            asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                         "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned long long* dticks, float* dsink)
{
    printf("%-34s", name);
    for (int W = 1; W <= 8; W *= 2) {
        // W waves per SIMD: blocks of 256 threads (one wave per SIMD each), W blocks per CU
        const int blocks = 256 * W;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dticks, dsink, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dticks, dsink, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t(blocks * 4);
        hipMemcpy(t.data(), dticks, t.size() * 8, hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        const double med = (double)t[t.size() / 2], n = (double)ITER * 8;
        // Two readings.  (a) one wave's own clock (s_memtime ticks between its first and last instruction) / its instructions:
        // what ONE wave sees -- only a per-SIMD figure if the W waves of a SIMD ran side by side, which the launch does not
        // guarantee (the 256 * W blocks are not co-resident from start to end; round 5 divided (a) by W and printed
        // 1.21 "cycles per instruction and SIMD" for v_fma_f32, below the 2-cycle floor).  (b) the LAUNCH: every SIMD executes
        // W x n instructions of this kind whatever the residency, so launch time x clock / (W n) is the SIMD's average issue
        // interval -- the throughput figure, valid for every W (it includes the launch's ramp: read it at W >= 2).
        const double clock_ghz = 2.4;
        printf(" | W=%d launch %6.1f us -> %5.2f cyc/instr/SIMD; one wave %5.2f cyc/instr by its own clock", W, ms * 1e3,
               ms * 1e-3 * clock_ghz * 1e9 / (n * W), med / n);
    }
    printf("\n");
}

int main()
{
    unsigned long long* dticks; float* dsink;
    hipMalloc(&dticks, 8 * 256 * 8 * 4); hipMalloc(&dsink, 4);
    run<0>("v_fma_f32", dticks, dsink);
    run<1>("v_add_f32", dticks, dsink);
    run<2>("v_mul_f32", dticks, dsink);
    run<14>("v_sub_f32 (1.0 - x)", dticks, dsink);
    run<13>("v_mov_b32", dticks, dsink);
    run<3>("v_min_f32 (vgpr operand)", dticks, dsink);
    run<4>("v_min_f32 (literal operand)", dticks, dsink);
    run<7>("v_cmp_lt_f32 -> sgpr pair", dticks, dsink);
    run<8>("v_cmp_lt_f32 -> vcc", dticks, dsink);
    run<9>("v_cndmask_b32 (vcc)", dticks, dsink);
    run<17>("v_cndmask_b32_e64 (sgpr pair)", dticks, dsink);
    run<18>("v_cndmask_b32_e64 0, x (sgpr pair)", dticks, dsink);
    run<19>("v_cmp_lt_f32 vcc + v_cndmask (2 instr)", dticks, dsink);
    run<20>("1 v_cmp -> vcc, then 8 v_cndmask (vcc)", dticks, dsink);
    run<21>("v_cndmask_b32_e64 ..., vcc (VOP3)", dticks, dsink);
    run<22>("v_cmp, v_mov, v_cndmask (3 instr)", dticks, dsink);
    run<23>("v_cmp, v_cndmask, v_cndmask (3 instr)", dticks, dsink);
    run<5>("v_exp_f32", dticks, dsink);
    run<6>("v_rcp_f32", dticks, dsink);
    run<10>("v_add_f32_dpp quad_perm", dticks, dsink);
    run<11>("v_add_f32_dpp row_ror:8", dticks, dsink);
    run<12>("v_add_f32_dpp row_ror:8 bank 0xc", dticks, dsink);
    run<15>("v_permlane32_swap_b32", dticks, dsink);
    run<16>("v_permlane16_swap_b32", dticks, dsink);
    return 0;
}
