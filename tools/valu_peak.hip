// valu_peak.hip — measures the fp32 VALU issue ceiling of the box every "VALU-bound" statement in DESIGN.md is divided by.
// Independent v_fma_f32 chains (8 accumulators per lane, so a wave never waits on its own result), all CUs, 1 / 2 / 4 / 8 waves per SIMD;
// variants: plain v_fma_f32, one v_rcp_f32 in eight instructions (quarter-rate transcendental), packed v_pk_fma_f32, and a dependent
// chain (1 accumulator) for the issue latency of a single wave.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak > profiles/r03_valu_peak.txt
// Output: wave-instructions/s and lane-operations/s (64 lanes per wave instruction; an FMA counts as ONE lane-operation here, as
// SQ_THREAD_CYCLES_VALU-derived figures in bench.py do), and the implied cycles per wave instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 64;          // instructions per loop body per accumulator group

template <int MODE> __global__ void __launch_bounds__(256) k_valu(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 0.001f;
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {               // 8 independent fp32 FMA chains
#pragma unroll
            for (int u = 0; u < UNROLL / 8; u++)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (MODE == 1) {        // one v_rcp_f32 in eight
#pragma unroll
            for (int u = 0; u < UNROLL / 8; u++)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if constexpr (MODE == 2) {        // packed: 4 independent v_pk_fma_f32 chains on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; const f2 mm = {m, m}, cc = {c, c};
#pragma unroll
            for (int u = 0; u < UNROLL / 4; u++)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if constexpr (MODE == 3) {        // ONE dependent chain: what a single wave can issue by itself
#pragma unroll
            for (int u = 0; u < UNROLL; u++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        } else {                                 // all quarter-rate: independent v_rcp_f32
#pragma unroll
            for (int u = 0; u < UNROLL / 8; u++)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// Explicit-register variants (register numbers fixed in the asm, so that the VGPR bank of every operand is known: bank = index mod 4):
//   10  v_fma_f32 vK, vK, v40, v41   K = 8..15   accumulators cycle through the banks, multiplier / addend in banks 0 and 1
//   11  v_fma_f32 vK, vK, v40, v44               multiplier and addend in the SAME bank
//   12  v_fma_f32 vK, vK, vK, vK                 one register per instruction
//   13  v_mul_f32 vK, vK, v40                    two-source VOP2
//   14  v_fma_f32 vK, vK, s4, v41                multiplier in an SGPR
//   15  v_fma_f32 vK, vK, 0.5, 0.5               inline constants
//   16  v_fma_f32 with K = 8, 12, 16, .. (all accumulators in bank 0), v41 (bank 1), v42 (bank 2): no instruction reads one bank twice
//   17  as 10 but dependent pairs: each accumulator is used twice in a row
template <int MODE> __global__ void __launch_bounds__(256) k_regs(float* out, int iters, float seed)
{
    float r = 0.f;
    for (int it = 0; it < iters; it++) {
#define REP8(X) X X X X X X X X
        if constexpr (MODE == 10) asm volatile(REP8("v_fma_f32 v8, v8, v40, v41\n v_fma_f32 v9, v9, v40, v41\n v_fma_f32 v10, v10, v40, v41\n v_fma_f32 v11, v11, v40, v41\n"
                                                    "v_fma_f32 v12, v12, v40, v41\n v_fma_f32 v13, v13, v40, v41\n v_fma_f32 v14, v14, v40, v41\n v_fma_f32 v15, v15, v40, v41\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40","v41");
        else if constexpr (MODE == 11) asm volatile(REP8("v_fma_f32 v8, v8, v40, v44\n v_fma_f32 v9, v9, v40, v44\n v_fma_f32 v10, v10, v40, v44\n v_fma_f32 v11, v11, v40, v44\n"
                                                    "v_fma_f32 v12, v12, v40, v44\n v_fma_f32 v13, v13, v40, v44\n v_fma_f32 v14, v14, v40, v44\n v_fma_f32 v15, v15, v40, v44\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40","v44");
        else if constexpr (MODE == 12) asm volatile(REP8("v_fma_f32 v8, v8, v8, v8\n v_fma_f32 v9, v9, v9, v9\n v_fma_f32 v10, v10, v10, v10\n v_fma_f32 v11, v11, v11, v11\n"
                                                    "v_fma_f32 v12, v12, v12, v12\n v_fma_f32 v13, v13, v13, v13\n v_fma_f32 v14, v14, v14, v14\n v_fma_f32 v15, v15, v15, v15\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15");
        else if constexpr (MODE == 13) asm volatile(REP8("v_mul_f32 v8, v8, v40\n v_mul_f32 v9, v9, v40\n v_mul_f32 v10, v10, v40\n v_mul_f32 v11, v11, v40\n"
                                                    "v_mul_f32 v12, v12, v40\n v_mul_f32 v13, v13, v40\n v_mul_f32 v14, v14, v40\n v_mul_f32 v15, v15, v40\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40");
        else if constexpr (MODE == 14) asm volatile(REP8("v_fma_f32 v8, v8, s4, v41\n v_fma_f32 v9, v9, s4, v41\n v_fma_f32 v10, v10, s4, v41\n v_fma_f32 v11, v11, s4, v41\n"
                                                    "v_fma_f32 v12, v12, s4, v41\n v_fma_f32 v13, v13, s4, v41\n v_fma_f32 v14, v14, s4, v41\n v_fma_f32 v15, v15, s4, v41\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v41","s4");
        else if constexpr (MODE == 15) asm volatile(REP8("v_fma_f32 v8, v8, 0.5, 0.5\n v_fma_f32 v9, v9, 0.5, 0.5\n v_fma_f32 v10, v10, 0.5, 0.5\n v_fma_f32 v11, v11, 0.5, 0.5\n"
                                                    "v_fma_f32 v12, v12, 0.5, 0.5\n v_fma_f32 v13, v13, 0.5, 0.5\n v_fma_f32 v14, v14, 0.5, 0.5\n v_fma_f32 v15, v15, 0.5, 0.5\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15");
        else if constexpr (MODE == 16) asm volatile(REP8("v_fma_f32 v8, v8, v41, v42\n v_fma_f32 v12, v12, v41, v42\n v_fma_f32 v16, v16, v41, v42\n v_fma_f32 v20, v20, v41, v42\n"
                                                    "v_fma_f32 v24, v24, v41, v42\n v_fma_f32 v28, v28, v41, v42\n v_fma_f32 v32, v32, v41, v42\n v_fma_f32 v36, v36, v41, v42\n") ::: "v8","v12","v16","v20","v24","v28","v32","v36","v41","v42");
        else if constexpr (MODE == 18) asm volatile(REP8("v_fmaak_f32 v8, v8, v40, 0x3f7fbe77\n v_fmaak_f32 v9, v9, v40, 0x3f7fbe77\n v_fmaak_f32 v10, v10, v40, 0x3f7fbe77\n v_fmaak_f32 v11, v11, v40, 0x3f7fbe77\n"
                                                    "v_fmaak_f32 v12, v12, v40, 0x3f7fbe77\n v_fmaak_f32 v13, v13, v40, 0x3f7fbe77\n v_fmaak_f32 v14, v14, v40, 0x3f7fbe77\n v_fmaak_f32 v15, v15, v40, 0x3f7fbe77\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40");
        else if constexpr (MODE == 19) asm volatile(REP8("v_mul_f32_e32 v8, s4, v8\n v_mul_f32_e32 v9, s4, v9\n v_mul_f32_e32 v10, s4, v10\n v_mul_f32_e32 v11, s4, v11\n"
                                                    "v_mul_f32_e32 v12, s4, v12\n v_mul_f32_e32 v13, s4, v13\n v_mul_f32_e32 v14, s4, v14\n v_mul_f32_e32 v15, s4, v15\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","s4");
        else if constexpr (MODE == 20) asm volatile(REP8("v_mul_f32_e32 v8, 0x3f7fbe77, v8\n v_mul_f32_e32 v9, 0x3f7fbe77, v9\n v_mul_f32_e32 v10, 0x3f7fbe77, v10\n v_mul_f32_e32 v11, 0x3f7fbe77, v11\n"
                                                    "v_mul_f32_e32 v12, 0x3f7fbe77, v12\n v_mul_f32_e32 v13, 0x3f7fbe77, v13\n v_mul_f32_e32 v14, 0x3f7fbe77, v14\n v_mul_f32_e32 v15, 0x3f7fbe77, v15\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15");
        else if constexpr (MODE == 21) asm volatile(REP8("v_perm_b32 v8, s4, v8, v40\n v_perm_b32 v9, s4, v9, v40\n v_perm_b32 v10, s4, v10, v40\n v_perm_b32 v11, s4, v11, v40\n"
                                                    "v_perm_b32 v12, s4, v12, v40\n v_perm_b32 v13, s4, v13, v40\n v_perm_b32 v14, s4, v14, v40\n v_perm_b32 v15, s4, v15, v40\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40","s4");
        else if constexpr (MODE == 22) asm volatile(REP8("v_perm_b32 v8, v41, v8, v40\n v_perm_b32 v9, v41, v9, v40\n v_perm_b32 v10, v41, v10, v40\n v_perm_b32 v11, v41, v11, v40\n"
                                                    "v_perm_b32 v12, v41, v12, v40\n v_perm_b32 v13, v41, v13, v40\n v_perm_b32 v14, v41, v14, v40\n v_perm_b32 v15, v41, v15, v40\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40","v41");
        else if constexpr (MODE == 23) asm volatile(REP8("v_cndmask_b32_e64 v8, v8, v40, s[4:5]\n v_cndmask_b32_e64 v9, v9, v40, s[4:5]\n v_cndmask_b32_e64 v10, v10, v40, s[4:5]\n v_cndmask_b32_e64 v11, v11, v40, s[4:5]\n"
                                                    "v_cndmask_b32_e64 v12, v12, v40, s[4:5]\n v_cndmask_b32_e64 v13, v13, v40, s[4:5]\n v_cndmask_b32_e64 v14, v14, v40, s[4:5]\n v_cndmask_b32_e64 v15, v15, v40, s[4:5]\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40","s4","s5");
        else if constexpr (MODE == 24) asm volatile(REP8("v_cmp_lt_f32_e64 s[4:5], v8, v40\n v_cmp_lt_f32_e64 s[6:7], v9, v40\n v_cmp_lt_f32_e64 s[4:5], v10, v40\n v_cmp_lt_f32_e64 s[6:7], v11, v40\n"
                                                    "v_cmp_lt_f32_e64 s[4:5], v12, v40\n v_cmp_lt_f32_e64 s[6:7], v13, v40\n v_cmp_lt_f32_e64 s[4:5], v14, v40\n v_cmp_lt_f32_e64 s[6:7], v15, v40\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40","s4","s5","s6","s7");
        else if constexpr (MODE == 25) asm volatile(REP8("v_max3_f32 v8, v8, v40, v41\n v_max3_f32 v9, v9, v40, v41\n v_max3_f32 v10, v10, v40, v41\n v_max3_f32 v11, v11, v40, v41\n"
                                                    "v_max3_f32 v12, v12, v40, v41\n v_max3_f32 v13, v13, v40, v41\n v_max3_f32 v14, v14, v40, v41\n v_max3_f32 v15, v15, v40, v41\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v40","v41");
        else if constexpr (MODE == 17) asm volatile(REP8("v_fma_f32 v8, v8, v40, v41\n v_fma_f32 v8, v8, v40, v41\n v_fma_f32 v10, v10, v40, v41\n v_fma_f32 v10, v10, v40, v41\n"
                               "v_fma_f32 v12, v12, v40, v41\n v_fma_f32 v12, v12, v40, v41\n v_fma_f32 v14, v14, v40, v41\n v_fma_f32 v14, v14, v40, v41\n") ::: "v8","v10","v12","v14","v40","v41");
        asm volatile("v_mov_b32 %0, v8" : "=v"(r));
    }
    out[blockIdx.x * 256 + threadIdx.x] = r + seed;
}
template <int MODE> static double run_regs(float* out, int blocks, int iters, int reps)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_regs<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
    CHECK(hipDeviceSynchronize());
    std::vector<double> ms;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_regs<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

template <int MODE> static double run(float* out, int blocks, int iters, int reps)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);      // warm-up
    CHECK(hipDeviceSynchronize());
    std::vector<double> ms;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double clockGHz = p.clockRate * 1e-6;
    printf("# device %s, %d CUs, clockRate %.3f GHz (hipDeviceProp; the sustained clock under load may be lower)\n", p.gcnArchName, cus, clockGHz);
    printf("# a block = 256 threads = 4 wavefronts = one per SIMD; grid = CUs x (waves per SIMD) blocks; median of 7 launches\n");
    printf("# %-34s %5s %10s %14s %14s %12s\n", "variant", "w/SIMD", "ms", "Gwave-inst/s", "Tlane-ops/s", "cyc/inst/SIMD");
    float* out; CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
    const int iters = 4096;
    const char* names[5] = {"v_fma_f32 x8 independent", "7 v_fma_f32 + 1 v_rcp_f32", "v_pk_fma_f32 x4 independent", "v_fma_f32 dependent chain", "v_rcp_f32 x8 independent"};
    for (int mode = 0; mode < 5; mode++) {
        for (int w : {1, 2, 4, 8}) {
            const int blocks = cus * w;
            double ms = 0;
            switch (mode) { case 0: ms = run<0>(out, blocks, iters, 7); break; case 1: ms = run<1>(out, blocks, iters, 7); break;
                            case 2: ms = run<2>(out, blocks, iters, 7); break; case 3: ms = run<3>(out, blocks, iters, 7); break; default: ms = run<4>(out, blocks, iters, 7); }
            const double instPerWave = (double)iters * UNROLL;                  // wave instructions of the measured kind per wavefront (loop overhead: 3 scalar per 64)
            const double waves = (double)blocks * 4;
            const double winst = instPerWave * waves / (ms * 1e-3);
            const double simds = cus * 4.0;
            printf("  %-34s %5d %10.3f %14.1f %14.2f %12.2f\n", names[mode], w, ms, winst * 1e-9, winst * 64 * 1e-12 * (mode == 2 ? 2 : 1), clockGHz * 1e9 * simds / winst);
        }
    }
    const char* rnames[16] = {"fma vK,vK,v40,v41 (banks k,0,1)", "fma vK,vK,v40,v44 (banks k,0,0)", "fma vK,vK,vK,vK", "v_mul_f32 vK,vK,v40 (VOP2)",
                             "fma vK,vK,s4,v41 (SGPR)", "fma vK,vK,0.5,0.5 (inline)", "fma v(4k),..,v41,v42 (banks 0,1,2)", "fma pairs (dependent in twos)",
                             "v_fmaak_f32 vK,vK,v40,literal", "v_mul_f32_e32 vK,s4,vK (VOP2 SGPR)", "v_mul_f32_e32 vK,literal,vK", "v_perm_b32 vK,s4,vK,v40", "v_perm_b32 vK,v41,vK,v40",
                             "v_cndmask_b32 vK,vK,v40,s[4:5]", "v_cmp_lt_f32 s[..],vK,v40", "v_max3_f32 vK,vK,v40,v41"};
    for (int mode = 10; mode < 26; mode++) {
        for (int w : {1, 2, 4, 8}) {
            const int blocks = cus * w;
            double ms = 0;
            switch (mode) { case 10: ms = run_regs<10>(out, blocks, iters, 7); break; case 11: ms = run_regs<11>(out, blocks, iters, 7); break; case 12: ms = run_regs<12>(out, blocks, iters, 7); break;
                            case 13: ms = run_regs<13>(out, blocks, iters, 7); break; case 14: ms = run_regs<14>(out, blocks, iters, 7); break; case 15: ms = run_regs<15>(out, blocks, iters, 7); break;
                            case 16: ms = run_regs<16>(out, blocks, iters, 7); break; case 17: ms = run_regs<17>(out, blocks, iters, 7); break;
                            case 18: ms = run_regs<18>(out, blocks, iters, 7); break; case 19: ms = run_regs<19>(out, blocks, iters, 7); break; case 20: ms = run_regs<20>(out, blocks, iters, 7); break;
                            case 21: ms = run_regs<21>(out, blocks, iters, 7); break; case 22: ms = run_regs<22>(out, blocks, iters, 7); break; case 23: ms = run_regs<23>(out, blocks, iters, 7); break;
                            case 24: ms = run_regs<24>(out, blocks, iters, 7); break; default: ms = run_regs<25>(out, blocks, iters, 7); }
            const double winst = (double)iters * 64 * blocks * 4 / (ms * 1e-3);
            printf("  %-34s %5d %10.3f %14.1f %14.2f %12.2f\n", rnames[mode - 10], w, ms, winst * 1e-9, winst * 64 * 1e-12, clockGHz * 1e9 * cus * 4.0 / winst);
        }
    }
    printf("# lane-ops: one per lane per instruction (an FMA = 1; packed = 2).  fp32 FLOP/s = 2 x that for FMA.\n");
    CHECK(hipFree(out));
    return 0;
}
