// Issue rates of the vector instructions the render kernels are made of, measured on the whole chip:
// every SIMD runs W waves, each a long run of independent instructions of one kind (8 accumulators per wave,
// so dependencies never stall).  Prints G wave-instructions/s chip-wide per kind and the cycles per instruction
// and SIMD that rate means at the clock measured in the same kernel (s_memtime ticks per s_memrealtime tick).
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(int iters, double *sink, unsigned long long *clk)
{
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double b = 1.0000001, c = 1e-9;
    float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    const float fb = 1.0000001f, fc = 1e-9f;
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define OPS(INS, A0, A1, A2, A3, A4, A5, A6, A7, B, C, RC)                                                        \
    asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n"     \
                 INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n"     \
                 : "+" RC(A0), "+" RC(A1), "+" RC(A2), "+" RC(A3), "+" RC(A4), "+" RC(A5), "+" RC(A6), "+" RC(A7)   \
                 : RC(B), RC(C))
#define OPS2(INS, A0, A1, A2, A3, A4, A5, A6, A7, B, RC)                                                          \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"                     \
                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n"                     \
                 : "+" RC(A0), "+" RC(A1), "+" RC(A2), "+" RC(A3), "+" RC(A4), "+" RC(A5), "+" RC(A6), "+" RC(A7)   \
                 : RC(B))
        if (KIND == 0) { REP8(OPS("v_fma_f64", a0, a1, a2, a3, a4, a5, a6, a7, b, c, "v");) }
        if (KIND == 1) { REP8(OPS2("v_add_f64", a0, a1, a2, a3, a4, a5, a6, a7, c, "v");) }
        if (KIND == 2) { REP8(OPS2("v_mul_f64", a0, a1, a2, a3, a4, a5, a6, a7, b, "v");) }
        if (KIND == 3) { REP8(OPS("v_fma_f32", f0, f1, f2, f3, f4, f5, f6, f7, fb, fc, "v");) }
        if (KIND == 4) { REP8(OPS2("v_add_u32", i0, i1, i2, i3, i4, i5, i6, i7, i0, "v");) }
        if (KIND == 5) { REP8(OPS2("v_max_f32", f0, f1, f2, f3, f4, f5, f6, f7, fb, "v");) }
        if (KIND == 6) {
            REP8(asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3\n"
                              "v_rcp_f64 %4, %4\nv_rcp_f64 %5, %5\nv_rcp_f64 %6, %6\nv_rcp_f64 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        if (KIND == 8) {
            unsigned long long m = 0x5555aaaa5555aaaaull ^ (unsigned long long)iters;
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %2, %2, %8, %9\n"
                              "v_cndmask_b32_e64 %3, %3, %8, %9\nv_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %5, %5, %8, %9\n"
                              "v_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %7, %7, %8, %9\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it), "s"(m));)
        }
        if (KIND == 14) {
            asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(i0), "v"(i1) : "vcc");
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\nv_cndmask_b32_e64 %1, %1, %8, vcc\nv_cndmask_b32_e64 %2, %2, %8, vcc\n"
                              "v_cndmask_b32_e64 %3, %3, %8, vcc\nv_cndmask_b32_e64 %4, %4, %8, vcc\nv_cndmask_b32_e64 %5, %5, %8, vcc\n"
                              "v_cndmask_b32_e64 %6, %6, %8, vcc\nv_cndmask_b32_e64 %7, %7, %8, vcc\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it) : "vcc");)
        }
        if (KIND == 15) {
            asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(i0), "v"(i1) : "vcc");
            REP8(asm volatile("v_cndmask_b32 %0, %1, %8, vcc\nv_cndmask_b32 %1, %2, %8, vcc\nv_cndmask_b32 %2, %3, %8, vcc\n"
                              "v_cndmask_b32 %3, %4, %8, vcc\nv_cndmask_b32 %4, %5, %8, vcc\nv_cndmask_b32 %5, %6, %8, vcc\n"
                              "v_cndmask_b32 %6, %7, %8, vcc\nv_cndmask_b32 %7, %0, %8, vcc\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it) : "vcc");)
        }
        if (KIND == 16) {
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\nv_add_co_u32 %1, vcc, %1, %8\nv_add_co_u32 %2, vcc, %2, %8\n"
                              "v_add_co_u32 %3, vcc, %3, %8\nv_add_co_u32 %4, vcc, %4, %8\nv_add_co_u32 %5, vcc, %5, %8\n"
                              "v_add_co_u32 %6, vcc, %6, %8\nv_add_co_u32 %7, vcc, %7, %8\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(it) : "vcc");)
        }
        if (KIND == 9) { REP8(OPS("v_max3_f32", f0, f1, f2, f3, f4, f5, f6, f7, fb, fc, "v");) }
        if (KIND == 10) { REP8(OPS2("v_mul_f32", f0, f1, f2, f3, f4, f5, f6, f7, fb, "v");) }
        if (KIND == 11) {
            REP8(asm volatile("v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));)
        }
        if (KIND == 12) {
            REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %1\nv_cmp_lt_f64 vcc, %1, %2\nv_cmp_lt_f64 vcc, %2, %3\nv_cmp_lt_f64 vcc, %3, %4\n"
                              "v_cmp_lt_f64 vcc, %4, %5\nv_cmp_lt_f64 vcc, %5, %6\nv_cmp_lt_f64 vcc, %6, %7\nv_cmp_lt_f64 vcc, %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
        }
        if (KIND == 13) {
            REP8(asm volatile("v_sqrt_f64 %0, %0\nv_sqrt_f64 %1, %1\nv_sqrt_f64 %2, %2\nv_sqrt_f64 %3, %3\n"
                              "v_sqrt_f64 %4, %4\nv_sqrt_f64 %5, %5\nv_sqrt_f64 %6, %6\nv_sqrt_f64 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        if (KIND == 7) {
            asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(i0), "v"(i1) : "vcc");
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\n"
                              "v_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\n"
                              "v_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0) : "vcc");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 +
                                                 i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int KIND>
static void run(const char *name, int waves_per_simd, int cus)
{
    const int blocks = cus * waves_per_simd, iters = 4000;      // a 256-thread block puts one wave on each SIMD of a CU
    double *sink; unsigned long long *clk;
    hipMalloc(&sink, sizeof(double) * blocks * 256);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<KIND><<<blocks, 256>>>(10, sink, clk);
    hipDeviceSynchronize();
    float best = 1e30f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k_rate<KIND><<<blocks, 256>>>(iters, sink, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
    }
    const double insts = (double)blocks * 4 * iters * 64;       // wave-instructions
    const double rate = insts / (best * 1e-3) / 1e9;
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0;      // s_memrealtime ticks at 100 MHz
    const double cyc = ghz * 1e9 * cus * 4 / (rate * 1e9);
    printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"g_wave_instr_per_s\": %.1f, \"clock_ghz\": %.3f, \"cycles_per_instr_per_simd\": %.2f}\n",
           name, waves_per_simd, rate, ghz, cyc);
    hipFree(sink); hipFree(clk);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int w : {1, 4}) {
        run<0>("v_fma_f64", w, cus); run<1>("v_add_f64", w, cus); run<2>("v_mul_f64", w, cus); run<3>("v_fma_f32", w, cus);
        run<4>("v_add_u32", w, cus); run<5>("v_max_f32", w, cus); run<6>("v_rcp_f64", w, cus); run<7>("v_cndmask_b32 vcc", w, cus);
        run<8>("v_cndmask_b32_e64 sgpr", w, cus); run<9>("v_max3_f32", w, cus); run<10>("v_mul_f32", w, cus); run<11>("v_mov_b32", w, cus);
        run<12>("v_cmp_lt_f64", w, cus); run<13>("v_sqrt_f64", w, cus); run<14>("v_cndmask_b32_e64 vcc", w, cus);
        run<15>("v_cndmask_b32 vcc, rotating regs", w, cus); run<16>("v_add_co_u32 vcc", w, cus);
    }
    return 0;
}
