// Effective shader clock under sustained FP64 load, by kernel length: the shader-clock counter against the
// constant 100 MHz counter, per launch.   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o build_variants/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void burn(long long iters, double *sink, unsigned long long *out)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4.0, a5 = 5.0, a6 = 6.0, a7 = 7.0;
    const double m = 1.0000001, b = 1e-9;
    for (long long i = 0; i < iters; ++i) {
        a0 = __builtin_fma(a0, m, b); a1 = __builtin_fma(a1, m, b); a2 = __builtin_fma(a2, m, b); a3 = __builtin_fma(a3, m, b);
        a4 = __builtin_fma(a4, m, b); a5 = __builtin_fma(a5, m, b); a6 = __builtin_fma(a6, m, b); a7 = __builtin_fma(a7, m, b);
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) sink[0] = a0;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main()
{
    const int blocks = 256;
    double *sink; unsigned long long *out;
    hipMalloc(&sink, 8); hipMalloc(&out, blocks * 16);
    std::vector<unsigned long long> h(blocks * 2);
    for (int rep = 0; rep < 2; ++rep)
        for (long long iters : {20000ll, 100000ll, 400000ll, 1600000ll, 400000ll, 100000ll, 20000ll}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(burn, dim3(blocks), dim3(1024), 0, 0, iters, sink, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
            double c = 0, w = 0;
            for (int i = 0; i < blocks; ++i) { c += (double)h[i * 2]; w += (double)h[i * 2 + 1]; }
            // 8 FMAs per iteration per wave, 4 waves per SIMD
            printf("iters %8lld: %.2f ms, shader clock %.0f MHz, %.2f cycles per FP64 FMA per SIMD\n", iters, ms, c / w * 100.0,
                   (c / blocks) / ((double)iters * 8 * 4));
        }
    return 0;
}
