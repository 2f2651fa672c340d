// Developer microbenchmark (GPU box): fp64 FMA issue rate and dependent-chain latency on this part, to turn
// "cycles" into microseconds when reading SQ counters.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void fma_kernel(double* out, double a, double b, int iters) {
    double v[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) v[k] = threadIdx.x * 1e-9 + k;
    for (int i = 0; i < iters; i += 32) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) v[k] = fma(v[k], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) s += v[k];
    if (s == 12345.678) out[0] = s;
}

template <int CHAINS>
static void run(int wavesPerSimd, int iters) {
    double* out; hipMalloc(&out, 8);
    const int blocks = 256 * 4 * wavesPerSimd;            // one wave per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, 1.0000001, 1e-9, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, 1.0000001, 1e-9, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fmasPerWave = (double)iters * CHAINS;
    const double nsPerFmaPerWave = ms * 1e6 / fmasPerWave;           // wall ns per wave-level FMA of one wave
    const double tflops = 2.0 * fmasPerWave * 64 * blocks / (ms * 1e-3) / 1e12;
    printf("chains %d waves/SIMD %d: %.3f ms, %.2f ns per dependent step, %.1f TFLOP/s fp64 (SIMD issue: %.2f ns per wave-FMA)\n",
           CHAINS, wavesPerSimd, ms, nsPerFmaPerWave * CHAINS, tflops, ms * 1e6 / (fmasPerWave * wavesPerSimd));
    hipFree(out);
}

int main() {
    run<1>(1, 204800);     // pure latency: one wave per SIMD, one dependent chain
    run<2>(1, 102400);
    run<4>(1, 51200);
    run<8>(1, 51200);      // one wave, 8 independent chains: issue rate of a single wave
    run<1>(2, 102400);
    run<1>(4, 102400);
    run<1>(6, 102400);
    run<1>(8, 102400);     // 8 waves x 1 chain
    run<8>(8, 20480);      // throughput
    return 0;
}
