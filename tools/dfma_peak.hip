// Micro-benchmark: sustained fp64 FMA rate of the vector pipe on this GPU
// (register operands only, one SGPR multiplicand as in corr_kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -o dfma_peak tools/dfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC>
__global__ __launch_bounds__(256) void dfma(double *out, const double *coef, int iters)
{
    double acc[NACC];
    double x[NACC];
    for (int i = 0; i < NACC; i++) {
        acc[i] = threadIdx.x * 1e-9 + i;
        x[i] = 1.0 + (threadIdx.x + i) * 1e-12;
    }
    for (int it = 0; it < iters; it++) {
        const double c0 = coef[(it & 7) * 2], c1 = coef[(it & 7) * 2 + 1];  // scalar loads
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = fma(rep & 1 ? c1 : c0, x[(i + rep) % NACC], acc[i]);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    const int iters = 20000;
    double *out, *coef;
    hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipMalloc(&coef, sizeof(double) * 16);
    std::vector<double> h(16, 1e-9);
    hipMemcpy(coef, h.data(), sizeof(double) * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocksPerCU : {1, 2, 4}) {
        const int grid = 256 * blocksPerCU;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(dfma<16>, dim3(grid), dim3(256), 0, 0, out, coef, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 16 * 8 * (double)iters * 256.0 * grid;
            printf("blocks/CU %d (waves/SIMD %d): %.3f ms  %.2f TFLOP/s fp64\n", blocksPerCU, blocksPerCU, ms,
                   flops / ms / 1e9);
        }
    }
    return 0;
}
