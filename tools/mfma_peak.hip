// Micro-benchmark: sustained f16 matrix-core rate of this GPU with register operands only, for the two
// instruction shapes (16x16x32 and 32x32x16), at 2 and 4 waves per SIMD, over short (0.2 ms) and long runs,
// with operands that are smooth small numbers, all zero, or pseudo-random in (-1, 1) with two alternating A
// operands (the rate the chip sustains depends on how many operand bits toggle: its clock under load does).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ float operand(int data, unsigned int seed, float smooth)
{
    if (data == 0) return smooth;
    if (data == 1) return 0.0f;
    unsigned int x = seed * 2654435761u + 12345u;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    const float r = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23));  // (-1, 1)
    if (data == 2) return r;
    // data = 3, 4: the same with the lowest 4 / 7 of the 10 mantissa bits of the half-precision value cleared
    _Float16 h = (_Float16)r;
    unsigned short bits;
    __builtin_memcpy(&bits, &h, 2);
    bits &= data == 3 ? 0xfff0 : 0xff80;
    __builtin_memcpy(&h, &bits, 2);
    return (float)h;
}

__global__ __launch_bounds__(512) void mfma16(float *out, int iters, int data)
{
    f32x4 acc[8];
    half8 a[2], b[8];
    const unsigned int g = blockIdx.x * 512 + threadIdx.x;
    for (int i = 0; i < 8; i++) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < 8; j++) b[i][j] = (_Float16)operand(data, g * 64 + i * 8 + j, 0.001f * (threadIdx.x + i + j));
    }
    for (int k = 0; k < 2; k++)
        for (int j = 0; j < 8; j++)
            a[k][j] = (_Float16)operand(data, 0x9e3779b9u + g * 16 + k * 8 + j, 0.002f * (threadIdx.x + j));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++)
#pragma unroll
            for (int t = 0; t < 8; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rep & 1], b[(t + rep) & 7], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void mfma32(float *out, int iters)
{
    f32x16 acc[2];
    half8 a, b[8];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) b[i][j] = (_Float16)(0.001f * (threadIdx.x + i + j));
    for (int j = 0; j < 8; j++) a[j] = (_Float16)(0.002f * (threadIdx.x + j));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16; rep++)
#pragma unroll
            for (int t = 0; t < 2; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[(t + rep) & 7], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, sizeof(float) * 512 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int data = 0; data < 5; data++)
    for (int shape = 0; shape < 2; shape++)
        for (int wgPerCU : {1, 2})
            for (int iters : {400, 20000}) {
                if (shape == 1 && data != 0) continue;
                const int grid = 256 * wgPerCU;
                float best = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    hipEventRecord(e0);
                    if (shape == 0)
                        hipLaunchKernelGGL(mfma16, dim3(grid), dim3(512), 0, 0, out, iters, data);
                    else
                        hipLaunchKernelGGL(mfma32, dim3(grid), dim3(512), 0, 0, out, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                // per wave and iteration: 64 x 16384 flop (16x16x32) or 32 x 32768 flop (32x32x16)
                const double flops = (double)iters * 64.0 * 16384.0 * 8.0 * grid;
                printf("%s  %s operands  waves/SIMD %d  iters %5d: %.3f ms  %.0f TFLOP/s\n",
                       shape ? "32x32x16" : "16x16x32", data == 0 ? "smooth" : data == 1 ? "zero  " : data == 2 ? "random" : data == 3 ? "random, 6 mantissa bits" : "random, 3 mantissa bits", 2 * wgPerCU,
                       iters, best, flops / best / 1e9);
            }
    return 0;
}
