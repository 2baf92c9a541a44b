// mfma_rate16.hip -- sustained rate of the two fp16 matrix instructions the product kernels can be built on, registers only:
// v_mfma_f32_16x16x32_f16 (what k_gemm8p_tn / k_gemm9_tn issue: 16 cycles each) against v_mfma_f32_32x32x16_f16 (32 cycles each).
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate16 mfma_rate16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// DATA 0: the same smooth operands in every instruction (few bits toggle between consecutive instructions);
// DATA 1: four pseudo-random operand pairs per lane, rotated from instruction to instruction (what a real product feeds the unit)
template <int KIND, int NACC, int DATA>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters) {
    f16v acc32[KIND == 0 ? NACC : 1];
    f4v acc16[KIND == 1 ? NACC : 1];
    for (int a = 0; a < (KIND == 0 ? NACC : 1); ++a) for (int e = 0; e < 16; ++e) acc32[a][e] = 0.f;
    for (int a = 0; a < (KIND == 1 ? NACC : 1); ++a) for (int e = 0; e < 4; ++e) acc16[a][e] = 0.f;
    h8 av[4], bv[4];
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) {
            if (DATA == 0) { av[j][e] = (_Float16)(threadIdx.x * 0.001f); bv[j][e] = (_Float16)(e * 0.01f); }
            else {
                x = x * 1664525u + 1013904223u; av[j][e] = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.001f);
                x = x * 1664525u + 1013904223u; bv[j][e] = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.0001f);
            }
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (KIND == 0) acc32[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[a & 3], bv[(a + (a >> 2)) & 3], acc32[a], 0, 0, 0);
            else acc16[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[a & 3], bv[(a + (a >> 2)) & 3], acc16[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < (KIND == 0 ? NACC : 1); ++a) s += acc32[a][0];
    for (int a = 0; a < (KIND == 1 ? NACC : 1); ++a) s += acc16[a][0];
    if (s == 12345.f) out[0] = s;
}

template <int KIND, int NACC, int DATA>
static double run(int threads) {
    float* d; hipMalloc(&d, 4);
    const int iters = 60000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<KIND, NACC, DATA>), dim3(256), dim3(threads), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<KIND, NACC, DATA>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    const double flops_per = KIND == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    return flops_per * NACC * iters * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
}

int main() {
    printf("instruction, independent accumulators per wave, waves per SIMD, operands -> TFLOP/s (chip)\n");
    printf("v_mfma_f32_32x32x16_f16  8 acc  1 wave/SIMD  smooth %.0f  random %.0f\n", run<0, 8, 0>(256), run<0, 8, 1>(256));
    printf("v_mfma_f32_32x32x16_f16  8 acc  2 waves/SIMD smooth %.0f  random %.0f\n", run<0, 8, 0>(512), run<0, 8, 1>(512));
    printf("v_mfma_f32_16x16x32_f16  8 acc  1 wave/SIMD  smooth %.0f  random %.0f\n", run<1, 8, 0>(256), run<1, 8, 1>(256));
    printf("v_mfma_f32_16x16x32_f16  8 acc  2 waves/SIMD smooth %.0f  random %.0f\n", run<1, 8, 0>(512), run<1, 8, 1>(512));
    printf("v_mfma_f32_16x16x32_f16 32 acc  2 waves/SIMD smooth %.0f  random %.0f\n", run<1, 32, 0>(512), run<1, 32, 1>(512));
    return 0;
}
