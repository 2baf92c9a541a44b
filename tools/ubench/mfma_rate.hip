// mfma_rate.hip -- sustained matrix-pipe rates on MI355X, registers only (no memory traffic): the fp16 instruction k_scan_wide
// uses (v_mfma_f32_32x32x16_f16) against the block-scaled fp8 one BASELINE configs[4] names (v_mfma_scale_f32_32x32x64_f8f6f4).
// An fp8 query cannot carry the exactness certificate (e4m3 rounds at 2^-4); a hi + lo split of the query (q = q_hi + q_lo, both
// e4m3 with a block scale) needs TWO fp8 MFMAs per 64 k -- this prints what that costs against FOUR fp16 MFMAs per 64 k.
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
    f16v acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    h8 ah, bh;
    i8v ai, bi;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(threadIdx.x * 0.001f); bh[e] = (_Float16)(e * 0.01f); ai[e] = threadIdx.x + e; bi[e] = e; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ai, bi, acc[a], 0, 0, 0, 127, 0, 127);   // fp8 e4m3 x e4m3, scales 2^0
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    if (s == 12345.f) out[0] = s;
}

template <int KIND, int NACC>
static double run(int waves_per_simd) {
    float* d; hipMalloc(&d, 64);
    const int iters = 4000, grid = 256 * waves_per_simd;   // 256-thread blocks = one wave per SIMD each
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_rate<KIND, NACC>), dim3(grid), dim3(256), 0, 0, d, 10);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_rate<KIND, NACC>), dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double kdepth = KIND == 0 ? 16.0 : 64.0;
    const double flops = 2.0 * 32 * 32 * kdepth * NACC * (double)iters * grid * 4;
    hipFree(d);
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    printf("%-44s %-10s %s\n", "instruction (8 independent accumulators/wave)", "waves/SIMD", "TFLOP/s (chip)");
    for (int w : {1, 2}) {
        printf("%-44s %-10d %.0f\n", "v_mfma_f32_32x32x16_f16", w, run<0, 8>(w));
        printf("%-44s %-10d %.0f\n", "v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)", w, run<1, 8>(w));
    }
    printf("per 64 k of a 32 x 32 tile: 4 fp16 MFMAs vs 2 fp8 MFMAs (hi + lo query) -> time ratio = (4 / rate_f16) : (2 * 4 / rate_f8) in k-normalised FLOP\n");
    return 0;
}
