// Instruction issue-cost micro-benchmark for gfx950 (one or two waves per SIMD): cycles per instruction of the streams the
// attention kernel is built from.  hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip ; ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define REP 64
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (lane + i);
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
    f16v acc0, acc1, acc2, acc3;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = acc2[i] = acc3[i] = 0.f;
    float s0 = 0.f, s1 = 0.f;
    const h2 one2 = {(_Float16)1.f, (_Float16)1.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {          // 16 independent v_exp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
        } else if (KIND == 1) {   // 16 v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f);
        } else if (KIND == 2) {   // 16 v_dot2c (two chains)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                h2 p = {(_Float16)x[i], (_Float16)x[i + 1]};
                asm volatile("" : "+v"(p));
                s0 = __builtin_amdgcn_fdot2(p, one2, s0, false);
                s1 = __builtin_amdgcn_fdot2(p, one2, s1, false);
            }
        } else if (KIND == 3) {   // 8 MFMA 32x32x16 f16, 4 independent accumulators
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
            }
        } else if (KIND == 4) {   // 8 MFMA, one dependent chain
#pragma unroll
            for (int i = 0; i < 8; ++i) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        } else if (KIND == 5) {   // 8 x {MFMA, 2 exp, cvt_pk, dot2c}: the attention phase's gap pattern
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                const h2 p = {(_Float16)__builtin_amdgcn_exp2f(x[2 * i]), (_Float16)__builtin_amdgcn_exp2f(x[2 * i + 1])};
                s0 = __builtin_amdgcn_fdot2(p, one2, s0, false);
                x[2 * i] = (float)p[0] * 0.5f;
                x[2 * i + 1] = (float)p[1] * 0.5f;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        } else if (KIND == 6) {   // 16 v_cvt_pk_f16_f32
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                h2 p = {(_Float16)x[i], (_Float16)x[i + 1]};
                asm volatile("" : "+v"(p));
                h2 q = {(_Float16)(x[i] + 1.f), (_Float16)(x[i + 1] + 1.f)};
                asm volatile("" : "+v"(q));
                s0 += (float)p[0];
            }
        } else if (KIND == 8) {   // 16 v_exp_f16 (low halves)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                _Float16 hx = (_Float16)x[i];
                asm volatile("v_exp_f16 %0, %0" : "+v"(hx));
                x[i] = (float)hx;
            }
        } else if (KIND == 9) {   // 16 x {v_exp_f16, cvt} vs KIND 8: isolates
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                _Float16 hx = (_Float16)x[i];
                asm volatile("" : "+v"(hx));
                x[i] = (float)hx;
            }
        } else if (KIND == 10) {  // 8 x {v_cvt_pk_f16_f32, 2 v_exp_f16_sdwa}: 16 exponentials in fp16
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                h2 p;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\t"
                             "v_exp_f16_sdwa %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"
                             "v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"
                             : "=&v"(p)
                             : "v"(x[i]), "v"(x[i + 1]));
                s0 = __builtin_amdgcn_fdot2(p, one2, s0, false);
            }
        } else if (KIND == 11) {  // 8 x {MFMA, cvt_pk, 2 v_exp_f16_sdwa, dot2c}
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                h2 p;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\t"
                             "v_exp_f16_sdwa %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"
                             "v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"
                             : "=&v"(p)
                             : "v"(x[2 * i]), "v"(x[2 * i + 1]));
                s0 = __builtin_amdgcn_fdot2(p, one2, s0, false);
            }
        } else if (KIND == 12) {  // 8 v_pk_fma_f32 (16 results)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2v v = {x[i], x[i + 1]};
                const f2v c0 = {0.999f, 0.998f}, c1 = {0.001f, 0.002f};
                v = __builtin_elementwise_fma(v, c0, c1);
                x[i] = v[0];
                x[i + 1] = v[1];
            }
        } else if (KIND == 13) {  // 16 v_rcp_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_rcpf(x[i] + 2.0f);
        } else if (KIND == 7) {   // 8 x {MFMA, 2 exp}
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                x[2 * i] = __builtin_amdgcn_exp2f(x[2 * i]);
                x[2 * i + 1] = __builtin_amdgcn_exp2f(x[2 * i + 1]);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = s0 + s1;
    for (int i = 0; i < 16; ++i) r += x[i] + acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter, int threads) {
    const int grid = 256, iters = 2000;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * grid * threads);
    hipMalloc(&cyc, 8 * grid * (threads / 64));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * (threads / 64));
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per = sum / h.size() / iters;
    printf("%-44s waves/SIMD %d : %8.1f cycles per iteration = %6.2f per instruction-slot (%d)\n", name, threads / 256, per,
           per / per_iter, per_iter);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("16 v_exp_f32", 16, threads);
        run<1>("16 v_fma_f32", 16, threads);
        run<2>("16 v_dot2c_f32_f16 (+8 pk cvt)", 16, threads);
        run<6>("cvt mix", 16, threads);
        run<8>("16 x {cvt f16, v_exp_f16, cvt f32}", 16, threads);
        run<9>("16 x {cvt f16, cvt f32}", 16, threads);
        run<10>("8 x {cvt_pk, 2 v_exp_f16_sdwa, nop, dot2c}", 8, threads);
        run<11>("8 x {MFMA, cvt_pk, 2 v_exp_f16_sdwa, nop, dot2c}", 8, threads);
        run<12>("8 v_pk_fma_f32", 8, threads);
        run<13>("16 x {v_add, v_rcp_f32}", 16, threads);
        run<3>("8 MFMA 32x32x16 f16, 4 accumulators", 8, threads);
        run<4>("8 MFMA 32x32x16 f16, one chain", 8, threads);
        run<7>("8 x {MFMA, 2 v_exp}", 8, threads);
        run<5>("8 x {MFMA, 2 v_exp, cvt_pk, dot2c, 2 cvt, 2 mul}", 8, threads);
    }
    return 0;
}
