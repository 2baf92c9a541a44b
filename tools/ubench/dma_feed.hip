// dma_feed.hip -- how fast can a CU pull a GEMM's operand tiles from L2 into LDS by LDS-DMA, and does the LAYOUT matter?
// 256 workgroups x 512 threads (one per CU), each "K-tile step" = 32 KB of A + 32 KB of W into a 2 x 64 KB LDS ring, two steps in
// flight, no compute.  Operands: A [51200][768] fp16 (78 MB: the Infinity Cache, shared by the 9 workgroups of an M-tile), W
// [2304][768] (3.5 MB: L2).  Patterns:
//   0  rows: both tiles as 256 rows x 128 B, row stride 1536 B (what k_gemm9_tn does: 8 rows per 1-KB DMA instruction)
//   1  W packed: W tiles pre-packed [nt][kt][256 x 128 B] contiguous 32 KB (a linear copy), A as rows
//   2  both packed (A cannot be in the product: its producer writes rows; the upper bound)
// hipcc --offload-arch=gfx950 -O3 -o dma_feed dma_feed.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
}

template <int PAT>
__global__ __launch_bounds__(512) void k(const char* A, const char* W, const char* Ap, const char* Wp, int Mt, int Nt, int NK, int rounds, float* out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // tile of this workgroup per round: XCD-contiguous groups of 4 M-tiles x all N-tiles (k_gemm9_tn's order)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int issued = 0;
    for (int r = 0; r < rounds; ++r) {
        const int p = (xcd * rounds + r) * 32 + slot;                 // 32 consecutive tiles per XCD and round
        const int g = p / (4 * Nt), rr = p % (4 * Nt), nt = rr / 4, mt = (g * 4 + rr % 4) % Mt;
        for (int kt = 0; kt < NK; ++kt) {
            const unsigned sb = lds0 + (unsigned)((issued & 1) * 65536) + (unsigned)wid * 8192u;
            // this wave's share: 4 instructions of A (32 rows), 4 of W
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wid * 32 + i * 8 + (lane >> 3);
                const char* pa = (PAT == 2) ? Ap + ((size_t)(mt * NK + kt) * 32768) + (size_t)(wid * 4 + i) * 1024 + lane * 16
                                            : A + ((size_t)(mt * 256 + row) * NK + kt) * 128 + (lane & 7) * 16;
                const char* pw = (PAT >= 1) ? Wp + ((size_t)(nt * NK + kt) * 32768) + (size_t)(wid * 4 + i) * 1024 + lane * 16
                                            : W + ((size_t)(nt * 256 + row) * NK + kt) * 128 + (lane & 7) * 16;
                dma16(pa, sb + i * 1024);
                dma16(pw, sb + 4096 + i * 1024);
            }
            ++issued;
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // the previous step has landed
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && out) out[blockIdx.x] = *(volatile float*)smem;
}

template <int PAT>
static double run(const char* A, const char* W, const char* Ap, const char* Wp, float* out) {
    const int Mt = 200, Nt = 9, NK = 12, rounds = 7;
    hipFuncSetAttribute((const void*)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 131072, 0, A, W, Ap, Wp, Mt, Nt, NK, 1, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 131072, 0, A, W, Ap, Wp, Mt, Nt, NK, rounds, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 5.0 * 256 * rounds * NK * 65536.0;
    return bytes / (ms * 1e-3) / 1e9;
}

int main() {
    const size_t asz = (size_t)51200 * 768 * 2, wsz = (size_t)2304 * 768 * 2;
    char *A, *W, *Ap, *Wp; float* out;
    hipMalloc(&A, asz); hipMalloc(&W, wsz); hipMalloc(&Ap, asz); hipMalloc(&Wp, wsz); hipMalloc(&out, 1024);
    hipMemset(A, 1, asz); hipMemset(W, 1, wsz); hipMemset(Ap, 1, asz); hipMemset(Wp, 1, wsz);
    for (int rep = 0; rep < 2; ++rep) {
        const double a = run<0>(A, W, Ap, Wp, out), b = run<1>(A, W, Ap, Wp, out), c = run<2>(A, W, Ap, Wp, out);
        printf("rows / rows: %.0f GB/s chip-wide = %.1f per CU   | W packed: %.0f = %.1f per CU   | both packed: %.0f = %.1f per CU\n", a, a / 256, b, b / 256, c, c / 256);
    }
    return 0;
}
