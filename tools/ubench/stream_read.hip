// stream_read.hip -- what bounds a once-through HBM read on MI355X: bytes in flight per CU, the per-instruction line
// pattern, or the number of CUs?  hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip ; ./stream_read
//   pattern 0 ("scan"): lane (r = lane & 31, h = lane >> 5) reads 16 B at row r, byte offset seg * 128 + h * 64 + i * 16 of a
//                       32-row x ROWB-byte tile: 32 lines touched per instruction, 32 B of each (k_scan's A-operand loads)
//   pattern 1 ("line"): lane l reads 16 B at chunk + l * 16: 8 whole 128-B lines per instruction
// D = dwordx4 loads in flight per lane before the first is consumed; W = waves per workgroup (one workgroup per CU: the
// kernel asks for 128 KB of LDS like k_scan); grid = CUs used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int D>
__global__ __launch_bounds__(512) void k_read(const char* __restrict__ base, long long bytes_per_wg, unsigned* sink) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, W = blockDim.x >> 6;
    const char* p0 = base + (long long)blockIdx.x * bytes_per_wg;
    // a wave walks its slice of the workgroup's range in units of D KB (D loads of 1 KB per wave-instruction)
    const long long unit = (long long)D * 1024;
    const long long units = bytes_per_wg / unit;
    u4 A[D], B[D];      // two statically named register stages (a runtime-indexed buf[2][D] lands in scratch)
    u4 acc = {0, 0, 0, 0};
    auto addr = [&](long long u, int j) -> const u4* {
        if (PATTERN == 0) {   // k_scan's A-operand loads over 48 KB tiles (32 rows x 1536 B); load n of the range = segment n / 4, piece n % 4
            const int r = lane & 31, h = lane >> 5;
            const long long n = u * D + j, tile = n / 48;
            const int jj = (int)(n % 48);
            return (const u4*)(p0 + tile * 49152 + (long long)r * 1536 + (jj / 4) * 128 + h * 64 + (jj % 4) * 16);
        }
        return (const u4*)(p0 + u * unit + j * 1024 + lane * 16);
    };
    auto clampu = [&](long long u) { return u < units ? u : units - 1; };
    long long u = wid;
#pragma unroll
    for (int j = 0; j < D; ++j) A[j] = *addr(clampu(u), j);
    for (; u < units; u += 2 * W) {
#pragma unroll
        for (int j = 0; j < D; ++j) B[j] = *addr(clampu(u + W), j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= A[j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < D; ++j) A[j] = *addr(clampu(u + 2 * W), j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= B[j];
        __builtin_amdgcn_sched_barrier(0);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;   // never true: keeps the loads alive
    if (threadIdx.x == 0 && lds[0] == 77) sink[1] = 1;
}

template <int PATTERN, int D>
static double run(const char* d_buf, long long total, int grid, int waves, unsigned* d_sink, hipStream_t st) {
    const long long per = total / grid / 49152 * 49152;
    hipFuncSetAttribute((const void*)k_read<PATTERN, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_read<PATTERN, D>), dim3(grid), dim3(waves * 64), 128 * 1024, st, d_buf, per, d_sink);
    hipEventRecord(a, st);
    const int it = 5;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k_read<PATTERN, D>), dim3(grid), dim3(waves * 64), 128 * 1024, st, d_buf, per, d_sink);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return (double)per * grid * it / (ms * 1e-3) / 1e12;
}

int main() {
    const long long total = 12ll << 30;
    char* d_buf; unsigned* d_sink;
    if (hipMalloc(&d_buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&d_sink, 64);
    hipMemset(d_buf, 1, total);
    hipStream_t st; hipStreamCreate(&st);
    printf("%-8s %-4s %-6s %-6s %s\n", "pattern", "D", "waves", "CUs", "TB/s");
    for (int grid : {256, 224, 192}) {
        for (int waves : {8, 4}) {
#define ROW(P, DD) printf("%-8s %-4d %-6d %-6d %.2f\n", P ? "line" : "scan", DD, waves, grid, run<P, DD>(d_buf, total, grid, waves, d_sink, st));
            ROW(0, 4) ROW(0, 8) ROW(0, 16) ROW(0, 24)
            ROW(1, 4) ROW(1, 8) ROW(1, 16) ROW(1, 24)
        }
    }
    return 0;
}
