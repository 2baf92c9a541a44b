// cu_mask_probe.hip -- does hipExtStreamCreateWithCUMask work on this box, and which mask bit is which (XCD, SE, CU)?
// hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip ; ./cu_mask_probe
// Each workgroup records its XCC_ID / HW_ID registers; the host prints the distinct (xcc, se, sh, cu) sets per mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void k_where(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the CU busy for a while so that a large grid spreads over every CU the queue may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static int run(const char* name, hipStream_t st, unsigned* d_out, int grid) {
    std::vector<unsigned> h(2 * grid);
    hipLaunchKernelGGL(k_where, dim3(grid), dim3(64), 0, st, d_out, 200000);
    if (hipStreamSynchronize(st) != hipSuccess) { printf("%s: launch failed\n", name); return -1; }
    hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    std::set<unsigned> per[16];
    for (int i = 0; i < grid; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        cus.insert(key); per[xcc].insert(key);
    }
    printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2zu", per[x].size());
    // CUs per shader engine of XCC 0 (a dispatch that hands every SE the same number of workgroups needs them equal)
    int per_se[8] = {0};
    for (unsigned key : per[0]) ++per_se[(key >> 8) & 7];
    printf("   XCC0 per SE:");
    for (int e = 0; e < 8; ++e) if (per_se[e]) printf(" se%d=%d", e, per_se[e]);
    printf("\n");
    (void)per_xcc;
    return (int)cus.size();
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, ncu);
    unsigned* d_out; const int grid = 4096;
    hipMalloc(&d_out, 2 * grid * 4);
    hipStream_t plain; hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    run("no mask", plain, d_out, grid);
    const int words = (ncu + 31) / 32;
    auto masked = [&](const char* name, int lo, int hi) {   // mask bits [lo, hi) set
        std::vector<uint32_t> m(words, 0u);
        for (int b = lo; b < hi; ++b) m[b / 32] |= 1u << (b % 32);
        hipStream_t s = nullptr;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, words, m.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask: %s\n", name, hipGetErrorString(e)); return; }
        run(name, s, d_out, grid);
        hipStreamDestroy(s);
    };
    masked("bits [0, ncu-8)", 0, ncu - 8);
    masked("bits [ncu-8, ncu)", ncu - 8, ncu);
    masked("bits [0, 8)", 0, 8);
    masked("bits [0, 32)", 0, 32);
    masked("bits [0, ncu/2)", 0, ncu / 2);
    masked("bits [0, ncu-16)", 0, ncu - 16);
    masked("bits [0, ncu-32)", 0, ncu - 32);
    masked("bits [ncu-32, ncu)", ncu - 32, ncu);
    return 0;
}
