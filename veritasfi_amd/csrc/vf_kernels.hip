// vf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for VeritasFi's dense retrieval
// hot path: corpus scan + cosine + fused top-k.  Replaces what the reference delegates
// to faiss.IndexFlatIP.search (src/utils/faissRetriever.py:37) and sklearn.cosine_similarity +
// np.argsort (experiments/retriever/step3_mul.py:275-283).
//
// Kernels
//   k_prep_rows        index build: canonical row norms, fp16 scan copy (+ per-row pow2 scale)
//   k_prep_queries     per batch: canonical query normalisation, fp16 LDS image for the MFMA scan
//   k_scan<NT,G,MODE>  THE hot kernel: streams the fp16 corpus once, 32x32x16 f16 MFMA against the
//                      LDS-resident query tile, threshold-filter epilogue (no score matrix written)
//   k_sel0             threshold seed from the sample scores
//   k_final            per query: top-k' of candidates, canonical fp32 re-score, sort, certificate
//   k_normalize_rows / k_dense_dot16 / k_sort_rows   exact dense path (small N, repairs)
//   k_merge_topk       multi-GPU / chunk merge;  k_fuse_rank  rank_chunk score fusion
#include "vf_internal.h"

#include <float.h>

namespace vf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 orderkey(float f) {
    f = f + 0.0f;  // -0 -> +0 so equal floats have equal keys
    const u32 b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unorderkey(u32 k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// canonical 16-lane tree: acc[l] += acc[l+8]; += [l+4]; += [l+2]; acc[0] + acc[1]  (lane 0 of the group)
__device__ __forceinline__ float group16_tree(float acc) {
    acc = acc + __shfl_down(acc, 8, 16);
    acc = acc + __shfl_down(acc, 4, 16);
    acc = acc + __shfl_down(acc, 2, 16);
    acc = acc + __shfl_down(acc, 1, 16);
    return acc;
}

__device__ __forceinline__ float canon_div(float x, float n) { return (float)((double)x / (double)n); }
__device__ __forceinline__ float canon_norm_from_sumsq(float s) {
    const float n = (float)sqrt((double)s);
    return n == 0.0f ? 1.0f : n;
}

__device__ __forceinline__ float load_elem(const void* rows, int is_half, long long idx) {
    return is_half ? (float)((const _Float16*)rows)[idx] : ((const float*)rows)[idx];
}

// histogram bin of a cosine score: 2048 uniform bins over [-1, 1]; monotone in s.
__device__ __forceinline__ float bin_x(float s) { return __builtin_fmaf(s, 0.5f * kHistBins, 0.5f * kHistBins); }
__device__ __forceinline__ int bin_of_x(float x) {
    int b = (int)floorf(x);
    return b < 0 ? 0 : (b > kHistBins - 1 ? kHistBins - 1 : b);
}

// workgroup bitonic sort, descending, n a power of two, keys in LDS
__device__ __forceinline__ void bitonic_sort_desc(u64* s, int n, int tid, int nthreads) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (n >> 1); i += nthreads) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int hi = lo | j;
                const bool desc = (lo & k) == 0;
                const u64 a = s[lo], b = s[hi];
                if ((a < b) == desc) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// ------------------------------------------------------------------------------------------------
// k_prep_rows: one 16-lane group per row.  Canonical norm = sqrt(dot16(x,x)); fp16 scan copy.
//   fp32 rows: scan = fp16(x * 2^e), e chosen so the row max lands in [2^13, 2^14)  (no overflow,
//   underflow below 2^-38 of the row max); inv_scan = 1 / (norm * 2^e).
//   fp16 rows: scan copy only when dp != d (zero padding); scale 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_rows(const void* rows, int is_half, long long n, int d, int dp,
                                                    _Float16* scan, float* norm, float* inv_scan) {
    const int l = threadIdx.x & 15;
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (r >= n) return;  // whole 16-lane group leaves together
    const long long base = r * (long long)d;
    float acc = 0.0f, mx = 0.0f;
    for (int j = l; j < d; j += 16) {
        const float x = load_elem(rows, is_half, base + j);
        acc = __builtin_fmaf(x, x, acc);
        mx = fmaxf(mx, fabsf(x));
    }
    acc = group16_tree(acc);
    acc = __shfl(acc, 0, 16);
    for (int o = 8; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    const float nm = canon_norm_from_sumsq(acc);
    float scale = 1.0f;
    if (!is_half && mx > 0.0f && mx <= FLT_MAX) {
        int e;
        frexpf(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
        scale = ldexpf(1.0f, 14 - e);
    }
    if (l == 0) {
        norm[r] = nm;
        inv_scan[r] = 1.0f / (nm * scale);
    }
    if (scan) {
        _Float16* out = scan + r * (long long)dp;
        for (int j = l; j < dp; j += 16) {
            const float x = j < d ? load_elem(rows, is_half, base + j) * scale : 0.0f;
            out[j] = (_Float16)x;
        }
    }
}

hipError_t launch_prep_rows(const void* rows, int is_half, long long n, int d, int dp, _Float16* scan,
                            float* norm, float* inv_scan, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long blocks = (n + 15) / 16;
    hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)blocks), dim3(256), 0, s, rows, is_half, n, d, dp, scan, norm,
                       inv_scan);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_prep_queries: block per query slot.  qn = q / canon_norm(q) (fp32, for the canonical re-score);
// qimg[(j/8) * QN*8 + slot*8 + j%8] = fp16(qn[j]) -- the layout k_scan copies verbatim into LDS so
// that a wave's B-operand read (32 queries x 8 halves) is 512 contiguous bytes, conflict-free.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_queries(const float* q, int nq, int d, int dp, int QN, float* qn,
                                                       _Float16* qimg) {
    __shared__ float s_norm;
    const int slot = blockIdx.x;
    const int tid = threadIdx.x;
    if (slot < nq) {
        const float* x = q + (long long)slot * d;
        if (tid < 16) {
            float acc = 0.0f;
            for (int j = tid; j < d; j += 16) acc = __builtin_fmaf(x[j], x[j], acc);
            acc = group16_tree(acc);
            if (tid == 0) s_norm = canon_norm_from_sumsq(acc);
        }
        __syncthreads();
        const float nm = s_norm;
        for (int j = tid; j < dp; j += 256) {
            float v = 0.0f;
            if (j < d) {
                v = canon_div(x[j], nm);
                qn[(long long)slot * d + j] = v;
            }
            qimg[((long long)(j >> 3) * QN + slot) * 8 + (j & 7)] = (_Float16)v;
        }
    } else {
        for (int j = tid; j < dp; j += 256) qimg[((long long)(j >> 3) * QN + slot) * 8 + (j & 7)] = (_Float16)0.0f;
    }
}

hipError_t launch_prep_queries(const float* q, int nq, int d, int dp, int qn_tile, float* qn, _Float16* qimg,
                               hipStream_t s) {
    hipLaunchKernelGGL(k_prep_queries, dim3(qn_tile), dim3(256), 0, s, q, nq, d, dp, qn_tile, qn, qimg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// exact dense path
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_normalize_rows(const void* rows, int is_half, long long row0,
                                                         long long nrows, int d, const float* norm, float* out) {
    const long long total = nrows * (long long)d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d;
        out[i] = canon_div(load_elem(rows, is_half, (row0 + r) * (long long)d + (i - r * d)), norm[row0 + r]);
    }
}

hipError_t launch_normalize_rows(const void* rows, int is_half, long long row0, long long nrows, int d,
                                 const float* norm, float* out, hipStream_t s) {
    if (nrows <= 0) return hipSuccess;
    long long blocks = (nrows * (long long)d + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_normalize_rows, dim3((unsigned)blocks), dim3(256), 0, s, rows, is_half, row0, nrows, d, norm,
                       out);
    return hipGetLastError();
}

// out[q * out_stride + r] = dot16(qn[q], cn[r]); one 16-lane group per row, looping over queries.
__global__ __launch_bounds__(256) void k_dense_dot16(const float* qn, int nq, const float* cn, long long nrows, int d,
                                                      float* out, long long out_stride) {
    const int l = threadIdx.x & 15;
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (r >= nrows) return;
    const float* c = cn + r * (long long)d;
    for (int q = 0; q < nq; ++q) {
        const float* a = qn + (long long)q * d;
        float acc = 0.0f;
        for (int j = l; j < d; j += 16) acc = __builtin_fmaf(a[j], c[j], acc);
        acc = group16_tree(acc);
        if (l == 0) out[(long long)q * out_stride + r] = acc;
    }
}

hipError_t launch_dense_dot16(const float* qn, int nq, const float* cn, long long nrows, int d, float* out,
                              long long out_stride, hipStream_t s) {
    if (nrows <= 0 || nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dense_dot16, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, s, qn, nq, cn, nrows, d, out,
                       out_stride);
    return hipGetLastError();
}

// Per query (block): rank n <= 16384 scores, descending, lower id first; write the best k.
__global__ __launch_bounds__(1024) void k_sort_rows(const float* scores, long long score_stride, int n, int k,
                                                     long long id_base, long long* out_ids, float* out_scores,
                                                     int out_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64* keys = (u64*)smem_raw;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int P = next_pow2(n > 1 ? n : 1);
    const float* s = scores + (long long)q * score_stride;
    for (int i = tid; i < P; i += 1024)
        keys[i] = i < n ? (((u64)orderkey(s[i]) << 32) | (u64)(0xFFFFFFFFu - (u32)i)) : 0ull;
    __syncthreads();
    bitonic_sort_desc(keys, P, tid, 1024);
    for (int i = tid; i < k; i += 1024) {
        long long id = -1;
        float sc = -FLT_MAX;
        if (i < n) {
            const u64 kv = keys[i];
            id = id_base + (long long)(0xFFFFFFFFu - (u32)kv);
            sc = unorderkey((u32)(kv >> 32));
        }
        out_ids[(long long)q * out_stride + i] = id;
        out_scores[(long long)q * out_stride + i] = sc;
    }
}

hipError_t launch_sort_rows(const float* scores, long long score_stride, int nq, int n, int k, long long id_base,
                            long long* out_ids, float* out_scores, int out_stride, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    int P = 1;
    while (P < n) P <<= 1;
    hipLaunchKernelGGL(k_sort_rows, dim3(nq), dim3(1024), (size_t)P * 8, s, scores, score_stride, n, k, id_base,
                       out_ids, out_scores, out_stride);
    return hipGetLastError();
}

// Merge nparts ranked lists per query (parts in ascending id-range order, -1 = padding).
__global__ __launch_bounds__(1024) void k_merge_topk(const long long* ids_parts, const float* score_parts, int nparts,
                                                      int nq, int k, long long* ids, float* scores) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64* keys = (u64*)smem_raw;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int m = nparts * k;
    const int P = next_pow2(m > 1 ? m : 1);
    for (int i = tid; i < P; i += 1024) {
        u64 kv = 0ull;
        if (i < m) {
            const int g = i / k, j = i - g * k;
            const long long o = ((long long)g * nq + q) * k + j;
            // equal scores: earlier part / earlier rank first == lower id first
            if (ids_parts[o] >= 0) kv = ((u64)orderkey(score_parts[o]) << 32) | (u64)(0xFFFFFFFFu - (u32)i);
        }
        keys[i] = kv;
    }
    __syncthreads();
    bitonic_sort_desc(keys, P, tid, 1024);
    for (int i = tid; i < k; i += 1024) {
        long long id = -1;
        float sc = -FLT_MAX;
        const u64 kv = i < P ? keys[i] : 0ull;
        if (kv != 0ull) {
            const int src = (int)(0xFFFFFFFFu - (u32)kv);
            const int g = src / k, j = src - g * k;
            const long long o = ((long long)g * nq + q) * k + j;
            id = ids_parts[o];
            sc = score_parts[o];
        }
        ids[(long long)q * k + i] = id;
        scores[(long long)q * k + i] = sc;
    }
}

hipError_t launch_merge_topk(const long long* ids_parts, const float* score_parts, int nparts, int nq, int k,
                             long long* ids, float* scores, hipStream_t s) {
    if (nq <= 0 || k <= 0) return hipSuccess;
    int P = 1;
    while (P < nparts * k) P <<= 1;
    if ((size_t)P * 8 > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_merge_topk, dim3(nq), dim3(1024), (size_t)P * 8, s, ids_parts, score_parts, nparts, nq, k, ids,
                       scores);
    return hipGetLastError();
}

// rank_chunk fusion (src/utils/vllmManager.py:454-457): out = a + b, order = argsort desc (stable).
__global__ __launch_bounds__(1024) void k_fuse_rank(const float* a, const float* b, int n, float* out,
                                                     long long* order) {
    __shared__ u64 keys[4096];
    const int tid = threadIdx.x;
    const int P = next_pow2(n > 1 ? n : 1);
    for (int i = tid; i < P; i += 1024) {
        u64 kv = 0ull;
        if (i < n) {
            const float s = a[i] + b[i];
            out[i] = s;
            kv = ((u64)orderkey(s) << 32) | (u64)(0xFFFFFFFFu - (u32)i);
        }
        keys[i] = kv;
    }
    __syncthreads();
    bitonic_sort_desc(keys, P, tid, 1024);
    for (int i = tid; i < n; i += 1024) order[i] = (long long)(0xFFFFFFFFu - (u32)keys[i]);
}

hipError_t launch_fuse_rank(const float* a, const float* b, int n, float* out, long long* order, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > 4096) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fuse_rank, dim3(1), dim3(1024), 0, s, a, b, n, out, order);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_scan: the HBM-bound hot kernel.
//
// Work split: rows [0,n) are divided evenly over TW waves (wave w owns [n*w/TW, n*(w+1)/TW)); the
// first `samp` rows of each range are the SAMPLE (MODE 0: dense approx scores -> s0, feeds k_sel0),
// the rest is the MAIN part (MODE 1: threshold filter + candidate append).  Every row is read
// exactly once per batch, by exactly one wave, straight from HBM into MFMA A-operand registers:
// lane (r = lane & 31, h = lane >> 5) owns corpus row r of the tile and reads 64 contiguous bytes
// (4 x dwordx4) of each 128-byte row segment -- two lanes cover a full cache line.  The MFMA k-slot
// (h, j) of step i in segment g is element (8g + 4h + i) * 8 + j of the row; the B operand (queries)
// uses the same map, so the contraction is a permuted-order dot product (order is irrelevant for the
// approximate score; the exact order lives in the canonical re-score).
//
// Accumulator layout (v_mfma_f32_32x32x16_f16 C/D): lane holds query (lane & 31) of N-tile nt and
// corpus rows (reg & 3) + 8 * (reg >> 2) + 4 * h, reg = 0..15.
// ------------------------------------------------------------------------------------------------
struct TileCursor {
    long long t0;  // first row of the tile
    int ss;        // superstep within the tile
};

template <int G>
__device__ __forceinline__ void issue_loads(h8 (&buf)[4 * G], const char* rows, long long row_bytes, long long myrow,
                                            int ss, int h) {
    const char* p = rows + myrow * row_bytes + (long long)(ss * G) * 128 + h * 64;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[g * 4 + i] = *(const h8*)(p + g * 128 + i * 16);
}

template <int NT, int G>
__device__ __forceinline__ void compute_superstep(f16v (&acc)[NT], const h8 (&buf)[4 * G], const char* lds_lane,
                                                  int ss) {
    constexpr int QN = NT * kQueryTile;
    const char* base = lds_lane + (long long)(ss * G) * (8 * QN * 16);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const h8 b = *(const h8*)(base + (g * 8 + i) * (QN * 16) + nt * (kQueryTile * 16));
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(buf[g * 4 + i], b, acc[nt], 0, 0, 0);
            }
}

// Recompute tau for query q from its histogram: the largest bin b with sum(hist[b..]) >= kprime.
// All 64 lanes cooperate (lane owns bins [32*lane, 32*lane+32)); returns -1 if fewer than kprime
// are counted.  Two passes so no per-lane array is live (this is a rare path; keep it out of the
// hot loop's register budget).  Counts only grow between the passes, which keeps the bound valid:
// the second pass can only reach kprime at the same or a higher bin than the first would have.
__device__ __attribute__((noinline)) int wave_tau_from_hist(const u32* hist_q, int kprime, int lane) {
    u32 sum = 0;
#pragma unroll 1
    for (int i = 0; i < 32; ++i)
        sum += __hip_atomic_load(hist_q + lane * 32 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 suf = sum;
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = __shfl_down(suf, off);
        if (lane + off < 64) suf += v;
    }
    const u32 above = suf - sum;
    int best = -1;
    if (above < (u32)kprime && suf >= (u32)kprime) {
        u32 run = above;
#pragma unroll 1
        for (int i = 31; i >= 0; --i) {
            run += __hip_atomic_load(hist_q + lane * 32 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (run >= (u32)kprime) { best = lane * 32 + i; break; }
        }
    }
#pragma unroll 1
    for (int off = 32; off; off >>= 1) best = max(best, __shfl_xor(best, off));
    return best;
}

// Per-tile values the epilogue needs, fetched BEFORE the next prefetch is issued so that waiting for
// them (vmcnt is in-order) never drains the prefetch: lane r (and r+32) holds 1/norm of tile row r.
template <int NT>
struct EpiRegs {
    float inv_lane;
    int tau[NT];
};

template <int NT, int MODE>
__device__ __forceinline__ void epi_prefetch(EpiRegs<NT>& e, const ScanArgs& a, long long t0, int lane) {
    e.inv_lane = a.inv_scan[t0 + (lane & 31)];  // inv_scan is padded by 64 entries past n
    if (MODE == kModeMain) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            e.tau[nt] = __hip_atomic_load(a.tau_bin + nt * kQueryTile + (lane & 31), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NT, int MODE>
__device__ __forceinline__ void tile_epilogue(const ScanArgs& a, const f16v (&acc)[NT], const EpiRegs<NT>& e,
                                              long long t0, long long lo, long long hi, long long gw, int lane) {
    const int r31 = lane & 31, h = lane >> 5;
    // inverse norms of this lane's 16 rows (reg -> row (reg&3) + 8*(reg>>2) + 4*h) via readlane
    float inv[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r0 = (reg & 3) + 8 * (reg >> 2);
        const float lo_half = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.inv_lane), r0));
        const float hi_half = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.inv_lane), r0 + 4));
        inv[reg] = h ? hi_half : lo_half;
    }
    if (MODE == kModeSample) {
        const long long slot0 = gw * (long long)a.samp + (t0 - lo);
        const long long s0_stride = (long long)a.total_waves * a.samp;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int q = nt * kQueryTile + r31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (t0 + rr < hi) a.s0[(long long)q * s0_stride + slot0 + rr] = acc[nt][reg] * inv[reg];
            }
        }
        return;
    }
    // ---- main mode: threshold filter ----
    float tb[NT];
    bool any = false;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = nt * kQueryTile + r31;
        const int t = e.tau[nt];
        tb[nt] = q < a.nq ? (t <= 0 ? -INFINITY : (float)t) : INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const float x = bin_x(acc[nt][reg] * inv[reg]);
            any |= (t0 + rr < hi) && (x >= tb[nt]);
        }
    }
    if (__ballot(any) == 0ull) return;

    // ---- rare path: append candidates, bump histogram, maybe refresh tau ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = nt * kQueryTile + r31;
        int c = 0;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const float x = bin_x(acc[nt][reg] * inv[reg]);
            c += ((t0 + rr < hi) && (x >= tb[nt])) ? 1 : 0;
        }
        bool need = false;
        if (c > 0) {
            u32 slot = atomicAdd(a.cnt + q, (u32)c);
            need = (slot / (u32)a.refresh_every) != ((slot + (u32)c) / (u32)a.refresh_every);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const float sc = acc[nt][reg] * inv[reg];
                const float x = bin_x(sc);
                if ((t0 + rr < hi) && (x >= tb[nt])) {
                    if (slot < (u32)a.cap)
                        a.cand[(long long)q * a.cap + slot] = ((u64)orderkey(sc) << 32) | (u64)(u32)(t0 + rr);
                    ++slot;
                    atomicAdd(a.hist + (long long)q * kHistBins + bin_of_x(x), 1u);
                }
            }
        }
        unsigned long long m = __ballot(need);
        while (m) {
            const int leader = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int qq = nt * kQueryTile + (leader & 31);
            const int nb = wave_tau_from_hist(a.hist + (long long)qq * kHistBins, a.kprime, lane);
            if (lane == 0 && nb > 0) atomicMax(a.tau_bin + qq, nb);
        }
    }
}

template <int NT, int G, int MODE>
__global__ __launch_bounds__(kScanThreads) void k_scan(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QN = NT * kQueryTile;
    const int tid = threadIdx.x;
    {   // query image -> LDS, verbatim
        const uint4* src = (const uint4*)a.qimg;
        uint4* dst = (uint4*)smem;
        const int nvec = (a.dp >> 3) * QN;
        for (int i = tid; i < nvec; i += kScanThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = tid & 63, wid = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const long long gw = (long long)blockIdx.x * (kScanThreads / 64) + wid;
    if (gw >= a.total_waves) return;
    const long long ra = a.n * gw / a.total_waves, rb = a.n * (gw + 1) / a.total_waves;
    const long long rs = (ra + a.samp < rb) ? ra + a.samp : rb;
    const long long lo = MODE == kModeSample ? ra : rs;
    const long long hi = MODE == kModeSample ? rs : rb;
    if (lo >= hi) return;
    const int SS = (a.dp >> 6) / G;
    const long long ntiles = (hi - lo + kRowTile - 1) / kRowTile;
    const long long total = ntiles * SS;
    const char* lds_lane = smem + ((4 * h) * QN + r31) * 16;

    f16v acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;

    // Two register stages (A0, A1), one flat sequence of supersteps over (tile, ss).  The prefetch of
    // step it+1 is issued UNCONDITIONALLY before step it is consumed (past the end it re-reads the last
    // step: harmless), so the compiler's counted vmcnt waits never cover the stage in flight.
    h8 A0[4 * G], A1[4 * G];
    EpiRegs<NT> epi;
    TileCursor cur = {lo, 0}, nxt;
    const long long last_t0 = lo + (ntiles - 1) * kRowTile;
    auto advance = [&](const TileCursor& c) {
        TileCursor o;
        if (c.ss + 1 == SS) { o.t0 = c.t0 + kRowTile; o.ss = 0; } else { o.t0 = c.t0; o.ss = c.ss + 1; }
        if (o.t0 > last_t0) { o.t0 = last_t0; o.ss = SS - 1; }
        return o;
    };
    auto rowof = [&](const TileCursor& c) {
        const long long r = c.t0 + r31;
        return r < hi ? r : hi - 1;
    };
#define VF_SCAN_STEP(CURBUF, NXTBUF)                                                       \
    {                                                                                      \
        const bool last_ss = (cur.ss + 1 == SS);                                           \
        if (last_ss) epi_prefetch<NT, MODE>(epi, a, cur.t0, lane);                         \
        nxt = advance(cur);                                                                \
        issue_loads<G>(NXTBUF, a.rows, a.row_bytes, rowof(nxt), nxt.ss, h);                \
        __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ABOVE the MFMAs */         \
        compute_superstep<NT, G>(acc, CURBUF, lds_lane, cur.ss);                           \
        if (last_ss) {                                                                     \
            tile_epilogue<NT, MODE>(a, acc, epi, cur.t0, lo, hi, gw, lane);                \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                              \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;          \
        }                                                                                  \
        cur = nxt;                                                                         \
    }
    issue_loads<G>(A0, a.rows, a.row_bytes, rowof(cur), cur.ss, h);
    long long it = 0;
    for (; it + 1 < total; it += 2) {
        VF_SCAN_STEP(A0, A1)
        VF_SCAN_STEP(A1, A0)
    }
    if (it < total) VF_SCAN_STEP(A0, A1)
#undef VF_SCAN_STEP
}

size_t scan_lds_bytes(int dp, int qn_tile) { return (size_t)dp * qn_tile * 2; }

template <int NT, int G, int MODE>
static hipError_t launch_scan_inst(const ScanArgs& a, int grid, hipStream_t s) {
    const size_t lds = scan_lds_bytes(a.dp, NT * kQueryTile);
    hipLaunchKernelGGL((k_scan<NT, G, MODE>), dim3(grid), dim3(kScanThreads), lds, s, a);
    return hipGetLastError();
}

template <int NT, int G>
static hipError_t launch_scan_mode(const ScanArgs& a, int mode, int grid, hipStream_t s) {
    return mode == kModeSample ? launch_scan_inst<NT, G, kModeSample>(a, grid, s)
                               : launch_scan_inst<NT, G, kModeMain>(a, grid, s);
}

// Segments (128 B of each row) per pipeline stage.  G=2 keeps the main-mode kernel spill-free
// (251 VGPRs at NT=2); larger G puts more bytes in flight per wave but spills on this compiler.
static int pick_G(int dp, int want) {
    const int segs = dp >> 6;
    if (want >= 1 && want <= 4 && segs % want == 0) return want;
    if (segs % 2 == 0) return 2;
    if (segs % 3 == 0) return 3;
    return 1;
}

hipError_t launch_scan(const ScanArgs& a, int mode, int qn_tile, int grid, int want_g, hipStream_t s) {
    const int G = pick_G(a.dp, want_g);
    const int NT = qn_tile / kQueryTile;
#define VF_CASE(NTV, GV) \
    if (NT == NTV && G == GV) return launch_scan_mode<NTV, GV>(a, mode, grid, s);
    VF_CASE(1, 1) VF_CASE(1, 2) VF_CASE(1, 3) VF_CASE(1, 4)
    VF_CASE(2, 1) VF_CASE(2, 2) VF_CASE(2, 3) VF_CASE(2, 4)
#undef VF_CASE
    return hipErrorInvalidValue;
}

template <int NT, int G, int MODE>
static hipError_t configure_one() {
    return hipFuncSetAttribute((const void*)k_scan<NT, G, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024);
}

// ------------------------------------------------------------------------------------------------
// k_sel0: per query, seed the threshold from the sample scores: LDS histogram -> tau bin with
// >= kprime sample rows at or above it -> emit those rows as the first candidates.
// ------------------------------------------------------------------------------------------------
template <int QN>
__global__ __launch_bounds__(1024) void k_sel0(ScanArgs a) {
    __shared__ u32 lh[kHistBins];
    __shared__ u32 lcnt;
    __shared__ int sbin;
    const int q = blockIdx.x, tid = threadIdx.x;
    for (int b = tid; b < kHistBins; b += 1024) lh[b] = 0;
    if (tid == 0) { lcnt = 0; sbin = 0; }
    __syncthreads();
    const long long len = (long long)a.total_waves * a.samp;
    const float* s = a.s0 + (long long)q * len;
    if (q < a.nq) {
        for (long long i = tid; i < len; i += 1024) {
            const float v = s[i];
            if (v > -INFINITY) atomicAdd(&lh[bin_of_x(bin_x(v))], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int nb = wave_tau_from_hist(lh, a.kprime, tid);
        if (tid == 0) sbin = nb > 0 ? nb : 0;
    }
    __syncthreads();
    const int tb = sbin;
    if (q < a.nq) {
        for (long long i = tid; i < len; i += 1024) {
            const float v = s[i];
            if (v > -INFINITY && bin_of_x(bin_x(v)) >= tb) {
                const u32 slot = atomicAdd(&lcnt, 1u);
                const long long w = i / a.samp;
                const long long row = a.n * w / a.total_waves + (i - w * a.samp);
                if (slot < (u32)a.cap) a.cand[(long long)q * a.cap + slot] = ((u64)orderkey(v) << 32) | (u64)(u32)row;
            }
        }
    }
    __syncthreads();
    for (int b = tid; b < kHistBins; b += 1024) a.hist[(long long)q * kHistBins + b] = (q < a.nq && b >= tb) ? lh[b] : 0u;
    if (tid == 0) {
        a.cnt[q] = lcnt;
        a.tau_bin[q] = q < a.nq ? tb : kHistBins;
    }
}

hipError_t launch_sel0(const ScanArgs& a, int qn_tile, hipStream_t s) {
    if (qn_tile == 32) hipLaunchKernelGGL(k_sel0<32>, dim3(32), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(k_sel0<64>, dim3(64), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_final: per query.  Sort the candidates by approximate score, re-score the best k' with the
// CANONICAL fp32 arithmetic (bit-identical to oracle/vf_oracle.c), sort by (canonical desc, id asc),
// write k results and the exactness certificate:
//   every row not re-scored has approx <= A (A = approx of the k'-th candidate), hence canonical
//   <= A + eps; if canonical_k > A + eps no such row can enter the top k  =>  result is exact.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_final(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64* ck = (u64*)smem_raw;
    const int q = blockIdx.x, tid = threadIdx.x;
    const u32 n_raw = a.cnt[q];
    const int n = n_raw < (u32)a.cap ? (int)n_raw : a.cap;
    const int P = next_pow2(n > 1 ? n : 1);
    const u64* cq = a.cand + (long long)q * a.cap;
    for (int i = tid; i < P; i += 1024) ck[i] = i < n ? cq[i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(ck, P, tid, 1024);
    const int m = n < a.kprime ? n : a.kprime;
    const int KP = next_pow2(m > 1 ? m : 1);
    // the re-scored keys live behind the first kprime sorted candidates (cap >= 2 * kprime)
    u64* rk = ck + (a.cap >> 1);
    const float approx_floor = m > 0 ? unorderkey((u32)(ck[m - 1] >> 32)) : -INFINITY;
    __syncthreads();
    const int g = tid >> 4, l = tid & 15;
    const float* qv = a.qn + (long long)q * a.d;
    for (int i = g; i < m; i += 64) {
        const u32 row = (u32)ck[i];
        const float nm = a.norm[row];
        const long long base = (long long)row * a.orig_row_elems;
        float acc = 0.0f;
        for (int j = l; j < a.d; j += 16)
            acc = __builtin_fmaf(qv[j], canon_div(load_elem(a.rows_orig, a.orig_is_half, base + j), nm), acc);
        acc = group16_tree(acc);
        if (l == 0) rk[i] = ((u64)orderkey(acc) << 32) | (u64)(0xFFFFFFFFu - row);
    }
    for (int i = m + tid; i < KP; i += 1024) rk[i] = 0ull;
    __syncthreads();
    bitonic_sort_desc(rk, KP, tid, 1024);
    for (int i = tid; i < a.k; i += 1024) {
        long long id = -1;
        float sc = -FLT_MAX;
        if (i < m) {
            const u64 kv = rk[i];
            id = a.id_offset + (long long)(0xFFFFFFFFu - (u32)kv);
            sc = unorderkey((u32)(kv >> 32));
        }
        a.out_ids[(long long)q * a.k + i] = id;
        a.out_scores[(long long)q * a.k + i] = sc;
    }
    if (tid == 0) {
        int flag = 0;
        if (n_raw > (u32)a.cap) flag = 2;
        else if ((long long)m < a.n_rows) {  // some row was not re-scored: need the certificate
            if (a.k > m) flag = 1;
            else {
                const float ck_k = unorderkey((u32)(rk[a.k - 1] >> 32));
                if (!(ck_k > approx_floor + a.eps)) flag = 1;
            }
        }
        a.flags[q] = flag;
        a.cand_count_out[q] = n_raw;
    }
}

hipError_t launch_final(const FinalArgs& a, int nq, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_final, dim3(nq), dim3(1024), (size_t)a.cap * 8, s, a);
    return hipGetLastError();
}

hipError_t scan_configure() {
    hipError_t e;
#define VF_CFG(NTV, GV) \
    if ((e = configure_one<NTV, GV, kModeSample>()) != hipSuccess) return e; \
    if ((e = configure_one<NTV, GV, kModeMain>()) != hipSuccess) return e;
    VF_CFG(1, 1) VF_CFG(1, 2) VF_CFG(1, 3) VF_CFG(1, 4)
    VF_CFG(2, 1) VF_CFG(2, 2) VF_CFG(2, 3) VF_CFG(2, 4)
#undef VF_CFG
    if ((e = hipFuncSetAttribute((const void*)k_sort_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_merge_topk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_final, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace vf
