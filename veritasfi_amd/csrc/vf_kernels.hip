// vf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for VeritasFi's dense retrieval
// hot path: corpus scan + cosine + fused top-k.  Replaces what the reference delegates
// to faiss.IndexFlatIP.search (src/utils/faissRetriever.py:37) and sklearn.cosine_similarity +
// np.argsort (experiments/retriever/step3_mul.py:275-283).
//
// Kernels
//   k_prep_rows        index build: canonical row norms, fp16 scan copy (+ per-row pow2 scale); fp8 rows stay bytes
//   k_prep_queries     per batch: canonical query normalisation, fp16 LDS image for the MFMA scan
//   k_scan<NT,G,MODE,F8>  THE hot kernel: streams the corpus once (fp16 rows, or fp8-e4m3 bytes converted in
//                      registers), 32x32x16 f16 MFMA against the LDS-resident query tile, threshold-filter
//                      epilogue (no score matrix written)
//   k_sel0             threshold seed from the sample scores
//   k_final            per query: top-k' of candidates, canonical fp32 re-score, sort, certificate
//   k_normalize_rows / k_dense_dot16 / k_sort_rows   exact dense path (small N, repairs)
//   k_merge_topk       multi-GPU / chunk merge;  k_fuse_rank  rank_chunk score fusion
#include "vf_internal.h"

#include <float.h>
#include <algorithm>
#include <type_traits>

#include <mutex>

namespace vf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

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

// canonical inverse norm: (float)(1.0 / (double)n), as faiss's fvec_renorm_L2 computes it
__device__ __forceinline__ float canon_inv(float n) { return (float)(1.0 / (double)n); }
__device__ __forceinline__ float canon_norm_from_sumsq(float s) {
    const float n = (float)sqrt((double)s);
    return n == 0.0f ? 1.0f : n;
}

__device__ __forceinline__ unsigned short e4m3_to_f16_bits(u32 b) {
    const u32 sign = (b & 0x80u) << 8, e = (b >> 3) & 0xFu, m = b & 7u;
    if (e == 0u) {  // zero / subnormal: m * 2^-9 = m * 2^-24 * 2^15 -> normalise by hand
        if (m == 0u) return (unsigned short)sign;
        const int sh = m >= 4u ? 0 : (m >= 2u ? 1 : 2);            // leading one at bit 2 - sh
        const u32 frac = ((m << (sh + 1)) & 7u) << 7;              // bits below the leading one -> fp16 mantissa
        return (unsigned short)(sign | ((u32)(15 - 7 - sh) << 10) | frac);  // value 1.f * 2^(-7 - sh)
    }
    if (e == 15u && m == 7u) return (unsigned short)(sign | 0x7E00u);  // NaN
    return (unsigned short)(sign | ((e + 8u) << 10) | (m << 7));       // exponent e - 7 + 15
}

__device__ __forceinline__ float e4m3_to_float(u32 b) {
    return (float)__builtin_bit_cast(_Float16, e4m3_to_f16_bits(b));
}

// element `idx` of a row-major corpus of storage dtype dt (VF_DTYPE_F32 / _F16 / _FP8_E4M3), as fp32 (exact)
__device__ __forceinline__ float load_elem(const void* rows, int dt, long long idx) {
    if (dt == VF_DTYPE_F16) return (float)((const _Float16*)rows)[idx];
    if (dt == VF_DTYPE_FP8_E4M3) return e4m3_to_float(((const unsigned char*)rows)[idx]);
    return ((const float*)rows)[idx];
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
//   fp8 (e4m3) rows: the scan copy stays fp8 BYTES (dp bytes per row, zero padded); scale 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_rows(const void* rows, int dt, long long n, int d, int dp,
                                                    void* scan, float* norm, float* inv_scan) {
    const int l = threadIdx.x & 15;
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (r >= n) return;  // whole 16-lane group leaves together
    const long long base = r * (long long)d;
    float acc = 0.0f, mx = 0.0f;
    for (int j = l; j < d; j += 16) {
        const float x = load_elem(rows, dt, base + j);
        acc = __builtin_fmaf(x, x, acc);
        mx = fmaxf(mx, fabsf(x));
    }
    acc = group16_tree(acc);
    acc = __shfl(acc, 0, 16);
    for (int o = 8; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    const float nm = canon_norm_from_sumsq(acc);
    float scale = 1.0f;
    if (dt == VF_DTYPE_F32 && mx > 0.0f && mx <= FLT_MAX) {
        int e;
        frexpf(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
        scale = ldexpf(1.0f, 14 - e);
    }
    if (l == 0) {
        norm[r] = nm;
        inv_scan[r] = 1.0f / (nm * scale);
    }
    if (scan && dt == VF_DTYPE_FP8_E4M3) {
        unsigned char* out = (unsigned char*)scan + r * (long long)dp;
        const unsigned char* in = (const unsigned char*)rows + base;
        for (int j = l; j < dp; j += 16) out[j] = j < d ? in[j] : (unsigned char)0;
    } else if (scan) {
        _Float16* out = (_Float16*)scan + r * (long long)dp;
        for (int j = l; j < dp; j += 16) {
            const float x = j < d ? load_elem(rows, dt, base + j) * scale : 0.0f;
            out[j] = (_Float16)x;
        }
    }
}

hipError_t launch_prep_rows(const void* rows, int dt, long long n, int d, int dp, void* scan,
                            float* norm, float* inv_scan, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long blocks = (n + 15) / 16;
    hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)blocks), dim3(256), 0, s, rows, dt, n, d, dp, scan, norm,
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
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = (float*)smem_raw;  // the query row, staged once
    __shared__ float s_norm;
    const int slot = blockIdx.x;
    const int tid = threadIdx.x;
    if (slot < nq) {
        const float* x = q + (long long)slot * d;
        for (int j = tid; j < d; j += 256) xs[j] = x[j];
        __syncthreads();
        if (tid < 16) {
            float acc = 0.0f;
            for (int j = tid; j < d; j += 16) acc = __builtin_fmaf(xs[j], xs[j], acc);
            acc = group16_tree(acc);
            if (tid == 0) s_norm = canon_norm_from_sumsq(acc);
        }
        __syncthreads();
        const float inv = canon_inv(s_norm);
        for (int j = tid; j < dp; j += 256) {
            float v = 0.0f;
            if (j < d) {
                v = xs[j] * inv;
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
    hipLaunchKernelGGL(k_prep_queries, dim3(qn_tile), dim3(256), (size_t)d * sizeof(float), s, q, nq, d, dp, qn_tile, qn, qimg);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// exact dense path
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_normalize_rows(const void* rows, int dt, long long row0,
                                                         long long nrows, int d, const float* norm, float* out) {
    const long long total = nrows * (long long)d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d;
        out[i] = load_elem(rows, dt, (row0 + r) * (long long)d + (i - r * d)) * canon_inv(norm[row0 + r]);
    }
}

hipError_t launch_normalize_rows(const void* rows, int dt, long long row0, long long nrows, int d,
                                 const float* norm, float* out, hipStream_t s) {
    if (nrows <= 0) return hipSuccess;
    long long blocks = (nrows * (long long)d + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_normalize_rows, dim3((unsigned)blocks), dim3(256), 0, s, rows, dt, row0, nrows, d, norm,
                       out);
    return hipGetLastError();
}

// the same for rows PICKED by index: out[i] = canonical normalisation of row sel[i] (vf_cosine_matrix_rows)
__global__ __launch_bounds__(256) void k_normalize_rows_gather(const void* rows, int dt, const long long* sel, int nsel, int d,
                                                                const float* norm, float* out) {
    const long long total = (long long)nsel * d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d, row = sel[r];
        out[i] = load_elem(rows, dt, row * (long long)d + (i - r * d)) * canon_inv(norm[row]);
    }
}

hipError_t launch_normalize_rows_gather(const void* rows, int dt, const long long* sel, int nsel, int d, const float* norm,
                                        float* out, hipStream_t s) {
    if (nsel <= 0) return hipSuccess;
    long long blocks = ((long long)nsel * d + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_normalize_rows_gather, dim3((unsigned)blocks), dim3(256), 0, s, rows, dt, sel, nsel, d, norm, out);
    return hipGetLastError();
}

// out[q * out_stride + r] = dot16(qn[q], cn[r]); one 16-lane group per row, looping over queries.
__global__ __launch_bounds__(256) void k_dense_dot16(const float* qn, int nq, const float* cn, long long nrows, int d,
                                                      float* out, long long out_stride) {
    const int l = threadIdx.x & 15;
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (r >= nrows) return;
    const float* c = cn + r * (long long)d;
    // four queries per pass over the row: the row is read once per four queries and the four accumulation chains are
    // independent (each keeps the canonical order: lane l sums j = l, l + 16, ... ascending, then the 16-lane tree)
    int q = 0;
    for (; q + 4 <= nq; q += 4) {
        const float *a0 = qn + (long long)q * d, *a1 = a0 + d, *a2 = a1 + d, *a3 = a2 + d;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        for (int j = l; j < d; j += 16) {
            const float cj = c[j];
            s0 = __builtin_fmaf(a0[j], cj, s0);
            s1 = __builtin_fmaf(a1[j], cj, s1);
            s2 = __builtin_fmaf(a2[j], cj, s2);
            s3 = __builtin_fmaf(a3[j], cj, s3);
        }
        s0 = group16_tree(s0);
        s1 = group16_tree(s1);
        s2 = group16_tree(s2);
        s3 = group16_tree(s3);
        if (l == 0) {
            out[(long long)q * out_stride + r] = s0;
            out[(long long)(q + 1) * out_stride + r] = s1;
            out[(long long)(q + 2) * out_stride + r] = s2;
            out[(long long)(q + 3) * out_stride + r] = s3;
        }
    }
    for (; q < nq; ++q) {
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

// k_topk_rows: the same result as k_sort_rows (the k best of a score row, ranked, ties to the lower id) without sorting
// the row.  The reference's per-request search shape is N ~ 1e4, k = 2048 (ensembleRetriever.py:64-66): a full bitonic
// sort of 16384 keys is 105 LDS passes (146 us, LDS-bandwidth bound); here a radix SELECT finds the k-th largest 64-bit
// key (8-bit digits from the top; a level is one LDS histogram and one suffix scan; it stops as soon as a digit's bin
// is taken whole, normally after the four score bytes), the keys >= it are compacted -- exactly k of them, keys are
// unique -- and only those are sorted (66 passes over 2048 keys, four waves).
constexpr int kTopkThreads = 256;
__global__ __launch_bounds__(kTopkThreads) void k_topk_rows(const float* scores, long long score_stride, int n, int k, int Pk,
                                                             long long id_base, long long* out_ids, float* out_scores,
                                                             int out_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64* keys = (u64*)smem_raw;                    // [n]
    u64* sel = keys + n;                           // [Pk]
    unsigned* hist = (unsigned*)(sel + Pk);        // [256]
    unsigned* wsum = hist + 256;                   // [4] wave totals, [4] digit, [5] above, [6] bin count, [7] compaction cursor
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* s = scores + (long long)q * score_stride;
    for (int i = tid; i < n; i += kTopkThreads) keys[i] = ((u64)orderkey(s[i]) << 32) | (u64)(0xFFFFFFFFu - (u32)i);
    const int kk = k < n ? k : n;                  // real results
    u64 prefix = 0;
    int need = kk;
    int shift = 56;
    for (;; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        const u64 hi_mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < n; i += kTopkThreads) {
            const u64 key = keys[i];
            if ((key & hi_mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // inclusive suffix sums over the 256 digits (thread t owns digit t): in-wave scan, then the waves' totals
        const unsigned c = hist[tid];
        unsigned sfx = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_down(sfx, o, 64);
            if (lane + o < 64) sfx += v;
        }
        if (lane == 0) wsum[wid] = sfx;
        __syncthreads();
        for (int w = wid + 1; w < kTopkThreads / 64; ++w) sfx += wsum[w];
        if (sfx >= (unsigned)need && sfx - c < (unsigned)need) {   // the digit that holds the need-th largest key
            wsum[4] = (unsigned)tid;
            wsum[5] = sfx - c;
            wsum[6] = c;
        }
        __syncthreads();
        prefix |= (u64)wsum[4] << shift;
        need -= (int)wsum[5];
        const bool whole_bin = (int)wsum[6] == need;   // every key of this bin is wanted: its lower digits do not matter
        if (whole_bin || shift == 0) break;
    }
    // keys >= prefix (lower digits zero) with the matched upper digits are exactly the kk best
    if (tid == 0) wsum[7] = 0;
    for (int i = kk + tid; i < Pk; i += kTopkThreads) sel[i] = 0ull;
    __syncthreads();
    for (int i = tid; i < n; i += kTopkThreads) {
        const u64 key = keys[i];
        if (key >= prefix) sel[atomicAdd(&wsum[7], 1u)] = key;
    }
    __syncthreads();
    bitonic_sort_desc(sel, Pk, tid, kTopkThreads);
    for (int i = tid; i < k; i += kTopkThreads) {
        long long id = -1;
        float sc = -FLT_MAX;
        if (i < kk) {
            const u64 kv = sel[i];
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
    int Pk = 1;
    while (Pk < (k < n ? k : n)) Pk <<= 1;
    const size_t topk_lds = (size_t)n * 8 + (size_t)Pk * 8 + 256 * 4 + 64;
    static const bool full_sort = getenv("VF_FULL_SORT") != nullptr;   // A/B switch
    if (!full_sort && n >= 2 && Pk * 2 <= P && topk_lds <= 160 * 1024) {
        hipLaunchKernelGGL(k_topk_rows, dim3(nq), dim3(kTopkThreads), topk_lds, s, scores, score_stride, n, k, Pk, id_base, out_ids,
                           out_scores, out_stride);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_sort_rows, dim3(nq), dim3(1024), (size_t)P * 8, s, scores, score_stride, n, k, id_base,
                       out_ids, out_scores, out_stride);
    return hipGetLastError();
}

// Merge nparts ranked lists per query (parts in ascending id-range order, -1 = padding).
// Part g's ids start at ids_base + g * part_stride_bytes, its scores at score_base + g * part_stride_bytes
// (two separate [G][nq][k] arrays, or one packed blob per part: [ids nq*k int64][scores nq*k fp32]).
__global__ __launch_bounds__(1024) void k_merge_topk(const char* ids_base, const char* score_base,
                                                      long long ids_stride, long long score_stride, int nparts, int nq,
                                                      int k, long long* ids, float* scores) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64* keys = (u64*)smem_raw;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int m = nparts * k;
    const int P = next_pow2(m > 1 ? m : 1);
    for (int i = tid; i < P; i += 1024) {
        u64 kv = 0ull;
        if (i < m) {
            const int g = i / k, j = i - g * k;
            const long long o = (long long)q * k + j;
            const long long id = ((const long long*)(ids_base + g * ids_stride))[o];
            // equal scores: earlier part / earlier rank first == lower id first
            if (id >= 0) kv = ((u64)orderkey(((const float*)(score_base + g * score_stride))[o]) << 32) | (u64)(0xFFFFFFFFu - (u32)i);
        }
        keys[i] = kv;
    }
    __syncthreads();
    if (m <= 2048) {
        // rank by counting (keys are unique; padding keys are 0): no barriers, ~m/16 LDS batches per thread.
        // A thread owns keys tid and tid + 1024; the key of rank r < k is written straight to output slot r.
        u64 mine[2];
        int rank[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) mine[u] = tid + u * 1024 < P ? keys[tid + u * 1024] : 0ull;
        for (int j = 0; j < P; j += 16) {
            u64 kj[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) kj[t] = j + t < P ? keys[j + t] : 0ull;
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) rank[u] += kj[t] > mine[u] ? 1 : 0;
        }
        // outputs default to padding; ranks of real keys are distinct and < number of real keys
        for (int i = tid; i < k; i += 1024) { ids[(long long)q * k + i] = -1; scores[(long long)q * k + i] = -FLT_MAX; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (mine[u] != 0ull && rank[u] < k) {
                const int src = (int)(0xFFFFFFFFu - (u32)mine[u]);
                const int g = src / k, j = src - g * k;
                const long long o = (long long)q * k + j;
                ids[(long long)q * k + rank[u]] = ((const long long*)(ids_base + g * ids_stride))[o];
                scores[(long long)q * k + rank[u]] = ((const float*)(score_base + g * score_stride))[o];
            }
        }
        return;
    }
    bitonic_sort_desc(keys, P, tid, 1024);
    for (int i = tid; i < k; i += 1024) {
        long long id = -1;
        float sc = -FLT_MAX;
        const u64 kv = i < P ? keys[i] : 0ull;
        if (kv != 0ull) {
            const int src = (int)(0xFFFFFFFFu - (u32)kv);
            const int g = src / k, j = src - g * k;
            const long long o = (long long)q * k + j;
            id = ((const long long*)(ids_base + g * ids_stride))[o];
            sc = ((const float*)(score_base + g * score_stride))[o];
        }
        ids[(long long)q * k + i] = id;
        scores[(long long)q * k + i] = sc;
    }
}

static hipError_t launch_merge_impl(const char* ib, const char* sb, long long is, long long ss, int nparts, int nq, int k,
                                    long long* ids, float* scores, hipStream_t s) {
    if (nq <= 0 || k <= 0) return hipSuccess;
    int P = 1;
    while (P < nparts * k) P <<= 1;
    if ((size_t)P * 8 > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_merge_topk, dim3(nq), dim3(1024), (size_t)P * 8, s, ib, sb, is, ss, nparts, nq, k, ids, scores);
    return hipGetLastError();
}

hipError_t launch_merge_topk(const long long* ids_parts, const float* score_parts, int nparts, int nq, int k,
                             long long* ids, float* scores, hipStream_t s) {
    return launch_merge_impl((const char*)ids_parts, (const char*)score_parts, (long long)nq * k * 8, (long long)nq * k * 4,
                             nparts, nq, k, ids, scores, s);
}

hipError_t launch_merge_topk_packed(const void* parts, int nparts, int nq, int k, long long* ids, float* scores,
                                    hipStream_t s) {
    const long long stride = packed_part_bytes(nq, k);
    return launch_merge_impl((const char*)parts, (const char*)parts + (long long)nq * k * 8, stride, stride, nparts, nq, k,
                             ids, scores, s);
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
// Corpus rows are read once per batch, which invites the non-temporal hint (global_load_dwordx4 ... nt).  MEASURED
// (round 3, tools/gpu_r03_nt.sh, same box, same process tree): with nt the main scan runs at 0.41 of peak instead of 0.77
// at 10M rows (4.73 vs 2.48 ms) and 0.37 instead of 0.66 at 1M -- a lane pair shares every 128-B line and four
// instructions walk it; without L1 allocation each of them goes out again.  Plain loads stay; -DVF_SCAN_NT=1 rebuilds the
// experiment.
#ifndef VF_SCAN_NT
#define VF_SCAN_NT 0
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if VF_SCAN_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

template <int G>
__device__ __forceinline__ void issue_loads(h8 (&buf)[4 * G], const char* rows, long long row_bytes, long long myrow,
                                            int ss /* superstep within the tile */, int h) {
    const char* p = rows + myrow * row_bytes + (long long)(ss * G) * 128 + h * 64;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[g * 4 + i] = stream_load((const h8*)(p + g * 128 + i * 16));
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

// ---- fp8 (e4m3) corpus rows: the same supersteps on HALF the bytes -------------------------------------------
// A 64-element segment is 64 B; lane (r, h) reads elements [32h, 32h + 32) = 32 B (2 x global_load_dwordx4) and
// converts them to the four h8 A-operands in registers (v_cvt_scalef32_pk_f16_fp8, scale 1: two codes -> two halves
// per VALU op; every e4m3 value is an fp16 value, so the MFMA sees exactly the corpus).  Query image, MFMAs,
// accumulators and epilogue are the fp16 kernel's.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h8 cvt8_e4m3(u32 w0, u32 w1) {
    const h2v a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, false);
    const h2v b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, true);
    const h2v c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, false);
    const h2v d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, true);
    h8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}

template <int G>
__device__ __forceinline__ void issue_loads_f8(uint4 (&buf)[2 * G], const char* rows, long long row_bytes, long long myrow,
                                               int ss, int h) {
    const char* p = rows + myrow * row_bytes + (long long)(ss * G) * 64 + h * 32;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const u4v w = stream_load((const u4v*)(p + g * 64 + i * 16));
            buf[g * 2 + i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
}

template <int NT, int G>
__device__ __forceinline__ void compute_superstep_f8(f16v (&acc)[NT], const uint4 (&buf)[2 * G], const char* lds_lane,
                                                     int ss) {
    constexpr int QN = NT * kQueryTile;
    const char* base = lds_lane + (long long)(ss * G) * (8 * QN * 16);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 w = buf[g * 2 + (i >> 1)];
            const h8 afrag = (i & 1) ? cvt8_e4m3(w.z, w.w) : cvt8_e4m3(w.x, w.y);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const h8 b = *(const h8*)(base + (g * 8 + i) * (QN * 16) + nt * (kQueryTile * 16));
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag, b, acc[nt], 0, 0, 0);
            }
        }
}

__global__ void k_debug_cvt_e4m3(const unsigned char* in, float* out, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per group of 8 codes
    if (i * 8 >= count) return;
    u32 w0 = 0, w1 = 0;
    for (int j = 0; j < 4; ++j) {
        w0 |= (u32)(i * 8 + j < count ? in[i * 8 + j] : 0) << (8 * j);
        w1 |= (u32)(i * 8 + 4 + j < count ? in[i * 8 + 4 + j] : 0) << (8 * j);
    }
    const h8 r = cvt8_e4m3(w0, w1);
    for (int j = 0; j < 8; ++j) if (i * 8 + j < count) out[i * 8 + j] = (float)r[j];
}

hipError_t launch_debug_cvt_e4m3(const unsigned char* in, float* out, int count, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    const int groups = (count + 7) / 8;
    hipLaunchKernelGGL(k_debug_cvt_e4m3, dim3((groups + 63) / 64), dim3(64), 0, s, in, out, count);
    return hipGetLastError();
}

// Recompute tau for query q: the largest fine bin b with sum(fine[b..]) >= kprime, found in two
// round trips through a two-level histogram (64 coarse bins of 32 fine bins): lane l reads coarse[l],
// a suffix scan finds the coarse bin L where the count crosses kprime, then lanes 0..31 read the 32
// fine bins of L.  Returns -1 if fewer than kprime are counted.  Coarse and fine counters are bumped
// by separate relaxed atomics, so a reader can see them out of step; that can only make tau LESS
// tight or -- rarely -- too tight, and a too-tight tau is caught a posteriori by k_final (it checks
// that >= kprime candidates sit at or above the final tau), which then takes the exact path.
__device__ __forceinline__ int wave_tau_two_level(const u32* coarse_q, const u32* fine_q, int kprime,
                                                            int lane) {
    const u32 c = __hip_atomic_load(coarse_q + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 suf = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = __shfl_down(suf, off);
        if (lane + off < 64) suf += v;
    }
    const u32 above = suf - c;
    const unsigned long long cross = __ballot(above < (u32)kprime && suf >= (u32)kprime);
    if (cross == 0ull) return -1;
    const int L = __ffsll((long long)cross) - 1;
    const u32 aboveL = (u32)__shfl((int)above, L);
    const u32 f = lane < 32 ? __hip_atomic_load(fine_q + L * 32 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    u32 sf = f;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const u32 v = __shfl_down(sf, off);
        if (lane + off < 32) sf += v;
    }
    const unsigned long long ok = __ballot(lane < 32 && aboveL + sf >= (u32)kprime);
    if (ok == 0ull) return -1;
    return L * 32 + (63 - __clzll((long long)ok));
}

// ---- LDS control block of k_scan (right after the query image) ---------------------------------
//   +0   u32 stage_cnt     candidates staged by this workgroup (main mode)
//   +4   u32 next_tile     next unclaimed tile of the workgroup's row range
//   +16  int tau_lds[64]   the workgroup's copy of the per-query threshold bins
//   +272 uint4 entries[stage_cap]
constexpr int kCtlBytes = 272;

template <int NT>
__device__ __forceinline__ constexpr int QN_of() { return NT * kQueryTile; }

// Per-tile values the epilogue needs, fetched BEFORE the next prefetch is issued so that waiting for
// them (vmcnt is in-order) never drains the prefetch: lane r (and r+32) holds 1/norm of tile row r.
// Every 8th tile (staggered over the 8 waves) a wave also re-reads the GLOBAL thresholds and folds
// them into the workgroup's LDS copy; the filter itself only ever reads the LDS copy, so the global
// tau lines -- which atomicMax keeps evicting from L2 -- are off the per-tile path.
template <int NT>
struct EpiRegs {
    float inv_lane;
    int tau_g[NT];
    bool sync_tau;
    const char* inv_lds = nullptr;   // k_scan2 / k_scan2r: the tile's 32 reciprocal norms as floats in LDS (the scratch half the DMA filled); nullptr: inv_lane
};

// Wave-wide OR / sum of a 32-bit value, result uniform (an SGPR): four DPP row rotations leave every lane with its row's
// (16 lanes') result, four readlanes combine the rows.  ~12 instructions, no LDS crossbar.
#define VF_DPP_ROR(v, n) (u32) __builtin_amdgcn_update_dpp(0, (int)(v), 0x120 + (n), 0xF, 0xF, false)
__device__ __forceinline__ u32 wave_or_u32(u32 v) {
    v |= VF_DPP_ROR(v, 1);
    v |= VF_DPP_ROR(v, 2);
    v |= VF_DPP_ROR(v, 4);
    v |= VF_DPP_ROR(v, 8);
    return (u32)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) |
                 __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
    v += VF_DPP_ROR(v, 1);
    v += VF_DPP_ROR(v, 2);
    v += VF_DPP_ROR(v, 4);
    v += VF_DPP_ROR(v, 8);
    return (u32)(__builtin_amdgcn_readlane((int)v, 0) + __builtin_amdgcn_readlane((int)v, 16) +
                 __builtin_amdgcn_readlane((int)v, 32) + __builtin_amdgcn_readlane((int)v, 48));
}

template <int NT, int MODE>
__device__ __forceinline__ void epi_prefetch(EpiRegs<NT>& e, const ScanArgs& a, long long t0, int lane, bool sync_tau) {
    e.inv_lane = a.inv_scan[t0 + (lane & 31)];  // inv_scan is padded by 64 entries past n
    e.sync_tau = sync_tau;
    if (MODE == kModeMain && sync_tau) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            e.tau_g[nt] = __hip_atomic_load(a.tau_bin + nt * kQueryTile + (lane & 31), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
    }
}

// PUBLISH = false (k_scan2): the wave only STAGES its candidates; publishing completed blocks to the global histograms and
// refreshing thresholds is the service wave's job (k_scan2_service), so that no global atomic or dependent load of this
// path ever sits in a streaming wave's in-order memory queue.
template <int NT, int MODE, bool PUBLISH = true, bool INV_LDS = false>
__device__ __forceinline__ void tile_epilogue(const ScanArgs& a, const f16v (&acc)[NT], const EpiRegs<NT>& e,
                                              long long t0, long long hi, long long s0_slot, int lane, char* ctl) {
    const int r31 = lane & 31, h = lane >> 5;
    // 1/norm of this lane's row for accumulator register `reg` (row (reg&3) + 8*(reg>>2) + 4*h): two
    // readlanes of the per-tile inv_lane and a select, recomputed at each use to keep registers free
    auto inv_of = [&](int reg) {
        const int r0 = (reg & 3) + 8 * (reg >> 2);
        const float lo_half = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.inv_lane), r0));
        const float hi_half = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.inv_lane), r0 + 4));
        return h ? hi_half : lo_half;
    };
    if (MODE == kModeSample) {
        const long long s0_stride = (long long)a.total_waves * a.samp;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int q = nt * kQueryTile + r31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (t0 + rr < hi) a.s0[(long long)q * s0_stride + s0_slot + rr] = acc[nt][reg] * inv_of(reg);
            }
        }
        return;
    }
    // ---- main mode: threshold filter ----
    int* tau_lds = (int*)(ctl + 16);
    if (e.sync_tau && lane < 32) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) atomicMax(tau_lds + nt * kQueryTile + lane, e.tau_g[nt]);
    }
    // The filter, written for the tile WITHOUT a candidate (round 6: the first form's per-element "(row exists) && (x >= tb)" compiled to
    // two exec-mask branches, a 64-bit add and compare and two readlanes per element -- ~700 instructions, 4 500-6 000 cycles per tile,
    // a third of a streaming wave's cycles by the cycle accounting of tools/stamps_scan2r.py; profiles/r06_scan2r_cycle_accounting.log):
    //  - the 16 reciprocal norms this lane's accumulator registers need (row (reg & 3) + 8 (reg >> 2) + 4 h) come as four 16-byte LDS
    //    reads where the tile's norms sit in LDS (k_scan2 / k_scan2r), else by two readlanes and a select each (k_scan);
    //  - rows past the part's end get a NaN norm (one scalar branch per tile, taken by a range's last tile only): every compare is false;
    //  - x = bin_x(acc / norm) for the 16 NT elements, their maximum per query tile, ONE compare per query tile, one ballot: no lane of
    //    the wave passes -> done (the same mul + fma as before: the scores, and so the candidates, are bit for bit what they were);
    //  - only a tile that holds a candidate builds the per-element mask (a compare and a select each).
    float inv16[16];
    if constexpr (INV_LDS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *(const float4*)(e.inv_lds + (8 * g + 4 * h) * 4);
            inv16[4 * g] = v.x; inv16[4 * g + 1] = v.y; inv16[4 * g + 2] = v.z; inv16[4 * g + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) inv16[reg] = inv_of(reg);
    }
    const long long left = hi - t0;
    const int lf = __builtin_amdgcn_readfirstlane((int)(left < kRowTile ? left : kRowTile));   // (wave-uniform by construction: a scalar branch)
    if (lf < kRowTile) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) inv16[reg] = ((reg & 3) + 8 * (reg >> 2) + 4 * h) < lf ? inv16[reg] : __builtin_nanf("");
    }
    // m = acc / norm, the approximate scores, as ONE register vector (the candidate loop below indexes it with a scalar); bin_x is
    // monotone, so "some bin_x(m) reaches the threshold" is asked of the maximum: one fma + one compare per query tile
    typedef float mvec __attribute__((ext_vector_type(NT * 16)));
    mvec m;
    float tb[NT];
    bool any = false;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = nt * kQueryTile + r31;
        const int t = tau_lds[q];
        tb[nt] = q < a.nq ? (t <= 0 ? -INFINITY : (float)t) : INFINITY;
        float top = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            m[nt * 16 + reg] = acc[nt][reg] * inv16[reg];
            top = fmaxf(top, m[nt * 16 + reg]);                        // (a NaN -- a row that does not exist -- never wins)
        }
        any |= bin_x(top) >= tb[nt];
    }
    if (__ballot(any) == 0ull) return;
    // bit (nt * 16 + reg) of `mask` = "this lane's score for (query nt*32 + r31, row reg) passes"
    u32 mask = 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) mask |= (bin_x(m[nt * 16 + reg]) >= tb[nt]) ? (1u << (nt * 16 + reg)) : 0u;
    // ---- a tile that holds a candidate: stage them in LDS; once per block of R staged entries ONE wave refreshes one tau and
    // publishes that block to the global histograms.
    // No global memory operation is issued per candidate: vmcnt retires in order, so even a no-return
    // atomic here would hold up the consumption of the corpus stage already in flight.  The
    // workgroup flushes its stage to the global lists once, at the end.
    // On a small shard nearly every tile takes this path with two or three candidates (1.25M x 768: 3.2 per tile), so it is written for
    // FEW candidates (round 6; the cycle accounting put it at ~1 700 cycles per tile, a tenth of the wave's time):
    //  - which (query tile, row register) pairs hold a passing lane anywhere in the wave, and how many candidates there are: a scalar
    //    loop over the lanes that hold one (a readlane each) while there are at most eight, the DPP reductions otherwise;
    //  - the block size is a power of two (begin_impl rounds it): shifts, not two 32-bit divisions.
    const unsigned long long hot = __ballot(mask != 0u);
    if (hot == 0ull) return;
    if (a.debug & 4) return;  // timing experiment: filter only
    u32 umask = 0u, total = 0u;
    if (__popcll(hot) <= 8) {
        for (unsigned long long hb = hot; hb != 0ull; hb &= hb - 1ull) {
            const u32 m = (u32)__builtin_amdgcn_readlane((int)mask, __builtin_ctzll(hb));
            umask |= m;
            total += (u32)__popc(m);
        }
    } else {
        umask = wave_or_u32(mask);
        total = wave_sum_u32((u32)__popc(mask));
    }
    u32* stage_cnt = (u32*)ctl;
    uint4* stage_ent = (uint4*)(ctl + kCtlBytes);
    const u32 rs = 31u - (u32)__builtin_clz((u32)a.refresh_every | 1u);   // block size R = 2^rs (1..256)
    const u32 R = 1u << rs;
    // one LDS atomic per wave claims the slots of all its candidates; lanes take theirs pair by pair (ballot prefix)
    u32 base = 0;
    if (lane == 0) base = atomicAdd(stage_cnt, total);
    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
    const bool need = (base >> rs) != ((base + total) >> rs);   // this wave completed a block
    if (a.debug & 8) return;  // timing experiment: claim slots, write nothing (flush skips w != 1)
    u32 run = base;
    const u32 edge = (((base + total) >> rs) << rs) - 1u;   // the slot that completed the block (meaningful when `need`)
    int qq = 0;               // the query whose tau is refreshed: that of the pair holding the completing slot (its first lane)
    // A scalar loop over the SET pairs: the pair's scores are read from the register vector with the pair number as a scalar index
    // (s_set_gpr_idx / v_movrel: no memory).  The first form tested all 16 NT pairs one by one, a scalar compare-and-branch each,
    // taken for every pair without a candidate -- the largest single piece of this path; a switch into unrolled bodies cost scratch.
    for (u32 um = umask; um != 0u; um &= um - 1u) {
        const int pair = __builtin_ctz(um);                           // scalar
        const int nt = pair >> 4, reg = pair & 15;
        const int q = nt * kQueryTile + r31;
        const bool mine = ((mask >> pair) & 1u) != 0u;
        const unsigned long long bal = __ballot(mine);
        if (mine) {
            const u32 slot = run + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
            const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const float sc = m[pair];
            const u32 bin = (u32)bin_of_x(bin_x(sc));
            const u32 key = orderkey(sc), row = (u32)(t0 + rr);
            if (slot < (u32)a.stage_cap) {
                stage_ent[slot] = make_uint4(row, key, (u32)q | (bin << 8), 1u);  // w = 1: entry written
            } else {  // stage full (hostile data): append straight to the global list
                const u32 gs = atomicAdd(a.cnt + q * kCntStride, 1u);
                if (gs < (u32)a.cap) a.cand[(long long)q * a.cap + gs] = ((u64)key << 32) | (u64)row;
                atomicAdd(a.hist + (long long)q * kHistBins + bin, 1u);
                atomicAdd(a.hist_coarse + q * 64 + (bin >> 5), 1u);
            }
        }
        const u32 nxt = run + (u32)__popcll(bal);
        if (run <= edge && edge < nxt) qq = nt * kQueryTile + ((__ffsll((long long)bal) - 1) & 31);
        run = nxt;
    }
    if (a.debug & 2) return;  // timing experiment: never publish / refresh
    if constexpr (!PUBLISH) return;
    if (need) {
        // (1) refresh one tau from what is published so far (two dependent L2 reads)
        const int nb = wave_tau_two_level(a.hist_coarse + qq * 64, a.hist + (long long)qq * kHistBins, a.kprime, lane);
        if (lane == 0 && nb > 0) {
            atomicMax(a.tau_bin + qq, nb);
            atomicMax(tau_lds + qq, nb);
        }
        // (2) publish the completed block: lane i takes entry blk*R + i, fire-and-forget atomics.
        //     Entries other waves claimed but have not written yet read w == 0 (the stage is zeroed at
        //     kernel start) and are skipped: the histogram then under-counts, which only makes tau
        //     less tight.
        //     EVERY block this wave completed is published (round 4): with small blocks (refresh_every scaled down to 4 for
        //     one or two queries) a wave's candidates can span several blocks, and the ones in the middle used to be lost.
        for (u32 blk = base >> rs; blk < (base + total) >> rs; ++blk)
        for (u32 j = (u32)lane; j < R; j += 64u) {
            const u32 idx = (blk << rs) + j;
            if (idx >= (u32)a.stage_cap) break;
            const uint4 en = stage_ent[idx];
            if (en.w == 1u) {
                const u32 q = en.z & 0xFFu, bin = en.z >> 8;
                if (q < (u32)QN_of<NT>() && bin < (u32)kHistBins) {
                    atomicAdd(a.hist + (long long)q * kHistBins + bin, 1u);
                    atomicAdd(a.hist_coarse + q * 64 + (bin >> 5), 1u);
                }
            }
        }
    }
}

// stage helpers of k_scan, picked by its F8 template flag (if constexpr keeps one kernel body for both element types)
#define VF_ISSUE(BUF, ...)                                                        \
    do {                                                                          \
        if constexpr (F8 != 0) issue_loads_f8<G>(BUF, __VA_ARGS__);               \
        else issue_loads<G>(BUF, __VA_ARGS__);                                    \
    } while (0)
#define VF_COMPUTE(ACC, BUF, ...)                                                 \
    do {                                                                          \
        if constexpr (F8 != 0) compute_superstep_f8<NT, G>(ACC, BUF, __VA_ARGS__); \
        else compute_superstep<NT, G>(ACC, BUF, __VA_ARGS__);                      \
    } while (0)

// k_scan: see the block comment above.  One workgroup (8 waves) per CU; the workgroup owns the row
// range [n*wg/grid, n*(wg+1)/grid); its first 8*samp rows are the sample part, the rest the main part.
// Waves claim 32-row tiles of the part dynamically from an LDS counter: a wave that stalls (tau
// refresh, unlucky memory channel) simply takes fewer tiles, so the workgroup finishes together.
template <int NT, int G, int MODE, int F8>
__global__ __launch_bounds__(kScanThreads) void k_scan(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QN = NT * kQueryTile;
    constexpr int WAVES = kScanThreads / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    // The corpus is cut into `grid` row ranges (one per MAIN-scan workgroup).  Main mode: gridDim.x == grid, a workgroup
    // owns its range.  Sample mode: gridDim.x may be SMALLER (a.scan_grid ranges, walked v = blockIdx.x, + gridDim.x, ...):
    // a few workgroups then stage the query image once and score the sample part of many ranges, so that the sample pass
    // of batch i + 1 fits the handful of CUs the main scan of batch i leaves free (vf_api.hip: CU-partitioned streams).
    const long long grid = (MODE == kModeSample && a.scan_grid > 0) ? a.scan_grid : gridDim.x;
    const long long swg = (long long)a.samp * WAVES;
    // row range of workgroup v's part in this mode: the sample part is the first swg rows of its range, the main part the rest
    long long lo, hi;
    int ntiles;
    auto part_of = [&](long long v, long long& plo, long long& phi) {
        const long long Ra = a.n * v / grid, Rb = a.n * (v + 1) / grid;
        const long long Rs = (Ra + swg < Rb) ? Ra + swg : Rb;
        plo = MODE == kModeSample ? Ra : Rs;
        phi = MODE == kModeSample ? Rs : Rb;
    };
    part_of(blockIdx.x, lo, hi);
    ntiles = (int)((hi - lo + kRowTile - 1) / kRowTile);  // per workgroup: < 2^31 / 32 rows
    // sample mode: tile T of this workgroup = tile (T % tps) of the sample part of range blockIdx.x + (T / tps) * gridDim.x
    const int tps = (int)((swg + kRowTile - 1) / kRowTile);
    if (MODE == kModeSample) {
        const long long mine = ((long long)grid - blockIdx.x + gridDim.x - 1) / gridDim.x;   // ranges this workgroup walks
        ntiles = (int)(mine > 0 ? mine : 0) * tps;
    }
    // (first row, end of the part, sample-slot base) of a sample tile
    auto sample_tile = [&](int tile, long long& t0, long long& thi, long long& sbase) {
        const int m = tile / tps, j = tile - m * tps;
        const long long v = blockIdx.x + (long long)m * gridDim.x;
        long long plo, phi;
        part_of(v < grid ? v : grid - 1, plo, phi);
        t0 = plo + (long long)j * kRowTile;
        thi = phi;
        sbase = v * swg + (long long)j * kRowTile;
    };
    // Main mode with a.tile_cnt: the LAST kPoolDiv-th of every range is a POOL that any workgroup may take tiles from
    // (global counters, set to 0 by k_sel0); the first part is the owner's alone (LDS counter: a global atomic per
    // tile on every tile measured 10 % slower -- its latency sits in the in-order memory queue ahead of the prefetch).
    constexpr int kPoolDiv = 12;
    const bool pooled = MODE == kModeMain && a.tile_cnt != nullptr;
    auto pool_of = [&](long long v, long long& plo, long long& phi) {
        part_of(v, plo, phi);
        const int nt_v = (int)((phi - plo + kRowTile - 1) / kRowTile);
        plo += (long long)(nt_v - nt_v / kPoolDiv) * kRowTile;   // the pool: tiles [nt - nt / 12, nt)
        if (plo > phi) plo = phi;
    };
    if (pooled) {
        const int nloc = ntiles - ntiles / kPoolDiv;
        hi = lo + (long long)nloc * kRowTile < hi ? lo + (long long)nloc * kRowTile : hi;
        ntiles = nloc;
    }
    const int SS = (a.dp >> 6) / G;
    char* ctl = smem + (size_t)a.dp * QN * 2;
    u32* next_tile = (u32*)(ctl + 4);
    // When a workgroup's own part is exhausted its waves take tiles from the pool with the most tiles left (their own
    // first, as a rule), so that the launch ends when the corpus does, not when the slowest CU does.
    u32* const own_cnt = next_tile;

    // Two register stages (A0, A1), one flat sequence of supersteps over (tile, ss).  The prefetch of
    // the next step is issued UNCONDITIONALLY before the current one is consumed (past the end it
    // re-reads the last step: harmless), so the compiler's counted vmcnt waits never cover the stage
    // in flight.
    // register stages: fp16 rows 4 x h8 per segment, fp8 rows 2 x uint4 (raw codes, converted at the MFMA)
    typedef typename std::conditional<F8 != 0, uint4, h8>::type stage_t;
    stage_t A0[F8 ? 2 * G : 4 * G], A1[F8 ? 2 * G : 4 * G];
    EpiRegs<NT> epi;
    int cur_tile = wid, claimed = 0x7fffffff, tiles_done = 0;
    bool active = cur_tile < ntiles;
    // row index of this lane in `tile`, clamped into the part (32-bit: a shard has < 2^32 rows)
    u32 lo32 = (u32)lo + (u32)r31, hi32m1 = (u32)(hi - 1);
    auto rowof = [&](int tile) {
        if (MODE == kModeSample) {
            long long t0s, this_, sb;
            sample_tile(tile < ntiles ? tile : (ntiles > 0 ? ntiles - 1 : 0), t0s, this_, sb);
            const long long r = t0s + r31;
            return r < this_ - 1 ? r : this_ - 1;
        }
        const u32 r = lo32 + (u32)tile * kRowTile;
        return (long long)(r < hi32m1 ? r : hi32m1);
    };
    // first corpus stage goes out before the query image is staged, so HBM latency overlaps the fill
    if (active) VF_ISSUE(A0, a.rows, a.row_bytes, rowof(cur_tile), 0, h);
    {   // query image -> LDS, verbatim; control block + candidate stage zeroed; tau copied
        const uint4* src = (const uint4*)a.qimg;
        uint4* dst = (uint4*)smem;
        const int nvec = (a.dp >> 3) * QN;
#pragma unroll 4
        for (int i = tid; i < nvec; i += kScanThreads) dst[i] = src[i];
        uint4* z = (uint4*)ctl;
        const int nz = (MODE == kModeMain) ? (kCtlBytes / 16 + a.stage_cap) : 1;
        for (int i = tid; i < nz; i += kScanThreads) z[i] = make_uint4(0u, (i == 0) ? (u32)WAVES : 0u, 0u, 0u);
    }
    __syncthreads();
    if (MODE == kModeMain && tid < QN) ((int*)(ctl + 16))[tid] = a.tau_bin[tid];
    __syncthreads();
    const char* lds_lane = smem + ((4 * h) * QN + r31) * 16;
    unsigned long long* dbg = (MODE == kModeMain && (a.debug & 128) && a.dbg && lane == 0)
                                  ? a.dbg + ((long long)blockIdx.x * WAVES + wid) * 4 : nullptr;
    if (dbg) dbg[0] = wall_clock64();

    u32* seg_cnt = own_cnt;          // the tile counter of the range this wave is working on
    long long s0_base = (long long)blockIdx.x * swg;
    for (;;) {   // one iteration per row range ("segment"): the workgroup's own, then stolen ones (main mode)
    if (active) {
        f16v acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;
        // SS is even (launch_scan picks G so): a tile is SS/2 pairs (A0 then A1) and always starts in A0.
        const int pairs = SS >> 1;
        while (true) {
            long long t0 = lo + (long long)cur_tile * kRowTile, t_hi = hi, t_s0 = s0_base + (t0 - lo);
            if (MODE == kModeSample) sample_tile(cur_tile, t0, t_hi, t_s0);
            {   // one claim per wave (LDS atomic, or an L2 atomic in main mode); the result is consumed at the tile's end
                int v_ = 0;
                if (lane == 0) v_ = (int)atomicAdd(seg_cnt, 1u);
                claimed = __builtin_amdgcn_readfirstlane(v_);
            }
            const long long myrow = rowof(cur_tile);
            for (int p = 0; p + 1 < pairs; ++p) {
                VF_ISSUE(A1, a.rows, a.row_bytes, myrow, 2 * p + 1, h);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE the MFMAs
                VF_COMPUTE(acc, A0, lds_lane, 2 * p);
                VF_ISSUE(A0, a.rows, a.row_bytes, myrow, 2 * p + 2, h);
                __builtin_amdgcn_sched_barrier(0);
                VF_COMPUTE(acc, A1, lds_lane, 2 * p + 1);
            }
            VF_ISSUE(A1, a.rows, a.row_bytes, myrow, SS - 1, h);
            __builtin_amdgcn_sched_barrier(0);
            VF_COMPUTE(acc, A0, lds_lane, SS - 2);
            // last superstep of the tile: epilogue operands first, then the next tile's first stage
            const bool more = claimed < ntiles;
            epi_prefetch<NT, MODE>(epi, a, t0, lane, ((tiles_done & (WAVES - 1)) == wid));
            VF_ISSUE(A0, a.rows, a.row_bytes, rowof(more ? claimed : cur_tile), 0, h);
            __builtin_amdgcn_sched_barrier(0);
            VF_COMPUTE(acc, A1, lds_lane, SS - 1);
            tile_epilogue<NT, MODE>(a, acc, epi, t0, t_hi, t_s0, lane, ctl);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;
            ++tiles_done;
            if (!more) break;
            cur_tile = claimed;
        }
    }
    if (!pooled) break;
    // ---- steal: the range with the most unclaimed tiles (lane l looks at ranges 4l .. 4l+3; counters only grow, so a
    // stale read over-estimates what is left and at worst costs a failed claim)
    {
        int best_left = 0, best_v = -1;
        for (int v0 = 0; v0 < (int)grid; v0 += 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = v0 + lane * 4 + j;
                if (v < (int)grid) {
                    long long plo, phi;
                    pool_of(v, plo, phi);
                    const int nt_v = (int)((phi - plo + kRowTile - 1) / kRowTile);
                    const int c = (int)__hip_atomic_load(a.tile_cnt + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int left = nt_v - c;
                    if (left > best_left) { best_left = left; best_v = v; }
                }
            }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const int ol = __shfl_xor(best_left, off), ov = __shfl_xor(best_v, off);
            if (ol > best_left || (ol == best_left && ov > best_v)) { best_left = ol; best_v = ov; }
        }
        if (best_left <= 0) break;   // nothing left anywhere: the wave is done
        const int v = __builtin_amdgcn_readfirstlane(best_v);
        pool_of(v, lo, hi);
        ntiles = (int)((hi - lo + kRowTile - 1) / kRowTile);
        lo32 = (u32)lo + (u32)r31; hi32m1 = (u32)(hi - 1);
        seg_cnt = a.tile_cnt + v;
        s0_base = 0;
        int t_ = 0;
        if (lane == 0) t_ = (int)atomicAdd(seg_cnt, 1u);
        cur_tile = __builtin_amdgcn_readfirstlane(t_);
        active = cur_tile < ntiles;
        if (active) VF_ISSUE(A0, a.rows, a.row_bytes, rowof(cur_tile), 0, h);
    }
    }
    if (dbg) { dbg[1] = wall_clock64(); }
    if (MODE == kModeMain) {
        // Flush the staged candidates to the per-query global lists, once per workgroup: rank the
        // entries per query in LDS, claim one contiguous range per non-empty query with ONE returning
        // global atomic (counters sit on separate 128-B lines), then write the entries.
        __syncthreads();
        if (dbg) dbg[2] = wall_clock64();
        const u32 staged = (a.debug & 16) ? 0u : *(const u32*)ctl;  // bit 4: timing experiment, no flush
        const u32 nst = staged < (u32)a.stage_cap ? staged : (u32)a.stage_cap;
        uint4* ent = (uint4*)(ctl + kCtlBytes);
        u32* qcnt = (u32*)smem;         // the query image is dead now: reuse its first bytes
        u32* qbase = qcnt + QN;
        if (tid < QN) qcnt[tid] = 0u;
        __syncthreads();
        for (u32 i = tid; i < nst; i += kScanThreads) {
            const uint4 e = ent[i];
            const u32 q = e.z & 0xFFu;
            if (e.w != 1u || q >= (u32)QN) continue;  // never index global memory with an unwritten entry
            ent[i].w = 2u + atomicAdd(qcnt + q, 1u);  // local rank, tagged
        }
        __syncthreads();
        if (tid < QN) {
            const u32 c = qcnt[tid];
            qbase[tid] = c ? atomicAdd(a.cnt + tid * kCntStride, c) : 0u;
        }
        __syncthreads();
        for (u32 i = tid; i < nst; i += kScanThreads) {
            const uint4 e = ent[i];
            if (e.w < 2u) continue;
            const u32 q = e.z & 0xFFu;
            const u32 gs = qbase[q] + (e.w - 2u);
            if (gs < (u32)a.cap) a.cand[(long long)q * a.cap + gs] = ((u64)e.y << 32) | (u64)e.x;
        }
        if (dbg) dbg[3] = wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------------
// k_scan2: the main scan with WHOLE-LINE corpus loads (round 3).  fp16 rows, main mode.
//
// What round 3 measured (tools/ubench/stream_read.hip, profiles/r03_stream_read.log): k_scan's A-operand loads -- lane
// (r, h) reads 16 B of ITS row, so one instruction touches 32 cache lines and takes 32 B of each, and four instructions
// walk every line -- stream at 5.4-5.8 TB/s on 256 CUs, 5.1-5.5 on 224 and 4.5-5.0 on 192, and DEEPER prefetch makes it
// worse: the pattern is bound per CU (every line stays allocated in the 32 KB L1 until its fourth instruction has come
// by).  The same bytes read as whole lines (8 lanes x 16 B per line, 8 lines per instruction) stream at 6.2-6.4 TB/s
// on ANY of those CU counts.  An MFMA A operand cannot be loaded that way into registers (a row lives in two lanes, a
// line would land in eight), so the rows go global -> LDS by LDS-DMA (global_load_lds_dwordx4: the lanes of a DMA
// instruction fetch whole lines, and WHICH piece a lane fetches is free, so the bank swizzle is applied through the
// source addresses) and the A fragments are read back with ds_read_b128, like the B fragments (queries).
//
// Geometry: 256 threads = one wave per SIMD; a wave owns 32-row tiles (claimed from the LDS counter as in k_scan) and
// a private ring of kRing 4-KB segment buffers (32 rows x 128 B): no barrier anywhere in the loop.  A segment is four
// DMA instructions; two segments (8 KB) stay in flight while the third is consumed, all three during a tile's epilogue.  LDS: query image (96 KB at d = 768)
// + ring 4 x 12 KB + per-wave scratch + control block + candidate stage (what is left: ~15 KB).
//   slot(row r, piece p) of a segment buffer = r * 8 + (p ^ ((r >> 1) & 7))   [16-byte slots]: a ds_read_b128 phase (16
//   lanes = rows 16 g .. 16 g + 15, one piece) then hits 16 distinct bank groups.
//   DMA instruction m (0..3) fills slots 64 m .. 64 m + 63: lane l -> row 8 m + (l >> 3), piece (l & 7) ^ ((row >> 1) & 7).
// The per-tile epilogue operands (1 / norm of the 32 rows, and the global thresholds when this wave's turn to sync them
// comes) travel by the same DMA queue into the wave's scratch at the START of the tile, so the loop contains no vector
// memory instruction the compiler knows about and none of its s_waitcnt vmcnt(0) (it cannot count hand-issued DMAs).
// MFMA layout, k-slot map, accumulators, threshold filter, candidate stage, refresh and flush are k_scan's.
// ------------------------------------------------------------------------------------------------
#ifndef VF_SCAN2_SERVICE
#define VF_SCAN2_SERVICE 0   // 1: a fifth wave publishes the staged blocks (measured 8 % SLOWER, round 3: DESIGN.md, k_scan2) -- A/B builds only
#endif
constexpr bool kScan2Service = VF_SCAN2_SERVICE != 0;
constexpr int kScan2Waves = 4, kScan2Threads = (kScan2Waves + (kScan2Service ? 1 : 0)) * 64, kRing = 3, kSegBytes = 4096, kScratchBytes = 1024;   // four streaming waves + the service wave; scratch: two 512-B halves (tile parity)

__device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void dma4(const void* g, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_base) : "memory");
}
// One 4-KB ring segment = four 1-KB pieces (eight rows each) in ONE asm statement: scalar 64-bit base + a 32-bit lane offset per piece
// (no 64-bit vector add per piece), M0 saved and restored once, the LDS destination stepped in M0 itself.
// The loads carry the NON-TEMPORAL policy: a corpus row is read once per batch, by exactly one of these instructions (whole 128-byte
// lines, the swizzle is in the source address), and 15 GB of them pass through per launch.  Same box, alternating
// (profiles/r06_dma_nt_ab.log): 10M x 768 fp16 2.42-2.50 -> 2.20-2.30 ms per batch = 0.77-0.80 -> **0.84-0.87 of 8 TB/s** -- the
// "copy ceiling" of ~6.3 TB/s that rounds 3-6 took for the bound is the DEFAULT policy's; 1.25M rows 0.336-0.344 -> 0.318-0.321,
// configs[1] 0.289-0.294 -> 0.275-0.276, 10M x 1024 e4m3 1.79-1.81 -> 1.748.  (k_scan's register loads LOSE half their rate with nt --
// a lane pair shares every line and four instructions walk it: round 3's measurement above stream_load -- which is why the hint had
// been written off before the whole-line DMA existed.)
#ifndef VF_ROW_POLICY
#define VF_ROW_POLICY "nt"   // (A/B builds: -DVF_ROW_POLICY='"sc1 nt"' ...; the other policies measured: profiles/r06_row_policy_ab.log)
#endif
__device__ __forceinline__ void dma16x4(unsigned long long ua, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5 " VF_ROW_POLICY "\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5 " VF_ROW_POLICY "\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5 " VF_ROW_POLICY "\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5 " VF_ROW_POLICY "\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(ua), "s"(lds_base) : "memory", "scc");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

size_t scan2_lds_bytes(int dp, int qn_tile, int stage_cap) {
    const size_t img = (size_t)dp * qn_tile * 2;
    return img + (size_t)kScan2Waves * (kRing * kSegBytes + kScratchBytes) + kCtlBytes + (size_t)stage_cap * 16;
}
int scan2_stage_cap(int dp, int qn_tile, int rows_are_fp8) {   // candidate-stage entries that fit beside image + rings; < 256 = "does not fit"
    const size_t fixed = scan2_lds_bytes(dp, qn_tile, 0);
    if ((dp >> (rows_are_fp8 ? 7 : 6)) < kRing) return 0;  // the ring holds three 128-byte segments of ONE row set at start-up
    if (fixed + 256 * 16 > 160 * 1024) return 0;
    const size_t area = std::min<size_t>(160 * 1024 - fixed, 32 * 1024);
    return (int)(area / 16);
}

// The SERVICE wave of k_scan2 (wave 4; it shares SIMD 0 with streaming wave 0).  Round 3 measured where a small shard's launch
// loses its time (tools/stamps_tiles.py, profiles/r03_stamps_tiles.log): the first tiles after the sample-seeded threshold take
// 10-13 us instead of 8 -- eight candidates per tile, and every completed block of staged candidates made ITS wave publish
// 2 R histogram atomics and refresh a threshold with two dependent loads, all of it in that wave's in-order vmcnt queue ahead of
// its prefetched segments: 37 us of a 330-us launch at 1.25M rows (11 %; with staging switched off the tiles run at 7.7 us
// from the fourth on).  k_scan's second wave per SIMD covered for that; here one wave per SIMD streams and nothing did.
// Now the streaming waves only stage (LDS), and this wave watches the stage counter: it publishes every completed block,
// refreshes the threshold of that block's last query, and folds the global thresholds into the workgroup's copy -- the
// sync_tau DMA of the streaming waves is gone with it.  It leaves when all four streaming waves have counted themselves out.
template <int NT>
__device__ __forceinline__ void k_scan2_service(const ScanArgs& a, char* ctl, int lane) {
    constexpr int QN = NT * kQueryTile;
    volatile u32* stage_cnt = (volatile u32*)ctl;
    volatile u32* waves_done = (volatile u32*)(ctl + 8);
    int* tau_lds = (int*)(ctl + 16);
    const uint4* stage_ent = (const uint4*)(ctl + kCtlBytes);
    const u32 R = (u32)a.refresh_every;
    u32 done_blk = 0;
    for (;;) {
        const u32 fin = *waves_done;
        const u32 staged = *stage_cnt;
        const u32 avail = staged < (u32)a.stage_cap ? staged : (u32)a.stage_cap;
        // a block is published once the NEXT one has started filling (its last entries have surely been written by then), or
        // at the end
        const u32 ready = fin == (u32)kScan2Waves ? avail / R : (avail >= R / 2 ? (avail - R / 2) / R : 0u);
        if (done_blk < ready) {
            const u32 blk = done_blk++;
            int qq = -1;
            for (u32 j = (u32)lane; j < R; j += 64u) {
                const uint4 en = stage_ent[blk * R + j];
                if (en.w == 1u) {
                    const u32 q = en.z & 0xFFu, bin = en.z >> 8;
                    if (q < (u32)QN && bin < (u32)kHistBins) {
                        atomicAdd(a.hist + (long long)q * kHistBins + bin, 1u);
                        atomicAdd(a.hist_coarse + q * 64 + (bin >> 5), 1u);
                        if (j == R - 1) qq = (int)q;
                    }
                }
            }
            // the block's last entry names the query whose threshold is refreshed (any lane may hold it)
#pragma unroll
            for (int o = 32; o; o >>= 1) qq = max(qq, __shfl_xor(qq, o));
            if (qq >= 0 && !(a.debug & 2)) {
                const int nb = wave_tau_two_level(a.hist_coarse + qq * 64, a.hist + (long long)qq * kHistBins, a.kprime, lane);
                if (lane == 0 && nb > 0) { atomicMax(a.tau_bin + qq, nb); atomicMax(tau_lds + qq, nb); }
            }
            continue;
        }
        if (fin == (u32)kScan2Waves) break;
        // nothing to publish: fold the other workgroups' thresholds in, then doze
        if (lane < QN) {
            const int tg = __hip_atomic_load(a.tau_bin + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(tau_lds + lane, tg);
        }
        __builtin_amdgcn_s_sleep(24);
    }
}

template <int NT, int F8>
__global__ __launch_bounds__(kScan2Threads) void k_scan2(ScanArgs a) {
    const unsigned long long t_entry = (a.debug & 512) ? wall_clock64() : 0ull;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int QN = NT * kQueryTile;
    constexpr int MODE = kModeMain;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const long long grid = gridDim.x;
    const long long swg = (long long)a.samp * (kScanThreads / 64);   // the sample part is defined by the sample pass's geometry
    long long range_v = blockIdx.x;
#ifdef VF_EXPERIMENTS
    // timing experiment (debug bit 5, INVALID RESULTS: ranges may be taken twice or not at all): the range a workgroup scans is a function
    // of the COMPUTE UNIT it landed on, so that a CU reads the same rows launch after launch -- is the slow start of a launch a per-CU
    // state (its translation cache) that a workgroup resident across batches would keep?
    if ((a.debug & 32) && a.sib) {   // a.sib: [1024] u32 table CU -> range, 0xFFFFFFFF = not seen yet (the first launch fills it from blockIdx)
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_id), "=s"(xcc_id));
        const unsigned key = (xcc_id & 0x7u) * 128u + ((hw_id >> 13) & 3u) * 32u + ((hw_id >> 12) & 1u) * 16u + ((hw_id >> 8) & 0xFu);
        const unsigned seen = __hip_atomic_load(a.sib + key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen != 0xFFFFFFFFu && seen < (unsigned)grid) range_v = seen;
        else if (threadIdx.x == 0) __hip_atomic_store(a.sib + key, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
    const long long Ra = a.n * range_v / grid, Rb = a.n * (range_v + 1) / grid;
    const long long lo = (Ra + swg < Rb) ? Ra + swg : Rb, hi = Rb;
    const int ntiles = (int)((hi - lo + kRowTile - 1) / kRowTile);
    const int S = a.dp >> (F8 ? 7 : 6);                               // 128-byte segments per row (64 halves, or 128 e4m3 codes)
    // LDS carve-up (offsets are multiples of 16; the rings of 1024)
    char* ring = smem + (size_t)wid * (kRing * kSegBytes);
    char* img = smem + kScan2Waves * kRing * kSegBytes;
    char* scratch = img + (size_t)a.dp * QN * 2 + (size_t)wid * kScratchBytes;   // per tile parity: [0,256) tau_bin[64] | [256,512) 1/norm of the tile's 32 rows (twice)
    char* ctl = img + (size_t)a.dp * QN * 2 + kScan2Waves * kScratchBytes;
    u32* next_tile = (u32*)(ctl + 4);
    // wave-uniform LDS bases of the DMAs (M0 is a scalar register: make the uniformity explicit)
    const unsigned ring_l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(ring)), scratch_l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(scratch));

    // this lane's part in the four DMA instructions of a segment: row 8 m + (lane >> 3), piece (lane & 7) ^ swizzle(row)
    const int drow = lane >> 3;
    // a tile's source: a scalar base (its first row) + per piece a 32-bit lane offset; a segment's four pieces go in one asm statement (dma16x4)
    struct TileSrc { unsigned long long ua; unsigned v[4]; };
    auto src_of = [&](int tile) -> TileSrc {
        const long long t0_ = lo + (long long)tile * kRowTile;
        TileSrc ts;
        ts.ua = (unsigned long long)(a.rows + t0_ * a.row_bytes);
        ts.ua = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ts.ua >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ts.ua);   // (wave-uniform by construction)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = 8 * m + drow;
            long long r = t0_ + row;
            r = r < hi - 1 ? r : hi - 1;                              // rows past the part's end re-read its last row (masked in the epilogue)
            ts.v[m] = (unsigned)((r - t0_) * a.row_bytes) + ((((lane & 7) ^ ((row >> 1) & 7))) << 4);
        }
        return ts;
    };
    auto issue_seg = [&](const TileSrc& src, int seg, int buf) {
        dma16x4(src.ua + (unsigned long long)seg * 128ull, src.v[0], src.v[1], src.v[2], src.v[3], ring_l + buf * kSegBytes);
    };
    // 1 / norm of a tile's rows (+ the global thresholds) into the scratch half of that tile's parity: a tile's operands are
    // issued while the PREVIOUS tile is still being consumed and may land before that tile's epilogue has read its own
    auto issue_epi = [&](long long t0, bool sync_tau, int par) {
        par = __builtin_amdgcn_readfirstlane(par);                    // (wave-uniform by construction; M0 wants to be told)
        dma4(a.inv_scan + t0 + r31, scratch_l + par * 512 + 256);     // 64 words: lanes 32..63 repeat the 32 rows
        if (sync_tau) dma4(a.tau_bin + (lane < QN ? lane : QN - 1), scratch_l + par * 512);
    };

    // (Round 6 measured two start-up orders and kept neither: every workgroup's range walked from a per-workgroup rotated tile instead of
    //  its first row, and the four waves of a workgroup started a quarter of a tile time apart -- the slow first tiles of a small
    //  shard's launch are neither a lockstep of the ranges nor of the waves: profiles/r06_scan2_rotation_and_stagger.log)
    const bool service = kScan2Service && wid == kScan2Waves;          // wave 4: publishes blocks, refreshes and syncs thresholds (k_scan2_service)
    int cur_tile = __builtin_amdgcn_readfirstlane(wid);                // (a scalar from here on: tile addresses stay in SGPRs)
    bool active = !service && cur_tile < ntiles;
    TileSrc src_cur{}, src_nxt{};
    if (active) {
        src_cur = src_of(cur_tile);
        issue_epi(lo + (long long)cur_tile * kRowTile, !kScan2Service, 0);
        for (int sg = 0; sg < kRing; ++sg) issue_seg(src_cur, sg, sg);   // the whole ring (S >= kRing: scan2_stage_cap)
    }
    {   // query image -> LDS verbatim; control block + candidate stage zeroed
        // (by LDS-DMA like everything else here: 1 KB per wave instruction, no register hop; the image is 32 or 64 queries x
        //  dp / 8 sixteen-byte units = a multiple of 256 units, and wave w of the streaming four takes every fourth KB)
        const char* srcq = (const char*)a.qimg;
        const int nkb = ((a.dp >> 3) * QN) >> 6;
        const unsigned img_l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(img));
        const int wu = __builtin_amdgcn_readfirstlane(wid);   // (M0 is scalar: make the wave index uniform for the compiler)
        if (wu < kScan2Waves)
            for (int c = wu; c < nkb; c += kScan2Waves) dma16(srcq + ((long long)c << 10) + (lane << 4), img_l + ((unsigned)c << 10));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share (and its ring) has landed; the barrier below covers the others'
        uint4* z = (uint4*)ctl;
        const int nz = kCtlBytes / 16 + a.stage_cap;
        for (int i = tid; i < nz; i += kScan2Threads) z[i] = make_uint4(0u, (i == 0) ? (u32)kScan2Waves : 0u, 0u, 0u);
    }
    __syncthreads();
    if (tid < QN) ((int*)(ctl + 16))[tid] = a.tau_bin[tid];
    __syncthreads();

    const char* lds_lane = img + ((4 * h) * QN + r31) * 16;
    const int asw = (r31 >> 1) & 7;
    const char* a_lane = ring + r31 * 128;                             // + buf * 4096 + ((4 h + i) ^ asw) * 16
    const int dbg_rec = (a.debug & 512) ? 72 : 4;   // bit 9: + the start time of the wave's first 64 tiles, [68] kernel entry, [69] tiles taken
    unsigned long long* dbg = ((a.debug & 128) && a.dbg && lane == 0 && !service) ? a.dbg + ((long long)blockIdx.x * kScan2Waves + wid) * dbg_rec : nullptr;
    if (dbg) dbg[0] = wall_clock64();
    if (dbg && dbg_rec > 4) {
        dbg[68] = t_entry;
        unsigned hw_id, xcc_id;   // where this wave ran (CU / SE / SIMD / wave slot, XCD): tools/stamps_gap.py pairs a CU's consecutive workgroups
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_id), "=s"(xcc_id));
        dbg[70] = ((unsigned long long)xcc_id << 32) | hw_id;
    }
    if (service) k_scan2_service<NT>(a, ctl, lane);
#ifdef VF_EXPERIMENTS
    // timing experiment (debug bit 6, INVALID RESULTS): the workgroup walks its range twice in one life -- is the slow start of a
    // launch tied to the workgroup's start (then the second pass runs at the steady rate) or to the start of a stream of rows?
    const int passes = (a.debug & 64) ? 2 : 1;
#else
    constexpr int passes = 1;
#endif
    int tiles_done = 0;
    for (int pass = 0; pass < passes; ++pass) {
    if (pass > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *next_tile = (u32)kScan2Waves;
        __syncthreads();
        cur_tile = wid;
        if (active) {
            src_cur = src_of(cur_tile);
            issue_epi(lo + (long long)cur_tile * kRowTile, !kScan2Service, tiles_done & 1);
#pragma unroll
            for (int sg = 0; sg < kRing; ++sg) issue_seg(src_cur, sg, sg);
        }
    }
    if (active) {
        f16v acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;
        int buf = 0;                                                   // ring buffer of the segment about to be consumed
        while (true) {
            const long long t0 = lo + (long long)cur_tile * kRowTile;
            int claimed;
            {
                int v_ = 0;
                if (lane == 0) v_ = (int)atomicAdd(next_tile, 1u);
                claimed = __builtin_amdgcn_readfirstlane(v_);
            }
            const bool more = claimed < ntiles;
            const int nxt = more ? claimed : cur_tile;
            src_nxt = src_of(nxt);
            // the service wave folds the global thresholds into tau_lds; without it the waves take turns
            const bool sync_now = !kScan2Service && (tiles_done & (kScan2Waves - 1)) == wid;
            const bool sync_next = !kScan2Service && ((tiles_done + 1) & (kScan2Waves - 1)) == wid;
            if (dbg && dbg_rec > 4 && tiles_done < 64) dbg[4 + tiles_done] = wall_clock64();
            for (int sg = 0; sg < S; ++sg) {
                // segments sg + 1 and sg + 2 (8 DMA instructions, and the <= 2 epilogue words issued among them) may still be in
                // flight; everything older has landed
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                const char* ab = a_lane + buf * kSegBytes;
                if constexpr (F8 == 0) {
                    const char* bb = lds_lane + (long long)sg * (8 * QN * 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const h8 af = *(const h8*)(ab + (((4 * h + i) ^ asw) << 4));
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const h8 bf = *(const h8*)(bb + i * (QN * 16) + nt * (kQueryTile * 16));
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[nt], 0, 0, 0);
                        }
                    }
                } else {
                    // e4m3 rows: a 128-byte segment is 128 elements = two 64-element chunks; in chunk c lane (r, h) owns the
                    // element blocks 8 c + 4 h + i (k_scan's map: the query image is shared), i.e. the two 16-byte pieces
                    // 4 c + 2 h, + 1 of its row: two conflict-free ds_read_b128, converted in registers (exactly: every e4m3
                    // value is an fp16 value)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const uint4 w0 = *(const uint4*)(ab + (((4 * c + 2 * h) ^ asw) << 4));
                        const uint4 w1 = *(const uint4*)(ab + (((4 * c + 2 * h + 1) ^ asw) << 4));
                        const char* bb = lds_lane + (long long)(sg * 16 + c * 8) * (QN * 16);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint4 w = i < 2 ? w0 : w1;
                            const h8 af = (i & 1) ? cvt8_e4m3(w.z, w.w) : cvt8_e4m3(w.x, w.y);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const h8 bf = *(const h8*)(bb + i * (QN * 16) + nt * (kQueryTile * 16));
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[nt], 0, 0, 0);
                            }
                        }
                    }
                }
                // the buffer just consumed (its fragments are in registers: the MFMAs that used them have issued) takes
                // segment sg + 3 -- past the tile's end the next tile's segment 0 / 1 / 2, preceded by that tile's epilogue
                // operands -- so that all THREE ring buffers are in flight while the epilogue below runs (with the issue in
                // front of the compute, as first written, a tile whose epilogue takes the candidate path left the queue dry:
                // a wave has no partner on its SIMD to cover for it)
                // The refill overwrites the LDS this segment's fragments were READ from: those ds_reads must have RETURNED, not merely
                // issued.  The matrix instructions above wait for them -- as long as the compiler keeps them above this point; it is
                // free not to (they touch registers only, the asm below is a memory barrier to memory operations only): in round 4's
                // fp8-instruction variant of this loop it sank all four MFMAs and their lgkmcnt waits BELOW the refill, and a DMA that
                // found its line in L2 beat the reads it should have followed -- one wrong score in ~10^4, a lost row in 12 of 40 runs
                // (found by the fuzz; that variant is gone).  In the shipped variants the wait is already lgkmcnt(0) here, so this costs
                // nothing; it turns an accident of scheduling into a guarantee.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int s3 = sg + kRing;
                if (s3 < S) issue_seg(src_cur, s3, buf);
                else {
                    if (s3 == S) issue_epi(lo + (long long)nxt * kRowTile, sync_next, (tiles_done + 1) & 1);
                    issue_seg(src_nxt, s3 - S, buf);
                }
                buf = buf + 1 == kRing ? 0 : buf + 1;
            }
            // epilogue operands of THIS tile landed long ago (issued before its first segment, into its parity's half)
            const char* sc = scratch + (tiles_done & 1) * 512;
            EpiRegs<NT> epi;
            epi.inv_lane = *(const float*)(sc + 256 + r31 * 4);
            epi.inv_lds = sc + 256;
            epi.sync_tau = sync_now;
            if (sync_now) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) epi.tau_g[nt] = *(const int*)(sc + (nt * kQueryTile + r31) * 4);
            }
            tile_epilogue<NT, MODE, !kScan2Service, true>(a, acc, epi, t0, hi, 0, lane, ctl);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;
            ++tiles_done;
            if (!more) break;
            cur_tile = nxt;
            src_cur = src_nxt;
        }
    }
    }
    if (dbg && dbg_rec > 4) dbg[69] = (unsigned long long)tiles_done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (unused) prefetches have landed before LDS is reused
    if (!service && lane == 0) atomicAdd((u32*)(ctl + 8), 1u);   // this streaming wave is done: the service wave leaves after the fourth
    if (dbg) dbg[1] = wall_clock64();
    // ---- flush the staged candidates (k_scan's, on this block size)
    __syncthreads();
    if (dbg) dbg[2] = wall_clock64();
    {
        const u32 staged = *(const u32*)ctl;
        const u32 nst = staged < (u32)a.stage_cap ? staged : (u32)a.stage_cap;
        uint4* ent = (uint4*)(ctl + kCtlBytes);
        u32* qcnt = (u32*)img;          // the query image is dead now
        u32* qbase = qcnt + QN;
        if (tid < QN) qcnt[tid] = 0u;
        __syncthreads();
        for (u32 i = tid; i < nst; i += kScan2Threads) {
            const uint4 e = ent[i];
            const u32 q = e.z & 0xFFu;
            if (e.w != 1u || q >= (u32)QN) continue;
            ent[i].w = 2u + atomicAdd(qcnt + q, 1u);
        }
        __syncthreads();
        if (tid < QN) {
            const u32 c = qcnt[tid];
            qbase[tid] = c ? atomicAdd(a.cnt + tid * kCntStride, c) : 0u;
        }
        __syncthreads();
        for (u32 i = tid; i < nst; i += kScan2Threads) {
            const uint4 e = ent[i];
            if (e.w < 2u) continue;
            const u32 q = e.z & 0xFFu;
            const u32 gs = qbase[q] + (e.w - 2u);
            if (gs < (u32)a.cap) a.cand[(long long)q * a.cap + gs] = ((u64)e.y << 32) | (u64)e.x;
        }
    }
    if (dbg) dbg[3] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// k_scan2r (round 6): k_scan2 with HALF of the query image in REGISTERS and a ring twice as deep.
//
// What round 6 measured on the 8-GPU rank's shard (1.25M x 768, tools/stamps_gap.py, profiles/r06_*): a wave of k_scan2 keeps three
// 4-KB segments in flight and its rate is ring / loaded latency -- 12 KB / 1.7-2.4 us -- so every cycle it spends off the streaming
// path (the candidate path of a tile with a candidate: ~0.3 us, entered in most tiles of a small shard) comes straight off its
// throughput: the ring is full, nothing more can be asked for, and nothing can be caught up afterwards.  A deeper ring has no room:
// at d = 768 the 64-query image alone takes 96 of the 160 KB.  But the image is the B operand of EVERY matrix instruction and a wave
// owns a whole SIMD's register file (one wave per SIMD: 512 registers, of which k_scan2 uses ~230): the B fragments of the first
// RB = S / 2 segments of a row (k = 0 .. 383 at d = 768) are loaded ONCE per workgroup life into 48 x 4 registers per lane and stay
// there; LDS holds only the other half of the image (48 KB) and the four rings grow from 3 to 6 segments (24 KB in flight per wave).
// Half of the B fragment reads leave the LDS pipe with it.  Everything else -- row DMA with the swizzle in the source address, tile
// claiming, epilogue operands by DMA, threshold filter, candidate stage, flush -- is k_scan2's, and so are the scores: the same
// products accumulate in the same order (segment by segment), so the approximate scores, the candidates and the results are
// bit-identical to k_scan2's.  Built for dp = 768 (S = 12); other widths keep k_scan2.
// The register half is fetched with ordinary loads BEFORE anything else and settled with the s_waitcnt BUILTIN (which the compiler's
// own wait insertion accounts for): no vector-memory load with a register destination is pending while tiles run (k_attention2's rule).
// ------------------------------------------------------------------------------------------------
// The shapes it is built for -- (row bytes per 128-byte segment count S, register segments RB, ring depth RING):
//   fp16 rows, dp =  768: S = 12, RB = 6 (192 registers of B fragments at 64 queries), RING = 6
//   fp16 rows, dp = 1024: S = 16, RB = 6, RING = 4 (80 KB of image in LDS: k_scan2 has no room for this width at all)
//   fp16 rows, dp =  512: S =  8, RB = 4, RING = 6;   dp = 384: S = 6, RB = 3, RING = 6
//   e4m3 rows, dp =  768: S =  6, RB = 3 (a segment is 128 elements: 16 KB of image, 64 registers), RING = 6
//   e4m3 rows, dp = 1024: S =  8, RB = 3 (with the 32 accumulators the 256 accumulator registers hold no fourth), RING = 4
// (e4m3 rows are converted in registers like k_scan2's F8 variant: every e4m3 value is an fp16 value, the image is shared.)
struct Scan2rShape { int S, RB, RING; };
static Scan2rShape scan2r_shape(int dp, int f8) {
    if (!f8 && dp == 768) return {12, 6, 6};
    if (!f8 && dp == 1024) return {16, 6, 4};   // (bge-m3 / bge-large rows: the reference's own width, config/example.yaml:3)
    if (!f8 && dp == 512) return {8, 4, 6};
    if (!f8 && dp == 384) return {6, 3, 6};
    if (f8 && dp == 768) return {6, 3, 6};
    if (f8 && dp == 1024) return {8, 3, 4};
    return {0, 0, 0};
}

size_t scan2r_lds_bytes(int dp, int qn_tile, int stage_cap, int f8) {
    const Scan2rShape sh = scan2r_shape(dp, f8);
    const size_t seg_img = (size_t)(f8 ? 128 : 64) * qn_tile * 2;
    return (size_t)(sh.S - sh.RB) * seg_img + (size_t)kScan2Waves * (sh.RING * kSegBytes + kScratchBytes) + kCtlBytes + (size_t)stage_cap * 16;
}
int scan2r_stage_cap(int dp, int qn_tile, int f8) {   // < 256 = "not this kernel"
    if (scan2r_shape(dp, f8).S == 0) return 0;
    const size_t fixed = scan2r_lds_bytes(dp, qn_tile, 0, f8);
    if (fixed + 256 * 16 > 160 * 1024) return 0;
    const size_t area = std::min<size_t>(160 * 1024 - fixed, 32 * 1024);
    return (int)(area / 16);
}

// AR = 1 (the shipped form): the register part of the image is PINNED to the accumulator half of the register file -- an empty asm with a
// "+a" operand per fragment, after which the fragment IS an accumulator-register value and the matrix instruction names it as its B
// operand directly (v_mfma ... v[6:9], a[32:35], ...) -- and every LDS read of a segment is issued before its first matrix instruction.
// Left to the allocator (AR = 0, the first form, kept under VF_EXPERIMENTS for the A/B) the fragments that did not fit the 256 ordinary
// registers were SPILLED to accumulator registers and copied back, four v_accvgpr_read per matrix instruction, nothing was left for
// temporaries, and every ds_read was followed by lgkmcnt(0) and ONE matrix instruction.
template <int NT, int MODE, int F8 = 0, int S = 12, int RB = 6, int RING = 6, int AR = 1>
__global__ __launch_bounds__(kScan2Waves * 64) void k_scan2r(ScanArgs a) {
    const unsigned long long t_entry = (a.debug & 512) ? wall_clock64() : 0ull;
    const unsigned long long c_entry = (a.debug & 512) ? __builtin_amdgcn_s_memtime() : 0ull;   // shader cycles: the in-kernel clock = [71] / ([3] - [68]) x 100 MHz
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int QN = NT * kQueryTile, THREADS = kScan2Waves * 64;
    constexpr int J = F8 ? 8 : 4;                       // B fragments (matrix instructions per query tile) per 128-byte row segment
    constexpr int SEGIMG = J * 2 * QN * 16;             // image bytes of a segment: 16 (e4m3: 128 elements) or 8 k-groups x QN x 16 B
    // fragment j of a segment: k-group 8 sg + 4 h + j (fp16 rows) or 16 sg + 8 (j >> 2) + 4 h + (j & 3) (e4m3 rows: two 64-element chunks)
    auto frag_off = [](int j) { return F8 ? ((j >> 2) * 8 + (j & 3)) * (QN * 16) : j * (QN * 16); };
    static_assert(S >= RING && RB <= S && RB <= 6, "ring fill and the spelled-out register segments");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    // Main mode: gridDim.x ranges, workgroup b owns range b minus its sample part.  SAMPLE mode (round 6): a.scan_grid ranges, workgroup b
    // walks the sample parts (the first samp x 8 rows) of ranges b, b + gridDim.x, ... -- k_scan's sample geometry and s0 slots, so k_sel0
    // and the main scan see what they saw -- with THIS kernel's operand path: the pass runs on the 32 CUs the main scans leave free, a
    // CU's rate is bytes in flight / latency, and k_scan's register-staged loads kept ~15 GB/s per CU (4 rounds of 128 workgroups, 70 us
    // for 11 MB at 8 rows per wave) where a six-segment ring per wave keeps 96 KB in flight.
    const long long grid = (MODE == kModeSample && a.scan_grid > 0) ? a.scan_grid : gridDim.x;
    const long long swg = (long long)a.samp * (kScanThreads / 64);
    const int tps = (int)((swg + kRowTile - 1) / kRowTile);            // tiles of a range's sample part
    long long lo = 0, hi = 0;
    int ntiles;
    if (MODE == kModeSample) {
        const long long mine = (grid - (long long)blockIdx.x + gridDim.x - 1) / gridDim.x;
        ntiles = (int)(mine > 0 ? mine : 0) * tps;
    } else {
        const long long Ra = a.n * blockIdx.x / grid, Rb = a.n * (blockIdx.x + 1) / grid;
        lo = (Ra + swg < Rb) ? Ra + swg : Rb; hi = Rb;
        ntiles = (int)((hi - lo + kRowTile - 1) / kRowTile);
    }
    // (first row, end of the part, sample-slot base) of a tile
    auto tile_rows = [&](int tile, long long& t0, long long& thi, long long& sbase) {
        if (MODE == kModeSample) {
            const int m = tile / tps, j = tile - m * tps;
            long long v = (long long)blockIdx.x + (long long)m * gridDim.x;
            v = v < grid ? v : grid - 1;
            const long long Ra = a.n * v / grid, Rb = a.n * (v + 1) / grid;
            thi = (Ra + swg < Rb) ? Ra + swg : Rb;
            t0 = Ra + (long long)j * kRowTile;
            sbase = v * swg + (long long)j * kRowTile;
        } else {
            t0 = lo + (long long)tile * kRowTile; thi = hi; sbase = 0;
        }
    };
    // ---- the register part of the image: B fragments of segments 0 .. RB - 1 (query 32 nt + r31), loaded first
    h8 breg[RB][J][NT];
    {
        const char* qb = (const char*)a.qimg + ((long long)(4 * h) * QN + r31) * 16;
#pragma unroll
        for (int sg = 0; sg < RB; ++sg)
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    breg[sg][j][nt] = *(const h8*)(qb + (long long)sg * SEGIMG + frag_off(j) + nt * (kQueryTile * 16));
    }
    // LDS carve-up: four rings of RING segments | the LDS half of the image (segments RB .. S - 1) | scratch | control block | stage
    char* ring = smem + (size_t)wid * (RING * kSegBytes);
    char* img = smem + kScan2Waves * RING * kSegBytes;
    constexpr size_t kImgBytes = (size_t)(S - RB) * SEGIMG;
    char* scratch = img + kImgBytes + (size_t)wid * kScratchBytes;
    char* ctl = img + kImgBytes + kScan2Waves * kScratchBytes;
    u32* next_tile = (u32*)(ctl + 4);
    const unsigned ring_l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(ring)), scratch_l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(scratch));
    const int drow = lane >> 3;
    // a tile's source: a scalar base (its first row) + per piece a 32-bit lane offset (row of the piece x row bytes + the swizzled 16-byte column)
    struct TileSrc { unsigned long long ua; unsigned v[4]; };
    auto src_of = [&](int tile) -> TileSrc {
        long long t0_, thi_, sb_;
        tile_rows(tile, t0_, thi_, sb_);
        TileSrc ts;
        // the base row is clamped INTO the part: a sample tile of a range shorter than its sample part starts past the range's end (t0_ >
        // thi_ - 1) -- past the corpus's for the last ranges (found by the fuzz as a memory fault: 21 845 rows in 2 048 ranges)
        const long long last_row = thi_ - 1 > 0 ? thi_ - 1 : 0;
        long long base_row = t0_ > 0 ? t0_ : 0;
        base_row = base_row < last_row ? base_row : last_row;
        ts.ua = (unsigned long long)(a.rows + base_row * a.row_bytes);
        ts.ua = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ts.ua >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ts.ua);   // (wave-uniform by construction)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = 8 * m + drow;
            long long r = t0_ + row;
            r = r < last_row ? r : last_row;                           // rows past the part's end re-read its last row (masked in the epilogue)
            r = r > base_row ? r : base_row;
            ts.v[m] = (unsigned)((r - base_row) * a.row_bytes) + ((((lane & 7) ^ ((row >> 1) & 7))) << 4);
        }
        return ts;
    };
    // first row of a tile for its epilogue operands (1 / norm of 32 rows from there; inv_scan is padded by 64 entries): clamped into the part like
    // the rows' base -- a sample tile past its range's end has no row the epilogue looks at
    auto first_row = [&](int tile) {
        long long t0_, thi_, sb_;
        tile_rows(tile, t0_, thi_, sb_);
        const long long last_row = thi_ - 1 > 0 ? thi_ - 1 : 0;
        return t0_ < last_row ? t0_ : last_row;
    };
    auto issue_seg = [&](const TileSrc& src, int seg, int buf) {
        dma16x4(src.ua + (unsigned long long)seg * 128ull, src.v[0], src.v[1], src.v[2], src.v[3], ring_l + buf * kSegBytes);
    };
    auto issue_epi = [&](long long t0, bool sync_tau, int par) {
        par = __builtin_amdgcn_readfirstlane(par);
        dma4(a.inv_scan + t0 + r31, scratch_l + par * 512 + 256);
        if (sync_tau) dma4(a.tau_bin + (lane < QN ? lane : QN - 1), scratch_l + par * 512);
    };
    int cur_tile = __builtin_amdgcn_readfirstlane(wid);                // (a scalar from here on: tile addresses stay in SGPRs)
    const bool active = cur_tile < ntiles;
    TileSrc src_cur{}, src_nxt{};
    if (active) {
        src_cur = src_of(cur_tile);
        issue_epi(first_row(cur_tile), MODE == kModeMain, 0);
#pragma unroll
        for (int sg = 0; sg < RING; ++sg) issue_seg(src_cur, sg, sg);   // the whole ring (S >= RING)
    }
    typedef unsigned int u4r __attribute__((ext_vector_type(4)));
    {   // the LDS part of the image, verbatim (it starts RB segments into the image); control block + candidate stage zeroed
        const char* srcq = (const char*)a.qimg + (long long)RB * SEGIMG;
        constexpr int nkb = (int)(kImgBytes >> 10);
        const unsigned img_l = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(img));
        const int wu = __builtin_amdgcn_readfirstlane(wid);
        for (int c = wu; c < nkb; c += kScan2Waves) dma16(srcq + ((long long)c << 10) + (lane << 4), img_l + ((unsigned)c << 10));
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), the BUILTIN: the register half, this wave's share of the LDS half and its ring have landed
        if constexpr (AR) {
#pragma unroll
            for (int sg = 0; sg < RB; ++sg)
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+a"(breg[sg][j][nt]));   // from here on this fragment lives in four accumulator registers
        }
        uint4* z = (uint4*)ctl;
        const int nz = kCtlBytes / 16 + (MODE == kModeMain ? a.stage_cap : 0);
        for (int i = tid; i < nz; i += THREADS) z[i] = make_uint4(0u, (i == 0) ? (u32)kScan2Waves : 0u, 0u, 0u);
    }
    __syncthreads();
    if (MODE == kModeMain) {
        if (tid < QN) ((int*)(ctl + 16))[tid] = a.tau_bin[tid];
        __syncthreads();
    }

    const char* lds_lane = img + ((4 * h) * QN + r31) * 16;            // + (sg - RB) * SEGIMG + frag_off(j) + nt * (32 * 16)
    const int asw = (r31 >> 1) & 7;
    const char* a_lane = ring + r31 * 128;
    const int dbg_rec = (a.debug & 512) ? 72 : 4;
    unsigned long long* dbg = (MODE == kModeMain && (a.debug & 128) && a.dbg && lane == 0) ? a.dbg + ((long long)blockIdx.x * kScan2Waves + wid) * dbg_rec : nullptr;
    if (dbg) dbg[0] = wall_clock64();
    if (dbg && dbg_rec > 4) {
        dbg[68] = t_entry;
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_id), "=s"(xcc_id));
        dbg[70] = ((unsigned long long)xcc_id << 32) | hw_id;
    }
#ifdef VF_EXPERIMENTS
    // cycle accounting of a wave's own timeline (debug bit 12; s_memtime + lgkmcnt(0) at each phase boundary, so the run is ~10 % slower than
    // the product's and the RATIOS are what to read): [4] waiting for a segment (vmcnt), [5] LDS reads + matrix instructions issued,
    // [6] the four LDS-DMA refills, [7] tile head (claim, addresses), [8] epilogue, [9] segments, in shader cycles -> dbg[4..9]
    const bool prof = (a.debug & 4096) != 0;
    unsigned long long pc_wait = 0, pc_use = 0, pc_fill = 0, pc_head = 0, pc_epi = 0, pc_n = 0, pt0 = 0, pt1 = 0;
    auto cyc = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; };
#define VF_PC(stmt) if (prof) { stmt; }
#else
#define VF_PC(stmt)
#endif
    int tiles_done = 0;
    if (active) {
        f16v acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;
        int buf = 0;
        while (true) {
            VF_PC(pt0 = cyc())
            long long t0, t_hi, t_s0;
            tile_rows(cur_tile, t0, t_hi, t_s0);
            int claimed;
            {
                int v_ = 0;
                if (lane == 0) v_ = (int)atomicAdd(next_tile, 1u);
                claimed = __builtin_amdgcn_readfirstlane(v_);
            }
            const bool more = claimed < ntiles;
            const int nxt = more ? claimed : cur_tile;
            src_nxt = src_of(nxt);
            const bool sync_now = MODE == kModeMain && (tiles_done & (kScan2Waves - 1)) == wid;
            const bool sync_next = MODE == kModeMain && ((tiles_done + 1) & (kScan2Waves - 1)) == wid;
            if (dbg && dbg_rec > 4 && tiles_done < 64) dbg[4 + tiles_done] = wall_clock64();
            VF_PC(pt1 = cyc(); pc_head += pt1 - pt0)
            // one body for both halves: BREG picks the B fragments from the registers (compile-time segment) or from LDS
            auto consume = [&](auto sg_c, const char* bb) {
                constexpr int sgc = decltype(sg_c)::value;        // >= 0: register segment; -1: LDS
                // segments sg + 1 .. sg + RING - 1 (4 (RING - 1) DMA instructions, and the <= 2 epilogue words issued among them) may still be in flight
                VF_PC(pt0 = cyc())
                if constexpr (RING == 6) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else if constexpr (RING == 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                static_assert(RING >= 4 && RING <= 6, "the counted waits above");
                VF_PC(pt1 = cyc(); pc_wait += pt1 - pt0; ++pc_n)
                const char* ab = a_lane + buf * kSegBytes;
                // B fragment j of query tile nt: from the accumulator registers, the allocator's registers (AR = 0) or LDS
                auto bfrag = [&](int j, int nt) -> h8 {
                    if constexpr (sgc >= 0) {
                        return breg[sgc < 0 ? 0 : sgc][j][nt];
                    } else {
                        return *(const h8*)(bb + frag_off(j) + nt * (kQueryTile * 16));
                    }
                };
                if constexpr (F8 == 0) {
                    if constexpr (AR) {
                        h8 af[4], bl[4][NT];
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[i] = *(const h8*)(ab + (((4 * h + i) ^ asw) << 4));
                        if constexpr (sgc < 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) bl[i][nt] = bfrag(i, nt);
                        }
                        __builtin_amdgcn_sched_barrier(0);                 // every LDS read of the segment is in flight before its first matrix instruction
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], sgc < 0 ? bl[i][nt] : bfrag(i, nt), acc[nt], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const h8 af = *(const h8*)(ab + (((4 * h + i) ^ asw) << 4));
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfrag(i, nt), acc[nt], 0, 0, 0);
                        }
                    }
                } else {
                    // e4m3 rows (k_scan2's F8 body): in chunk c lane (r, h) owns the two 16-byte pieces 4 c + 2 h, + 1 of its row
                    uint4 wv[4];
                    h8 bl[8][NT];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        wv[2 * c] = *(const uint4*)(ab + (((4 * c + 2 * h) ^ asw) << 4));
                        wv[2 * c + 1] = *(const uint4*)(ab + (((4 * c + 2 * h + 1) ^ asw) << 4));
                    }
                    if constexpr (sgc < 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) bl[j][nt] = bfrag(j, nt);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
#ifdef VF_EXPERIMENTS
                        if (c == 1 && (a.debug & 32)) break;          // timing experiment (INVALID RESULTS): half of the matrix instructions and conversions
#endif
                        const uint4 w0 = wv[2 * c], w1 = wv[2 * c + 1];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint4 w = i < 2 ? w0 : w1;
                            h8 af;
#ifdef VF_EXPERIMENTS
                            if (a.debug & 64) {                       // timing experiment (INVALID RESULTS): no conversions, the raw words as halves
                                const u4r raw = {(i & 1) ? w.z : w.x, (i & 1) ? w.w : w.y, (i & 1) ? w.z : w.x, (i & 1) ? w.w : w.y};
                                af = __builtin_bit_cast(h8, raw);
                            } else
#endif
                            af = (i & 1) ? cvt8_e4m3(w.z, w.w) : cvt8_e4m3(w.x, w.y);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, sgc < 0 ? bl[4 * c + i][nt] : bfrag(4 * c + i, nt), acc[nt], 0, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the refill overwrites the LDS these fragments were read from (k_scan2's note)
                VF_PC(pt0 = cyc(); pc_use += pt0 - pt1)
            };
            auto refill = [&](int sg) {
                const int s3 = sg + RING;
                if (s3 < S) issue_seg(src_cur, s3, buf);
                else {
                    if (s3 == S) issue_epi(first_row(nxt), sync_next, (tiles_done + 1) & 1);
                    issue_seg(src_nxt, s3 - S, buf);
                }
                buf = buf + 1 == RING ? 0 : buf + 1;
                VF_PC(pt1 = cyc(); pc_fill += pt1 - pt0)
            };
#define VF_R_SEG(N) if constexpr (RB > N) { consume(std::integral_constant<int, N>{}, nullptr); refill(N); }
            VF_R_SEG(0) VF_R_SEG(1) VF_R_SEG(2) VF_R_SEG(3) VF_R_SEG(4) VF_R_SEG(5)
#undef VF_R_SEG
            for (int sg = RB; sg < S; ++sg) {
                consume(std::integral_constant<int, -1>{}, lds_lane + (long long)(sg - RB) * SEGIMG);
                refill(sg);
            }
            VF_PC(pt0 = cyc())
            const char* sc = scratch + (tiles_done & 1) * 512;
            EpiRegs<NT> epi;
            epi.inv_lane = *(const float*)(sc + 256 + r31 * 4);
            epi.inv_lds = sc + 256;
            epi.sync_tau = sync_now;
            if (sync_now) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) epi.tau_g[nt] = *(const int*)(sc + (nt * kQueryTile + r31) * 4);
            }
            tile_epilogue<NT, MODE, true, true>(a, acc, epi, t0, t_hi, t_s0, lane, ctl);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.0f;
            VF_PC(pt1 = cyc(); pc_epi += pt1 - pt0)
            ++tiles_done;
            if (!more) break;
            cur_tile = nxt;
            src_cur = src_nxt;
        }
    }
    if (dbg && dbg_rec > 4) dbg[69] = (unsigned long long)tiles_done;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == kModeSample) return;                                   // no candidates in the sample pass: nothing to flush
    if (lane == 0) atomicAdd((u32*)(ctl + 8), 1u);
    if (dbg) dbg[1] = wall_clock64();
    __syncthreads();
    if (dbg) dbg[2] = wall_clock64();
    {   // flush the staged candidates (k_scan2's)
        const u32 staged = *(const u32*)ctl;
        const u32 nst = staged < (u32)a.stage_cap ? staged : (u32)a.stage_cap;
        uint4* ent = (uint4*)(ctl + kCtlBytes);
        u32* qcnt = (u32*)img;          // the LDS half of the image is dead now
        u32* qbase = qcnt + QN;
        if (tid < QN) qcnt[tid] = 0u;
        __syncthreads();
        for (u32 i = tid; i < nst; i += THREADS) {
            const uint4 e = ent[i];
            const u32 q = e.z & 0xFFu;
            if (e.w != 1u || q >= (u32)QN) continue;
            ent[i].w = 2u + atomicAdd(qcnt + q, 1u);
        }
        __syncthreads();
        if (tid < QN) {
            const u32 c = qcnt[tid];
            qbase[tid] = c ? atomicAdd(a.cnt + tid * kCntStride, c) : 0u;
        }
        __syncthreads();
        for (u32 i = tid; i < nst; i += THREADS) {
            const uint4 e = ent[i];
            if (e.w < 2u) continue;
            const u32 q = e.z & 0xFFu;
            const u32 gs = qbase[q] + (e.w - 2u);
            if (gs < (u32)a.cap) a.cand[(long long)q * a.cap + gs] = ((u64)e.y << 32) | (u64)e.x;
        }
    }
    if (dbg && dbg_rec > 4) dbg[71] = __builtin_amdgcn_s_memtime() - c_entry;
#ifdef VF_EXPERIMENTS
    if (dbg && dbg_rec > 4 && prof) { dbg[4] = pc_wait; dbg[5] = pc_use; dbg[6] = pc_fill; dbg[7] = pc_head; dbg[8] = pc_epi; dbg[9] = pc_n; }
#endif
#undef VF_PC
    if (dbg) dbg[3] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------
// k_scan_wide: the scan for LARGE query batches (nq > 128; BASELINE configs[4]: B = 1024, k = 1000).
//
// With more than ~100 queries per pass the contraction is MFMA-bound, not HBM-bound (2*nq/elt
// FLOP per byte), so the shard must be read ONCE per as many queries as the accumulators hold instead of once per
// 64.  A workgroup of FOUR waves -- one per SIMD, so that each wave may use the whole 512-entry register file --
// owns a 256-query tile jt and a row group rg; a wave owns 64 rows (two MFMA row tiles) of each 256-row super-tile
// and all 256 queries: 2 x 8 accumulator tiles of 32x32 (256 accumulator registers).  Every query fragment read
// from LDS feeds two MFMAs, which keeps the LDS array at a quarter of its bandwidth.  The k dimension is walked in
// chunks of 64 elements:
//   * corpus rows go HBM -> registers directly as in k_scan (each byte is used by exactly one wave; fp8 codes are
//     converted to fp16 in registers), two register stages, a stage ahead;
//   * the query chunk [8 k-groups][256 queries][8 halves] = 32 KB is shared by the waves: global (L2-resident
//     image) -> registers -> LDS, one chunk ahead, three LDS buffers so that ONE barrier per chunk suffices
//     (the buffer written in iteration c was last read in iteration c - 2).
// Epilogue, candidate stage, threshold refresh and histograms are k_scan's, over 256 queries; the stage is flushed
// to the global lists whenever it is half full (k = 1000 yields ~40 candidates per 32-row tile).
// Grid: block b -> jt = (b / 8) % J, rg = b % 8 + 8 * (b / (8 J)): the J workgroups that scan the same rows for
// different query tiles have equal b % 8 (one XCD under round-robin placement: their row reads share that L2;
// speed only) and are adjacent in dispatch order.
// ------------------------------------------------------------------------------------------------
constexpr int kWideQ = 256, kWideNT = kWideQ / kQueryTile, kWideKC = 64, kWideRows = 256;
#ifndef VF_WIDE_THREADS
#define VF_WIDE_THREADS 512
#endif
constexpr int kWideThreads = VF_WIDE_THREADS, kWideWaves = kWideThreads / 64, kWideM = kWideRows / kWideWaves / kRowTile;   // row tiles per wave: 1 (8 waves) or 2 (4 waves)
constexpr int kWideBuf = (kWideKC / 8) * kWideQ * 16;   // 32 KB per query chunk
constexpr int kWideCtl = 16 + 3 * kWideQ * 4;            // stage_cnt | tau_lds[256] | qcnt[256] | qbase[256]
constexpr int kSampWaves = kScanThreads / 64;            // sample rows per row group = samp * 8 (k_sel0's slot map)

// One register stage of corpus data = DC consecutive 64-element chunks of this lane's two rows: fp8 rows 2 chunks
// (2 x 32 B per row), fp16 rows 1 chunk (64 B per row): 32 VGPRs either way.  Chunk kc, MFMA step i, lane half h
// covers elements kc*64 + (4h + i)*8 .. +8 -- the k-group order of the query chunk in LDS.
template <int F8> struct WideStage { uint4 w[kWideM][4]; };
template <int F8> __device__ __forceinline__ constexpr int wide_dc() { return F8 ? 2 : 1; }

template <int F8>
__device__ __forceinline__ void wide_issue_a(WideStage<F8>& st, const char* rows, long long row_bytes, const long long (&row)[kWideM],
                                             int unit, int h) {
#pragma unroll
    for (int m = 0; m < kWideM; ++m) {
        if constexpr (F8 != 0) {
            const char* p = rows + row[m] * row_bytes + (long long)unit * 128 + h * 32;
            st.w[m][0] = *(const uint4*)p;
            st.w[m][1] = *(const uint4*)(p + 16);
            st.w[m][2] = *(const uint4*)(p + 64);
            st.w[m][3] = *(const uint4*)(p + 80);
        } else {
            const char* p = rows + row[m] * row_bytes + (long long)unit * 128 + h * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) st.w[m][i] = *(const uint4*)(p + i * 16);
        }
    }
}

// 32 x kWideM MFMAs: chunk `CC` (0 .. DC-1) of the stage against the query chunk at lds_lane
// 32 x kWideM MFMAs: chunk CC (0 .. DC-1) of the stage against the query chunk at lds_lane.
// The query fragments run through a 4-deep register ring that is carried ACROSS chunks: fragment s of this chunk sits in
// bq[s % 4] (32 fragments per chunk, so the ring position is the same at every chunk start); while pair s computes, the
// read of fragment s + 3 is issued -- for the last three pairs from the NEXT chunk's buffer (next_lane), which is
// already visible (it was written two chunks ago).  Pinned with sched_group_barrier: per pair [kWideM MFMAs][1 read].
template <int F8, int CC>
__device__ __forceinline__ void wide_compute(f16v (&acc)[kWideM][kWideNT], const WideStage<F8>& st, const char* lds_lane,
                                             const char* next_lane, h8 (&bq)[4]) {
    auto frag_at = [&](int sn) {
        const char* base = sn < 4 * kWideNT ? lds_lane : next_lane;
        const int s2 = sn < 4 * kWideNT ? sn : sn - 4 * kWideNT;
        return *(const h8*)(base + (s2 / kWideNT) * (kWideQ * 16) + (s2 % kWideNT) * (kQueryTile * 16));
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h8 afrag[kWideM];
#pragma unroll
        for (int m = 0; m < kWideM; ++m) {
            if constexpr (F8 != 0) {
                const uint4 w = st.w[m][2 * CC + (i >> 1)];
                afrag[m] = (i & 1) ? cvt8_e4m3(w.z, w.w) : cvt8_e4m3(w.x, w.y);
            } else {
                afrag[m] = __builtin_bit_cast(h8, st.w[m][i]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < kWideNT; ++nt) {
            const int sidx = i * kWideNT + nt;
#pragma unroll
            for (int m = 0; m < kWideM; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afrag[m], bq[sidx % 4], acc[m][nt], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, kWideM, 0);
            bq[(sidx + 3) % 4] = frag_at(sidx + 3);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
}

// WG-wide: move the staged candidates to the per-query global lists (see k_scan's final flush), publish the entries
// the block-wise publisher has not reached yet, and empty the stage for reuse.  Called by every thread.
template <int Q = kWideQ, int THREADS = kWideThreads>
__device__ __forceinline__ void wide_flush(const ScanArgs& a, char* ctl, int jt, int tid) {
    constexpr int kWideQ = Q, kWideThreads = THREADS, kWideCtl = 16 + 3 * Q * 4;   // (shadow the file-level constants: the body is unchanged)
    u32* stage_cnt = (u32*)ctl;
    u32* qcnt = (u32*)(ctl + 16 + kWideQ * 4);
    u32* qbase = qcnt + kWideQ;
    uint4* ent = (uint4*)(ctl + kWideCtl);
    __syncthreads();
    const u32 staged = *stage_cnt;
    const u32 nst = staged < (u32)a.stage_cap ? staged : (u32)a.stage_cap;
    if (tid < kWideQ) qcnt[tid] = 0u;
    __syncthreads();
    const u32 published = staged & ~((u32)a.refresh_every - 1u);   // complete blocks were published by whoever completed them
    for (u32 i = tid; i < nst; i += kWideThreads) {
        const uint4 e = ent[i];
        const u32 q = e.z & 0xFFu;
        if (e.w != 1u) continue;
        ent[i].w = 2u + atomicAdd(qcnt + q, 1u);
        if (i >= published) {
            const u32 bin = e.z >> 8, gq = (u32)jt * kWideQ + q;
            if (bin < (u32)kHistBins) {
                atomicAdd(a.hist + (long long)gq * kHistBins + bin, 1u);
                atomicAdd(a.hist_coarse + gq * 64 + (bin >> 5), 1u);
            }
        }
    }
    __syncthreads();
    if (tid < kWideQ) {
        const u32 c = qcnt[tid];
        qbase[tid] = c ? atomicAdd(a.cnt + ((long long)jt * kWideQ + tid) * kCntStride, c) : 0u;
    }
    __syncthreads();
    for (u32 i = tid; i < nst; i += kWideThreads) {
        const uint4 e = ent[i];
        if (e.w >= 2u) {
            const u32 q = e.z & 0xFFu;
            const u32 gs = qbase[q] + (e.w - 2u);
            if (gs < (u32)a.cap) a.cand[((long long)jt * kWideQ + q) * a.cap + gs] = ((u64)e.y << 32) | (u64)e.x;
        }
        ent[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0) *stage_cnt = 0u;
    __syncthreads();
}

// epilogue of ONE 32-row tile (rows t0 .. t0+31) against the 256 queries

template <int MODE, int NT = kWideNT, int Q = kWideQ>
__device__ __forceinline__ void wide_epilogue(const ScanArgs& a, const f16v (&acc)[NT], float inv_lane, long long t0,
                                              long long hi, long long s0_slot, int jt, int lane, char* ctl, bool sync_tau,
                                              int q0 = 0 /* first query (of the workgroup's Q) of this wave's NT tiles */,
                                              unsigned long long* ph = nullptr /* debug: [4] ticks in thresholds + pass 1a, 1b, 2, publish */) {
    constexpr int kWideQ = Q, kWideCtl = 16 + 3 * Q * 4;   // (shadow the file-level constants: the body is unchanged)
    const int r31 = lane & 31, h = lane >> 5;
    unsigned long long pc0 = ph ? wall_clock64() : 0ull;
    auto ph_mark = [&](int i) { if (ph) { const unsigned long long n_ = wall_clock64(); ph[i] += n_ - pc0; pc0 = n_; } };
    auto inv_of = [&](int reg) {
        const int r0 = (reg & 3) + 8 * (reg >> 2);
        const float lo_half = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv_lane), r0));
        const float hi_half = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv_lane), r0 + 4));
        return h ? hi_half : lo_half;
    };
    const int qg0 = jt * kWideQ;   // first global query of this workgroup's tile
    if (MODE == kModeSample) {
        const long long s0_stride = (long long)a.rgroups * a.samp * kSampWaves;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int q = qg0 + q0 + nt * kQueryTile + r31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (t0 + rr < hi) a.s0[(long long)q * s0_stride + s0_slot + rr] = acc[nt][reg] * inv_of(reg);
            }
        }
        return;
    }
    int* tau_lds = (int*)(ctl + 16);
    if (sync_tau) {   // one wave per super-tile folds the global thresholds into the workgroup's copy
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int tg = __hip_atomic_load(a.tau_bin + qg0 + q0 + nt * kQueryTile + r31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 32) atomicMax(tau_lds + q0 + nt * kQueryTile + lane, tg);
        }
    }
    u32* stage_cnt = (u32*)ctl;
    uint4* stage_ent = (uint4*)(ctl + kWideCtl);
    const u32 Rm1 = (u32)a.refresh_every - 1u;   // refresh_every is a power of two on this path
    bool need = false;
    u32 myslot = 0u;
    int myq = 0;
    // the thresholds are read ONCE per tile (other waves raise them concurrently; the masks and the claim below must agree)
    float tb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ql = q0 + nt * kQueryTile + r31;
        const int t = tau_lds[ql];
        tb[nt] = (qg0 + ql) < a.nq ? (t <= 0 ? -INFINITY : (float)t) : INFINITY;
#if defined(VF_WIDE_NOCAND)
        tb[nt] = INFINITY;   // timing experiments: the filter runs, nothing passes (results invalid)
#endif
    }
    // Pass 1a, branch-free over this lane's 16 NT scores: one bit per score -- did it pass? -- and nothing else.
    // Round 4 measured the earlier form (a compare + s_and_saveexec + branch per score, the value noted inside): 5.9 us per
    // 256-row super-tile and wave with NOTHING passing -- a dozen issue slots per score, mostly VALU -> SALU hazards
    // (profiles/r04_wide8_phases.log).  Now two instructions per score:
    //   e = acc * (-S / norm) + (tb - O - 2^-9)        one v_fma_f32  (bin_x(s) = S s + O, S a power of two)
    //   mask = (mask << 1) | sign(e)                    one v_alignbit_b32
    // e < 0  <=>  S acc / norm + O > tb - 2^-9: the scores the compare bin_x(acc / norm) >= tb passes, plus those within 2^-9 of a
    // bin below the edge (the two roundings of the old form move x by < 2^-12 bins): a superset, which is all the certificate asks
    // of the filter -- a candidate's bin is computed from its noted value as before.  Padded queries carry tb = +inf (e = +inf),
    // rows past the end a NaN inverse norm (e = that NaN, sign clear): neither passes.  Bit (nt & 1) * 16 + reg of word nt >> 1, as
    // pass 2 reads them: scores are taken in descending position order so that the first one shifted in ends up highest.
    // (v_pk_fma_f32 on register pairs would halve the fma count, but hipcc 7.2 miscompiles the sign extraction of the pair's SECOND
    //  element -- both shifts read element 0, also behind an opaque asm copy -- so the fma stays scalar: 2 instructions per score)
    u32 mk[4] = {0u, 0u, 0u, 0u};
    {
        float ninv[16];   // -S / norm of the row behind accumulator register reg (two readlanes and a select each: once per call)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ninv[reg] = inv_of(reg) * (-0.5f * kHistBins);
#pragma unroll
        for (int w = 0; w < NT / 2; ++w) {
            u32 m = 0u;
#pragma unroll
            for (int hn = 1; hn >= 0; --hn) {
                const float nc = tb[2 * w + hn] - 0.5f * kHistBins - 0.001953125f;
#pragma unroll
                for (int reg = 15; reg >= 0; --reg) {
                    const float e = __builtin_fmaf(acc[2 * w + hn][reg], ninv[reg], nc);
                    m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(u32, e), 31);
                }
            }
            mk[w] = m;
        }
    }
    const u32 cnt = (u32)(__popc(mk[0]) + __popc(mk[1]) + __popc(mk[2]) + __popc(mk[3]));
    ph_mark(0);
    if (__ballot(cnt != 0u) == 0ull) return;
    u32 anyw[4] = {0u, 0u, 0u, 0u};   // which positions passed in ANY lane (scalar)
#pragma unroll
    for (int w = 0; w < NT / 2; ++w) anyw[w] = wave_or_u32(mk[w]);
    // Pass 2: the passing scores become stage entries, ALL LANES AT ONCE: per query tile with a passing score somewhere (scalar test
    // on the wave-wide OR of the masks) every lane takes ITS lowest set bit, picks that accumulator register with a four-level
    // select tree (15 v_cndmask: the register differs from lane to lane, so it cannot be an indexed read) and the row's inverse norm
    // with one ds_bpermute; slots come from one LDS atomic per call, handed out by ballot rank.  A trip serves up to 64 candidates;
    // a second trip on the same tile only if some lane has two passing scores in it.  Cost per call ~45 instructions per query tile
    // whatever the candidate count -- the earlier loop over POSITIONS (value by indexed register read, one position per trip,
    // ~60 instructions and their scalar <-> vector hand-overs each) paid per candidate: 0.28 us, 1.7 of 16.5 ms at 10M rows and a
    // quarter of the launch at the 8-GPU shard, where candidates are eight times denser (profiles/r04_wide8_phases.log).
    // The candidate that gets the last slot of a block of R publishes that block and refreshes one threshold (below).
    {
        const u32 total = wave_sum_u32(cnt);
        u32 run = 0u;
        if (lane == 0) run = atomicAdd(stage_cnt, total);
        run = (u32)__builtin_amdgcn_readfirstlane((int)run);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if ((u32)__builtin_amdgcn_readfirstlane((int)((anyw[nt >> 1] >> ((nt & 1) * 16)) & 0xFFFFu)) == 0u) continue;
            u32 mine = (mk[nt >> 1] >> ((nt & 1) * 16)) & 0xFFFFu;
            const int ql = q0 + nt * kQueryTile + r31;
            for (;;) {
                const bool take = mine != 0u;
                const unsigned long long bal = __ballot(take);
                if (bal == 0ull) break;
                const int reg = take ? __builtin_ctz(mine) : 0;
                mine &= mine - 1u;
                // (bitwise selects -- v_bfi_b32 with an all-ones / all-zeros lane mask per bit of reg: written as ?: on the vector's
                //  elements hipcc turns the tree back into a dynamic extractelement and expands THAT as sixteen compare + select pairs, four times over)
                const u32 m0 = 0u - ((u32)reg & 1u), m1 = 0u - (((u32)reg >> 1) & 1u), m2 = 0u - (((u32)reg >> 2) & 1u), m3 = 0u - (((u32)reg >> 3) & 1u);
                u32 t8[8], t4[4], t2[2];
#pragma unroll
                for (int i = 0; i < 8; ++i) t8[i] = (m0 & __float_as_uint(acc[nt][2 * i + 1])) | (~m0 & __float_as_uint(acc[nt][2 * i]));
#pragma unroll
                for (int i = 0; i < 4; ++i) t4[i] = (m1 & t8[2 * i + 1]) | (~m1 & t8[2 * i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) t2[i] = (m2 & t4[2 * i + 1]) | (~m2 & t4[2 * i]);
                const float v = __uint_as_float((m3 & t2[1]) | (~m3 & t2[0]));
                const int r0 = (reg & 3) + 8 * (reg >> 2) + 4 * h;                  // the row of that register: lane r0 holds its inverse norm
                const float iv = __shfl(inv_lane, r0);
                if (take) {
                    const u32 slot = run + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                    const float sc = v * iv;
                    const u32 row = (u32)(t0 + r0);
                    const u32 bin = (u32)bin_of_x(bin_x(sc)), key = orderkey(sc);
                    // rows past the part's end are kept out by their NaN inverse norm (sign clear in pass 1a) -- but IEEE leaves the
                    // sign of a propagated NaN open, so the row bound is tested here as well: its claimed slot stays empty (w == 0:
                    // skipped by the flush and the publisher), nothing out of range can reach k_final's gather
                    if (t0 + r0 >= hi) {
                    } else if (slot < (u32)a.stage_cap) {
                        stage_ent[slot] = make_uint4(row, key, (u32)ql | (bin << 8), 1u);
                    } else {   // stage full (the first tiles after a loose seed, or hostile data): straight to the global list
                        const long long gq = qg0 + ql;
                        const u32 gs = atomicAdd(a.cnt + gq * kCntStride, 1u);
                        if (gs < (u32)a.cap) a.cand[gq * a.cap + gs] = ((u64)key << 32) | (u64)row;
                        atomicAdd(a.hist + gq * kHistBins + bin, 1u);
                        atomicAdd(a.hist_coarse + gq * 64 + (bin >> 5), 1u);
                    }
                    if ((slot & Rm1) == Rm1) { need = true; myslot = slot; myq = ql; }
                }
                run += (u32)__popcll(bal);
            }
        }
    }
    ph_mark(2);
    unsigned long long m = __ballot(need);
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const u32 blk0 = (u32)__shfl((int)myslot, leader) & ~Rm1;   // first entry of the completed block
        const int qq = qg0 + __shfl(myq, leader);
        const int nb = wave_tau_two_level(a.hist_coarse + qq * 64, a.hist + (long long)qq * kHistBins, a.kprime, lane);
        if (lane == 0 && nb > 0) {
            atomicMax(a.tau_bin + qq, nb);
            atomicMax(tau_lds + (qq - qg0), nb);
        }
        for (u32 j = (u32)lane; j <= Rm1; j += 64u) {
            const u32 idx = blk0 + j;
            if (idx >= (u32)a.stage_cap) break;
            const uint4 en = stage_ent[idx];
            if (en.w == 1u) {   // claimed-but-unwritten entries read w == 0: skipped, the histogram under-counts (safe)
                const u32 bin = en.z >> 8;
                const long long gq = qg0 + (en.z & 0xFFu);
                if (bin < (u32)kHistBins) {
                    atomicAdd(a.hist + gq * kHistBins + bin, 1u);
                    atomicAdd(a.hist_coarse + gq * 64 + (bin >> 5), 1u);
                }
            }
        }
    }
    ph_mark(3);
}

template <int MODE, int F8>
__global__ __launch_bounds__(kWideThreads) void k_scan_wide(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const int J = a.jtiles;
    const int jt = ((int)blockIdx.x >> 3) % J;
    const int rg = ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x / (8 * J));
    if (rg >= a.rgroups) return;   // grid is padded to a multiple of 8 J; whole workgroups leave before any barrier
    const long long Ra = a.n * rg / a.rgroups, Rb = a.n * (rg + 1) / a.rgroups;
    const long long swg = (long long)a.samp * kSampWaves;
    const long long Rs = (Ra + swg < Rb) ? Ra + swg : Rb;
    const long long lo = MODE == kModeSample ? Ra : Rs;
    const long long hi = MODE == kModeSample ? Rs : Rb;
    const int nst = (int)((hi - lo + kWideRows - 1) / kWideRows);
    const int NCH = a.dp >> 6;
    char* ctl = smem + 3 * kWideBuf;
    {
        uint4* z = (uint4*)ctl;
        const int nz = kWideCtl / 16 + (MODE == kModeMain ? a.stage_cap : 0);
        for (int i = tid; i < nz; i += kWideThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (MODE == kModeMain && tid < kWideQ) ((int*)(ctl + 16))[tid] = a.tau_bin[jt * kWideQ + tid];
    if (nst == 0) return;
    constexpr int DC = wide_dc<F8>();
    const int UN = NCH / DC;           // register stages per tile; even (launch requires dp % 256 == 0 for fp8, % 128 for fp16)
    WideStage<F8> A0, A1;
    // query-chunk copy: the 32 KB chunk [8 k-groups][256 q][8 halves] is 2048 units of 16 bytes; thread t moves units
    // t + kWideThreads v: k-group row (t >> 8) + (kWideThreads / 256) v, query t & 255
    // (named registers, not an array: an array here was demoted to scratch memory by the compiler)
    constexpr int GSTEP = kWideThreads / 256;      // k-group rows between a thread's consecutive units
    uint4 rb0, rb1, rb2, rb3;
#if VF_WIDE_THREADS == 256
    uint4 rb4, rb5, rb6, rb7;
#endif
    const long long qn8 = (long long)a.qn_total * 8;   // halves per k-group row of the global image
    const _Float16* bsrc = a.qimg + ((long long)(tid >> 8) * a.qn_total + (long long)jt * kWideQ + (tid & 255)) * 8;
#if VF_WIDE_THREADS == 256
#define VF_ISSUE_B(KC)                                                        \
    do {                                                                      \
        const _Float16* p_ = bsrc + (long long)(KC) * 8 * qn8;                \
        rb0 = *(const uint4*)(p_);           rb1 = *(const uint4*)(p_ + qn8);     \
        rb2 = *(const uint4*)(p_ + 2 * qn8); rb3 = *(const uint4*)(p_ + 3 * qn8); \
        rb4 = *(const uint4*)(p_ + 4 * qn8); rb5 = *(const uint4*)(p_ + 5 * qn8); \
        rb6 = *(const uint4*)(p_ + 6 * qn8); rb7 = *(const uint4*)(p_ + 7 * qn8); \
    } while (0)
#define VF_WRITE_B(BUF)                                                       \
    do {                                                                      \
        uint4* d_ = (uint4*)(smem + (BUF) * kWideBuf) + tid;                  \
        d_[0] = rb0; d_[256] = rb1; d_[512] = rb2; d_[768] = rb3;             \
        d_[1024] = rb4; d_[1280] = rb5; d_[1536] = rb6; d_[1792] = rb7;       \
    } while (0)
#else
#define VF_ISSUE_B(KC)                                                        \
    do {                                                                      \
        const _Float16* p_ = bsrc + (long long)(KC) * 8 * qn8;                \
        rb0 = *(const uint4*)(p_);           rb1 = *(const uint4*)(p_ + 2 * qn8); \
        rb2 = *(const uint4*)(p_ + 4 * qn8); rb3 = *(const uint4*)(p_ + 6 * qn8); \
    } while (0)
#define VF_WRITE_B(BUF)                                                       \
    do {                                                                      \
        uint4* d_ = (uint4*)(smem + (BUF) * kWideBuf) + tid;                  \
        d_[0] = rb0; d_[512] = rb1; d_[1024] = rb2; d_[1536] = rb3;           \
    } while (0)
#endif
    static_assert(GSTEP == 1 || GSTEP == 2, "256 or 512 threads");
    const u32 row0 = (u32)lo + (u32)(wid * (kWideM * kRowTile) + r31), hi32m1 = (u32)(hi - 1);
    auto rows_of = [&](int st, long long (&r)[kWideM]) {
#pragma unroll
        for (int m = 0; m < kWideM; ++m) {
            const u32 x = row0 + (u32)st * kWideRows + (u32)(m * kRowTile);
            r[m] = (long long)(x < hi32m1 ? x : hi32m1);
        }
    };
    const char* lds_lane0 = smem + ((4 * h) * kWideQ + r31) * 16;
    f16v acc[kWideM][kWideNT];
#pragma unroll
    for (int m = 0; m < kWideM; ++m)
#pragma unroll
        for (int nt = 0; nt < kWideNT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][nt][e] = 0.0f;
    // flat chunk counter state: buffer being read (br), chunk-in-tile of the chunk being computed (kc)
    int br = 0, kc = 0;
    h8 bq[4];
    // One chunk c: queries of chunk c + 2 -> LDS (the buffer read in chunk c - 1: every wave finished with it before the
    // last barrier), fetch chunk c + 3's into registers, 32 MFMAs on chunk c while the fragment ring runs on into
    // chunk c + 1's buffer (written during chunk c - 1, visible since the last barrier), ONE barrier.
#ifndef VF_WIDE_EXP
#define VF_WIDE_EXP 0   /* timing experiments (results invalid): bit 0 no query-chunk traffic, bit 1 no corpus loads, bit 2 no barrier, bit 3 no epilogue, bit 4 no sibling pacing / flush */
#endif
#define VF_CHUNK(STG, CC)                                                     \
    do {                                                                      \
        const int b1_ = br == 2 ? 0 : br + 1;          /* buffer of chunk c + 1 */ \
        const int b2_ = b1_ == 2 ? 0 : b1_ + 1;        /* buffer of chunk c + 2 (= the one chunk c - 1 used) */ \
        if (!(VF_WIDE_EXP & 1)) { VF_WRITE_B(b2_); }                          \
        int kn_ = kc + 3; kn_ = kn_ >= NCH ? kn_ - NCH : kn_; kn_ = kn_ >= NCH ? kn_ - NCH : kn_; \
        if (!(VF_WIDE_EXP & 1)) { VF_ISSUE_B(kn_); }                          \
        __builtin_amdgcn_sched_barrier(0);                                    \
        wide_compute<F8, CC>(acc, STG, lds_lane0 + br * kWideBuf, lds_lane0 + b1_ * kWideBuf, bq); \
        if (!(VF_WIDE_EXP & 4)) __syncthreads();                              \
        br = b1_;                                                             \
        kc = kc + 1 == NCH ? 0 : kc + 1;                                      \
    } while (0)
    long long myrow[kWideM], nxrow[kWideM];
    bool sib_on = J > 1;
    rows_of(0, myrow);
    wide_issue_a<F8>(A0, a.rows, a.row_bytes, myrow, 0, h);
    VF_ISSUE_B(0);
    VF_WRITE_B(0);
    VF_ISSUE_B(1);
    VF_WRITE_B(1);
    VF_ISSUE_B(2 < NCH ? 2 : 2 - NCH);
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) bq[s_] = *(const h8*)(lds_lane0 + (s_ / kWideNT) * (kWideQ * 16) + (s_ % kWideNT) * (kQueryTile * 16));
    bq[3] = bq[2];
    for (int st = 0; st < nst; ++st) {
        const long long t0 = lo + (long long)st * kWideRows + wid * (kWideM * kRowTile);
        float inv_lane[kWideM];
        for (int u = 0; u < UN; u += 2) {
            if (!(VF_WIDE_EXP & 2)) wide_issue_a<F8>(A1, a.rows, a.row_bytes, myrow, u + 1, h);
            __builtin_amdgcn_sched_barrier(0);
            VF_CHUNK(A0, 0);
            if constexpr (DC == 2) VF_CHUNK(A0, 1);
            if (VF_WIDE_EXP & 2) {
                if (u + 2 >= UN) {
#pragma unroll
                    for (int m = 0; m < kWideM; ++m) inv_lane[m] = 1.0f;
                    rows_of(st + 1 < nst ? st + 1 : st, nxrow);
                }
            } else if (u + 2 < UN) wide_issue_a<F8>(A0, a.rows, a.row_bytes, myrow, u + 2, h);
            else {
                // 1 / norm of this lane's rows; rows past the part's end get NaN: their scores compare false
#pragma unroll
                for (int m = 0; m < kWideM; ++m)
                    inv_lane[m] = (t0 + m * kRowTile + r31 < hi) ? a.inv_scan[t0 + m * kRowTile + r31] : __builtin_nanf("");
                rows_of(st + 1 < nst ? st + 1 : st, nxrow);
                wide_issue_a<F8>(A0, a.rows, a.row_bytes, nxrow, 0, h);
            }
            __builtin_amdgcn_sched_barrier(0);
            VF_CHUNK(A1, 0);
            if constexpr (DC == 2) VF_CHUNK(A1, 1);
        }
        // The epilogue's per-lane constants (row offsets, query ids, LDS addresses) must not be hoisted out of the tile
        // loop by LICM: kept live across the MFMA loop they push accumulators into scratch.  Opaque copies pin them here.
        int lane_e = lane, jt_e = jt;
        char* ctl_e = ctl;
        asm volatile("" : "+v"(lane_e), "+s"(jt_e));
#pragma unroll
        for (int m = 0; m < kWideM; ++m) {
            myrow[m] = nxrow[m];
            if (VF_WIDE_EXP & 8) {   // keep the accumulators alive at the price of 128 adds
                float sm = 0.f;
#pragma unroll
                for (int nt = 0; nt < kWideNT; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sm += acc[m][nt][e];
                if (sm == 12345.678f) a.s0[0] = sm;
            } else
            wide_epilogue<MODE>(a, acc[m], inv_lane[m], t0 + m * kRowTile, hi, (long long)rg * swg + (t0 + m * kRowTile - lo), jt_e,
                                lane_e, ctl_e,
                                m == 0 && (st & (kWideWaves - 1)) == wid);
#pragma unroll
            for (int nt = 0; nt < kWideNT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][nt][e] = 0.0f;
        }
        if (MODE == kModeMain && !(VF_WIDE_EXP & 16)) {
            // The J workgroups that scan the same rows (one XCD, dispatch permitting) read them once from HBM only while they
            // stay within an L2's reach of each other (8 row groups x 256 KB per super-tile against 4 MB): left alone they
            // drift and the shard was read 1.33 times.  Each publishes the super-tiles it has finished and waits (bounded --
            // this is a speed hint, never a correctness condition; a workgroup whose siblings are not resident stops waiting)
            // until the slowest sibling is at most sib_slack behind.
            if (a.sib && tid == 0 && sib_on) {
                u32* pr = a.sib + rg * 4;
                __hip_atomic_store(pr + jt, (u32)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 need = (u32)(st + 1) > (u32)a.sib_slack ? (u32)(st + 1) - (u32)a.sib_slack : 0u;
                int spins = 0;
                for (;;) {
                    u32 m = 0xffffffffu;
                    for (int j = 0; j < J; ++j) {
                        const u32 v = __hip_atomic_load(pr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        m = v < m ? v : m;
                    }
                    if (m >= need) break;
                    if (++spins > 512) { sib_on = false; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
            if (*(const u32*)ctl >= (u32)(a.stage_cap >> 1)) wide_flush(a, ctl, jt, tid);
        }
    }
    if (MODE == kModeMain) wide_flush(a, ctl, jt, tid);
#undef VF_CHUNK
#undef VF_WRITE_B
#undef VF_ISSUE_B
}

size_t scan_wide_lds_bytes(int stage_cap);
// ------------------------------------------------------------------------------------------------
// k_scan_wide8: the wide scan on the instruction BASELINE configs[4] names -- v_mfma_scale_f32_32x32x64_f8f6f4 -- for e4m3 rows.
//
// The row bytes ARE the A operand (no conversion); the query is split q = hi * 2^-8 + lo * 2^-12 + delta with hi, lo e4m3 codes
// (k_prep_wide8), the two powers of two ride in the instruction's block-scale operand, and ||delta||_2 -- known exactly per query --
// is that query's certificate bound (FinalArgs::eps_q) instead of the fp16 path's 2^-11.  Two MFMAs of 64 k per (row tile, query
// tile, K-tile): the matrix time of the fp16 form, without its 32 conversions per chunk and with a quarter of its instructions.
//
// What round 4 learned on the transformer products (k_gemm9_tn) shapes the rest: BOTH operands reach LDS by DMA
// (global_load_lds_dwordx4, no register hop, hand-counted waits, no vector-memory instruction the compiler knows about in the loop);
// a wave owns 64 rows x 128 queries (2 x 4 accumulator tiles, 128 registers), so that a fragment read feeds two or four MFMAs:
// 320 B of LDS reads per lane and K-tile against the old form's 512; a K-tile (64 elements) is 16 KB of rows + 2 x 16 KB of query
// codes, two stages of 48 KB: the same 96 KB k_scan_wide spends on its three query buffers, so control block, candidate stage and
// lane lists sit where they sat and the epilogue / flush / sibling pacing are k_scan_wide's own code.
//   LDS slot (16 B) of (row r, piece p of its 64 bytes) = 4 r + (p ^ ((r >> 2) & 3)): a ds_read_b128 phase (16 rows, one piece)
//   covers the 16 bank groups; DMA instruction I (1 KB) fills rows 16 I .. 16 I + 15, lane l fetching the piece its slot holds.
//   The query image is laid out in exactly this form per (query tile, K-tile) by k_prep_wide8, so its DMA is a linear copy.
// One barrier per K-tile: [my DMAs of tile t have landed] barrier [issue tile t + 1 into the other stage] 20 fragment reads, 16 MFMAs.
// ------------------------------------------------------------------------------------------------
#ifndef VF_W8_STAMPS
#define VF_W8_STAMPS 0
#endif
constexpr bool kW8Stamps = VF_W8_STAMPS != 0;
constexpr int kW8Stage = 48 * 1024, kW8NT = 4;

__device__ __forceinline__ void dma16s(unsigned long long ua, unsigned voff, unsigned lds_base) {   // scalar base + 32-bit lane offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(ua), "s"(lds_base) : "memory");
}

// The wave's DMA instructions of one K-tile in TWO asm statements instead of one per instruction (each of which saved and restored
// M0 and had its 64-bit base and LDS address computed by the compiler: ~12 instructions per DMA, 100 per K-tile in the 4-wave form).
// Rows: instruction i fetches the rows 16 i below instruction 0's -- the step goes into a scratch lane offset (an address register
// is read when the instruction issues: it may be overwritten right behind it); LDS destination + 1 KB each.
// (Default cache policy: this kernel reads a row tile once per 256-query super-tile, four times per launch, three of them out of L2 / the
//  Infinity Cache -- with `nt`, which the narrow scans' once-read rows gained 10 % from in round 6, configs[4] went from 17.0 to 18.8 ms:
//  profiles/r06_dma_nt_wide8_ab.log)
template <int N>
__device__ __forceinline__ void dma_rows(unsigned long long ua, unsigned voff0, unsigned lds_base, unsigned row_step) {
    unsigned keep, vt;
    if constexpr (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_add_u32 %1, %5, %2\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&v"(vt) : "v"(voff0), "s"(ua), "s"(lds_base), "s"(row_step) : "memory", "scc");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_add_u32 %1, %5, %2\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\tv_add_u32 %1, %5, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\tv_add_u32 %1, %5, %1\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&v"(vt) : "v"(voff0), "s"(ua), "s"(lds_base), "s"(row_step) : "memory", "scc");
}
// Query codes: 2 KB of hi codes, then the lo codes 16 KB further on in the image and in LDS.  The second instruction of each pair
// goes through the instruction's immediate offset, which the hardware adds to the global address AND to the LDS address
// (LDS address = M0 base + instruction offset + 16 x lane): + 1 KB on both sides, M0 untouched.
__device__ __forceinline__ void dma_codes(unsigned long long ub, unsigned voff, unsigned lds_base) {
    unsigned keep, vt;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_add_u32 %1, 0x4000, %2\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                 "s_add_u32 m0, m0, 0x4000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&v"(vt) : "v"(voff), "s"(ub), "s"(lds_base) : "memory");
}

// hi / lo e4m3 image of the normalised queries + each query's certificate bound.
//   img8[((jt * NK + kt) * 2 + part) * 16384 + (4 ql + ((b >> 4) ^ ((ql >> 2) & 3))) * 16 + (b & 15)]   (part 0 = hi, 1 = lo;
//   ql = query in its 256-tile jt, b = byte of K-tile kt): the LDS form of k_scan_wide8, copied verbatim.
// The encode is the hardware's (v_cvt_pk_fp8_f32); delta is computed from the DECODED codes, so the bound holds whatever the
// rounding did: |approx - q . c| <= ||delta||_2 for a unit row c (Cauchy-Schwarz).
__global__ __launch_bounds__(256) void k_prep_wide8(const float* qn, int nq, int d, int dp, int qtot, unsigned char* img8, float* eps_q) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    const int jt = slot >> 8, ql = slot & 255, NK = dp >> 6;
    __shared__ float red[256];
    float ss = 0.0f;
    for (int j = tid; j < dp; j += 256) {
        const float v = (slot < nq && j < d) ? qn[(long long)slot * d + j] : 0.0f;
        const int hc = __builtin_amdgcn_cvt_pk_fp8_f32(v * 256.0f, 0.0f, 0, false) & 0xFF;
        const float hv = __builtin_amdgcn_cvt_f32_fp8(hc, 0) * (1.0f / 256.0f);
        const float r = v - hv;
        const int lc = __builtin_amdgcn_cvt_pk_fp8_f32(r * 4096.0f, 0.0f, 0, false) & 0xFF;
        const float lv = __builtin_amdgcn_cvt_f32_fp8(lc, 0) * (1.0f / 4096.0f);
        const float dl = r - lv;
        ss = __builtin_fmaf(dl, dl, ss);
        const int kt = j >> 6, b = j & 63;
        const long long base = ((long long)(jt * NK + kt) * 2) * 16384 + (4 * ql + ((b >> 4) ^ ((ql >> 2) & 3))) * 16 + (b & 15);
        img8[base] = (unsigned char)hc;
        img8[base + 16384] = (unsigned char)lc;
    }
    red[tid] = ss;
    __syncthreads();
    for (int o = 128; o; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    // + the accumulation of 2 x dp products in the matrix unit and the canonical dot product's own rounding (generous: 8 d ulps of
    // a unit-scale sum), + the subnormal floor of the codes (covered by delta itself), + slack for this sum's own rounding
    if (tid == 0) eps_q[slot] = sqrtf(red[0]) * 1.002f + 8.0f * (float)dp * 5.9604645e-8f + 1e-6f;
}

hipError_t launch_prep_wide8(const float* qn, int nq, int d, int dp, int qtot, unsigned char* img8, float* eps_q, hipStream_t s) {
    hipLaunchKernelGGL(k_prep_wide8, dim3(qtot), dim3(256), 0, s, qn, nq, d, dp, qtot, img8, eps_q);
    return hipGetLastError();
}

// ---- the K-tile bodies of k_scan_wide8 in asm with FIXED registers (generated by tools/gen_w8_asm.py; do not edit by hand) --------
// Accumulators acc[m][nt] = v[64 m + 16 nt : +15]; five fragment slots of 16 registers S0 .. S4 = v[128 + 16 i : +15], each two
// 8-register operands (A: row tiles m = 0 | 1; B: hi codes | lo codes).  A tile reads A -> S1 (even tile) / S0 (odd), B0 -> S2,
// B1 -> S3, B2 -> the other A slot (the previous tile's A, dead once its held-back MFMAs have issued), B3 -> S4, and HOLDS BACK
// the four MFMAs of query tile 3: they are issued first thing in the NEXT tile's body, interleaved with that tile's first reads,
// so the matrix pipe has work while those reads are in flight (the fragment reads of tile t + 1 cannot start before the barrier
// that certifies its operands).  The C++ source could not express this: with 128 accumulators the allocator spilled 530 bytes per
// lane (reloads inside the MFMA sequence).  The compiler is told where everything lives through physical-register constraints.
//   E0: first tile of a super-tile (nothing held back before it; its MFMAs START the sums: C = 0, the accumulators are never cleared)
//   O1: the second tile (the held-back four of tile 0 start theirs)   EM / OM: even / odd tile in the middle   OL: last tile (odd; runs its own tile 3)
// Each body comes in two pieces: _A (the first held-back MFMAs and the first reads) is issued right behind the barrier, then the wave's
// six DMA instructions for the next tile (C++ between the two asm statements: ~50 instructions that now run under those MFMAs), then _B.
#define VF8_ASM_E0_A \
    "ds_read_b128 v[144:147], %[pa0] offset:0\n\t" \
    "ds_read_b128 v[148:151], %[pa1] offset:0\n\t" \
    "ds_read_b128 v[152:155], %[pa0] offset:2048\n\t" \
    "ds_read_b128 v[156:159], %[pa1] offset:2048\n\t" \
    "ds_read_b128 v[160:163], %[pb0] offset:0\n\t" \
    "ds_read_b128 v[164:167], %[pb1] offset:0\n\t" \
    "ds_read_b128 v[168:171], %[pb0] offset:16384\n\t" \
    "ds_read_b128 v[172:175], %[pb1] offset:16384\n\t"

#define VF8_ASM_E0_B \
    "ds_read_b128 v[176:179], %[pb0] offset:2048\n\t" \
    "ds_read_b128 v[180:183], %[pb1] offset:2048\n\t" \
    "ds_read_b128 v[184:187], %[pb0] offset:18432\n\t" \
    "ds_read_b128 v[188:191], %[pb1] offset:18432\n\t" \
    "ds_read_b128 v[128:131], %[pb0] offset:4096\n\t" \
    "ds_read_b128 v[132:135], %[pb1] offset:4096\n\t" \
    "ds_read_b128 v[136:139], %[pb0] offset:20480\n\t" \
    "ds_read_b128 v[140:143], %[pb1] offset:20480\n\t" \
    "ds_read_b128 v[192:195], %[pb0] offset:6144\n\t" \
    "ds_read_b128 v[196:199], %[pb1] offset:6144\n\t" \
    "ds_read_b128 v[200:203], %[pb0] offset:22528\n\t" \
    "ds_read_b128 v[204:207], %[pb1] offset:22528\n\t" \
    "s_waitcnt lgkmcnt(12)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[144:151], v[160:167], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[152:159], v[160:167], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[144:151], v[168:175], v[0:15], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[152:159], v[168:175], v[64:79], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(8)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[144:151], v[176:183], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[152:159], v[176:183], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[144:151], v[184:191], v[16:31], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[152:159], v[184:191], v[80:95], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[144:151], v[128:135], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[152:159], v[128:135], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[144:151], v[136:143], v[32:47], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[152:159], v[136:143], v[96:111], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"

#define VF8_ASM_O1_A \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[144:151], v[192:199], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[128:131], %[pa0] offset:0\n\t" \
    "ds_read_b128 v[132:135], %[pa1] offset:0\n\t" \
    "ds_read_b128 v[136:139], %[pa0] offset:2048\n\t" \
    "ds_read_b128 v[140:143], %[pa1] offset:2048\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[152:159], v[192:199], 0, %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[160:163], %[pb0] offset:0\n\t" \
    "ds_read_b128 v[164:167], %[pb1] offset:0\n\t" \
    "ds_read_b128 v[168:171], %[pb0] offset:16384\n\t" \
    "ds_read_b128 v[172:175], %[pb1] offset:16384\n\t"

#define VF8_ASM_O1_B \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[144:151], v[200:207], v[48:63], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[176:179], %[pb0] offset:2048\n\t" \
    "ds_read_b128 v[180:183], %[pb1] offset:2048\n\t" \
    "ds_read_b128 v[184:187], %[pb0] offset:18432\n\t" \
    "ds_read_b128 v[188:191], %[pb1] offset:18432\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[152:159], v[200:207], v[112:127], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[144:147], %[pb0] offset:4096\n\t" \
    "ds_read_b128 v[148:151], %[pb1] offset:4096\n\t" \
    "ds_read_b128 v[152:155], %[pb0] offset:20480\n\t" \
    "ds_read_b128 v[156:159], %[pb1] offset:20480\n\t" \
    "ds_read_b128 v[192:195], %[pb0] offset:6144\n\t" \
    "ds_read_b128 v[196:199], %[pb1] offset:6144\n\t" \
    "ds_read_b128 v[200:203], %[pb0] offset:22528\n\t" \
    "ds_read_b128 v[204:207], %[pb1] offset:22528\n\t" \
    "s_waitcnt lgkmcnt(12)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[128:135], v[160:167], v[0:15], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[136:143], v[160:167], v[64:79], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[128:135], v[168:175], v[0:15], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[136:143], v[168:175], v[64:79], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(8)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[128:135], v[176:183], v[16:31], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[136:143], v[176:183], v[80:95], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[128:135], v[184:191], v[16:31], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[136:143], v[184:191], v[80:95], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[128:135], v[144:151], v[32:47], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[136:143], v[144:151], v[96:111], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[128:135], v[152:159], v[32:47], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[136:143], v[152:159], v[96:111], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"

#define VF8_ASM_EM_A \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[128:135], v[192:199], v[48:63], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[144:147], %[pa0] offset:0\n\t" \
    "ds_read_b128 v[148:151], %[pa1] offset:0\n\t" \
    "ds_read_b128 v[152:155], %[pa0] offset:2048\n\t" \
    "ds_read_b128 v[156:159], %[pa1] offset:2048\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[136:143], v[192:199], v[112:127], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[160:163], %[pb0] offset:0\n\t" \
    "ds_read_b128 v[164:167], %[pb1] offset:0\n\t" \
    "ds_read_b128 v[168:171], %[pb0] offset:16384\n\t" \
    "ds_read_b128 v[172:175], %[pb1] offset:16384\n\t"

#define VF8_ASM_EM_B \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[128:135], v[200:207], v[48:63], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[176:179], %[pb0] offset:2048\n\t" \
    "ds_read_b128 v[180:183], %[pb1] offset:2048\n\t" \
    "ds_read_b128 v[184:187], %[pb0] offset:18432\n\t" \
    "ds_read_b128 v[188:191], %[pb1] offset:18432\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[136:143], v[200:207], v[112:127], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[128:131], %[pb0] offset:4096\n\t" \
    "ds_read_b128 v[132:135], %[pb1] offset:4096\n\t" \
    "ds_read_b128 v[136:139], %[pb0] offset:20480\n\t" \
    "ds_read_b128 v[140:143], %[pb1] offset:20480\n\t" \
    "ds_read_b128 v[192:195], %[pb0] offset:6144\n\t" \
    "ds_read_b128 v[196:199], %[pb1] offset:6144\n\t" \
    "ds_read_b128 v[200:203], %[pb0] offset:22528\n\t" \
    "ds_read_b128 v[204:207], %[pb1] offset:22528\n\t" \
    "s_waitcnt lgkmcnt(12)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[144:151], v[160:167], v[0:15], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[152:159], v[160:167], v[64:79], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[144:151], v[168:175], v[0:15], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[152:159], v[168:175], v[64:79], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(8)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[144:151], v[176:183], v[16:31], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[152:159], v[176:183], v[80:95], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[144:151], v[184:191], v[16:31], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[152:159], v[184:191], v[80:95], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[144:151], v[128:135], v[32:47], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[152:159], v[128:135], v[96:111], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[144:151], v[136:143], v[32:47], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[152:159], v[136:143], v[96:111], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"

#define VF8_ASM_OM_A \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[144:151], v[192:199], v[48:63], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[128:131], %[pa0] offset:0\n\t" \
    "ds_read_b128 v[132:135], %[pa1] offset:0\n\t" \
    "ds_read_b128 v[136:139], %[pa0] offset:2048\n\t" \
    "ds_read_b128 v[140:143], %[pa1] offset:2048\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[152:159], v[192:199], v[112:127], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[160:163], %[pb0] offset:0\n\t" \
    "ds_read_b128 v[164:167], %[pb1] offset:0\n\t" \
    "ds_read_b128 v[168:171], %[pb0] offset:16384\n\t" \
    "ds_read_b128 v[172:175], %[pb1] offset:16384\n\t"

#define VF8_ASM_OM_B \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[144:151], v[200:207], v[48:63], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[176:179], %[pb0] offset:2048\n\t" \
    "ds_read_b128 v[180:183], %[pb1] offset:2048\n\t" \
    "ds_read_b128 v[184:187], %[pb0] offset:18432\n\t" \
    "ds_read_b128 v[188:191], %[pb1] offset:18432\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[152:159], v[200:207], v[112:127], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[144:147], %[pb0] offset:4096\n\t" \
    "ds_read_b128 v[148:151], %[pb1] offset:4096\n\t" \
    "ds_read_b128 v[152:155], %[pb0] offset:20480\n\t" \
    "ds_read_b128 v[156:159], %[pb1] offset:20480\n\t" \
    "ds_read_b128 v[192:195], %[pb0] offset:6144\n\t" \
    "ds_read_b128 v[196:199], %[pb1] offset:6144\n\t" \
    "ds_read_b128 v[200:203], %[pb0] offset:22528\n\t" \
    "ds_read_b128 v[204:207], %[pb1] offset:22528\n\t" \
    "s_waitcnt lgkmcnt(12)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[128:135], v[160:167], v[0:15], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[136:143], v[160:167], v[64:79], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[128:135], v[168:175], v[0:15], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[136:143], v[168:175], v[64:79], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(8)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[128:135], v[176:183], v[16:31], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[136:143], v[176:183], v[80:95], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[128:135], v[184:191], v[16:31], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[136:143], v[184:191], v[80:95], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[128:135], v[144:151], v[32:47], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[136:143], v[144:151], v[96:111], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[128:135], v[152:159], v[32:47], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[136:143], v[152:159], v[96:111], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"

#define VF8_ASM_OL_A \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[144:151], v[192:199], v[48:63], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[128:131], %[pa0] offset:0\n\t" \
    "ds_read_b128 v[132:135], %[pa1] offset:0\n\t" \
    "ds_read_b128 v[136:139], %[pa0] offset:2048\n\t" \
    "ds_read_b128 v[140:143], %[pa1] offset:2048\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[152:159], v[192:199], v[112:127], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[160:163], %[pb0] offset:0\n\t" \
    "ds_read_b128 v[164:167], %[pb1] offset:0\n\t" \
    "ds_read_b128 v[168:171], %[pb0] offset:16384\n\t" \
    "ds_read_b128 v[172:175], %[pb1] offset:16384\n\t"

#define VF8_ASM_OL_B \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[144:151], v[200:207], v[48:63], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[176:179], %[pb0] offset:2048\n\t" \
    "ds_read_b128 v[180:183], %[pb1] offset:2048\n\t" \
    "ds_read_b128 v[184:187], %[pb0] offset:18432\n\t" \
    "ds_read_b128 v[188:191], %[pb1] offset:18432\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[152:159], v[200:207], v[112:127], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "ds_read_b128 v[144:147], %[pb0] offset:4096\n\t" \
    "ds_read_b128 v[148:151], %[pb1] offset:4096\n\t" \
    "ds_read_b128 v[152:155], %[pb0] offset:20480\n\t" \
    "ds_read_b128 v[156:159], %[pb1] offset:20480\n\t" \
    "ds_read_b128 v[192:195], %[pb0] offset:6144\n\t" \
    "ds_read_b128 v[196:199], %[pb1] offset:6144\n\t" \
    "ds_read_b128 v[200:203], %[pb0] offset:22528\n\t" \
    "ds_read_b128 v[204:207], %[pb1] offset:22528\n\t" \
    "s_waitcnt lgkmcnt(12)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[128:135], v[160:167], v[0:15], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[136:143], v[160:167], v[64:79], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[0:15], v[128:135], v[168:175], v[0:15], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[64:79], v[136:143], v[168:175], v[64:79], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(8)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[128:135], v[176:183], v[16:31], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[136:143], v[176:183], v[80:95], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[16:31], v[128:135], v[184:191], v[16:31], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[80:95], v[136:143], v[184:191], v[80:95], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[128:135], v[144:151], v[32:47], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[136:143], v[144:151], v[96:111], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[32:47], v[128:135], v[152:159], v[32:47], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[96:111], v[136:143], v[152:159], v[96:111], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[128:135], v[192:199], v[48:63], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[136:143], v[192:199], v[112:127], %[sa], %[sh] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[48:63], v[128:135], v[200:207], v[48:63], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "v_mfma_scale_f32_32x32x64_f8f6f4 v[112:127], v[136:143], v[200:207], v[112:127], %[sa], %[sl] op_sel_hi:[0,0,0]\n\t" \
    "s_nop 15\n\t" \
    "s_nop 7\n\t"

// WAVES = 8: one workgroup of 512 threads per CU, tile 256 rows x 256 queries (wave (wr, wc) = 64 rows x 128 queries), two stages of
//   48 KB [rows 16 K | hi codes 16 K | lo codes 16 K]; a.jtiles = 256-query tiles.
// WAVES = 4 (round 5): TWO workgroups of 256 threads per CU, tile 256 rows x 128 queries (wave wr = 64 rows x the 128 queries: the same
//   wave tile, the same asm bodies), a.jtiles counts 128-query tiles.  The two workgroups of a CU are independent barrier domains, so
//   the hardware de-phases them: one's epilogue (VALU) and K-tile boundaries (barrier, DMA wait, first fragment reads) run under the
//   other's matrix instructions -- with ONE 8-wave workgroup all eight waves meet at every boundary and reach the epilogue together
//   (3.4 + ~3 ms of a 16.5-ms launch at 10M rows with the matrix pipes idle).  Two stages of 32 KB, laid out so
//   that the asm bodies' "lo codes = hi codes + 16384" still holds:
//     [rows s0 16 K | hi s0 8 K | hi s1 8 K | lo s0 8 K | lo s1 8 K | rows s1 16 K] = 64 KB, + control block + candidate stage <= 80 KB.
//   Price: a row's bytes reach LDS once per 128 instead of once per 256 queries (L2 -> LDS traffic per product + 33 %).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void k_scan_wide8(ScanArgs a) {
    static_assert(WAVES == 8 || WAVES == 4, "k_scan_wide8: 8 waves (256 x 256 tile) or 4 waves (256 x 128 tile, two workgroups per CU)");
    constexpr int kW8Threads = 64 * WAVES, Q = WAVES == 8 ? 256 : 128, kCtl = 16 + 3 * Q * 4, kDmaRows = 256 / WAVES / 16;
    // LDS byte offsets of the two stages' row / hi-code areas (the lo codes sit 16384 above the hi codes in both layouts)
    constexpr unsigned kRowsAt0 = 0u, kRowsAt1 = WAVES == 8 ? (unsigned)kW8Stage : 49152u;
    constexpr unsigned kHiAt0 = 16384u, kHiAt1 = WAVES == 8 ? (unsigned)kW8Stage + 16384u : 24576u;
    constexpr int kOperandBytes = WAVES == 8 ? 2 * kW8Stage : 65536;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const int wr = WAVES == 8 ? wid >> 1 : wid, wc = WAVES == 8 ? wid & 1 : 0;   // rows 64 wr .. + 63 of a super-tile, queries 128 wc .. + 127 of the tile
    const int J = a.jtiles;
    const int jt = ((int)blockIdx.x >> 3) % J;
    const int rg = ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x / (8 * J));
    if (rg >= a.rgroups) return;
    const long long Ra = a.n * rg / a.rgroups, Rb = a.n * (rg + 1) / a.rgroups;
    const long long swg = (long long)a.samp * kSampWaves;
    const long long lo = (Ra + swg < Rb) ? Ra + swg : Rb, hi = Rb;
    const int nst = (int)((hi - lo + kWideRows - 1) / kWideRows);
    const int NK = a.dp >> 6;
    char* ctl = smem + kOperandBytes;
    {
        uint4* z = (uint4*)ctl;
        const int nz = kCtl / 16 + a.stage_cap;
        for (int i = tid; i < nz; i += kW8Threads) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (tid < Q) ((int*)(ctl + 16))[tid] = a.tau_bin[jt * Q + tid];
    if (nst == 0) return;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(smem));
    // ---- DMA duties of this wave per K-tile: rows (256 / WAVES) wid .. (kDmaRows instructions of 16 rows), hi and lo codes of queries
    //      32 wid .. + 31 (two instructions each).  The query image holds 256-query tiles (k_prep_wide8): a 128-query tile is one half
    //      of one -- a query's 64 bytes stay together under the swizzle, so the half is a contiguous 8 KB of each part.
    // Register budget: the asm bodies own v[0:207]; what the compiler keeps across the K loop must fit the other 48.  So ONE lane
    // offset serves all of the wave's row instructions (instruction i fetches the rows 16 i further down: + 16 i rows on the SCALAR
    // base) -- except in a super-tile that crosses the part's end, where every row is clamped on its own (recomputed per K-tile in
    // that one super-tile) -- and the fragment addresses below exist once and are toggled between the stages.
    const unsigned apiece = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);   // (row >> 2) & 3 of row 16 x + (lane >> 2)
    const unsigned long long rows_base = (unsigned long long)(a.rows + lo * a.row_bytes);
    const unsigned long long bimg = (unsigned long long)a.qimg + (WAVES == 8 ? (unsigned long long)jt * NK * 32768ull
                                                                           : (unsigned long long)(jt >> 1) * NK * 32768ull + (unsigned long long)(jt & 1) * 8192ull);
    const unsigned voffB = (unsigned)(2 * wid) * 1024u + (unsigned)lane * 16u;
    const long long span = hi - lo;
    const unsigned row_bytes32 = (unsigned)a.row_bytes;             // (a row group's bytes fit a 32-bit lane offset: checked by the host)
    unsigned voffA0 = 0u, row_first = 0u;
    bool tail_tile = false;
    auto set_rows = [&](int st) {
        row_first = (unsigned)st * kWideRows + (unsigned)(16 * kDmaRows * wid);
        tail_tile = (long long)st * kWideRows + kWideRows > span;
        const unsigned r = row_first + (unsigned)(lane >> 2), last = (unsigned)(span - 1);
        voffA0 = (r < last ? r : last) * row_bytes32 + apiece;      // rows past the part's end re-read its last row (NaN inverse norm below)
    };
    auto issue = [&](int kt, int stage) {
        const unsigned sa = lds0 + (stage ? kRowsAt1 : kRowsAt0) + (unsigned)(kDmaRows * wid) * 1024u;
        const unsigned sh = lds0 + (stage ? kHiAt1 : kHiAt0) + (unsigned)(2 * wid) * 1024u;
        const unsigned long long ua = rows_base + (unsigned long long)kt * 64ull;
        const unsigned long long ub = bimg + (unsigned long long)kt * 32768ull;
        if (!tail_tile) {
            dma_rows<kDmaRows>(ua, voffA0, sa, 16u * row_bytes32);
        } else {
            const unsigned last = (unsigned)(span - 1);
#pragma unroll
            for (int i = 0; i < kDmaRows; ++i) {
                const unsigned r = row_first + (unsigned)(16 * i) + (unsigned)(lane >> 2);
                dma16s(ua, (r < last ? r : last) * row_bytes32 + apiece, sa + 1024u * i);
            }
        }
        dma_codes(ub, voffB, sh);
    };
    // ---- fragment addresses of this lane (relative to a stage): row tiles m = 0, 1; query tiles nt = 0 .. 3; pieces 2 h, 2 h + 1
    unsigned fa[2][2], fb[kW8NT][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int row = wr * 64 + m * 32 + r31;
#pragma unroll
        for (int j = 0; j < 2; ++j) fa[m][j] = (unsigned)((4 * row + ((2 * h + j) ^ ((row >> 2) & 3))) << 4);
    }
#pragma unroll
    for (int nt = 0; nt < kW8NT; ++nt) {
        const int ql = wc * 128 + nt * 32 + r31;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[nt][j] = (unsigned)((4 * ql + ((2 * h + j) ^ ((ql >> 2) & 3))) << 4);   // relative to the stage's hi-code area
    }
    f16v acc[2][kW8NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nt = 0; nt < kW8NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][nt][e] = 0.0f;
    bool sib_on = J > 1;
    // debug bit 7: per wave [0] kernel time, [1] time waiting for its own DMAs at the K-tile boundary, [2] time in the K-tile barrier,
    // [3] time in the epilogues (filter, candidates, sibling pacing, flush) -- ticks of the 100 MHz counter
    // (compiled in by -DVF_W8_STAMPS=1 only -- tools/build_variant.sh stamps "-DVF_W8_STAMPS=1": the ten 64-bit counters cost the
    //  shipped kernel 20 registers it does not have)
    unsigned long long* dbg = (kW8Stamps && (a.debug & 128) && a.dbg && lane == 0) ? a.dbg + ((long long)blockIdx.x * WAVES + wid) * 16 : nullptr;
    unsigned long long t_wait = 0, t_bar = 0, t_epi = 0, t_inv = 0, t_flt = 0, t_ph[4] = {0, 0, 0, 0};
    const unsigned long long t_begin = dbg ? wall_clock64() : 0ull;
    set_rows(0);
    issue(0, 0);
    int stage = 0;                       // stage of the tile about to be computed
    float inv_lane[2] = {0.0f, 0.0f};
    // "Top" of tile (st_, kt_), whose operands went on their way one tile ago: this wave's DMAs have landed; barrier -- everyone's
    // have, and every wave is done reading the other stage; the tile AFTER it goes on its way into that stage.
    // The top of a super-tile's FIRST tile is taken BEFORE the previous super-tile's epilogue (below): the waves meet while they are
    // still in step, and a wave that has little to note in its epilogue runs ahead into the next super-tile's first K-tile while its
    // partner on the SIMD is still filtering -- matrix and vector work overlap instead of all eight waves waiting for the slowest
    // epilogue (round 4 stamps, profiles/r04_wide8_phases.log: 3.8 of 20 ms in that barrier).
    auto top = [&](int st_, int kt_) {
        unsigned long long c0 = 0, c1 = 0;
        if (dbg) c0 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dbg) c1 = wall_clock64();
        __syncthreads();
        if (dbg) { const unsigned long long c2 = wall_clock64(); t_wait += c1 - c0; t_bar += c2 - c1; }
    };
    auto issue_after = [&](int st_, int kt_) {   // the tile after (st_, kt_) goes on its way into the other stage
        if (kt_ + 1 < NK) issue(kt_ + 1, stage ^ 1);
        else if (st_ + 1 < nst) { set_rows(st_ + 1); issue(0, stage ^ 1); }
    };
    // fragment addresses per stage (tile kt lives in stage kt & 1: NK is even) and the three block scales (E8M0: 2^(x - 127)):
    // rows as they are, hi codes x 2^-8, lo codes x 2^-12
    // (one copy, stepped to the other stage behind every tile -- VF8_STEP; the opaque asm keeps the compiler from recognising the two
    //  values of each and holding both across the loop, which is what the array form [2] amounted to: 8 registers instead of 4)
    unsigned pa0 = lds0 + kRowsAt0 + fa[0][0], pa1 = lds0 + kRowsAt0 + fa[0][1], pb0 = lds0 + kHiAt0 + fb[0][0], pb1 = lds0 + kHiAt0 + fb[0][1];
    constexpr unsigned kStepA = kRowsAt1 - kRowsAt0, kStepB = kHiAt1 - kHiAt0;
    const int sc_a = 127, sc_h = 119, sc_l = 115;
    i8v s0a, s0b, s1a, s1b, s2a, s2b, s3a, s3b, s4a, s4b;
#define VF8_ACC_OPS "+{v[0:15]}"(acc[0][0]), "+{v[16:31]}"(acc[0][1]), "+{v[32:47]}"(acc[0][2]), "+{v[48:63]}"(acc[0][3]), \
                    "+{v[64:79]}"(acc[1][0]), "+{v[80:95]}"(acc[1][1]), "+{v[96:111]}"(acc[1][2]), "+{v[112:127]}"(acc[1][3])
#define VF8_IN_OPS(SG) [pa0] "v"(pa0), [pa1] "v"(pa1), [pb0] "v"(pb0), [pb1] "v"(pb1), [sa] "v"(sc_a), [sh] "v"(sc_h), [sl] "v"(sc_l)
#define VF8_STEP(SG)                                                                                                                  \
    do {                                                                                                                              \
        if (SG) { pa0 -= kStepA; pa1 -= kStepA; pb0 -= kStepB; pb1 -= kStepB; }                                                       \
        else { pa0 += kStepA; pa1 += kStepA; pb0 += kStepB; pb1 += kStepB; }                                                          \
        asm volatile("" : "+v"(pa0), "+v"(pa1), "+v"(pb0), "+v"(pb1));                                                                \
    } while (0)
#define VF8_SLOTS_OUT "=&{v[128:135]}"(s0a), "=&{v[136:143]}"(s0b), "=&{v[144:151]}"(s1a), "=&{v[152:159]}"(s1b), "=&{v[160:167]}"(s2a),      \
                      "=&{v[168:175]}"(s2b), "=&{v[176:183]}"(s3a), "=&{v[184:191]}"(s3b), "=&{v[192:199]}"(s4a), "=&{v[200:207]}"(s4b)
#define VF8_SLOTS_IO "+{v[128:135]}"(s0a), "+{v[136:143]}"(s0b), "+{v[144:151]}"(s1a), "+{v[152:159]}"(s1b), "+{v[160:167]}"(s2a),            \
                     "+{v[168:175]}"(s2b), "+{v[176:183]}"(s3a), "+{v[184:191]}"(s3b), "+{v[192:199]}"(s4a), "+{v[200:207]}"(s4b)
    // tile 0 of a super-tile: its top (and the issue of tile 1) was taken before the previous epilogue
#define VF8_TILE_FIRST()                                                                                                              \
    do {                                                                                                                              \
        asm volatile(VF8_ASM_E0_A : VF8_ACC_OPS, VF8_SLOTS_OUT : VF8_IN_OPS(0) : "memory");                                           \
        asm volatile(VF8_ASM_E0_B : VF8_ACC_OPS, VF8_SLOTS_IO : VF8_IN_OPS(0) : "memory");                                            \
        VF8_STEP(0);                                                                                                                  \
        stage ^= 1;                                                                                                                   \
    } while (0)
#define VF8_TILE_NEXT(NAME, SG, KT)                                                                                                   \
    do {                                                                                                                              \
        top(st, (KT));                                                                                                                \
        asm volatile(NAME##_A : VF8_ACC_OPS, VF8_SLOTS_IO : VF8_IN_OPS(SG) : "memory");                                               \
        issue_after(st, (KT));                                                                                                        \
        asm volatile(NAME##_B : VF8_ACC_OPS, VF8_SLOTS_IO : VF8_IN_OPS(SG) : "memory");                                               \
        VF8_STEP(SG);                                                                                                                 \
        stage ^= 1;                                                                                                                   \
    } while (0)
    top(0, 0);
    issue_after(0, 0);
    for (int st = 0; st < nst; ++st) {
        const long long t0 = lo + (long long)st * kWideRows + wr * 64;
        VF8_TILE_FIRST();
        VF8_TILE_NEXT(VF8_ASM_O1, 1, 1);
        for (int kt = 2; kt < NK - 2; kt += 2) {
            VF8_TILE_NEXT(VF8_ASM_EM, 0, kt);
            VF8_TILE_NEXT(VF8_ASM_OM, 1, kt + 1);
        }
        VF8_TILE_NEXT(VF8_ASM_EM, 0, NK - 2);
        {   // 1 / norm of this lane's rows, a K-tile ahead of its use; rows past the part's end get NaN: their scores never pass
            // (scalar base + 32-bit lane index: as 64-bit per-lane row numbers the compiler kept (long long)r31 in a register pair across
            //  the K loop -- in scratch, that is, reloaded here behind the next tile's DMAs)
            const long long rem = hi - t0;
            const int remi = rem > 2 * kRowTile ? 2 * kRowTile : (int)rem;
            const float* ivp = a.inv_scan + t0;
#pragma unroll
            for (int m = 0; m < 2; ++m) inv_lane[m] = (m * kRowTile + r31 < remi) ? ivp[m * kRowTile + r31] : __builtin_nanf("");
        }
        VF8_TILE_NEXT(VF8_ASM_OL, 1, NK - 1);
        if (st + 1 < nst) {
            // The candidate stage is flushed when half full.  The decision must be ONE value for the whole workgroup (wide_flush has
            // barriers inside): thread 0 copies the count into a spare control word BEFORE the barrier -- every epilogue of super-tile
            // st - 1 ended before the barrier of this super-tile's second K-tile and none of super-tile st starts before the barrier
            // below, so the count is at rest -- and everyone tests the copy behind it.  (Reading stage_cnt itself behind the barrier
            // raced with the epilogue of a wave that had run ahead: two waves could see values on either side of the limit.)
            if (tid == 0) ((volatile u32*)ctl)[1] = *(const volatile u32*)ctl;
            top(st + 1, 0);
            if (((const volatile u32*)ctl)[1] >= (u32)(a.stage_cap >> 1)) {
                // (opaque copies: with the plain ids LICM computes the flush's per-thread addresses -- qcnt + tid, qbase + tid, the
                //  stage entry, the global counter of query tid -- once in front of the super-tile loop, and with 48 registers to
                //  its name the allocator parks them in scratch: 7 registers, reloaded in every epilogue BEHIND the next tile's
                //  DMAs, i.e. with a vmcnt(0) that waits for them.  The flush runs once in a dozen super-tiles: it can add.)
                int tid_f = tid, jt_f = jt;
                asm volatile("" : "+v"(tid_f), "+s"(jt_f));
                wide_flush<Q, kW8Threads>(a, ctl, jt_f, tid_f);
            }
            issue_after(st + 1, 0);
        }
        const unsigned long long e0 = dbg ? wall_clock64() : 0ull;
        int lane_e = lane, jt_e = jt;
        char* ctl_e = ctl;
        asm volatile("" : "+v"(lane_e), "+s"(jt_e));
        unsigned long long e1 = 0;
        if (dbg) { asm volatile("" :: "v"(inv_lane[0]), "v"(inv_lane[1])); e1 = wall_clock64(); t_inv += e1 - e0; }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            wide_epilogue<kModeMain, kW8NT, Q>(a, acc[m], inv_lane[m], t0 + m * kRowTile, hi, 0, jt_e, lane_e, ctl_e,
                                            m == 0 && (st & 3) == wr, wc * 128, dbg ? t_ph : nullptr);
        }
        if (dbg) t_flt += wall_clock64() - e1;
        if (a.sib && tid == 0 && sib_on) {   // sibling pacing: k_scan_wide's (a speed hint, bounded)
            u32* pr = a.sib + rg * 8;   // (eight progress words per row group: up to eight 128-query siblings)
            __hip_atomic_store(pr + jt, (u32)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 need = (u32)(st + 1) > (u32)a.sib_slack ? (u32)(st + 1) - (u32)a.sib_slack : 0u;
            int spins = 0;
            for (;;) {
                u32 mn = 0xffffffffu;
                for (int j = 0; j < J; ++j) {
                    const u32 v = __hip_atomic_load(pr + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mn = v < mn ? v : mn;
                }
                if (mn >= need) break;
                if (++spins > 512) { sib_on = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (dbg) t_epi += wall_clock64() - e0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wide_flush<Q, kW8Threads>(a, ctl, jt, tid);
    if (dbg) {   // where and when this wave ran: absolute start (100 MHz), HW_ID (CU / SE / SIMD / wave slot) and XCC_ID registers
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_id), "=s"(xcc_id));
        dbg[10] = t_begin; dbg[11] = ((unsigned long long)xcc_id << 32) | hw_id;
    }
    if (dbg) { dbg[0] = wall_clock64() - t_begin; dbg[1] = t_wait; dbg[2] = t_bar; dbg[3] = t_epi; dbg[4] = t_inv; dbg[5] = t_flt; dbg[6] = t_ph[0]; dbg[7] = t_ph[1]; dbg[8] = t_ph[2]; dbg[9] = t_ph[3]; }
}

#undef VF8_TILE_NEXT
#undef VF8_TILE_FIRST
#undef VF8_SLOTS_IO
#undef VF8_SLOTS_OUT
#undef VF8_IN_OPS
#undef VF8_STEP
#undef VF8_ACC_OPS

// waves = 8: a.jtiles 256-query tiles, one 512-thread workgroup per CU; waves = 4: a.jtiles 128-query tiles, two 256-thread workgroups per CU
size_t scan_wide8_lds_bytes(int waves, int stage_cap) {
    return waves == 8 ? scan_wide_lds_bytes(stage_cap) : (size_t)65536 + (16 + 3 * 128 * 4) + (size_t)stage_cap * 16;
}
int scan_wide8_stage_cap(int waves) { return waves == 8 ? 3584 : 896; }   // what the operand stages and the control block leave of 160 / 80 KB

// resident workgroups per CU of the wide8 kernel with that many candidate-stage entries (hipOccupancyMaxActiveBlocksPerMultiprocessor)
int scan_wide8_occupancy(int waves, int stage_cap) {
    int n = -1;
    const size_t lds = scan_wide8_lds_bytes(waves, stage_cap);
#ifdef VF_EXPERIMENTS
    hipError_t e = waves == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scan_wide8<4>, 256, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scan_wide8<8>, 512, lds);
#else
    if (waves != 8) return -1;   // the 4-wave form (measured and rejected) is built with -DVF_EXPERIMENTS only
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_scan_wide8<8>, 512, lds);
#endif
    return e == hipSuccess ? n : -(int)e;
}

hipError_t launch_scan_wide8(const ScanArgs& a, int waves, hipStream_t s) {
    const int grid = 8 * a.jtiles * ((a.rgroups + 7) / 8);
#ifdef VF_EXPERIMENTS
    static const size_t pad4 = [] { const char* e = getenv("VF_W8_PAD_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();   // experiment: extra LDS per 4-wave workgroup (forces one per CU)
    if (waves == 4) hipLaunchKernelGGL(k_scan_wide8<4>, dim3(grid), dim3(256), scan_wide8_lds_bytes(4, a.stage_cap) + pad4, s, a);
    else
#else
    if (waves != 8) return hipErrorInvalidValue;
#endif
    hipLaunchKernelGGL(k_scan_wide8<8>, dim3(grid), dim3(512), scan_wide8_lds_bytes(8, a.stage_cap), s, a);
    return hipGetLastError();
}

size_t scan_wide_lds_bytes(int stage_cap) {
    return (size_t)3 * kWideBuf + kWideCtl + (size_t)stage_cap * 16;
}

hipError_t launch_scan_wide(const ScanArgs& a, int mode, int rows_are_fp8, hipStream_t s) {
    const int grid = 8 * a.jtiles * ((a.rgroups + 7) / 8);
    const size_t lds = scan_wide_lds_bytes(mode == kModeMain ? a.stage_cap : 0);
#define VF_WCASE(MODEV, F8V) \
    if (mode == MODEV && (rows_are_fp8 != 0) == (F8V != 0)) { hipLaunchKernelGGL((k_scan_wide<MODEV, F8V>), dim3(grid), dim3(kWideThreads), lds, s, a); return hipGetLastError(); }
    VF_WCASE(kModeSample, 0) VF_WCASE(kModeSample, 1) VF_WCASE(kModeMain, 0) VF_WCASE(kModeMain, 1)
#undef VF_WCASE
    return hipErrorInvalidValue;
}

// dynamic LDS of k_scan: query image + (main mode) candidate stage
size_t scan_lds_bytes(int dp, int qn_tile) { return (size_t)dp * qn_tile * 2 + kCtlBytes; }
int scan_stage_cap(int dp, int qn_tile) {
    const size_t used = scan_lds_bytes(dp, qn_tile);
    const size_t freeb = used < 160 * 1024 ? 160 * 1024 - used : 0;
    const size_t area = freeb < 32 * 1024 ? freeb : 32 * 1024;
    return (int)(area / 16);
}

template <int NT, int G, int MODE, int F8>
static hipError_t launch_scan_inst(const ScanArgs& a, int grid, hipStream_t s) {
    size_t lds = scan_lds_bytes(a.dp, NT * kQueryTile);
    if (MODE == kModeMain) lds += (size_t)a.stage_cap * 16;
    hipLaunchKernelGGL((k_scan<NT, G, MODE, F8>), dim3(grid), dim3(kScanThreads), lds, s, a);
    return hipGetLastError();
}

// Segments (128 B of each row) per pipeline stage; the superstep count dp/64/G must be even (the
// tile loop alternates two register stages).  Main mode: G=2 keeps the kernel spill-free with its
// filter epilogue (234 VGPRs at NT=2).  Sample mode is latency-bound (a workgroup has only 128 rows):
// the deepest spill-free G (4) puts a third of a 32-row tile in flight per stage.
static int pick_G(int dp, int want, int mode, int f8) {
    const int segs = dp >> 6;  // dp is a multiple of 128, so segs is even
    auto ok = [&](int g) { return g >= 1 && segs % g == 0 && ((segs / g) & 1) == 0; };
    if (f8 && !(want >= 1 && want <= 4 && ok(want))) {
        // fp8 rows: a stage of G segments holds half the bytes of the fp16 kernel's, so go as deep as divides
        for (int g : {4, 3, 2}) if (ok(g)) return g;
        return 1;
    }
    if (mode == kModeSample) {
        for (int g : {4, 3, 2}) if (ok(g)) return g;
        return 1;
    }
    if (want >= 1 && want <= 4 && ok(want)) return want;
    if (ok(2)) return 2;
    if (ok(3)) return 3;
    return 1;  // segs even => always ok
}

hipError_t launch_scan(const ScanArgs& a, int mode, int qn_tile, int grid, int want_g, int rows_are_fp8, hipStream_t s) {
    const int G = pick_G(a.dp, want_g, mode, rows_are_fp8);
    const int NT = qn_tile / kQueryTile;
#define VF_CASE(NTV, GV, MODEV)                                                                         \
    if (NT == NTV && G == GV && mode == MODEV)                                                          \
        return rows_are_fp8 ? launch_scan_inst<NTV, GV, MODEV, 1>(a, grid, s) : launch_scan_inst<NTV, GV, MODEV, 0>(a, grid, s);
    VF_CASE(1, 1, kModeMain) VF_CASE(1, 2, kModeMain) VF_CASE(1, 3, kModeMain) VF_CASE(1, 4, kModeMain)
    VF_CASE(2, 1, kModeMain) VF_CASE(2, 2, kModeMain) VF_CASE(2, 3, kModeMain) VF_CASE(2, 4, kModeMain)
    VF_CASE(1, 1, kModeSample) VF_CASE(1, 2, kModeSample) VF_CASE(1, 3, kModeSample) VF_CASE(1, 4, kModeSample)
    VF_CASE(2, 1, kModeSample) VF_CASE(2, 2, kModeSample) VF_CASE(2, 3, kModeSample) VF_CASE(2, 4, kModeSample)
#undef VF_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_scan2(const ScanArgs& a, int qn_tile, int grid, int rows_are_fp8, hipStream_t s) {
    const size_t lds = scan2_lds_bytes(a.dp, qn_tile, a.stage_cap);
    if (qn_tile == kQueryTile) {
        if (rows_are_fp8) hipLaunchKernelGGL((k_scan2<1, 1>), dim3(grid), dim3(kScan2Threads), lds, s, a);
        else hipLaunchKernelGGL((k_scan2<1, 0>), dim3(grid), dim3(kScan2Threads), lds, s, a);
    } else {
        if (rows_are_fp8) hipLaunchKernelGGL((k_scan2<2, 1>), dim3(grid), dim3(kScan2Threads), lds, s, a);
        else hipLaunchKernelGGL((k_scan2<2, 0>), dim3(grid), dim3(kScan2Threads), lds, s, a);
    }
    return hipGetLastError();
}

// every (query tile, mode, shape) instantiation of k_scan2r, as X(NT, MODE, F8, S, RB, RING)
#define VF_SCAN2R_PRODUCT(X, NT, MODE) X(NT, MODE, 0, 12, 6, 6, 1) X(NT, MODE, 0, 16, 6, 4, 1) X(NT, MODE, 0, 8, 4, 6, 1) X(NT, MODE, 0, 6, 3, 6, 1) X(NT, MODE, 1, 6, 3, 6, 1) X(NT, MODE, 1, 8, 3, 4, 1)
#ifdef VF_EXPERIMENTS
#define VF_SCAN2R_SHAPES(X, NT, MODE) VF_SCAN2R_PRODUCT(X, NT, MODE) X(NT, MODE, 0, 12, 6, 6, 0) X(NT, MODE, 1, 6, 3, 4, 1)   // (+ the first form: debug bit 10; + a four-segment ring on e4m3 rows of 768: bit 11)
#else
#define VF_SCAN2R_SHAPES(X, NT, MODE) VF_SCAN2R_PRODUCT(X, NT, MODE)
#endif
#define VF_SCAN2R_ALL(X) VF_SCAN2R_SHAPES(X, 1, kModeMain) VF_SCAN2R_SHAPES(X, 2, kModeMain) VF_SCAN2R_SHAPES(X, 1, kModeSample) VF_SCAN2R_SHAPES(X, 2, kModeSample)
static hipError_t launch_scan2r_any(const ScanArgs& a, int qn_tile, int mode, int grid, int f8, size_t lds, hipStream_t s) {
    Scan2rShape sh = scan2r_shape(a.dp, f8);
#ifdef VF_EXPERIMENTS
    if (f8 && a.dp == 768 && (a.debug & 2048)) sh.RING = 4;   // timing experiment: is the scan bound by what a wave keeps in flight?  (LDS sized for six: harmless)
#endif
    const int nt = qn_tile / kQueryTile;
    bool done = false;
    const int ar = (!f8 && a.dp == 768 && (a.debug & 1024)) ? 0 : 1;
#define VF_X(NT, MODE, F8, S_, RB_, RING_, AR_) \
    if (!done && nt == NT && mode == MODE && f8 == F8 && sh.S == S_ && sh.RB == RB_ && sh.RING == RING_ && ar == AR_) { \
        hipLaunchKernelGGL((k_scan2r<NT, MODE, F8, S_, RB_, RING_, AR_>), dim3(grid), dim3(kScan2Waves * 64), lds, s, a); done = true; }
    VF_SCAN2R_ALL(VF_X)
#undef VF_X
    return done ? hipGetLastError() : hipErrorInvalidValue;
}
hipError_t launch_scan2r(const ScanArgs& a, int qn_tile, int grid, int f8, hipStream_t s) {
    return launch_scan2r_any(a, qn_tile, kModeMain, grid, f8, scan2r_lds_bytes(a.dp, qn_tile, a.stage_cap, f8), s);
}
// the sample pass on k_scan2r's operand path: `grid` workgroups walk the sample parts of a.scan_grid ranges (a.samp rows per wave of k_scan's geometry)
hipError_t launch_scan2r_sample(const ScanArgs& a, int qn_tile, int grid, int f8, hipStream_t s) {
    return launch_scan2r_any(a, qn_tile, kModeSample, grid, f8, scan2r_lds_bytes(a.dp, qn_tile, 0, f8), s);
}

template <int NT, int G, int MODE>
static hipError_t configure_one() {
    hipError_t e = hipFuncSetAttribute((const void*)k_scan<NT, G, MODE, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_scan<NT, G, MODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// ------------------------------------------------------------------------------------------------
// k_sel0: per query, seed the threshold from the sample scores: LDS histogram -> tau bin with
// >= kprime sample rows at or above it -> emit those rows as the first candidates, publish the
// fine / coarse histograms the main scan keeps counting into.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_tau_from_lds(const u32* lh, int kprime, int lane) {
    u32 hv[32];
    u32 sum = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) { hv[i] = lh[lane * 32 + i]; sum += hv[i]; }
    u32 suf = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = __shfl_down(suf, off);
        if (lane + off < 64) suf += v;
    }
    const u32 above = suf - sum;
    int best = -1;
    if (above < (u32)kprime && suf >= (u32)kprime) {
        u32 run = above;
        bool found = false;
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            run += hv[i];
            if (!found && run >= (u32)kprime) { best = lane * 32 + i; found = true; }
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) best = max(best, __shfl_xor(best, off));
    return best;
}

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
    const long long swg = (long long)a.samp * (kScanThreads / 64);
    constexpr int kKeep = 32;  // sample scores per thread kept in registers between the two phases
    float v[kKeep];
    const bool in_regs = len <= (long long)kKeep * 1024;
    // Pre-threshold from bucket maxima: each wave takes the r-th largest (r = ceil(k'/16)) of its 64 per-lane
    // maxima with r wave-max rounds; the minimum of the 16 wave values has >= 16 r >= k' sample scores at or
    // above it, so only scores >= it can matter.  This keeps a few hundred values out of 32K for the LDS
    // histogram (whose atomics pile onto ~100 hot bins otherwise: 14 of this kernel's 21 us).
    __shared__ float wave_thr[16];
    float pre = -INFINITY;
    if (q < a.nq) {
        if (in_regs) {
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < kKeep; ++u) {
                const long long i = tid + (long long)u * 1024;
                const float x = s[i < len ? i : len - 1];
                v[u] = i < len ? x : -INFINITY;
                mx = fmaxf(mx, v[u]);
            }
            if (a.kprime <= 1024) {
                const int rounds = (a.kprime + 15) / 16;
                const int lane = tid & 63;
                float cur = mx, thr = -INFINITY;
                for (int i = 0; i < rounds; ++i) {
                    float wm = cur;
#pragma unroll
                    for (int o = 32; o; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
                    thr = wm;
                    const unsigned long long bal = __ballot(cur == wm);
                    if (bal && lane == __ffsll((long long)bal) - 1) cur = -INFINITY;  // drop ONE holder of the max
                }
                if (lane == 0) wave_thr[tid >> 6] = thr;
                __syncthreads();
                pre = wave_thr[0];
#pragma unroll
                for (int w = 1; w < 16; ++w) pre = fminf(pre, wave_thr[w]);
            }
#pragma unroll
            for (int u = 0; u < kKeep; ++u)
                if (v[u] > -INFINITY && v[u] >= pre) atomicAdd(&lh[bin_of_x(bin_x(v[u]))], 1u);
        } else {
            for (long long i = tid; i < len; i += 1024) {
                const float x = s[i];
                if (x > -INFINITY) atomicAdd(&lh[bin_of_x(bin_x(x))], 1u);
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int nb = wave_tau_from_lds(lh, a.kprime, tid);
        if (tid == 0) sbin = nb > 0 ? nb : 0;
    }
    __syncthreads();
    const int tb = sbin;
    auto emit = [&](float x, long long i) {
        if (x > -INFINITY && bin_of_x(bin_x(x)) >= tb) {
            const u32 slot = atomicAdd(&lcnt, 1u);
            const u32 wg = (u32)i / (u32)swg;  // 32-bit: the sample has < 2^31 slots
            const long long row = a.wg_base[wg] + (long long)((u32)i - wg * (u32)swg);
            if (slot < (u32)a.cap) a.cand[(long long)q * a.cap + slot] = ((u64)orderkey(x) << 32) | (u64)(u32)row;
        }
    };
    if (q < a.nq) {
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < kKeep; ++u) emit(v[u], tid + (long long)u * 1024);
        } else {
            for (long long i = tid; i < len; i += 1024) emit(s[i], i);
        }
    }
    __syncthreads();
    for (int b = tid; b < kHistBins; b += 1024) a.hist[(long long)q * kHistBins + b] = (q < a.nq && b >= tb) ? lh[b] : 0u;
    if (tid < 64) {
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int b = tid * 32 + i;
            c += (q < a.nq && b >= tb) ? lh[b] : 0u;
        }
        a.hist_coarse[q * 64 + tid] = c;
    }
    if (tid == 0) {
        a.cnt[q * kCntStride] = lcnt;
        a.tau_bin[q] = (q < a.nq && !(a.debug & 1)) ? tb : kHistBins;
    }
    // the main scan's pool counters (tiles any workgroup may claim from the tail of every range) start at 0
    if (q == 0 && a.tile_cnt)
        for (int i = tid; i < a.scan_grid; i += 1024) a.tile_cnt[i] = 0u;
}

hipError_t launch_sel0(const ScanArgs& a, int qn_tile, hipStream_t s) {
    hipLaunchKernelGGL(k_sel0, dim3(qn_tile), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_final: per query.  Keep the candidates at or above the scan's final threshold, take the best k' of them by
// approximate score, re-score those with the CANONICAL fp32 arithmetic (bit-identical to oracle/vf_oracle.c), rank
// by (canonical desc, id asc), write k results and the exactness certificate:
//   every row not re-scored has approx <= A (A = approx of the k'-th candidate), hence canonical
//   <= A + eps; if canonical_k > A + eps no such row can enter the top k  =>  result is exact.
// LDS holds only the survivors (<= sel_cap, typically 1.2-2 x k') and the ranked k': 12 KB at k = 100, 46 KB at
// k = 1000 -- the candidate list itself (up to 16384 per query) is read once from global memory.  Ranking is by
// counting (rank = number of larger keys; keys are unique): O(n^2 / threads) compares against LDS-broadcast keys
// but NO barriers inside.  If the threshold turns out not to be a valid bound and the list does not fit the
// survivor area, the query is flagged and recomputed by the exact path (hostile data only).
// ------------------------------------------------------------------------------------------------
constexpr int kFinalThreads = 256;

// out[r] = the key of rank r (descending) for every r < limit; keys[0..n) unique, in LDS.
__device__ __forceinline__ void rank_select_desc(const u64* keys, int n, u64* out, int limit, int tid) {
    for (int i0 = tid; i0 < n; i0 += 4 * kFinalThreads) {
        u64 mine[4];
        int rank[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kFinalThreads;
            mine[u] = i < n ? keys[i] : 0ull;
            rank[u] = 0;
        }
        // 16 keys per batch are read (same address in every lane: LDS broadcast) before any compare,
        // so the loop runs at LDS throughput instead of one LDS latency per key
        int j = 0;
        for (; j + 16 <= n; j += 16) {
            u64 kj[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) kj[t] = keys[j + t];
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) rank[u] += kj[t] > mine[u] ? 1 : 0;
        }
        for (; j < n; ++j) {
            const u64 kj = keys[j];
#pragma unroll
            for (int u = 0; u < 4; ++u) rank[u] += kj > mine[u] ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * kFinalThreads < n && rank[u] < limit) out[rank[u]] = mine[u];
    }
}

// Canonical cosine of one corpus row against the (already normalised) query by TWO lanes: lane half h owns the
// partial sums acc[8h .. 8h+7] (elements j with (j & 15) in that range, in ascending j: the oracle's order), then
// the fixed tree.  A lane streams its 8 elements of every 16-element block with one 16-byte load (fp16 rows; two
// for fp32 rows, one 8-byte load for fp8 rows), kBlk blocks in flight at once; all loads are issued
// UNCONDITIONALLY (block index clamped, contribution masked): a runtime condition around a load makes hipcc
// branch around each one and wait vmcnt(0) per element.  qs = the query vector in LDS (broadcast reads).
template <int DT> struct RowVec;
template <> struct RowVec<VF_DTYPE_F16> { typedef uint4 type; };
template <> struct RowVec<VF_DTYPE_FP8_E4M3> { typedef uint2 type; };
template <> struct RowVec<VF_DTYPE_F32> { struct type { uint4 a, b; }; };

template <int DT>
__device__ __forceinline__ void decode8(const typename RowVec<DT>::type& v, float (&x)[8]) {
    if constexpr (DT == VF_DTYPE_F16) {
        const h8 hv = __builtin_bit_cast(h8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)hv[e];
    } else if constexpr (DT == VF_DTYPE_FP8_E4M3) {
        const h8 hv = cvt8_e4m3(v.x, v.y);   // the scan's own conversion (pinned against the oracle's table)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)hv[e];
    } else {
        x[0] = __uint_as_float(v.a.x); x[1] = __uint_as_float(v.a.y); x[2] = __uint_as_float(v.a.z); x[3] = __uint_as_float(v.a.w);
        x[4] = __uint_as_float(v.b.x); x[5] = __uint_as_float(v.b.y); x[6] = __uint_as_float(v.b.z); x[7] = __uint_as_float(v.b.w);
    }
}

template <int DT>
__device__ __forceinline__ float rescore_pair(const void* rows, long long row, int d, const float* qs, float nm, int h) {
    constexpr int ESZ = DT == VF_DTYPE_F32 ? 4 : (DT == VF_DTYPE_F16 ? 2 : 1);
    constexpr int kBlk = DT == VF_DTYPE_F32 ? 8 : 16;   // 16-element blocks in flight per lane
    typedef typename RowVec<DT>::type vec_t;
    const float inv = canon_inv(nm);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    const char* base = (const char*)rows + row * (long long)d * ESZ;
    const int nblk = d >> 4;
    if constexpr (DT == VF_DTYPE_FP8_E4M3) {
        if ((d & 15) == 0) {
            // e4m3 rows: a 16-element block is 16 bytes.  Lane h loads block 2 p + h WHOLE (one 16-byte load instead of two lanes x
            // 8 bytes: half the load instructions, twice the bytes in flight per lane -- the gather is latency-bound: round 4,
            // profiles/r04_stamps_final_c5.log: 503 of 863 us of a k = 1000 workgroup at 10M rows) and the pair swaps halves by DPP:
            // lane 0 keeps elements 0..7 of its block and takes elements 0..7 of the partner's, lane 1 the upper halves.  Blocks are
            // consumed in ascending order, so every partial sum sees its elements in the oracle's order.
            constexpr int kPair = 16;   // block pairs in flight per lane (256 bytes)
            const int npair = (nblk + 1) >> 1;
            for (int p0 = 0; p0 < npair; p0 += kPair) {
                uint4 v[kPair];
#pragma unroll
                for (int u = 0; u < kPair; ++u) {
                    int b = 2 * (p0 + u) + h;
                    b = b < nblk ? b : nblk - 1;
                    v[u] = *(const uint4*)(base + (long long)b * 16);
                }
#pragma unroll
                for (int u = 0; u < kPair; ++u) {
                    const u32 sx = h ? v[u].x : v[u].z, sy = h ? v[u].y : v[u].w;          // what the partner needs of my block
                    const u32 rx = (u32)__shfl_xor((int)sx, 1), ry = (u32)__shfl_xor((int)sy, 1);
                    const uint2 blk0 = h ? make_uint2(rx, ry) : make_uint2(v[u].x, v[u].y);      // my 8 elements of block 2 p
                    const uint2 blk1 = h ? make_uint2(v[u].z, v[u].w) : make_uint2(rx, ry);      // ... of block 2 p + 1
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int bb = 2 * (p0 + u) + t;
                        const bool live = bb < nblk;
                        const int b = live ? bb : nblk - 1;
                        float x[8];
                        decode8<DT>(t ? blk1 : blk0, x);
                        const float4 q0 = *(const float4*)(qs + b * 16 + 8 * h), q1 = *(const float4*)(qs + b * 16 + 8 * h + 4);
                        const float qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float tt = __builtin_fmaf(qq[e], x[e] * inv, acc[e]);
                            acc[e] = live ? tt : acc[e];
                        }
                    }
                }
            }
            goto tail;
        }
    }
    if ((((long long)d * ESZ) & 15) == 0) {
        for (int b0 = 0; b0 < nblk; b0 += kBlk) {
            vec_t v[kBlk];
#pragma unroll
            for (int u = 0; u < kBlk; ++u) {
                const int b = b0 + u < nblk ? b0 + u : nblk - 1;
                v[u] = *(const vec_t*)(base + ((long long)b * 16 + 8 * h) * ESZ);
            }
#pragma unroll
            for (int u = 0; u < kBlk; ++u) {
                const bool live = b0 + u < nblk;
                const int b = live ? b0 + u : nblk - 1;
                float x[8];
                decode8<DT>(v[u], x);
                const float4 q0 = *(const float4*)(qs + b * 16 + 8 * h), q1 = *(const float4*)(qs + b * 16 + 8 * h + 4);
                const float qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = __builtin_fmaf(qq[e], x[e] * inv, acc[e]);
                    acc[e] = live ? t : acc[e];
                }
            }
        }
    } else {  // rows that are not 16-byte aligned (odd d): element loads, same order
        for (int b = 0; b < nblk; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = b * 16 + 8 * h + e;
                acc[e] = __builtin_fmaf(qs[j], load_elem(rows, DT, row * (long long)d + j) * inv, acc[e]);
            }
    }
tail:
    // tail block (d % 16 elements): element j = 16 nblk + 8 h + e exists while j < d
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = nblk * 16 + 8 * h + e;
        if (j < d) acc[e] = __builtin_fmaf(qs[j], load_elem(rows, DT, row * (long long)d + j) * inv, acc[e]);
    }
    // canonical tree: s8[l] = acc[l] + acc[l + 8] needs the partner half's sums; then 4, 2, 1 in-lane
    float s8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s8[e] = acc[e] + __shfl_xor(acc[e], 1);
    const float s4[4] = {s8[0] + s8[4], s8[1] + s8[5], s8[2] + s8[6], s8[3] + s8[7]};
    return (s4[0] + s4[2]) + (s4[1] + s4[3]);
}

__global__ __launch_bounds__(kFinalThreads) void k_final(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u64* top = (u64*)smem_raw;                // [top_cap] ranked approx keys, then ranked canonical keys
    u64* sel = top + a.top_cap;               // [sel_cap] survivors of the threshold pre-filter
    u64* rk = sel;                            // re-scored keys (unordered): reuses sel once `top` is built
    float* qs = (float*)(sel + a.sel_cap);    // [d] this query's canonical normalised vector (+ 4 spare words)
    u32* lh = (u32*)(qs + ((a.d + 3) & ~3));  // [kHistBins] histogram of the candidates' score bins
    u32& s_nsel = lh[kHistBins];              // all LDS is dynamic: the kernel may be given the full 160 KB
    int& s_bin = *(int*)(lh + kHistBins + 1);
    u32& s_minkey = lh[kHistBins + 2];        // smallest approximate key among the survivors
    const int q = blockIdx.x, tid = threadIdx.x;
    unsigned long long* dbg = (a.dbg && tid == 0) ? a.dbg + (long long)q * 8 : nullptr;
    if (dbg) dbg[0] = wall_clock64();
    const u32 n_raw = a.cnt[q * kCntStride];
    const int n = n_raw < (u32)a.cap ? (int)n_raw : a.cap;
    const int mwant = n < a.kprime ? n : a.kprime;   // how many of the best candidates must be re-scored
    const u64* cq = a.cand + (long long)q * a.cap;
    const int tbin = a.tau_bin[q];
    if (n_raw > (u32)a.cap) {   // overflowed list: not every slot need be written -- nothing here can be trusted, exact path
        for (int i = tid; i < a.k; i += kFinalThreads) { a.out_ids[(long long)q * a.k + i] = -1; a.out_scores[(long long)q * a.k + i] = -FLT_MAX; }
        if (tid == 0) { a.flags[q] = 2; a.cand_count_out[q] = n_raw; }
        return;
    }
    for (int j = tid; j < a.d; j += kFinalThreads) qs[j] = a.qn[(long long)q * a.d + j];
    for (int b = tid; b < kHistBins; b += kFinalThreads) lh[b] = 0u;
    if (tid == 0) { s_nsel = 0u; s_bin = 0; s_minkey = 0xFFFFFFFFu; }
    __syncthreads();
    // Select by the list's OWN histogram, in FINE bins: 16 per threshold bin, counted from the scan's final threshold
    // (every candidate sits at or just below it; nearly all of a k = 1000 list falls into ~40 threshold bins, so a
    // histogram at that resolution would serialise its LDS atomics on those few words and leave ~100 candidates in the
    // boundary bin to be ranked).  The highest fine bin b* with >= m candidates at or above it splits the list: the
    // survivors (m plus the handful that share b*) are ALL re-scored -- no approximate ranking is needed at all; A, the
    // certificate's "best score a row that was not re-scored can have", is the smallest survivor.
    const float fscale = tbin > 0 ? 16.0f : 1.0f, forig = tbin > 0 ? (float)tbin : 0.0f;
    auto fine_bin = [&](u64 kv) {
        const float x = (bin_x(unorderkey((u32)(kv >> 32))) - forig) * fscale;
        const int fb = (int)floorf(x) + 1;   // bin 0 is for what lies BELOW the origin only (entries kept under an earlier, looser threshold)
        return fb < 0 ? 0 : (fb > kHistBins - 1 ? kHistBins - 1 : fb);
    };
    for (int i = tid; i < n; i += kFinalThreads) atomicAdd(&lh[fine_bin(cq[i])], 1u);
    __syncthreads();
    if (tid < 64) {
        const int nb = wave_tau_from_lds(lh, mwant > 0 ? mwant : 1, tid);
        if (tid == 0) s_bin = nb > 0 ? nb : 0;
    }
    __syncthreads();
    const int sbin = s_bin;
    u32 mymin = 0xFFFFFFFFu;
    for (int i = tid; i < n; i += kFinalThreads) {
        const u64 kv = cq[i];
        if (fine_bin(kv) >= sbin) {
            const u32 sl = atomicAdd(&s_nsel, 1u);
            if (sl < (u32)a.sel_cap) sel[sl] = kv;
            const u32 ak = (u32)(kv >> 32);
            mymin = ak < mymin ? ak : mymin;
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) { const u32 o = (u32)__shfl_xor((int)mymin, off); mymin = o < mymin ? o : mymin; }
    if ((tid & 63) == 0) atomicMin(&s_minkey, mymin);
    __syncthreads();
    const int ns = (int)s_nsel;
    if (ns > a.sel_cap) {  // more than sel_cap candidates share the boundary bin (hostile data): exact path
        for (int i = tid; i < a.k; i += kFinalThreads) { a.out_ids[(long long)q * a.k + i] = -1; a.out_scores[(long long)q * a.k + i] = -FLT_MAX; }
        if (tid == 0) { a.flags[q] = n_raw > (u32)a.cap ? 2 : 1; a.cand_count_out[q] = n_raw; }
        return;
    }
    if (dbg) dbg[1] = wall_clock64();
    const float approx_floor = ns > 0 ? unorderkey(s_minkey) : -INFINITY;
    // canonical re-score of every survivor: two lanes per row, a block of 128 rows at a time.  The key of row i replaces
    // its approximate key in place (sel[i] is read by the pair that writes it).
    const int pr = tid >> 1, h = tid & 1;
    for (int i0 = 0; i0 < ns; i0 += kFinalThreads / 2) {
        const int i = i0 + pr;
        u32 row = (u32)sel[i < ns ? i : ns - 1];   // idle pairs shadow the last row (whole wave stays converged)
        row = (long long)row < a.n_rows ? row : (u32)(a.n_rows - 1);   // never index the corpus with a value the scan did not produce
        const float nm = a.norm[row];
        const float acc = a.orig_dtype == VF_DTYPE_F16       ? rescore_pair<VF_DTYPE_F16>(a.rows_orig, row, a.d, qs, nm, h)
                          : a.orig_dtype == VF_DTYPE_FP8_E4M3 ? rescore_pair<VF_DTYPE_FP8_E4M3>(a.rows_orig, row, a.d, qs, nm, h)
                                                              : rescore_pair<VF_DTYPE_F32>(a.rows_orig, row, a.d, qs, nm, h);
        __builtin_amdgcn_wave_barrier();   // every pair of this wave has read its row id before any pair overwrites a slot
        if (h == 0 && i < ns) rk[i] = ((u64)orderkey(acc) << 32) | (u64)(0xFFFFFFFFu - row);
    }
    __syncthreads();
    if (dbg) dbg[2] = wall_clock64();
    // rank by (canonical desc, id asc): counting for short lists, an LDS bitonic network beyond 256 keys (counting is
    // O(n^2 / threads): 60 us at n = 1280; the network's 66 barriers cost ~10 us with four waves)
    const u64* ranked;
    if (ns <= 256) {
        rank_select_desc(rk, ns, top, ns < a.top_cap ? ns : a.top_cap, tid);
        ranked = top;
    } else {
        const int P = next_pow2(ns);   // <= sel_cap (a power of two)
        for (int i = ns + tid; i < P; i += kFinalThreads) rk[i] = 0ull;
        __syncthreads();
        bitonic_sort_desc(rk, P, tid, kFinalThreads);
        ranked = rk;
    }
    __syncthreads();
    if (dbg) dbg[3] = wall_clock64();
    const int m = ns;   // rows re-scored
    for (int i = tid; i < a.k; i += kFinalThreads) {
        long long id = -1;
        float sc = -FLT_MAX;
        if (i < m) {
            const u64 kv = ranked[i];
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
            // (1) the final tau must be a valid bound: >= kprime candidates at or above it
            if (tbin > 0 && (mwant < a.kprime || bin_of_x(bin_x(approx_floor)) < tbin)) flag = 1;
            // (2) the k-th canonical score must clear every row that was not re-scored
            if (a.k > m) flag = 1;
            else if (flag == 0) {
                const float ck_k = unorderkey((u32)(ranked[a.k - 1] >> 32));
                if (!(ck_k > approx_floor + (a.eps_q ? fmaxf(a.eps_q[q], a.eps) : a.eps))) flag = 1;
            }
        }
        // flags / counts live in host-mapped pinned memory: no device-to-host copy kernel needed
        a.flags[q] = flag;
        a.cand_count_out[q] = n_raw;
        if (dbg) dbg[4] = wall_clock64();
    }
}

hipError_t launch_final(FinalArgs a, int nq, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    a.top_cap = (a.kprime + 1) & ~1;
    int sc = 1024;
    // survivors = k' + what shares the boundary fine bin (1 / 16 of a threshold bin: a handful of rows on smooth data; a list that
    // does not fit is recomputed by the exact path).  Sized at 1.25 k' rounded up to a power of two: at k' = 1504 that is 2048 entries
    // instead of 4096 -- 40 instead of 57 KB of LDS per workgroup, four workgroups per CU instead of two for a latency-bound gather
    while (sc < a.kprime + a.kprime / 4 && sc < 4096) sc <<= 1;
    a.sel_cap = sc > a.top_cap ? sc : a.top_cap;
    const size_t lds = ((size_t)a.top_cap + a.sel_cap) * 8 + ((size_t)a.d + 4 + kHistBins + 4) * 4;
    hipLaunchKernelGGL(k_final, dim3(nq), dim3(kFinalThreads), lds, s, a);
    return hipGetLastError();
}

// Opt the kernels into > 64 KB of dynamic LDS.  Once per device per process (hipFuncSetAttribute is not free
// and the merge entry points sit on the per-batch path of the multi-GPU loop).
hipError_t scan_configure() {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
#define VF_CFG(NTV, GV) \
    if ((e = configure_one<NTV, GV, kModeSample>()) != hipSuccess) return e; \
    if ((e = configure_one<NTV, GV, kModeMain>()) != hipSuccess) return e;
    VF_CFG(1, 1) VF_CFG(1, 2) VF_CFG(1, 3) VF_CFG(1, 4)
    VF_CFG(2, 1) VF_CFG(2, 2) VF_CFG(2, 3) VF_CFG(2, 4)
#undef VF_CFG
#define VF_X(NT, MODE, F8, S_, RB_, RING_, AR_) \
    if ((e = hipFuncSetAttribute((const void*)k_scan2r<NT, MODE, F8, S_, RB_, RING_, AR_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    VF_SCAN2R_ALL(VF_X)
#undef VF_X
    if ((e = hipFuncSetAttribute((const void*)k_scan2<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan2<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan2<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan2<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan_wide<kModeSample, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan_wide<kModeSample, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan_wide<kModeMain, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan_wide<kModeMain, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_scan_wide8<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
#ifdef VF_EXPERIMENTS
    if ((e = hipFuncSetAttribute((const void*)k_scan_wide8<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
#endif
    if ((e = hipFuncSetAttribute((const void*)k_sort_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_topk_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_merge_topk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_final, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if (dev >= 0 && dev < 64) done[dev] = true;
    return hipSuccess;
}

}  // namespace vf
