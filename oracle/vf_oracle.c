/*
 * vf_oracle.c -- CPU ORACLE for the VeritasFi dense-retrieval hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (veritasfi_amd/) never links, imports or falls back to it.
 *
 * What it restates (reference = /root/reference, read as text only):
 *   - cosine + top-k of  experiments/retriever/step3_mul.py:233-253 (select_top_chunks),
 *     :255-289 (select_top_chunks_batch) and continuous_retrieval.py:154-167:
 *       sklearn.cosine_similarity(E, C)  ==  normalize(E) @ normalize(C).T
 *       (row_norm = sqrt(sum x^2); zero norm -> divide by 1; x / norm elementwise),
 *       then per row  argsort(sim)[-k:][::-1]  (all rows sorted when k == -1).
 *   - the index/search surface of src/utils/faissRetriever.py:14-24,33-38
 *     (normalise corpus at build, normalise queries, exact inner-product top-k,
 *      k > N pads ids with -1 and scores with -FLT_MAX).
 *   - the n x n similarity matrix of src/utils/ensembleRetriever.py:275-279.
 *
 * Third-party arithmetic the reference delegates to and that is NOT under /root/reference:
 * scikit-learn (cosine_similarity; unpinned, 1.7.2 present in the build container),
 * faiss (IndexFlatIP / normalize_L2; unpinned, absent).  Both reduce to an fp32 dot
 * product whose summation ORDER is an implementation detail of the BLAS in use, so
 * it is not a property of the reference.  This oracle therefore fixes a CANONICAL
 * order (below) that the HIP path reproduces bit-for-bit; parity with the real
 * reference import is pinned on tie-free golden vectors (tests/golden/, made by
 * tools/gen_golden.py) where ids must match exactly and scores to 1e-6.
 *
 * CANONICAL ARITHMETIC (shared with veritasfi_amd/csrc, see DESIGN.md "Canonical score"):
 *   dot16(a,b,d): 16 interleaved partial sums, acc[j & 15] = fmaf(a[j], b[j], acc[j & 15])
 *                 for j = 0..d-1 in increasing j, then the fixed tree
 *                 acc[l] += acc[l+8] (l<8); acc[l] += acc[l+4] (l<4);
 *                 acc[l] += acc[l+2] (l<2); result = acc[0] + acc[1].
 *   norm(x)     : n = (float)sqrt((double)dot16(x,x,d));  n == 0 -> 1.
 *   xn[j]       : x[j] * inv,  inv = (float)(1.0 / (double)n)   -- faiss's own form
 *                 (fvec_renorm_L2: inv_nr = 1.0 / sqrtf(nr); x[i] *= inv_nr); sklearn divides
 *                 instead, which differs in the last bit only (goldens are compared at 2e-6)
 *   cos(q,c)    : dot16(qn, cn, d)
 *   ranking     : descending score, ties broken by LOWER id first.
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -mf16c -ffp-contract=off -fopenmp).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VF_L 16

/* ---- canonical primitives ------------------------------------------------ */

static inline float dot16(const float* a, const float* b, int d) {
    float acc[VF_L];
    for (int l = 0; l < VF_L; ++l) acc[l] = 0.0f;
    int j = 0;
    for (; j + VF_L <= d; j += VF_L)
        for (int l = 0; l < VF_L; ++l) acc[l] = __builtin_fmaf(a[j + l], b[j + l], acc[l]);
    for (int l = 0; j < d; ++j, ++l) acc[l] = __builtin_fmaf(a[j], b[j], acc[l]);
    for (int l = 0; l < 8; ++l) acc[l] = acc[l] + acc[l + 8];
    for (int l = 0; l < 4; ++l) acc[l] = acc[l] + acc[l + 4];
    for (int l = 0; l < 2; ++l) acc[l] = acc[l] + acc[l + 2];
    return acc[0] + acc[1];
}

static inline float canon_norm(const float* x, int d) {
    float n = (float)sqrt((double)dot16(x, x, d));
    return n == 0.0f ? 1.0f : n;
}

static inline void canon_normalize(const float* x, int d, float* out) {
    const float inv = (float)(1.0 / (double)canon_norm(x, d));
    for (int j = 0; j < d; ++j) out[j] = x[j] * inv;
}

static inline void half_row_to_float(const uint16_t* h, int d, float* out) {
    int j = 0;
    for (; j + 8 <= d; j += 8)
        _mm256_storeu_ps(out + j, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(h + j))));
    for (; j < d; ++j) out[j] = _cvtsh_ss(h[j]);
}

/* exported for unit tests of the primitive itself */
float vf_oracle_dot16(const float* a, const float* b, int d) { return dot16(a, b, d); }

int vf_oracle_row_norms_f32(const float* x, int64_t n, int d, float* out) {
    if (!x || !out || n < 0 || d <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = canon_norm(x + i * (int64_t)d, d);
    return 0;
}

int vf_oracle_normalize_f32(const float* x, int64_t n, int d, float* out) {
    if (!x || !out || n < 0 || d <= 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) canon_normalize(x + i * (int64_t)d, d, out + i * (int64_t)d);
    return 0;
}

/* dense canonical cosine matrix out[na, nb]  (ensembleRetriever.py:275-279 when a == b;
 * step3_mul.py:275 similarities_matrix in general) */
int vf_oracle_cosine_f32(const float* a, int64_t na, const float* b, int64_t nb, int d, float* out) {
    if (!a || !b || !out || na < 0 || nb < 0 || d <= 0) return -1;
    float* an = (float*)malloc(sizeof(float) * (size_t)(na > 0 ? na : 1) * d);
    if (!an) return -2;
    vf_oracle_normalize_f32(a, na, d, an);
#pragma omp parallel
    {
        float* bn = (float*)malloc(sizeof(float) * d);
#pragma omp for schedule(static)
        for (int64_t j = 0; j < nb; ++j) {
            canon_normalize(b + j * (int64_t)d, d, bn);
            for (int64_t i = 0; i < na; ++i) out[i * nb + j] = dot16(an + i * (int64_t)d, bn, d);
        }
        free(bn);
    }
    free(an);
    return 0;
}

/* ---- top-k --------------------------------------------------------------- */

typedef struct { float s; int64_t id; } vf_hit;

/* "a ranks before b": higher score first, ties -> lower id first */
static inline int hit_before(vf_hit a, vf_hit b) { return a.s > b.s || (a.s == b.s && a.id < b.id); }

/* heap[0] is the WORST kept hit (the one every other kept hit ranks before) */
static void heap_sift_down(vf_hit* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && hit_before(h[w], h[l])) w = l;
        if (r < n && hit_before(h[w], h[r])) w = r;
        if (w == i) return;
        vf_hit t = h[i]; h[i] = h[w]; h[w] = t; i = w;
    }
}
static void heap_sift_up(vf_hit* h, int i) {
    while (i > 0) {
        int p = (i - 1) / 2;
        if (hit_before(h[p], h[i])) { vf_hit t = h[i]; h[i] = h[p]; h[p] = t; i = p; } else return;
    }
}
static inline void heap_offer(vf_hit* h, int* n, int k, vf_hit x) {
    if (*n < k) { h[*n] = x; heap_sift_up(h, *n); ++*n; }
    else if (hit_before(x, h[0])) { h[0] = x; heap_sift_down(h, k, 0); }
}
static int hit_cmp(const void* pa, const void* pb) {
    vf_hit a = *(const vf_hit*)pa, b = *(const vf_hit*)pb;
    return hit_before(a, b) ? -1 : (hit_before(b, a) ? 1 : 0);
}

/* Sort one score row: ids/scores of the k best of n, descending, lower id wins ties.
 * k > n pads with id -1 / score -FLT_MAX (faiss contract, faissRetriever.py:37). */
int vf_oracle_topk_row(const float* scores, int64_t n, int k, int64_t* ids, float* out) {
    if (!scores || !ids || !out || n < 0 || k < 0) return -1;
    int kk = (int64_t)k < n ? k : (int)n;
    vf_hit* h = (vf_hit*)malloc(sizeof(vf_hit) * (size_t)(kk > 0 ? kk : 1));
    if (!h) return -2;
    int cnt = 0;
    for (int64_t i = 0; i < n && kk > 0; ++i) { vf_hit x = {scores[i], i}; heap_offer(h, &cnt, kk, x); }
    qsort(h, (size_t)cnt, sizeof(vf_hit), hit_cmp);
    for (int i = 0; i < k; ++i) {
        if (i < cnt) { ids[i] = h[i].id; out[i] = h[i].s; } else { ids[i] = -1; out[i] = -FLT_MAX; }
    }
    free(h);
    return 0;
}

/* ---- exact search: FaissRetriever.__init__ + .invoke arithmetic ------------ */

#define ROW_BLOCK 256

static int search_impl(const void* corpus, int is_half, int64_t n, int d, const float* queries,
                       int nq, int k, int64_t id_offset, int64_t* ids, float* scores) {
    if (!corpus || !queries || !ids || !scores || n < 0 || d <= 0 || nq < 0 || k < 0) return -1;
    if (nq == 0 || k == 0) return 0;
    const int kk = (int64_t)k < n ? k : (int)n;
    float* qn = (float*)malloc(sizeof(float) * (size_t)nq * d);
    if (!qn) return -2;
    vf_oracle_normalize_f32(queries, nq, d, qn);

    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    const size_t heap_elems = (size_t)nq * (size_t)(kk > 0 ? kk : 1);
    vf_hit* heaps = (vf_hit*)malloc(sizeof(vf_hit) * heap_elems * nthreads);
    int* counts = (int*)calloc((size_t)nq * nthreads, sizeof(int));
    if (!heaps || !counts) { free(qn); free(heaps); free(counts); return -2; }

#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        vf_hit* myh = heaps + heap_elems * tid;
        int* myc = counts + (size_t)nq * tid;
        float* rowf = (float*)malloc(sizeof(float) * d);
        float* cn = (float*)malloc(sizeof(float) * d);
#pragma omp for schedule(dynamic, ROW_BLOCK)
        for (int64_t r = 0; r < n; ++r) {
            const float* src;
            if (is_half) { half_row_to_float((const uint16_t*)corpus + r * (int64_t)d, d, rowf); src = rowf; }
            else src = (const float*)corpus + r * (int64_t)d;
            canon_normalize(src, d, cn);
            for (int q = 0; q < nq; ++q) {
                vf_hit x = { dot16(qn + (size_t)q * d, cn, d), r };
                if (kk > 0) heap_offer(myh + (size_t)q * kk, &myc[q], kk, x);
            }
        }
        free(rowf); free(cn);
    }

    /* merge the per-thread heaps: concatenate, sort, keep the first kk */
    vf_hit* all = (vf_hit*)malloc(sizeof(vf_hit) * (size_t)(kk > 0 ? kk : 1) * nthreads);
    for (int q = 0; q < nq; ++q) {
        int m = 0;
        for (int t = 0; t < nthreads; ++t) {
            const vf_hit* h = heaps + heap_elems * t + (size_t)q * kk;
            const int c = counts[(size_t)nq * t + q];
            memcpy(all + m, h, sizeof(vf_hit) * (size_t)c);
            m += c;
        }
        qsort(all, (size_t)m, sizeof(vf_hit), hit_cmp);
        for (int i = 0; i < k; ++i) {
            if (i < m && i < kk) { ids[(size_t)q * k + i] = all[i].id + id_offset; scores[(size_t)q * k + i] = all[i].s; }
            else { ids[(size_t)q * k + i] = -1; scores[(size_t)q * k + i] = -FLT_MAX; }
        }
    }
    free(all); free(heaps); free(counts); free(qn);
    return 0;
}

int vf_oracle_search_f32(const float* corpus, int64_t n, int d, const float* queries, int nq, int k,
                         int64_t id_offset, int64_t* ids, float* scores) {
    return search_impl(corpus, 0, n, d, queries, nq, k, id_offset, ids, scores);
}

int vf_oracle_search_f16(const uint16_t* corpus, int64_t n, int d, const float* queries, int nq, int k,
                         int64_t id_offset, int64_t* ids, float* scores) {
    return search_impl(corpus, 1, n, d, queries, nq, k, id_offset, ids, scores);
}

/* Merge per-shard results (multi-GPU all-gather): parts[g] holds [nq,k]
 * (ids global, padded with -1).  Output = best k over the union, same ordering rule. */
int vf_oracle_merge_topk(const int64_t* ids_in, const float* scores_in, int nparts, int nq, int k,
                         int64_t* ids, float* scores) {
    if (!ids_in || !scores_in || !ids || !scores || nparts <= 0 || nq < 0 || k < 0) return -1;
    vf_hit* all = (vf_hit*)malloc(sizeof(vf_hit) * (size_t)(nparts * k > 0 ? nparts * k : 1));
    if (!all) return -2;
    for (int q = 0; q < nq; ++q) {
        int m = 0;
        for (int g = 0; g < nparts; ++g)
            for (int i = 0; i < k; ++i) {
                const size_t o = ((size_t)g * nq + q) * k + i;
                if (ids_in[o] >= 0) { all[m].s = scores_in[o]; all[m].id = ids_in[o]; ++m; }
            }
        qsort(all, (size_t)m, sizeof(vf_hit), hit_cmp);
        for (int i = 0; i < k; ++i) {
            if (i < m) { ids[(size_t)q * k + i] = all[i].id; scores[(size_t)q * k + i] = all[i].s; }
            else { ids[(size_t)q * k + i] = -1; scores[(size_t)q * k + i] = -FLT_MAX; }
        }
    }
    free(all);
    return 0;
}

int vf_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void vf_oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
