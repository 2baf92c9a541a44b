/* Sanitizer driver for the CPU oracle (TEST INFRASTRUCTURE).  Built together with vf_oracle.c under
 * -fsanitize=address,undefined (oracle/Makefile target `sanitize`) and run by tests/test_oracle_golden.py:
 * ragged sizes, k > n, d not a multiple of 16, fp16 rows, merge of uneven parts.  Exits 0 when every call
 * returns 0 and the sanitizers stay silent (they abort the process otherwise). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

float vf_oracle_dot16(const float* a, const float* b, int d);
int vf_oracle_row_norms_f32(const float* x, int64_t n, int d, float* out);
int vf_oracle_normalize_f32(const float* x, int64_t n, int d, float* out);
int vf_oracle_cosine_f32(const float* a, int64_t na, const float* b, int64_t nb, int d, float* out);
int vf_oracle_topk_row(const float* scores, int64_t n, int k, int64_t* ids, float* out);
int vf_oracle_search_f32(const float* corpus, int64_t n, int d, const float* queries, int nq, int k, int64_t id_offset,
                         int64_t* out_ids, float* out_scores);
int vf_oracle_search_f16(const uint16_t* corpus, int64_t n, int d, const float* queries, int nq, int k, int64_t id_offset,
                         int64_t* out_ids, float* out_scores);
int vf_oracle_merge_topk(const int64_t* ids_in, const float* scores_in, int nparts, int nq, int k, int64_t* ids_out,
                         float* scores_out);

static unsigned long long st = 88172645463325252ull;
static float rnd(void) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    return (float)((double)(st % 2000001ull) / 1000000.0 - 1.0);
}

int main(void) {
    const int dims[] = {1, 7, 16, 17, 100, 768};
    int bad = 0;
    for (unsigned di = 0; di < sizeof(dims) / sizeof(dims[0]); ++di) {
        const int d = dims[di];
        const int64_t ns[] = {0, 1, 5, 333};
        for (unsigned ni = 0; ni < 4; ++ni) {
            const int64_t n = ns[ni];
            const int nq = 3, k = 10;  /* k > n for the small cases */
            float* c = malloc(sizeof(float) * (size_t)(n * d + 1));
            uint16_t* ch = malloc(sizeof(uint16_t) * (size_t)(n * d + 1));
            float* q = malloc(sizeof(float) * (size_t)(nq * d));
            for (int64_t i = 0; i < n * d; ++i) { c[i] = rnd(); ch[i] = (uint16_t)(0x3000u + (st & 0x3FFu)); }
            for (int i = 0; i < nq * d; ++i) q[i] = rnd();
            if (n > 0) for (int j = 0; j < d; ++j) c[j] = 0.0f;  /* a zero row */
            int64_t* ids = malloc(sizeof(int64_t) * nq * k);
            float* sc = malloc(sizeof(float) * nq * k);
            bad |= vf_oracle_search_f32(c, n, d, q, nq, k, 1000, ids, sc);
            bad |= vf_oracle_search_f16(ch, n, d, q, nq, k, 0, ids, sc);
            float* norms = malloc(sizeof(float) * (size_t)(n + 1));
            float* cn = malloc(sizeof(float) * (size_t)(n * d + 1));
            float* sim = malloc(sizeof(float) * (size_t)(nq * n + 1));
            bad |= vf_oracle_row_norms_f32(c, n, d, norms);
            bad |= vf_oracle_normalize_f32(c, n, d, cn);
            bad |= vf_oracle_cosine_f32(q, nq, c, n, d, sim);
            if (n > 0) bad |= vf_oracle_topk_row(sim, n, k, ids, sc);
            (void)vf_oracle_dot16(q, q, d);
            free(c); free(ch); free(q); free(ids); free(sc); free(norms); free(cn); free(sim);
        }
    }
    {   /* merge of 3 parts of k = 4 for 2 queries, with padding entries */
        const int nparts = 3, nq = 2, k = 4;
        int64_t ids_in[3 * 2 * 4]; float sc_in[3 * 2 * 4]; int64_t ids_out[2 * 4]; float sc_out[2 * 4];
        for (int i = 0; i < nparts * nq * k; ++i) { ids_in[i] = (i % 5 == 4) ? -1 : i; sc_in[i] = (i % 5 == 4) ? -3.4028234e38f : rnd(); }
        bad |= vf_oracle_merge_topk(ids_in, sc_in, nparts, nq, k, ids_out, sc_out);
    }
    printf(bad ? "oracle returned an error\n" : "sanitize_check ok\n");
    return bad ? 1 : 0;
}
