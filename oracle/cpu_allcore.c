/* cpu_allcore.c — the "best CPU" comparison line of SURVEY.md §8(d)(iii): an OpenMP CSR sum-aggregation over all host
 * cores.  TEST / BENCH INFRASTRUCTURE ONLY (same rules as gnn_oracle.c).  This is NOT a restatement of the reference (whose
 * CPU propagate is single-threaded: NNlib gather/scatter and SparseArrays `*` do not multithread); it exists so that the
 * GPU number is also shown next to what the same host could do with every core, CSR built once. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int ac_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* dst-sorted CSR of 1-based (s, t) with self loops appended: rowptr[n+1], col[E+n] (0-based sources), stable */
void ac_build_csr(const int64_t *s, const int64_t *t, int64_t E, int64_t n, int64_t *rowptr, int32_t *col) {
    memset(rowptr, 0, sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t k = 0; k < E; ++k) rowptr[t[k]]++;
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1]++; /* self loop */
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    memcpy(fill, rowptr, sizeof(int64_t) * (size_t)n);
    for (int64_t k = 0; k < E; ++k) col[fill[t[k] - 1]++] = (int32_t)(s[k] - 1);
    for (int64_t i = 0; i < n; ++i) col[fill[i]++] = (int32_t)i;
    free(fill);
}

/* out[i] = c[i] * sum_{p in row i} c[col[p]] * x[col[p]]   (GCN's normalised aggregation; c may be NULL) */
void ac_spmm_csr_omp(const int64_t *rowptr, const int32_t *col, const float *x, const float *c, int64_t n, int64_t D,
                     float *out) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
        float *o = out + i * D;
        for (int64_t d = 0; d < D; ++d) o[d] = 0.0f;
        for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const float *xr = x + (int64_t)col[p] * D;
            const float cj = c ? c[col[p]] : 1.0f;
            for (int64_t d = 0; d < D; ++d) o[d] += cj * xr[d];
        }
        if (c)
            for (int64_t d = 0; d < D; ++d) o[d] *= c[i];
    }
}
