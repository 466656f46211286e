/* harness.c — drives libgnnmp.so through its C ABI from a plain C program: no torch, no Python, device memory owned through
 * hipMalloc.  It makes the calls the Julia extension makes (graphneuralnetworks.jl_amd/julia/GNNlibGnnmpExt.jl), with the
 * argument conventions of a Julia host: 1-based Int64 COO vectors, column-major (D, N) features = row-major [N][D], a
 * (Dout, Din) column-major weight = w_layout 1 with ldw = Dout, `a` (2C, H) column-major = [H][2C].
 *   plan_create(validate) -> plan_info / plan_export -> degree -> propagate(copy_xj | w_mul_xj; +, mean, max) ->
 *   dense -> fused_conv -> gat_conv -> plan_destroy, and the EBOUNDS error path.
 * Expected values come from plain host loops in this file (edge order, separately rounded products: bit-exact where the
 * library promises bits).  Prints C_HARNESS_OK and exits 0 on success.   Built by __graft_entry__.build() / tests. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnnmp.h"

int gnnmp_tune(int knob, int value);   /* experiment hook of the library (not part of the drop-in surface, hence not in gnnmp.h) */

#define CHECK_HIP(e)                                                                  \
    do {                                                                              \
        hipError_t e__ = (e);                                                         \
        if (e__ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d hip error %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)
#define CHECK_G(e)                                                                    \
    do {                                                                              \
        int s__ = (e);                                                                \
        if (s__ != GNNMP_OK) {                                                        \
            fprintf(stderr, "%s:%d gnnmp status %d: %s\n", __FILE__, __LINE__, s__, gnnmp_last_error()); \
            exit(3);                                                                  \
        }                                                                             \
    } while (0)
#define REQUIRE(c, ...)                                                               \
    do {                                                                              \
        if (!(c)) {                                                                   \
            fprintf(stderr, "%s:%d FAILED: ", __FILE__, __LINE__);                    \
            fprintf(stderr, __VA_ARGS__);                                             \
            fprintf(stderr, "\n");                                                    \
            exit(4);                                                                  \
        }                                                                             \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}
static float rndf(void) { return (float)(rnd() & 0xFFFFFF) / (float)0x1000000 * 2.0f - 1.0f; }

static void *dev_copy(const void *h, size_t bytes) {
    void *d = NULL;
    CHECK_HIP(hipMalloc(&d, bytes ? bytes : 16));
    if (bytes) CHECK_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return d;
}
static void *dev_alloc(size_t bytes) {
    void *d = NULL;
    CHECK_HIP(hipMalloc(&d, bytes ? bytes : 16));
    return d;
}
static void to_host(void *h, const void *d, size_t bytes) { CHECK_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); }

static double rel_err(const float *a, const float *b, size_t n) {
    double num = 0, den = 0;
    for (size_t i = 0; i < n; ++i) {
        num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]);
        den += (double)b[i] * b[i];
    }
    return sqrt(num / (den > 0 ? den : 1));
}

int main(void) {
    REQUIRE(gnnmp_version() == GNNMP_VERSION, "library / header version mismatch");
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    REQUIRE(ndev > 0, "no GPU");
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));       /* the caller's own stream, like AMDGPU.jl's task-local one */

    /* ---- a small multigraph: n nodes, random edges with duplicates and self loops, one hub above the split threshold ---- */
    const int64_t n = 300, E0 = 4000, HUB = 150, E = E0 + HUB;
    const int D = 100, Dout = 64, H = 4, C = 16;
    int64_t *s = malloc(sizeof(int64_t) * E), *t = malloc(sizeof(int64_t) * E);
    for (int64_t k = 0; k < E0; ++k) { s[k] = 1 + rnd() % n; t[k] = 1 + rnd() % (n - 5); }   /* last 5 nodes: no in-edges */
    for (int64_t k = E0; k < E; ++k) { s[k] = 1 + rnd() % n; t[k] = 7; }
    float *x = malloc(sizeof(float) * n * D), *w = malloc(sizeof(float) * E);
    for (int64_t i = 0; i < n * D; ++i) x[i] = rndf();
    for (int64_t k = 0; k < E; ++k) w[k] = 0.25f + 0.5f * (rndf() + 1.0f);
    void *ds = dev_copy(s, sizeof(int64_t) * E), *dt = dev_copy(t, sizeof(int64_t) * E);
    float *dx = dev_copy(x, sizeof(float) * n * D), *dw = dev_copy(w, sizeof(float) * E);

    /* ---- plan ---- */
    gnnmp_graph_t *plan = NULL;
    CHECK_G(gnnmp_plan_create(&plan, ds, dt, 8, 1, n, n, E, 0, 1, stream));
    int64_t info[8];
    CHECK_G(gnnmp_plan_info(plan, info));
    REQUIRE(info[0] == n && info[1] == n && info[2] == E && info[3] == E, "plan_info sizes");
    int32_t *rowptr = malloc(4 * (n + 1)), *col = malloc(4 * E), *eid = malloc(4 * E);
    void *drp = dev_alloc(4 * (n + 1)), *dcol = dev_alloc(4 * E), *deid = dev_alloc(4 * E);
    CHECK_G(gnnmp_plan_export(plan, drp, dcol, deid, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    to_host(rowptr, drp, 4 * (n + 1)); to_host(col, dcol, 4 * E); to_host(eid, deid, 4 * E);
    {   /* stable destination sort: slots of a row list the row's edges in original order */
        int64_t p = 0;
        for (int64_t i = 0; i < n; ++i) {
            REQUIRE(rowptr[i] == p, "rowptr[%lld]", (long long)i);
            for (int64_t k = 0; k < E; ++k)
                if (t[k] == i + 1) {
                    REQUIRE(eid[p] == k && col[p] == s[k] - 1, "slot %lld of row %lld", (long long)p, (long long)i);
                    ++p;
                }
        }
        REQUIRE(rowptr[n] == E && p == E, "rowptr end");
    }
    const int thresh = (int)info[7];
    REQUIRE(info[4] >= HUB && info[5] >= 1, "the hub row is split (max degree %lld, split rows %lld)", (long long)info[4], (long long)info[5]);

    /* ---- degree + propagate, all against host loops in edge order ---- */
    float *deg = malloc(4 * n), *ddeg = dev_alloc(4 * n);
    CHECK_G(gnnmp_degree_f32(plan, NULL, ddeg, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    to_host(deg, ddeg, 4 * n);
    for (int64_t i = 0; i < n; ++i) REQUIRE(deg[i] == (float)(rowptr[i + 1] - rowptr[i]), "degree[%lld]", (long long)i);

    float *ref = malloc(4 * n * D), *got = malloc(4 * n * D), *dout = dev_alloc(4 * n * D);
    for (int pass = 0; pass < 3; ++pass) {       /* 0: copy_xj +   1: w_mul_xj mean   2: copy_xj max */
        for (int64_t i = 0; i < n; ++i)
            for (int f = 0; f < D; ++f) ref[i * D + f] = pass == 2 ? -INFINITY : 0.0f;
        for (int64_t k = 0; k < E; ++k) {
            const float *xr = x + (s[k] - 1) * D;
            float *o = ref + (t[k] - 1) * D;
            for (int f = 0; f < D; ++f) {
                if (pass == 0) o[f] = o[f] + xr[f];
                else if (pass == 1) { volatile float m = w[k] * xr[f]; o[f] = o[f] + m; }
                else o[f] = xr[f] > o[f] ? xr[f] : o[f];
            }
        }
        if (pass == 1)
            for (int64_t i = 0; i < n; ++i) {
                const float cnt = (float)(rowptr[i + 1] - rowptr[i]);
                if (cnt > 0) for (int f = 0; f < D; ++f) ref[i * D + f] = 0.0f + ref[i * D + f] / cnt;
            }
        CHECK_G(gnnmp_propagate_f32(plan, pass == 1 ? GNNMP_W_MUL_XJ : GNNMP_COPY_XJ,
                                    pass == 0 ? GNNMP_SUM : (pass == 1 ? GNNMP_MEAN : GNNMP_MAX), dx, pass == 1 ? dw : NULL, NULL,
                                    NULL, dout, D, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        to_host(got, dout, 4 * n * D);
        for (int64_t i = 0; i < n; ++i) {
            const int len = rowptr[i + 1] - rowptr[i];
            if (len <= thresh || pass == 2)      /* unsplit rows (and max everywhere): bits */
                REQUIRE(memcmp(got + i * D, ref + i * D, 4 * D) == 0, "propagate pass %d row %lld (len %d) not bit-exact", pass, (long long)i, len);
        }
        REQUIRE(pass == 2 || rel_err(got, ref, n * D) <= 1e-5, "propagate pass %d", pass);
    }

    /* ---- dense: Julia (Dout, Din) column-major weight = C [Din][Dout], w_layout = 1, ldw = Dout ---- */
    float *Wjl = malloc(4 * Dout * D), *b = malloc(4 * Dout);
    for (int i = 0; i < Dout * D; ++i) Wjl[i] = 0.2f * rndf();         /* Wjl[k * Dout + j] = W(j, k) */
    for (int j = 0; j < Dout; ++j) b[j] = 0.1f * rndf();
    float *dW = dev_copy(Wjl, 4 * Dout * D), *db = dev_copy(b, 4 * Dout);
    float *dy = dev_alloc(4 * n * Dout), *y = malloc(4 * n * Dout), *yref = malloc(4 * n * Dout);
    /* aggregate of pass 0 (copy_xj, +) as the dense input */
    CHECK_G(gnnmp_propagate_f32(plan, GNNMP_COPY_XJ, GNNMP_SUM, dx, NULL, NULL, NULL, dout, D, stream));
    CHECK_G(gnnmp_dense_f32(dout, dW, D, Dout, NULL, NULL, 0, 0, 1, db, GNNMP_ACT_RELU, dy, n, Dout, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    to_host(got, dout, 4 * n * D);
    to_host(y, dy, 4 * n * Dout);
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < Dout; ++j) {
            double acc = 0;
            for (int k = 0; k < D; ++k) acc += (double)Wjl[k * Dout + j] * got[i * D + k];
            acc += b[j];
            yref[i * Dout + j] = acc < 0 ? 0.0f : (float)acc;
        }
    REQUIRE(rel_err(y, yref, n * Dout) <= 1e-5, "dense rel err %g", rel_err(y, yref, n * Dout));

    /* ---- the same layer in one kernel (force it: the graph is far below the size at which the library fuses by itself) ---- */
    CHECK_G(gnnmp_tune(14, 16));
    float *dy2 = dev_alloc(4 * n * Dout), *dagg = dev_alloc(4 * n * D), *y2 = malloc(4 * n * Dout), *agg = malloc(4 * n * D);
    CHECK_G(gnnmp_fused_conv_f32(plan, GNNMP_SUM, dx, NULL, NULL, NULL, NULL, NULL, D, NULL, 0, NULL, 0, dW, Dout, 1, db,
                                 GNNMP_ACT_RELU, dy2, Dout, dagg, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_G(gnnmp_tune(14, 0));
    to_host(y2, dy2, 4 * n * Dout);
    to_host(agg, dagg, 4 * n * D);
    REQUIRE(memcmp(agg, got, 4 * n * D) == 0, "fused_conv's aggregate is not bit-identical to propagate's");
    REQUIRE(rel_err(y2, yref, n * Dout) <= 1e-5, "fused_conv rel err %g", rel_err(y2, yref, n * Dout));

    /* ---- gat_conv on a self-looped plan: Wx [n][H*C], a = Julia (2C, H) column-major = [H][2C] ---- */
    gnnmp_graph_t *plan_l = NULL;
    CHECK_G(gnnmp_plan_create(&plan_l, ds, dt, 8, 1, n, n, E, 1, 1, stream));
    const int HC = H * C;
    float *Wx = malloc(4 * n * HC), *a = malloc(4 * H * 2 * C), *bg = malloc(4 * HC);
    for (int64_t i = 0; i < n * HC; ++i) Wx[i] = rndf();
    for (int i = 0; i < H * 2 * C; ++i) a[i] = 0.5f * rndf();
    for (int i = 0; i < HC; ++i) bg[i] = 0.1f * rndf();
    float *dWx = dev_copy(Wx, 4 * n * HC), *da = dev_copy(a, 4 * H * 2 * C), *dbg = dev_copy(bg, 4 * HC), *dgo = dev_alloc(4 * n * HC);
    CHECK_G(gnnmp_gat_conv_f32(plan_l, dWx, NULL, da, 0.2f, dbg, GNNMP_ACT_RELU, dgo, H, C, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    float *go = malloc(4 * n * HC), *gref = malloc(4 * n * HC);
    to_host(go, dgo, 4 * n * HC);
    for (int64_t i = 0; i < n; ++i)
        for (int h = 0; h < H; ++h) {
            /* edges into i in original order, then the self loop (transform.jl:12-28) */
            double sd = 0, mx = -INFINITY, den = 0, acc[64] = {0};
            for (int c = 0; c < C; ++c) sd += (double)a[h * 2 * C + c] * Wx[i * HC + h * C + c];
            for (int rep = 0; rep < 2; ++rep)
                for (int64_t k = 0; k <= E; ++k) {
                    int64_t j;
                    if (k < E) { if (t[k] != i + 1) continue; j = s[k] - 1; } else j = i;
                    double ss = 0;
                    for (int c = 0; c < C; ++c) ss += (double)a[h * 2 * C + C + c] * Wx[j * HC + h * C + c];
                    double l = sd + ss;
                    l = l > 0 ? l : 0.2 * l;
                    if (rep == 0) { if (l > mx) mx = l; }
                    else {
                        const double p = exp(l - mx);
                        den += p;
                        for (int c = 0; c < C; ++c) acc[c] += p * Wx[j * HC + h * C + c];
                    }
                }
            for (int c = 0; c < C; ++c) {
                const double v = acc[c] / den + bg[h * C + c];
                gref[i * HC + h * C + c] = v < 0 ? 0.0f : (float)v;
            }
        }
    REQUIRE(rel_err(go, gref, n * HC) <= 1e-5, "gat_conv rel err %g", rel_err(go, gref, n * HC));

    /* ---- error contract: an index outside 1..n is refused with GNNMP_EBOUNDS and a message (convert.jl:47-54) ---- */
    int64_t bad = n + 1;
    CHECK_HIP(hipMemcpy((char *)ds + 8 * 5, &bad, 8, hipMemcpyHostToDevice));
    gnnmp_graph_t *pbad = NULL;
    const int st = gnnmp_plan_create(&pbad, ds, dt, 8, 1, n, n, E, 0, 1, stream);
    REQUIRE(st == GNNMP_EBOUNDS && pbad == NULL && strlen(gnnmp_last_error()) > 0, "out-of-range index: status %d", st);

    CHECK_G(gnnmp_plan_destroy(plan));
    CHECK_G(gnnmp_plan_destroy(plan_l));
    CHECK_HIP(hipStreamDestroy(stream));
    printf("C_HARNESS_OK n=%lld E=%lld split_threshold=%d\n", (long long)n, (long long)E, thresh);
    return 0;
}
