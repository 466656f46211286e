/* harness.c — drives libgnnmp.so through its C ABI from a plain C program: no torch, no Python, device memory owned through
 * hipMalloc.  It makes the calls the Julia extension makes (graphneuralnetworks.jl_amd/julia/GNNlibGnnmpExt.jl), with the
 * argument conventions of a Julia host: 1-based Int64 COO vectors, column-major (D, N) features = row-major [N][D], a
 * (Dout, Din) column-major weight = w_layout 1 with ldw = Dout, `a` (2C, H) column-major = [H][2C].
 *   plan_create(validate) -> plan_info / plan_export -> degree -> propagate(copy_xj | w_mul_xj; +, mean, max) ->
 *   dense -> fused_conv -> gat_conv -> plan_destroy, the EBOUNDS error path, and (round 4) a batch taken from a resident dataset:
 *   plan_select == plan_create on the concatenated COO, plan_edge_index, the node map, chain_jobs_pack / export, plan_release;
 *   plan_from_csc == plan_create on findnz(A) for a sparse-matrix graph; (round 5) the fused steps of the chain's pullback:
 *   dense_grad_w2 == two dense_grad_w calls, pool_grad_act and propagate_add_mask against host loops.
 * Expected values come from plain host loops in this file (edge order, separately rounded products: bit-exact where the
 * library promises bits).  Prints C_HARNESS_OK and exits 0 on success.   Built by __graft_entry__.build() / tests. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnnmp.h"


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

    /* ---- the same propagate for Float64 features (gnnmp_propagate_f64: round 6), w_mul_xj + against the host loop in edge order ---- */
    {
        double *x64 = malloc(8 * n * D), *w64 = malloc(8 * E), *ref64 = calloc(n * D, 8), *got64 = malloc(8 * n * D);
        for (int64_t i = 0; i < n * D; ++i) x64[i] = (double)x[i] + 1e-9 * (double)rndf();      /* values that are NOT representable in fp32 */
        for (int64_t k = 0; k < E; ++k) w64[k] = (double)w[k];
        for (int64_t k = 0; k < E; ++k)
            for (int f = 0; f < D; ++f) {
                volatile double m = w64[k] * x64[(s[k] - 1) * D + f];
                ref64[(t[k] - 1) * D + f] = ref64[(t[k] - 1) * D + f] + m;
            }
        double *dx64 = dev_copy(x64, 8 * n * D), *dw64 = dev_copy(w64, 8 * E), *do64 = dev_alloc(8 * n * D);
        CHECK_G(gnnmp_propagate_f64(plan, GNNMP_W_MUL_XJ, GNNMP_SUM, dx64, dw64, NULL, NULL, do64, D, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        to_host(got64, do64, 8 * n * D);
        double worst = 0.0, scale = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const int len = rowptr[i + 1] - rowptr[i];
            if (len <= thresh) REQUIRE(memcmp(got64 + i * D, ref64 + i * D, 8 * D) == 0, "propagate_f64 row %lld (len %d) not bit-exact", (long long)i, len);
            for (int f = 0; f < D; ++f) {
                const double d = fabs(got64[i * D + f] - ref64[i * D + f]), a = fabs(ref64[i * D + f]);
                if (d > worst) worst = d;
                if (a > scale) scale = a;
            }
        }
        REQUIRE(worst <= 1e-12 * scale, "propagate_f64 split rows: %g of %g", worst, scale);
        CHECK_HIP(hipFree(dx64)); CHECK_HIP(hipFree(dw64)); CHECK_HIP(hipFree(do64));
        free(x64); free(w64); free(ref64); free(got64);
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

    /* ---- a new batch every step (round 4): a dataset of member graphs batched ONCE (MLUtils.batch, transform.jl:682-709), then
     * batch(gs[ids]) as a selection — gnnmp_plan_select must equal gnnmp_plan_create on the COO a host would have concatenated; the
     * features come through the node map; the fused chain's wave jobs are packed on the device ---- */
    {
        enum { GM = 40, KSEL = 25 };
        int64_t mn[GM], me[GM], nptr[GM + 1], eptr[GM + 1];
        nptr[0] = eptr[0] = 0;
        for (int g = 0; g < GM; ++g) {
            mn[g] = 1 + rnd() % 40;
            me[g] = rnd() % (4 * mn[g] + 1);
            nptr[g + 1] = nptr[g] + mn[g];
            eptr[g + 1] = eptr[g] + me[g];
        }
        const int64_t Nd = nptr[GM], Ed = eptr[GM];
        int64_t *S = malloc(8 * (Ed + 1)), *T = malloc(8 * (Ed + 1));          /* the dataset's batched COO, 1-based */
        for (int g = 0; g < GM; ++g)
            for (int64_t k = eptr[g]; k < eptr[g + 1]; ++k) {
                S[k] = nptr[g] + 1 + rnd() % mn[g];
                T[k] = nptr[g] + 1 + rnd() % mn[g];
            }
        float *X = malloc(4 * Nd * 3);
        for (int64_t i = 0; i < Nd * 3; ++i) X[i] = rndf();
        int64_t *dS = dev_copy(S, 8 * Ed), *dT = dev_copy(T, 8 * Ed), *dnptr = dev_copy(nptr, 8 * (GM + 1));
        float *dX = dev_copy(X, 4 * Nd * 3);
        gnnmp_graph_t *pds = NULL;
        CHECK_G(gnnmp_plan_create(&pds, dS, dT, 8, 1, Nd, Nd, Ed, 0, 1, stream));
        int64_t ids[KSEL], nb = 0, eb = 0, mx = 0;
        for (int k = 0; k < KSEL; ++k) { ids[k] = 1 + rnd() % GM; nb += mn[ids[k] - 1]; eb += me[ids[k] - 1]; if (mn[ids[k] - 1] > mx) mx = mn[ids[k] - 1]; }
        int64_t *dids = dev_copy(ids, 8 * KSEL), *dseg = dev_alloc(8 * (KSEL + 1)), *dgi = dev_alloc(8 * nb);
        int32_t *dnmap = dev_alloc(4 * nb);
        gnnmp_graph_t *pb = NULL;
        CHECK_G(gnnmp_plan_select(&pb, pds, dnptr, GM, dids, 8, 1, KSEL, nb, eb, dseg, dnmap, dgi, stream));
        CHECK_G(gnnmp_plan_status(pb, stream));
        /* what MLUtils.batch(gs[ids]) holds: members in the order of ids, local ids shifted by the nodes before them */
        int64_t *Sb = malloc(8 * (eb + 1)), *Tb = malloc(8 * (eb + 1)), *seg = malloc(8 * (KSEL + 1)), at = 0, ro = 0;
        for (int k = 0; k < KSEL; ++k) {
            const int g = (int)ids[k] - 1;
            seg[k] = ro;
            for (int64_t q = eptr[g]; q < eptr[g + 1]; ++q, ++at) { Sb[at] = S[q] - nptr[g] + ro; Tb[at] = T[q] - nptr[g] + ro; }
            ro += mn[g];
        }
        seg[KSEL] = ro;
        int64_t *dSb = dev_copy(Sb, 8 * eb), *dTb = dev_copy(Tb, 8 * eb);
        gnnmp_graph_t *pr = NULL;
        CHECK_G(gnnmp_plan_create(&pr, dSb, dTb, 8, 1, nb, nb, eb, 0, 1, stream));
        int32_t *e1[3], *e2[3];
        for (int q = 0; q < 3; ++q) { e1[q] = dev_alloc(4 * (nb + eb + 1)); e2[q] = dev_alloc(4 * (nb + eb + 1)); }
        CHECK_G(gnnmp_plan_export(pb, e1[0], e1[1], e1[2], stream));
        CHECK_G(gnnmp_plan_export(pr, e2[0], e2[1], e2[2], stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        const size_t lens[3] = {(size_t)nb + 1, (size_t)eb, (size_t)eb};
        for (int q = 0; q < 3; ++q) {
            int32_t *h1 = malloc(4 * lens[q] + 4), *h2 = malloc(4 * lens[q] + 4);
            to_host(h1, e1[q], 4 * lens[q]); to_host(h2, e2[q], 4 * lens[q]);
            REQUIRE(memcmp(h1, h2, 4 * lens[q]) == 0, "plan_select: array %d differs from plan_create on the batched COO", q);
            free(h1); free(h2);
        }
        int64_t *hseg = malloc(8 * (KSEL + 1));
        to_host(hseg, dseg, 8 * (KSEL + 1));
        REQUIRE(memcmp(hseg, seg, 8 * (KSEL + 1)) == 0, "plan_select: seg_ptr");
        /* s, t back from the plan = the batched COO */
        int64_t *dS2 = dev_alloc(8 * eb), *dT2 = dev_alloc(8 * eb), *hS2 = malloc(8 * (eb + 1)), *hT2 = malloc(8 * (eb + 1));
        CHECK_G(gnnmp_plan_edge_index(pb, 8, 1, dS2, dT2, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        to_host(hS2, dS2, 8 * eb); to_host(hT2, dT2, 8 * eb);
        REQUIRE(memcmp(hS2, Sb, 8 * eb) == 0 && memcmp(hT2, Tb, 8 * eb) == 0, "plan_edge_index: the batch's COO");
        /* features through the node map */
        float *dXb = dev_alloc(4 * nb * 3), *hXb = malloc(4 * nb * 3);
        CHECK_G(gnnmp_gather_f32(dX, dnmap, 4, 0, nb, dXb, 3, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        to_host(hXb, dXb, 4 * nb * 3);
        for (int k = 0; k < KSEL; ++k)
            REQUIRE(memcmp(hXb + 3 * seg[k], X + 3 * nptr[ids[k] - 1], 12 * mn[ids[k] - 1]) == 0, "collated features of member %d", k);
        /* the wave jobs, packed on the device: every row exactly once, member graphs whole */
        gnnmp_chain_jobs_t *jobs = NULL;
        CHECK_G(gnnmp_chain_jobs_pack(&jobs, dseg, KSEL, nb, mx, 0, stream));
        int32_t *dtab = dev_alloc(4 * 64 * KSEL), *dhdr = dev_alloc(4 * 32), hdr[32], *tab = malloc(4 * 64 * KSEL);
        CHECK_HIP(hipMemset(dtab, 0xff, 4 * 64 * KSEL));
        CHECK_G(gnnmp_chain_jobs_export(jobs, dtab, KSEL, dhdr, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        to_host(hdr, dhdr, 4 * 32); to_host(tab, dtab, 4 * 64 * KSEL);
        REQUIRE(hdr[2] == 0 && hdr[0] > 0 && hdr[0] <= KSEL && hdr[5] == KSEL, "chain_jobs_pack: header %d %d %d", hdr[0], hdr[2], hdr[5]);
        int *seen = calloc(nb, sizeof(int));
        for (int j = 0; j < hdr[0]; ++j) {
            int used = 0;
            while (used < 64 && tab[64 * j + used] >= 0) ++used;
            REQUIRE(used > 0, "job %d is empty", j);
            for (int q = used; q < 64; ++q) REQUIRE(tab[64 * j + q] == -1, "job %d has a hole", j);
            for (int q = 0; q < used;) {                      /* a whole member graph at a time */
                const int r0 = tab[64 * j + q];
                int k = 0;
                while (k < KSEL && seg[k] != r0) ++k;
                REQUIRE(k < KSEL, "job %d slot %d does not start a member graph", j, q);
                for (int64_t r = seg[k]; r < seg[k + 1]; ++r, ++q) { REQUIRE(q < used && tab[64 * j + q] == r, "job %d: member %d not whole", j, k); ++seen[r]; }
            }
        }
        for (int64_t r = 0; r < nb; ++r) REQUIRE(seen[r] == 1, "row %lld packed %d times", (long long)r, seen[r]);
        CHECK_G(gnnmp_chain_jobs_release(jobs, stream));
        CHECK_G(gnnmp_plan_release(pb, stream));               /* stream-ordered: back to the pool, no host synchronisation */
        CHECK_G(gnnmp_plan_destroy(pr));
        CHECK_G(gnnmp_plan_destroy(pds));
    }

    /* ---- a sparse-matrix graph (GNNGraph{SPARSE_T}): the CSC arrays of A (A[s, t] != 0 for s -> t) ARE the plan — gnnmp_plan_from_csc
     * must equal gnnmp_plan_create on findnz(A)'s COO, and slot k must be edge k (GNNGraphs/src/query.jl:14, convert.jl:62-73).  A
     * SparseMatrixCSC has no duplicates: the multigraph above is de-duplicated column by column. ---- */
    {
        int64_t *colptr = malloc(8 * (n + 1)), *rowval = malloc(8 * E);
        unsigned char *seen = calloc((size_t)n, 1);
        int64_t nnz = 0;
        for (int64_t j = 0; j < n; ++j) {                    /* column j = sources of j's incoming edges, ascending, once each */
            colptr[j] = nnz + 1;
            memset(seen, 0, (size_t)n);
            for (int64_t p = rowptr[j]; p < rowptr[j + 1]; ++p) seen[col[p]] = 1;
            for (int64_t i = 0; i < n; ++i)
                if (seen[i]) rowval[nnz++] = i + 1;
        }
        colptr[n] = nnz + 1;
        int64_t *fs = malloc(8 * nnz), *ft = malloc(8 * nnz);          /* findnz(A): columns in order */
        for (int64_t j = 0, k = 0; j < n; ++j)
            for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p, ++k) { fs[k] = rowval[p]; ft[k] = j + 1; }
        void *dcp = dev_copy(colptr, 8 * (n + 1)), *drv = dev_copy(rowval, 8 * nnz), *dfs = dev_copy(fs, 8 * nnz), *dft = dev_copy(ft, 8 * nnz);
        gnnmp_graph_t *pc = NULL, *pf = NULL;
        CHECK_G(gnnmp_plan_from_csc(&pc, dcp, drv, 8, 1, n, n, nnz, 1, stream));
        CHECK_G(gnnmp_plan_create(&pf, dfs, dft, 8, 1, n, n, nnz, 0, 1, stream));
        int32_t *h1 = malloc(4 * (nnz + n + 1)), *h2 = malloc(4 * (nnz + n + 1));
        void *d1 = dev_alloc(4 * (n + 1)), *d2 = dev_alloc(4 * nnz), *d3 = dev_alloc(4 * nnz);
        const gnnmp_graph_t *both[2] = {pc, pf};
        int32_t *arr[2][3];
        for (int w2 = 0; w2 < 2; ++w2) {
            CHECK_G(gnnmp_plan_export(both[w2], d1, d2, d3, stream));
            CHECK_HIP(hipStreamSynchronize(stream));
            arr[w2][0] = malloc(4 * (n + 1)); arr[w2][1] = malloc(4 * nnz); arr[w2][2] = malloc(4 * nnz);
            to_host(arr[w2][0], d1, 4 * (n + 1)); to_host(arr[w2][1], d2, 4 * nnz); to_host(arr[w2][2], d3, 4 * nnz);
        }
        REQUIRE(memcmp(arr[0][0], arr[1][0], 4 * (n + 1)) == 0 && memcmp(arr[0][1], arr[1][1], 4 * nnz) == 0 &&
                memcmp(arr[0][2], arr[1][2], 4 * nnz) == 0, "plan_from_csc differs from plan_create on findnz(A)");
        for (int64_t k = 0; k < nnz; ++k) REQUIRE(arr[0][2][k] == k, "plan_from_csc: slot %lld is not edge %lld", (long long)k, (long long)k);
        colptr[3] = colptr[2] - 1;                               /* a decreasing column pointer is refused */
        CHECK_HIP(hipMemcpy(dcp, colptr, 8 * (n + 1), hipMemcpyHostToDevice));
        gnnmp_graph_t *pbadc = NULL;
        const int stc = gnnmp_plan_from_csc(&pbadc, dcp, drv, 8, 1, n, n, nnz, 1, stream);
        REQUIRE(stc == GNNMP_EBOUNDS && pbadc == NULL, "plan_from_csc: bad colptr gave status %d", stc);
        CHECK_G(gnnmp_plan_destroy(pc));
        CHECK_G(gnnmp_plan_destroy(pf));
        (void)h1; (void)h2;
    }

    /* ---- the fused steps of the graph-classification chain's pullback (round 5; examples/graph_classification_tudataset.jl:79-82,97-104):
     * each must give the bits of the separate calls / of a host loop.  Shapes: Δz (n x 64), x1 = 64 columns, x2 = 100 columns. ---- */
    {
        const int Do = 64, K1 = 64, K2 = D, G = 4;
        float *dzh = malloc(4 * n * Do), *x1h = malloc(4 * n * K1);
        for (int64_t i = 0; i < n * Do; ++i) dzh[i] = rndf();
        for (int64_t i = 0; i < n * K1; ++i) x1h[i] = rndf();
        float *ddz = dev_copy(dzh, 4 * n * Do), *dx1 = dev_copy(x1h, 4 * n * K1);
        /* (a) ΔW1, ΔW2, Δb from one read of Δz == two gnnmp_dense_grad_w_f32 calls, bit for bit */
        const int64_t L = (int64_t)Do * (K1 + K2) + Do;
        const int64_t nws2 = gnnmp_dense_grad_w2_workspace(n, Do, K1, K2), nws1 = gnnmp_dense_grad_workspace(n, Do, K2 > K1 ? K2 : K1);
        float *dws2 = dev_alloc(4 * nws2), *dws1 = dev_alloc(4 * nws1), *dout2 = dev_alloc(4 * L);
        float *dW1 = dev_alloc(4 * Do * K1), *dW2 = dev_alloc(4 * Do * K2), *dbb = dev_alloc(4 * Do);
        CHECK_G(gnnmp_dense_grad_w2_f32(ddz, dx1, K1, dx, K2, n, Do, dout2, dws2, nws2, stream));
        CHECK_G(gnnmp_dense_grad_w_f32(ddz, dx1, n, Do, K1, dW1, dbb, dws1, nws1, stream));
        CHECK_G(gnnmp_dense_grad_w_f32(ddz, dx, n, Do, K2, dW2, NULL, dws1, nws1, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        float *o2 = malloc(4 * L), *o1 = malloc(4 * L);
        to_host(o2, dout2, 4 * L);
        to_host(o1, dW1, 4 * Do * K1); to_host(o1 + Do * K1, dW2, 4 * Do * K2); to_host(o1 + Do * (K1 + K2), dbb, 4 * Do);
        REQUIRE(memcmp(o1, o2, 4 * L) == 0, "dense_grad_w2 differs from two dense_grad_w calls");
        double ref00 = 0.0;                                   /* and one element against a host sum: ΔW1[0][0] = Σ_n Δz[n][0] x1[n][0] */
        for (int64_t i = 0; i < n; ++i) ref00 += (double)dzh[i * Do] * (double)x1h[i * K1];
        REQUIRE(fabs(o2[0] - ref00) <= 1e-5 * (fabs(ref00) + 1.0), "dense_grad_w2: dW1[0][0] = %g, host %g", o2[0], ref00);
        int st2 = gnnmp_dense_grad_w2_f32(ddz, dx, K2, dx1, K1, n, Do, dout2, dws2, nws2, stream);      /* K1 = 100: not a multiple of 16 */
        REQUIRE(st2 == GNNMP_EUNSUPPORTED, "dense_grad_w2 with K1 = 100 gave status %d", st2);
        /* (b) Δz = relu'(y) .* Δpool[g(i)] .* inv[g(i)] — bits of the host loop */
        int64_t *gi = malloc(8 * n);
        float *dp = malloc(4 * G * Do), *inv = malloc(4 * G), *yy = malloc(4 * n * Do), *want = malloc(4 * n * Do), *have = malloc(4 * n * Do);
        for (int64_t i = 0; i < n; ++i) gi[i] = 1 + i * G / n;
        for (int i = 0; i < G * Do; ++i) dp[i] = rndf();
        for (int g2 = 0; g2 < G; ++g2) inv[g2] = 1.0f / (float)(3 + g2);
        for (int64_t i = 0; i < n * Do; ++i) yy[i] = rndf();
        for (int64_t i = 0; i < n; ++i)
            for (int d = 0; d < Do; ++d) {
                volatile float r = inv[gi[i] - 1] * dp[(gi[i] - 1) * Do + d];
                want[i * Do + d] = yy[i * Do + d] > 0.0f ? r : 0.0f;
            }
        void *dgi2 = dev_copy(gi, 8 * n);
        float *ddp = dev_copy(dp, 4 * G * Do), *dinv = dev_copy(inv, 4 * G), *dyy = dev_copy(yy, 4 * n * Do), *dhave = dev_alloc(4 * n * Do);
        CHECK_G(gnnmp_pool_grad_act_f32(ddp, dgi2, 8, 1, dinv, dyy, GNNMP_ACT_RELU, dhave, n, G, Do, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        to_host(have, dhave, 4 * n * Do);
        REQUIRE(memcmp(have, want, 4 * n * Do) == 0, "pool_grad_act differs from the host loop");
        /* (c) out = relu'(mask) .* (addend + Σ_{j -> i} xj) on the plan: bits of propagate + add + mask on the unsplit rows */
        float *add = malloc(4 * n * D), *msk = malloc(4 * n * D);
        for (int64_t i = 0; i < n * D; ++i) { add[i] = rndf(); msk[i] = rndf(); }
        float *dadd = dev_copy(add, 4 * n * D), *dmsk = dev_copy(msk, 4 * n * D);
        CHECK_G(gnnmp_propagate_add_mask_f32(plan, GNNMP_SUM, dx, NULL, dadd, dmsk, dout, D, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        to_host(got, dout, 4 * n * D);
        for (int64_t i = 0; i < n; ++i)
            for (int f = 0; f < D; ++f) ref[i * D + f] = 0.0f;
        for (int64_t k = 0; k < E; ++k)
            for (int f = 0; f < D; ++f) ref[(t[k] - 1) * D + f] = ref[(t[k] - 1) * D + f] + x[(s[k] - 1) * D + f];
        for (int64_t i = 0; i < n; ++i) {
            const int len = rowptr[i + 1] - rowptr[i];
            for (int f = 0; f < D; ++f) {
                volatile float v = add[i * D + f] + ref[i * D + f];
                ref[i * D + f] = msk[i * D + f] > 0.0f ? v : 0.0f;
            }
            if (len <= thresh) REQUIRE(memcmp(got + i * D, ref + i * D, 4 * D) == 0, "propagate_add_mask row %lld not bit-exact", (long long)i);
        }
        REQUIRE(rel_err(got, ref, n * D) <= 1e-5, "propagate_add_mask");
    }

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
