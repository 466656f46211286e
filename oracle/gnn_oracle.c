/*
 * gnn_oracle.c — CPU restatement of the reference's message-passing hot path.  TEST INFRASTRUCTURE ONLY:
 * nothing under graphneuralnetworks.jl_amd/ may import, link or call this; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do (as the checker / the reported CPU baseline, never as the product).
 *
 * The reference (GraphNeuralNetworks.jl @ 2024-12-18) is 100 % Julia and Julia is not installed here, so the
 * reference cannot be executed in this container (no oracle/_ref).  Its arithmetic for this path lives in
 * un-vendored third-party code — NNlib.jl (compat "0.9", GNNlib/Project.toml:38; exact patch version unpinned: no
 * Manifest.toml is committed) gather/scatter, and Julia 1.10's SparseArrays `sparse` + dense*CSC `*` — which this
 * file restates from their published algorithms, anchored on the reference's own call sites (cited per function).
 *
 * PARITY PIN STATUS: pinned by the reference's own known-answer tests (tests/test_oracle_reference_pins.py encodes
 * each with its file:line): GCNConv closed form (GraphNeuralNetworks/test/layers/conv.jl:30-44), degree
 * (GNNGraphs/test/query.jl:49-87), add_self_loops adjacency (GNNGraphs/test/transform.jl:1-17), batch
 * (GNNGraphs/test/transform.jl:29-39), conv_weight zeros (conv.jl:55-65), softmax_edge_neighbors
 * (GNNlib/test/utils.jl:58-67), propagate == X*Adj (GNNlib/test/msgpass.jl:69-116), reduce_nodes mean
 * (GNNlib/test/utils.jl:13-20), and the EXACT assertion of the reference's micro-benchmark, isequal(propagate(e_mul_xj, g, +; xj = B,
 * e), B * A) in Float64 (GraphNeuralNetworks/perf/bench_gnn.jl:38-40) — the one place where the reference itself fixes the summation order
 * of scatter(+): edge order from zero, product rounded first.  PARITY UNPINNED for: the summation ORDER beyond that assertion (restated as
 * NNlib's CPU loop: sequential in edge order), the value of empty destinations under max/min (-Inf/+Inf = NNlib's identity fill) and
 * under mean (0) — no reference test asserts these (risk R1).
 *
 * Layout: Julia column-major (D, N) == C row-major [N][D].  Index arrays are int64, 1-based, exactly as Julia holds
 * them.  Compile with -ffp-contract=off: the reference materialises every product before adding.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_SUM = 0, ORC_MEAN = 1, ORC_MAX = 2, ORC_MIN = 3 };

/* Julia Base.max / Base.min for floats: NaN-propagating; max(-0.0, 0.0) = 0.0; min(0.0, -0.0) = -0.0 */
static float jl_max(float x, float y) {
    if (x != x) return x;
    if (y != y) return y;
    if (y > x) return y;
    if (x > y) return x;
    return signbit(x) ? y : x;
}
static float jl_min(float x, float y) {
    if (x != x) return x;
    if (y != y) return y;
    if (y < x) return y;
    if (x < y) return x;
    return signbit(x) ? x : y;
}

/* NNlib.gather(x, idx): out[:, k] = x[:, idx[k]]        — GNNGraphs/src/gatherscatter.jl:4 */
int orc_gather(const float *x, int64_t n, int64_t D, const int64_t *idx, int64_t K, float *out) {
    for (int64_t k = 0; k < K; ++k) {
        int64_t r = idx[k] - 1;
        if (r < 0 || r >= n) return -1;
        memcpy(out + k * D, x + r * D, sizeof(float) * (size_t)D);
    }
    return 0;
}

/* NNlib.scatter(aggr, src, idx; dstsize = (D, n))        — GNNGraphs/src/gatherscatter.jl:12-18
 * dst filled with the identity (+ -> 0, max -> typemin = -Inf, min -> typemax = +Inf, mean -> 0), then
 * for k = 1..K in order: dst[:, idx[k]] = op(dst[:, idx[k]], src[:, k]).
 * mean: Ns = scatter(+, ones); dst_ = scatter(+, src); dst = 0 .+ safe_div.(dst_, Ns), safe_div(x, 0) = x. */
int orc_scatter(int aggr, const float *src, int64_t D, const int64_t *idx, int64_t K, int64_t n,
                float *out) {
    const float init = aggr == ORC_MAX ? -INFINITY : (aggr == ORC_MIN ? INFINITY : 0.0f);
    for (int64_t i = 0; i < n * D; ++i) out[i] = init;
    int64_t *cnt = NULL;
    if (aggr == ORC_MEAN) {
        cnt = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
        if (!cnt) return -2;
    }
    for (int64_t k = 0; k < K; ++k) {
        int64_t r = idx[k] - 1;
        if (r < 0 || r >= n) {
            free(cnt);
            return -1;
        }
        float *d = out + r * D;
        const float *s = src + k * D;
        if (aggr == ORC_SUM || aggr == ORC_MEAN) {
            for (int64_t f = 0; f < D; ++f) d[f] = d[f] + s[f];
        } else if (aggr == ORC_MAX) {
            for (int64_t f = 0; f < D; ++f) d[f] = jl_max(d[f], s[f]);
        } else {
            for (int64_t f = 0; f < D; ++f) d[f] = jl_min(d[f], s[f]);
        }
        if (cnt) cnt[r] += 1;
    }
    if (cnt) {
        for (int64_t r = 0; r < n; ++r) {
            float c = (float)cnt[r];
            float *d = out + r * D;
            for (int64_t f = 0; f < D; ++f) d[f] = 0.0f + (cnt[r] == 0 ? d[f] : d[f] / c);
        }
        free(cnt);
    }
    return 0;
}

/* degree(g, Float32; dir = :in | :out, edge_weight)       — GNNGraphs/src/query.jl:355-369
 * zeros(T, n) .+ scatter(+, w_or_ones, idx; dstsize = (n,)) */
int orc_degree(const int64_t *idx, const float *w, int64_t E, int64_t n, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = 0.0f;
    for (int64_t k = 0; k < E; ++k) {
        int64_t r = idx[k] - 1;
        if (r < 0 || r >= n) return -1;
        out[r] = out[r] + (w ? w[k] : 1.0f);
    }
    for (int64_t i = 0; i < n; ++i) out[i] = 0.0f + out[i];
    return 0;
}

/* add_self_loops(g::GNNGraph{COO})                         — GNNGraphs/src/transform.jl:12-28
 * s = [s; 1:n], t = [t; 1:n], w = [w; ones(n)] (if weighted); never de-duplicates. */
void orc_add_self_loops(const int64_t *s, const int64_t *t, const float *w, int64_t E, int64_t n,
                        int64_t *s2, int64_t *t2, float *w2) {
    memcpy(s2, s, sizeof(int64_t) * (size_t)E);
    memcpy(t2, t, sizeof(int64_t) * (size_t)E);
    if (w && w2) memcpy(w2, w, sizeof(float) * (size_t)E);
    for (int64_t i = 0; i < n; ++i) {
        s2[E + i] = i + 1;
        t2[E + i] = i + 1;
        if (w && w2) w2[E + i] = 1.0f;
    }
}

/* MLUtils.batch(::Vector{GNNGraph{COO}}) index part        — GNNGraphs/src/transform.jl:682-709
 * member graph g has edges [edge_ptr[g], edge_ptr[g+1]) and nodes [node_ptr[g], node_ptr[g+1]). */
void orc_batch(const int64_t *s, const int64_t *t, const int64_t *edge_ptr, const int64_t *node_ptr,
               int64_t G, int64_t *s2, int64_t *t2, int64_t *gi) {
    for (int64_t g = 0; g < G; ++g) {
        for (int64_t k = edge_ptr[g]; k < edge_ptr[g + 1]; ++k) {
            s2[k] = s[k] + node_ptr[g];
            t2[k] = t[k] + node_ptr[g];
        }
        for (int64_t v = node_ptr[g]; v < node_ptr[g + 1]; ++v) gi[v] = g + 1;
    }
}

/* Generic propagate: aggregate_neighbors(g, aggr, apply_edges(f, g, xi, xj, e))   — GNNlib/src/msgpass.jl:71-79
 * with f = copy_xj (:162) or w_mul_xj / e_mul_xj with vector e (:191-208), materialising the (D, E) message array
 * exactly as the reference does: gather (:125-126), broadcast multiply, scatter (:148). */
int orc_propagate(int aggr, const int64_t *s, const int64_t *t, int64_t E, int64_t n_src,
                  int64_t n_dst, const float *xj, int64_t D, const float *w, float *out) {
    float *m = (float *)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1) * (size_t)(D > 0 ? D : 1));
    if (!m) return -2;
    int rc = orc_gather(xj, n_src, D, s, E, m);
    if (rc == 0 && w) {
        for (int64_t k = 0; k < E; ++k)
            for (int64_t f = 0; f < D; ++f) m[k * D + f] = w[k] * m[k * D + f];
    }
    if (rc == 0) rc = orc_scatter(aggr, m, D, t, E, n_dst, out);
    free(m);
    return rc;
}

/* ---- the same three loops for Float64 features (round 6: the reference's message passing is eltype-generic; its micro-benchmark
 * GraphNeuralNetworks/perf/bench_gnn.jl:9-27 runs propagate(e_mul_xj, g, +) on `rand(100, n)` — Float64 — and asserts isequal with B * A) */
static double jl_max64(double x, double y) {
    if (x != x) return x;
    if (y != y) return y;
    if (y > x) return y;
    if (x > y) return x;
    return signbit(x) ? y : x;
}
static double jl_min64(double x, double y) {
    if (x != x) return x;
    if (y != y) return y;
    if (y < x) return y;
    if (x < y) return x;
    return signbit(x) ? x : y;
}
int orc_gather64(const double *x, int64_t n, int64_t D, const int64_t *idx, int64_t K, double *out) {
    for (int64_t k = 0; k < K; ++k) {
        int64_t r = idx[k] - 1;
        if (r < 0 || r >= n) return -1;
        memcpy(out + k * D, x + r * D, sizeof(double) * (size_t)D);
    }
    return 0;
}
int orc_scatter64(int aggr, const double *src, int64_t D, const int64_t *idx, int64_t K, int64_t n, double *out) {
    const double init = aggr == ORC_MAX ? -INFINITY : (aggr == ORC_MIN ? INFINITY : 0.0);
    for (int64_t i = 0; i < n * D; ++i) out[i] = init;
    int64_t *cnt = NULL;
    if (aggr == ORC_MEAN) {
        cnt = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
        if (!cnt) return -2;
    }
    for (int64_t k = 0; k < K; ++k) {
        int64_t r = idx[k] - 1;
        if (r < 0 || r >= n) {
            free(cnt);
            return -1;
        }
        double *d = out + r * D;
        const double *s = src + k * D;
        if (aggr == ORC_SUM || aggr == ORC_MEAN) {
            for (int64_t f = 0; f < D; ++f) d[f] = d[f] + s[f];
        } else if (aggr == ORC_MAX) {
            for (int64_t f = 0; f < D; ++f) d[f] = jl_max64(d[f], s[f]);
        } else {
            for (int64_t f = 0; f < D; ++f) d[f] = jl_min64(d[f], s[f]);
        }
        if (cnt) cnt[r] += 1;
    }
    if (cnt) {
        for (int64_t r = 0; r < n; ++r) {
            double c = (double)cnt[r];
            double *d = out + r * D;
            for (int64_t f = 0; f < D; ++f) d[f] = 0.0 + (cnt[r] == 0 ? d[f] : d[f] / c);
        }
        free(cnt);
    }
    return 0;
}
int orc_propagate64(int aggr, const int64_t *s, const int64_t *t, int64_t E, int64_t n_src, int64_t n_dst, const double *xj, int64_t D,
                    const double *w, double *out) {
    double *m = (double *)malloc(sizeof(double) * (size_t)(E > 0 ? E : 1) * (size_t)(D > 0 ? D : 1));
    if (!m) return -2;
    int rc = orc_gather64(xj, n_src, D, s, E, m);
    if (rc == 0 && w) {
        for (int64_t k = 0; k < E; ++k)
            for (int64_t f = 0; f < D; ++f) m[k * D + f] = w[k] * m[k * D + f];
    }
    if (rc == 0) rc = orc_scatter64(aggr, m, D, t, E, n_dst, out);
    free(m);
    return rc;
}

/* The CPU fast path: propagate(copy_xj | w_mul_xj | e_mul_xj, g, +) = xj * adjacency_matrix(g)
 *   — GNNlib/src/msgpass.jl:215-238 -> GNNGraphs/src/query.jl:220-231 -> convert.jl:221-237
 * A = sparse(s, t, val, n, n): CSC (column = destination), row indices ascending inside a column, duplicate (s,t)
 * entries SUMMED (Julia 1.10 SparseArrays.sparse default combine = +; for val = ones(Int) that is the multiplicity).
 * Product (SparseArrays mul!(C, X::Dense, A::CSC)): for col, for k in nzrange(col): C[:, col] += X[:, row_k] * val_k,
 * multiply and add rounded separately. */
typedef struct {
    int64_t t, s, k;
} orc_triple;
static int cmp_triple(const void *a, const void *b) {
    const orc_triple *x = (const orc_triple *)a, *y = (const orc_triple *)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    return x->k < y->k ? -1 : (x->k > y->k ? 1 : 0);
}
int orc_spmm_csc(const int64_t *s, const int64_t *t, const float *w, int64_t E, int64_t n,
                 const float *x, int64_t D, float *out) {
    orc_triple *tr = (orc_triple *)malloc(sizeof(orc_triple) * (size_t)(E > 0 ? E : 1));
    if (!tr) return -2;
    for (int64_t k = 0; k < E; ++k) {
        if (s[k] < 1 || s[k] > n || t[k] < 1 || t[k] > n) {
            free(tr);
            return -1;
        }
        tr[k].t = t[k];
        tr[k].s = s[k];
        tr[k].k = k;
    }
    qsort(tr, (size_t)E, sizeof(orc_triple), cmp_triple);
    for (int64_t i = 0; i < n * D; ++i) out[i] = 0.0f;
    int64_t k = 0;
    while (k < E) {
        /* combine duplicates of (s, t) in original order: val = ((v1 + v2) + v3) ... */
        int64_t j = k;
        float val = w ? w[tr[k].k] : 1.0f;
        while (j + 1 < E && tr[j + 1].t == tr[k].t && tr[j + 1].s == tr[k].s) {
            ++j;
            val = val + (w ? w[tr[j].k] : 1.0f);
        }
        float *d = out + (tr[k].t - 1) * D;
        const float *xr = x + (tr[k].s - 1) * D;
        for (int64_t f = 0; f < D; ++f) d[f] = d[f] + xr[f] * val;
        k = j + 1;
    }
    free(tr);
    return 0;
}

/* softmax_edge_neighbors(g, e)                             — GNNlib/src/utils.jl:84-97
 *   max_ = gather(scatter(max, e, t), t); num = exp.(e .- max_); den = gather(scatter(+, num, t), t); num ./ den
 * e and out are [E][H]. */
int orc_softmax_edge_neighbors(const int64_t *t, int64_t E, int64_t n, const float *e, int64_t H,
                               float *out) {
    size_t nh = (size_t)(n > 0 ? n : 1) * (size_t)H, eh = (size_t)(E > 0 ? E : 1) * (size_t)H;
    float *mx = (float *)malloc(sizeof(float) * nh);
    float *g = (float *)malloc(sizeof(float) * eh);
    float *num = (float *)malloc(sizeof(float) * eh);
    float *den = (float *)malloc(sizeof(float) * nh);
    int rc = (!mx || !g || !num || !den) ? -2 : 0;
    if (rc == 0) rc = orc_scatter(ORC_MAX, e, H, t, E, n, mx);
    if (rc == 0) rc = orc_gather(mx, n, H, t, E, g);
    if (rc == 0) {
        for (int64_t i = 0; i < E * H; ++i) num[i] = expf(e[i] - g[i]);
        rc = orc_scatter(ORC_SUM, num, H, t, E, n, den);
    }
    if (rc == 0) rc = orc_gather(den, n, H, t, E, g);
    if (rc == 0)
        for (int64_t i = 0; i < E * H; ++i) out[i] = num[i] / g[i];
    free(mx);
    free(g);
    free(num);
    free(den);
    return rc;
}

/* gat_message logits                                        — GNNlib/src/layers/conv.jl:152-167
 *   Wxx = vcat(Wxi, Wxj)  (2C, H, E);  aWW = sum(l.a .* Wxx, dims = 1);  logα = leakyrelu.(aWW, slope)
 * Wxi, Wxj are the edge-materialised [E][H][C] arrays; a is [H][2C] (Julia (2C, H)); the 2C products are rounded,
 * then summed sequentially c = 1..2C (Julia's dims=1 reduction order; its @simd may reassociate — unpinned). */
void orc_gat_logits(const float *Wxi, const float *Wxj, const float *a, int64_t E, int64_t H,
                    int64_t C, float slope, float *logit) {
    for (int64_t k = 0; k < E; ++k)
        for (int64_t h = 0; h < H; ++h) {
            const float *xi = Wxi + (k * H + h) * C, *xj = Wxj + (k * H + h) * C;
            const float *ah = a + h * 2 * C;
            float acc = 0.0f;
            for (int64_t c = 0; c < C; ++c) acc = acc + ah[c] * xi[c];
            for (int64_t c = 0; c < C; ++c) acc = acc + ah[C + c] * xj[c];
            logit[k * H + h] = acc > 0.0f ? acc : acc * slope; /* NNlib.leakyrelu */
        }
}

/* β = α .* m.Wxj                                            — GNNlib/src/layers/conv.jl:140
 * alpha [E][H] broadcast over the C channels of Wxj [E][H][C]. */
void orc_gat_weight_messages(const float *alpha, const float *Wxj, int64_t E, int64_t H, int64_t C,
                             float *beta) {
    for (int64_t k = 0; k < E * H; ++k)
        for (int64_t c = 0; c < C; ++c) beta[k * C + c] = alpha[k] * Wxj[k * C + c];
}

/* out[n][:] = x[n][:] * c[n]   (`xj .* cout'`, `x .* cin'`  — GNNlib/src/layers/conv.jl:59,67) */
void orc_scale_rows(const float *x, const float *c, int64_t n, int64_t D, float *out) {
    for (int64_t i = 0; i < n; ++i)
        for (int64_t f = 0; f < D; ++f) out[i * D + f] = x[i * D + f] * c[i];
}

/* norm_fn default d -> 1 ./ sqrt.(d)                        — GraphNeuralNetworks/src/layers/conv.jl:99 */
void orc_inv_sqrt(const float *d, int64_t n, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = 1.0f / sqrtf(d[i]);
}

/* y[n][o] = sum_k W[o][k] * x[n][k]  — the dense `weight * x` with a plain k-ordered loop (the reference calls BLAS
 * sgemm, whose blocking/FMA order is unspecified: compared with a tolerance, never bit-wise). */
void orc_matmul(const float *W, const float *x, int64_t N, int64_t Dout, int64_t Din, float *y) {
    for (int64_t n = 0; n < N; ++n)
        for (int64_t o = 0; o < Dout; ++o) {
            float acc = 0.0f;
            const float *wr = W + o * Din, *xr = x + n * Din;
            for (int64_t k = 0; k < Din; ++k) acc = acc + wr[k] * xr[k];
            y[n * Dout + o] = acc;
        }
}
