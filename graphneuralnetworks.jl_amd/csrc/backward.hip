// backward.hip — adjoints of the hot path ("next" after the forward bar).
//
// What training through the reference runs today is Zygote over NNlib's rrules:
//   ∇gather(x, idx)            : Δx   = scatter(+, Δ, idx)
//   ∇scatter(+,    src, idx)   : Δsrc = gather(Δ, idx)
//   ∇scatter(mean, src, idx)   : Δsrc = gather(Δ, idx) ./ count[idx]
//   ∇scatter(max|min, src, idx): Δsrc = (src .== gather(dst, idx)) .* gather(Δ, idx)
// and the `adjacency_matrix` rrule of the CPU fast path (GNNGraphs/src/query.jl:244-278).
//
// For the FUSED propagate  out_i = sd_i * Σ_{k: t_k = i} w_k * ss_{s_k} * x_{s_k}   the adjoint w.r.t. x is the SAME
// kernel on the TRANSPOSED plan (a plan built from (t, s)):
//     Δx_j = ss_j * Σ_{k: s_k = j} w_k * sd_{t_k} * Δ_{t_k}
// i.e. gnnmp_propagate_f32(plan_T, msg, SUM, Δ, w, scale_src = sd (or sd / count for mean), scale_dst = ss): no new
// kernel, same edge order as NNlib's scatter(+, gather(Δ, t) .* w, s), hence the same bits.  What does need kernels:
//   - the adjoint w.r.t. the edge weights,  Δw_k = Δ_{t_k} · (ss_{s_k} x_{s_k}) * sd_{t_k}  — an edge-wise dot product
//     of two gathered rows (SDDMM): edge_dot_kernel;
//   - the max / min adjoint, which compares the forward input with the forward output per feature: maxmin_grad_kernel
//     on the transposed plan.
#include "common.h"

namespace gnnmp {

// out[k] = Σ_d a[dst_k][d] * b[src_k][d]; one group of G lanes per edge, lanes stride the feature dimension.
template <int VEC>
__global__ void __launch_bounds__(256) edge_dot_kernel(const float *a, const float *b, const void *src,
                                                       const void *dst, int idx_bytes, int base, int64_t E,
                                                       int D, int log2g, float *out) {
    const int G = 1 << log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> log2g;
    if (k >= E) return;
    const int64_t i = load_index(dst, k, idx_bytes, base);
    const int64_t j = load_index(src, k, idx_bytes, base);
    const float *ra = a + i * D, *rb = b + j * D;
    float acc = 0.0f;
    for (int f = lig * VEC; f < D; f += G * VEC) {
        float va[VEC], vb[VEC];
        Vec<VEC>::load(ra + f, va);
        Vec<VEC>::load(rb + f, vb);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc = fmaf(va[q], vb[q], acc);
    }
    for (int o = 1; o < G; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (lig == 0) out[k] = acc;
}

// The same dot products in the plan's destination-sorted order: the destination row a[i] is loaded once per row (per
// chunk for split rows) and stays in registers, only b[col_p] is gathered per edge — half the traffic of the COO-order
// kernel above (measured 12.0 -> 6.6 ms, LABNOTES.md §7).  Results are written back in ORIGINAL edge order through eid.
struct EdgeDotRowsArgs {
    const uint32_t *rowptr;
    const int32_t *col, *eid;
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    const float *a, *b;
    float *out;
    int n_chunks, long_thresh;
    uint32_t n_edges;
    int D, n_rows, log2g, waves;
};
// U dot products' partial sums, one set per lane of a G-lane group -> lane l of the group gets the TOTAL of product (l mod U): a
// transposing butterfly.  At step s a lane keeps the products whose index has bit s equal to ITS bit s and hands the others to lane ^ 2^s:
// U/2 + U/4 + ... + 1 exchanges, then log2(G / U) plain steps on the one value left — 9 exchanges for 8 products over 32 lanes where a
// butterfly per product costs 40.  (Measured at the products size: 6.7 -> 6.6 ms.  The kernel is bound by its 4-byte stores in original
// edge order — a write request each — not by the reductions; the gather alone needs 4.5 ms.)
template <int U, int G>
__device__ __forceinline__ float reduce_transposed(float (&d)[U], int lane) {
    static_assert(U <= G, "one product per lane at most");
#pragma unroll
    for (int s = 0; (U >> s) > 1; ++s) {
        const int o = 1 << s;
        const bool hi = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < (U >> (s + 1)); ++i) {
            const float keep = hi ? d[2 * i + 1] : d[2 * i];
            const float send = hi ? d[2 * i] : d[2 * i + 1];
            d[i] = keep + __shfl_xor(send, o, 64);
        }
    }
    float r = d[0];
#pragma unroll
    for (int o = U; o < G; o <<= 1) r += __shfl_xor(r, o, 64);
    return r;
}

template <int VEC, int LOG2G>
__global__ void __launch_bounds__(256) edge_dot_rows_kernel(const EdgeDotRowsArgs a) {
    constexpr int G = 1 << LOG2G;
    constexpr int U = G < 8 ? G : 8;               // products per batch: loads in flight per lane, and the transposed reduction's width
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lig = lane & (G - 1);
    const int grp = lane >> LOG2G;
    const int gbase = lane - lig;
    constexpr int rpw = 64 >> LOG2G;
    const int64_t v64 = ((int64_t)blockIdx.x * a.waves + wave) * rpw + grp;
    if (v64 >= (int64_t)a.n_rows + a.n_chunks) return;
    const int v = (int)v64;
    int row;
    uint32_t beg, end;
    if (v < a.n_chunks) {
        row = a.chunk_row[v];
        beg = a.chunk_beg[v];
        end = a.chunk_end[v];
    } else {
        row = v - a.n_chunks;
        beg = a.rowptr[row];
        end = a.rowptr[row + 1];
        if (end - beg > a.long_thresh) return;   // its chunks are separate virtual rows (outputs are per edge: no combine)
    }
    const int f0 = lig * VEC;                      // one feature tile: G * VEC >= D (checked by the launcher)
    const bool active = f0 < a.D;
    float av[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) av[q] = 0.0f;
    if (active) Vec<VEC>::load(a.a + (int64_t)row * a.D + f0, av);
    for (uint32_t base = beg; base < end; base += G) {
        const uint32_t p = base + lig;
        int c = 0;
        uint32_t e = 0;
        if (p < end) {
            c = a.col[p];
            e = (uint32_t)a.eid[p];
        }
        const int n = (int)min((uint32_t)G, end - base);
        float mine = 0.0f;                         // lane lig keeps the result of slot base + lig
        for (int j = 0; j < n; j += U) {
            float bv[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cj = __shfl(c, gbase + min(j + u, n - 1), 64);
                if (active) {
                    Vec<VEC>::load(a.b + (int64_t)cj * a.D + f0, bv[u]);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) bv[u][q] = 0.0f;
                }
            }
            float d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = 0.0f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) d[u] = fmaf(av[q], bv[u][q], d[u]);
            }
            const float r = reduce_transposed<U, G>(d, lane);      // the total of product j + (lig mod U)
            if ((lig & ~(U - 1)) == j) mine = r;
        }
        if (p < end && e < a.n_edges) a.out[e] = mine;   // plan-added self loops carry no weight: no output slot
    }
}

struct MaxMinGradArgs {
    const uint32_t *rowptr;  // transposed plan: row j = source node, slots = the edges j -> i in original order
    const int32_t *col;     // destination i of each slot
    const float *x;         // [n_src][D] forward input
    const float *y;         // [n_dst][D] forward output (max / min over incoming messages)
    const float *dy;        // [n_dst][D]
    float *dx;              // [n_src][D]
    float *partial;         // [n_chunks][D] (rows of the transposed plan longer than its threshold are split)
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    int n_chunks, long_thresh;
    int D, n_rows, log2g, waves;
};

// Δx_j[d] = Σ_{slots p of row j} (x_j[d] == y_{col_p}[d]) ? Δ_{col_p}[d] : 0     (ties: every maximiser gets Δ, like NNlib)
template <int VEC, int U>
__global__ void __launch_bounds__(256) maxmin_grad_kernel(const MaxMinGradArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = 1 << a.log2g;
    const int lig = lane & (G - 1);
    const int grp = lane >> a.log2g;
    const int gbase = lane - lig;
    const int rpw = 64 >> a.log2g;
    const int64_t v64 = ((int64_t)blockIdx.x * a.waves + wave) * rpw + grp;
    if (v64 >= (int64_t)a.n_rows + a.n_chunks) return;
    const int v = (int)v64;
    const bool is_chunk = v < a.n_chunks;
    int row;
    uint32_t beg, end;
    if (is_chunk) {
        row = a.chunk_row[v];
        beg = a.chunk_beg[v];
        end = a.chunk_end[v];
    } else {
        row = v - a.n_chunks;
        beg = a.rowptr[row];
        end = a.rowptr[row + 1];
        if (end - beg > a.long_thresh) return;   // split row: chunks are virtual rows, folded by csr_combine_kernel
    }
    const int f0 = ((int)blockIdx.y * G + lig) * VEC;
    const bool active = f0 < a.D;
    float xv[VEC], acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) { xv[q] = 0.0f; acc[q] = 0.0f; }
    if (active) Vec<VEC>::load(a.x + (int64_t)row * a.D + f0, xv);
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float yv[U][VEC], dv[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cj = __shfl(c, gbase + min(j + u, n - 1), 64);
                if (active) {   // clamped, unconditional within the lane's activity: no per-element branch + wait
                    Vec<VEC>::load(a.y + (int64_t)cj * a.D + f0, yv[u]);
                    Vec<VEC>::load(a.dy + (int64_t)cj * a.D + f0, dv[u]);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) { yv[u][q] = 0.0f; dv[u][q] = 0.0f; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < n) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[q] = acc[q] + (xv[q] == yv[u][q] ? dv[u][q] : 0.0f);
                }
            }
        }
    }
    if (active) Vec<VEC>::store((is_chunk ? a.partial + (int64_t)v * a.D : a.dx + (int64_t)row * a.D) + f0, acc);
}

int run_combine_sum(gnnmp_graph_t *p, float *out, int64_t D, hipStream_t stream);  // propagate.hip

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_edge_dot_f32(const float *a_dst, const float *b_src, const void *src, const void *dst,
                       int idx_bytes, int index_base, int64_t n_edges, int64_t D, float *out,
                       gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "edge_dot: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "edge_dot: index_base %d", index_base);
    if (n_edges < 0 || D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "edge_dot: bad size");
    if (n_edges == 0) return GNNMP_OK;
    if (!a_dst || !b_src || !src || !dst || !out) return fail(GNNMP_EINVAL, "edge_dot: null pointer");
    const int vec = pick_vec(D, a_dst, b_src);
    const int log2g = pick_log2g((D + vec - 1) / vec);
    const int64_t threads = n_edges << log2g;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    switch (vec) {
        case 4: edge_dot_kernel<4><<<nb, 256, 0, stream>>>(a_dst, b_src, src, dst, idx_bytes, index_base, n_edges, (int)D, log2g, out); break;
        case 2: edge_dot_kernel<2><<<nb, 256, 0, stream>>>(a_dst, b_src, src, dst, idx_bytes, index_base, n_edges, (int)D, log2g, out); break;
        default: edge_dot_kernel<1><<<nb, 256, 0, stream>>>(a_dst, b_src, src, dst, idx_bytes, index_base, n_edges, (int)D, log2g, out); break;
    }
    GNNMP_LAUNCH_CHECK("edge_dot_kernel");
    return GNNMP_OK;
}

int gnnmp_edge_dot_plan_f32(gnnmp_graph_t *plan, const float *a_dst, const float *b_src, float *out,
                            int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!plan) return fail(GNNMP_EINVAL, "edge_dot_plan: null plan");
    if (D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "edge_dot_plan: bad D");
    if (plan->n_total == 0 || plan->n_dst == 0) return GNNMP_OK;
    if (!a_dst || !b_src || !out) return fail(GNNMP_EINVAL, "edge_dot_plan: null pointer");
    const int vec = pick_vec(D, a_dst, b_src);
    const int lanes = (int)((D + vec - 1) / vec);
    if (lanes > 64) return fail(GNNMP_EUNSUPPORTED, "edge_dot_plan: D = %lld needs more than one wave per row; use gnnmp_edge_dot_f32", (long long)D);
    EdgeDotRowsArgs a;
    a.rowptr = plan->rowptr;
    a.col = plan->col;
    a.eid = plan->eid;
    a.chunk_row = plan->chunk_row;
    a.chunk_beg = plan->chunk_beg;
    a.chunk_end = plan->chunk_end;
    a.a = a_dst;
    a.b = b_src;
    a.out = out;
    a.n_chunks = plan->n_chunks;
    a.long_thresh = plan->long_thresh;
    a.n_edges = (uint32_t)plan->n_edges;
    a.D = (int)D;
    a.n_rows = (int)plan->n_dst;
    a.log2g = 0;
    while ((1 << a.log2g) < lanes) ++a.log2g;
    a.waves = 4;
    const int rows_per_block = (64 >> a.log2g) * a.waves;
    const unsigned nb = (unsigned)(((int64_t)a.n_rows + a.n_chunks + rows_per_block - 1) / rows_per_block);
#define EDR(V, LG) edge_dot_rows_kernel<V, LG><<<nb, 256, 0, stream>>>(a)
#define EDR_LG(V)                                                                                                       \
    switch (a.log2g) {                                                                                                  \
        case 0: EDR(V, 0); break; case 1: EDR(V, 1); break; case 2: EDR(V, 2); break; case 3: EDR(V, 3); break;          \
        case 4: EDR(V, 4); break; case 5: EDR(V, 5); break; default: EDR(V, 6); break;                                   \
    }
    switch (vec) {
        case 4: EDR_LG(4); break;
        case 2: EDR_LG(2); break;
        default: EDR_LG(1); break;
    }
#undef EDR_LG
#undef EDR
    GNNMP_LAUNCH_CHECK("edge_dot_rows_kernel");
    return GNNMP_OK;
}

int gnnmp_propagate_maxmin_grad_f32(gnnmp_graph_t *plan_t, const float *x, const float *y, const float *dy,
                                    float *dx, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!plan_t) return fail(GNNMP_EINVAL, "maxmin_grad: null plan");
    if (D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "maxmin_grad: bad D");
    if (plan_t->n_dst == 0) return GNNMP_OK;
    if (!x || !dx || (plan_t->n_total > 0 && (!y || !dy))) return fail(GNNMP_EINVAL, "maxmin_grad: null pointer");
    if (plan_t->n_chunks > 0) {
        if (int rc = ensure_workspace(plan_t, (size_t)plan_t->n_chunks * (size_t)D)) return rc;
    }
    MaxMinGradArgs a;
    a.partial = plan_t->ws;
    a.chunk_row = plan_t->chunk_row;
    a.chunk_beg = plan_t->chunk_beg;
    a.chunk_end = plan_t->chunk_end;
    a.n_chunks = plan_t->n_chunks;
    a.long_thresh = plan_t->long_thresh;
    a.rowptr = plan_t->rowptr;
    a.col = plan_t->col;
    a.x = x;
    a.y = y;
    a.dy = dy;
    a.dx = dx;
    a.D = (int)D;
    a.n_rows = (int)plan_t->n_dst;
    uintptr_t m = reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy);
    int vec = pick_vec(D, x, dx);
    if (vec == 4 && (m & 15)) vec = 1;
    if (vec == 2 && (m & 7)) vec = 1;
    a.log2g = pick_log2g((D + vec - 1) / vec);
    a.waves = 4;
    const int G = 1 << a.log2g;
    const int rows_per_block = (64 / G) * a.waves;
    const int lanes_needed = (int)((D + vec - 1) / vec);
    dim3 grid((unsigned)(((int64_t)a.n_rows + a.n_chunks + rows_per_block - 1) / rows_per_block),
              (unsigned)((lanes_needed + G - 1) / G));
    switch (vec) {
        case 4: maxmin_grad_kernel<4, 4><<<grid, 256, 0, stream>>>(a); break;
        case 2: maxmin_grad_kernel<2, 4><<<grid, 256, 0, stream>>>(a); break;
        default: maxmin_grad_kernel<1, 4><<<grid, 256, 0, stream>>>(a); break;
    }
    GNNMP_LAUNCH_CHECK("maxmin_grad_kernel");
    return run_combine_sum(plan_t, dx, D, stream);
}

}  // extern "C"
