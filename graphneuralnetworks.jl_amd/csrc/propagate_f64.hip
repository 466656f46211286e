// propagate_f64.hip — the seam's methods and their leaves for Float64 features (round 6).
//   propagate(copy_xj | w_mul_xj | e_mul_xj (vector e), g, + | mean | max | min)   GNNlib/src/msgpass.jl:71-79, 215-238
//   _gather / _scatter                                                               GNNGraphs/src/gatherscatter.jl:4,12-18
// The reference's generic path is eltype-generic and its own micro-benchmark runs in Float64 (GraphNeuralNetworks/perf/bench_gnn.jl:9-10:
// `B = rand(100, n)`, and asserts isequal(propagate(e_mul_xj, g, +; xj = B, e), B * A)); rounds 1-5 took Float32 only, so such calls
// fell through to the reference's three-pass path ("missing" 6).  This file is the Float64 spelling of the same plan walk:
// one lane group of G = 2^k lanes per destination row, a lane owns TWO doubles (16 bytes) of every feature tile of 2 G columns, source ids
// loaded coalesced and broadcast inside the group, U = 8 row loads (16 bytes each when D is even) in flight, adds in ORIGINAL edge order with separately rounded
// products — bit-identical to NNlib's sequential CPU loop on every row the plan does not split.  Split rows: chunk partials (virtual rows)
// folded in chunk order by a second small kernel (the two-kernel scheme of rounds 1-4: deterministic, no cross-workgroup hand-off inside
// a launch).  Bound: HBM / the fabric's line-request rate like the fp32 kernels — a row of D doubles is 8 D bytes; no effort was spent
// on tuning beyond the shared structure (the fp32 kernels are the measured product path; this one is about not falling back).
#include "common.h"

namespace gnnmp {

// (v_max_f64 / v_min_f64 order -0 < +0 and return the other operand for a NaN, like their f32 twins: common.h jl_max)
__device__ __forceinline__ double jl_max64(double x, double y) {
    double m;
    asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    const double nan_pick = (x != x) ? x : y;
    return __builtin_isunordered(x, y) ? nan_pick : m;
}
__device__ __forceinline__ double jl_min64(double x, double y) {
    double m;
    asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    const double nan_pick = (x != x) ? x : y;
    return __builtin_isunordered(x, y) ? nan_pick : m;
}
template <int OP>
__device__ __forceinline__ double op_identity64() {
    return OP == OP_SUM ? 0.0 : (OP == OP_MAX ? -__builtin_inf() : __builtin_inf());
}
template <int OP>
__device__ __forceinline__ double op_apply64(double a, double b) {
    if (OP == OP_SUM) return a + b;
    if (OP == OP_MAX) return jl_max64(a, b);
    return jl_min64(a, b);
}

struct R64Args {
    const uint32_t *rowptr;
    const int32_t *idx;       // per slot: row of x to read (plan->col, or plan->eid for _scatter)
    const int32_t *eid;       // per slot: original edge position (weights)
    const double *x;          // [n_src][D]
    const double *w;          // [n_edges] original edge order, nullable; plan-added self loops weigh 1
    const double *ss, *sd;    // [n_src] / [n_dst] nullable: the GCN-style source / destination factors
    double *out;              // [n_dst][D]
    double *partial;          // [n_chunks][D]
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    const int32_t *long_rows, *long_cptr;
    int n_chunks, n_long, D, n_rows, log2g, mean, long_thresh;
    uint32_t n_edges;
};

// PAIR: D is even and x is 16-byte aligned — a lane's two doubles are ONE 16-byte load (otherwise two 8-byte loads)
template <int OP, bool SCALED, bool PAIR>
__global__ void __launch_bounds__(256) csr_rows_f64_kernel(const R64Args a) {
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const int G = 1 << a.log2g;
    const int lig = lane & (G - 1), gbase = lane - lig;
    const int64_t v64 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> a.log2g;
    if (v64 >= (int64_t)a.n_rows + a.n_chunks) return;
    const int v = (int)v64;
    int row;
    uint32_t beg, end;
    const bool is_chunk = v < a.n_chunks;
    if (is_chunk) {
        row = a.chunk_row[v]; beg = a.chunk_beg[v]; end = a.chunk_end[v];
    } else {
        row = v - a.n_chunks; beg = a.rowptr[row]; end = a.rowptr[row + 1];
        if (end - beg > (uint32_t)a.long_thresh) return;      // split row: its chunks are virtual rows, folded by csr_combine_f64_kernel
    }
    const int D = a.D;
    for (int ft = 0; ft < D; ft += 2 * G) {                    // feature tiles of 2 G columns (one pass over the row per tile)
        // EVERY lane of the group walks the edges (it supplies source ids to the others); only its loads and sums depend on its columns
        const int f0 = ft + 2 * lig;
        const bool one = f0 < D, two = f0 + 1 < D;             // (odd D: the last active lane of the last tile owns one column)
        double acc0 = op_identity64<OP>(), acc1 = op_identity64<OP>();
        for (uint32_t base = beg; base < end; base += G) {
            const uint32_t p = base + lig;
            uint32_t c = 0;
            double wv = 1.0, sv = 1.0;
            if (p < end) {
                c = (uint32_t)a.idx[p];
                if (SCALED) {
                    if (a.w) {
                        const uint32_t e = (uint32_t)a.eid[p];
                        if (e < a.n_edges) wv = a.w[e];
                    }
                    if (a.ss) sv = a.ss[c];
                }
            }
            const int n = (int)min((uint32_t)G, end - base);
            for (int j = 0; j < n; j += U) {
                uint32_t cj[U];
                double wj[U], sj[U], v0[U], v1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int jj = min(j + u, n - 1);
                    cj[u] = (uint32_t)__shfl((int)c, gbase + jj, 64);
                    if (SCALED) {
                        wj[u] = __shfl(wv, gbase + jj, 64);
                        sj[u] = __shfl(sv, gbase + jj, 64);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double *px = a.x + (int64_t)cj[u] * D + f0;
                    if (PAIR) {
                        double2 q = make_double2(0.0, 0.0);
                        if (one) q = *reinterpret_cast<const double2 *>(px);
                        v0[u] = q.x; v1[u] = q.y;
                    } else {
                        v0[u] = one ? px[0] : 0.0;
                        v1[u] = two ? px[1] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (j + u < n) {
                        double t0 = v0[u], t1 = v1[u];
                        if (SCALED) {
                            t0 = t0 * sj[u]; t1 = t1 * sj[u];          // xj .* cout'  (conv.jl:59), rounded
                            t0 = wj[u] * t0; t1 = wj[u] * t1;          // w .* xj      (msgpass.jl:203-208), rounded
                        }
                        acc0 = op_apply64<OP>(acc0, t0);
                        acc1 = op_apply64<OP>(acc1, t1);
                    }
                }
            }
        }
        if (!one) continue;
        if (is_chunk) {
            a.partial[(int64_t)v * D + f0] = acc0;
            if (two) a.partial[(int64_t)v * D + f0 + 1] = acc1;
        } else {
            const uint32_t len = end - beg;
            if (OP == OP_SUM && a.mean) {      // NNlib scatter(mean): dst = 0 .+ sum ./ count; count == 0 keeps the sum (0)
                const double cnt = (double)len;
                acc0 = 0.0 + (len == 0 ? acc0 : acc0 / cnt);
                acc1 = 0.0 + (len == 0 ? acc1 : acc1 / cnt);
            }
            if (a.sd) { const double s = a.sd[row]; acc0 = acc0 * s; acc1 = acc1 * s; }
            a.out[(int64_t)row * D + f0] = acc0;
            if (two) a.out[(int64_t)row * D + f0 + 1] = acc1;
        }
    }
}

// the chunk partials of each split row, in chunk order from the identity, then the row's epilogue; one thread per (long row, column)
template <int OP>
__global__ void __launch_bounds__(256) csr_combine_f64_kernel(const R64Args a) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)a.n_long * a.D) return;
    const int r = (int)(t / a.D), f = (int)(t - (int64_t)r * a.D);
    const int row = a.long_rows[r];
    double acc = op_identity64<OP>();
    for (int c = a.long_cptr[r]; c < a.long_cptr[r + 1]; ++c) acc = op_apply64<OP>(acc, a.partial[(int64_t)c * a.D + f]);
    const uint32_t len = a.rowptr[row + 1] - a.rowptr[row];
    if (OP == OP_SUM && a.mean) acc = 0.0 + acc / (double)len;
    if (a.sd) acc = acc * a.sd[row];
    a.out[(int64_t)row * a.D + f] = acc;
}

__global__ void __launch_bounds__(256) gather_f64_kernel(const double *x, const void *idx, int idx_bytes, int base, int64_t K, double *out,
                                                         int64_t D) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * D) return;
    const int64_t k = t / D, f = t - k * D;
    out[t] = x[load_index(idx, k, idx_bytes, base) * D + f];
}

template <int OP, bool SCALED>
static int launch_rows64(const R64Args &a, hipStream_t stream) {
    const int64_t groups = (int64_t)a.n_rows + a.n_chunks;
    const int64_t threads = groups << a.log2g;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    if ((a.D & 1) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0)
        csr_rows_f64_kernel<OP, SCALED, true><<<nb, 256, 0, stream>>>(a);
    else
        csr_rows_f64_kernel<OP, SCALED, false><<<nb, 256, 0, stream>>>(a);
    GNNMP_LAUNCH_CHECK("csr_rows_f64_kernel");
    if (a.n_long > 0) {
        const int64_t tc = (int64_t)a.n_long * a.D;
        csr_combine_f64_kernel<OP><<<(unsigned)((tc + 255) / 256), 256, 0, stream>>>(a);
        GNNMP_LAUNCH_CHECK("csr_combine_f64_kernel");
    }
    return GNNMP_OK;
}

static int run_reduce64(gnnmp_graph_t *p, const int32_t *idx, int aggr, const double *x, const double *w, const double *ss,
                        const double *sd, double *out, int64_t D, hipStream_t stream) {
    if (p->n_dst == 0 || D == 0) return GNNMP_OK;
    if (D > (int64_t)INT32_MAX / 4) return fail(GNNMP_EUNSUPPORTED, "propagate_f64: D too large");
    R64Args a = {};
    a.rowptr = p->rowptr; a.idx = idx; a.eid = p->eid; a.x = x; a.w = w; a.ss = ss; a.sd = sd; a.out = out;
    a.chunk_row = p->chunk_row; a.chunk_beg = p->chunk_beg; a.chunk_end = p->chunk_end; a.long_rows = p->long_rows; a.long_cptr = p->long_cptr;
    a.n_chunks = p->n_chunks; a.n_long = p->n_long; a.D = (int)D; a.n_rows = (int)p->n_dst; a.n_edges = (uint32_t)p->n_edges;
    a.mean = (aggr == GNNMP_MEAN); a.long_thresh = p->long_thresh;
    int l = 0;
    while ((2 << l) < D && l < 6) ++l;            // 2 G >= D up to a whole wave
    a.log2g = l;
    if (p->n_chunks > 0) {
        if (int rc = ensure_workspace(p, 2 * (size_t)p->n_chunks * (size_t)D + 2)) return rc;      // (doubles in the float workspace: hipMalloc is 256-byte aligned)
        a.partial = reinterpret_cast<double *>(p->ws);
    }
    const bool scaled = w || ss;
    switch (aggr) {
        case GNNMP_SUM:
        case GNNMP_MEAN: return scaled ? launch_rows64<OP_SUM, true>(a, stream) : launch_rows64<OP_SUM, false>(a, stream);
        case GNNMP_MAX: return scaled ? launch_rows64<OP_MAX, true>(a, stream) : launch_rows64<OP_MAX, false>(a, stream);
        default: return scaled ? launch_rows64<OP_MIN, true>(a, stream) : launch_rows64<OP_MIN, false>(a, stream);
    }
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_propagate_f64(gnnmp_graph_t *p, int msg, int aggr, const double *xj, const double *w, const double *scale_src,
                        const double *scale_dst, double *out, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "propagate_f64: null plan");
    if (aggr < GNNMP_SUM || aggr > GNNMP_MIN) return fail(GNNMP_EINVAL, "propagate_f64: bad aggr %d", aggr);
    if (msg != GNNMP_COPY_XJ && msg != GNNMP_W_MUL_XJ) return fail(GNNMP_EINVAL, "propagate_f64: bad msg %d", msg);
    if (D < 0) return fail(GNNMP_EINVAL, "propagate_f64: negative D");
    if (p->n_dst == 0 || D == 0) return GNNMP_OK;
    if (!out || (!xj && p->n_total > 0)) return fail(GNNMP_EINVAL, "propagate_f64: null pointer");
    if (msg == GNNMP_W_MUL_XJ && !w && p->n_edges > 0) return fail(GNNMP_EINVAL, "propagate_f64: W_MUL_XJ without weights");
    return run_reduce64(p, p->col, aggr, xj, msg == GNNMP_W_MUL_XJ ? w : nullptr, scale_src, scale_dst, out, D, stream);
}

int gnnmp_scatter_f64(gnnmp_graph_t *p, int aggr, const double *m, double *out, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "scatter_f64: null plan");
    if (aggr < GNNMP_SUM || aggr > GNNMP_MIN) return fail(GNNMP_EINVAL, "scatter_f64: bad aggr %d", aggr);
    if (D < 0) return fail(GNNMP_EINVAL, "scatter_f64: negative D");
    if (p->n_dst == 0 || D == 0) return GNNMP_OK;
    if (!out || (!m && p->n_total > 0)) return fail(GNNMP_EINVAL, "scatter_f64: null pointer");
    return run_reduce64(p, p->eid, aggr, m, nullptr, nullptr, nullptr, out, D, stream);      // rows of m by original edge position
}

int gnnmp_gather_f64(const double *x, const void *idx, int idx_bytes, int index_base, int64_t K, double *out, int64_t D,
                     gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "gather_f64: idx_bytes must be 4 or 8 (got %d)", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "gather_f64: index_base must be 0 or 1");
    if (K < 0 || D < 0) return fail(GNNMP_EINVAL, "gather_f64: negative size");
    if (K == 0 || D == 0) return GNNMP_OK;
    if (!x || !idx || !out) return fail(GNNMP_EINVAL, "gather_f64: null pointer");
    const int64_t n = K * D;
    gather_f64_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(x, idx, idx_bytes, index_base, K, out, D);
    GNNMP_LAUNCH_CHECK("gather_f64_kernel");
    return GNNMP_OK;
}

}  // extern "C"
