// leaf.hip — the leaf ops of GNNGraphs/src/gatherscatter.jl (NNlib.gather :4, NNlib.scatter :12-18) for callers
// that go through the generic (closure) path of propagate, the NNlib-style atomic scatter kept as the measured
// comparator, and reduce_nodes over a sorted graph_indicator (GNNlib/src/utils.jl:12-16).
#include "common.h"

namespace gnnmp {

// out[k][:] = x[idx[k]][:]; one group of G lanes per gathered row, VEC floats per lane per step.
template <int VEC>
__global__ void __launch_bounds__(256) gather_kernel(const float *x, const void *idx, int idx_bytes,
                                                     int base, int64_t K, float *out, int D,
                                                     int log2g) {
    const int G = 1 << log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> log2g;
    if (k >= K) return;
    const int64_t r = load_index(idx, k, idx_bytes, base);
    const float *srow = x + r * D;
    float *drow = out + k * D;
    for (int f = lig * VEC; f < D; f += G * VEC) {
        float v[VEC];
        Vec<VEC>::load(srow + f, v);
        Vec<VEC>::store(drow + f, v);
    }
}

// out[k][:] = sign * (xi[t_k][:] - xj[s_k][:]) — apply_edges(xi_sub_xj | xj_sub_xi) (GNNlib/src/msgpass.jl:177-185) in one
// pass: both rows gathered and subtracted in registers instead of two (D, E) gathers and a broadcast.
template <int VEC>
__global__ void __launch_bounds__(256) edge_sub_kernel(const float *xi, const float *xj, const void *s, const void *t,
                                                       int idx_bytes, int base, int64_t K, int swap, float *out, int D,
                                                       int log2g) {
    const int G = 1 << log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> log2g;
    if (k >= K) return;
    const float *irow = xi + load_index(t, k, idx_bytes, base) * D;
    const float *jrow = xj + load_index(s, k, idx_bytes, base) * D;
    float *drow = out + k * D;
    for (int f = lig * VEC; f < D; f += G * VEC) {
        float a[VEC], b[VEC];
        Vec<VEC>::load(irow + f, a);
        Vec<VEC>::load(jrow + f, b);
#pragma unroll
        for (int q = 0; q < VEC; ++q) a[q] = swap ? b[q] - a[q] : a[q] - b[q];
        Vec<VEC>::store(drow + f, a);
    }
}

__device__ __forceinline__ void atomic_max_f32(float *addr, float v) {
    // CAS loop with Julia max semantics (comparator path only)
    unsigned int *ua = reinterpret_cast<unsigned int *>(addr);
    unsigned int old = *ua, assumed;
    do {
        assumed = old;
        float cur = __uint_as_float(assumed);
        float nv = jl_max(cur, v);
        if (__float_as_uint(nv) == assumed) break;
        old = atomicCAS(ua, assumed, __float_as_uint(nv));
    } while (old != assumed);
}
__device__ __forceinline__ void atomic_min_f32(float *addr, float v) {
    unsigned int *ua = reinterpret_cast<unsigned int *>(addr);
    unsigned int old = *ua, assumed;
    do {
        assumed = old;
        float cur = __uint_as_float(assumed);
        float nv = jl_min(cur, v);
        if (__float_as_uint(nv) == assumed) break;
        old = atomicCAS(ua, assumed, __float_as_uint(nv));
    } while (old != assumed);
}

// NNlib's GPU scatter: one thread per (feature, edge) element, one atomic each.
template <int OP>
__global__ void __launch_bounds__(256) scatter_atomic_kernel(const float *m, const void *idx,
                                                             int idx_bytes, int base, int64_t K,
                                                             float *out, int D) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * D) return;
    const int64_t k = t / D;
    const int f = (int)(t - k * D);
    const int64_t r = load_index(idx, k, idx_bytes, base);
    float *dst = out + r * D + f;
    const float v = m[t];
    if (OP == OP_SUM)
        atomicAdd(dst, v);
    else if (OP == OP_MAX)
        atomic_max_f32(dst, v);
    else
        atomic_min_f32(dst, v);
}

// reduce_nodes over a sorted indicator: group of G lanes per graph, boundaries by binary search,
// node rows are contiguous => coalesced streaming reads, sum in node order.
template <int VEC, int OP>
__global__ void __launch_bounds__(256) segment_pool_kernel(const float *x, const void *seg,
                                                           int idx_bytes, int base, float *out,
                                                           int D, int64_t N, int64_t Gn, int log2g,
                                                           int mean) {
    const int G = 1 << log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> log2g;
    if (g >= Gn) return;
    // first node with seg >= g, first node with seg >= g+1
    int64_t lo = 0, hi = N;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (load_index(seg, mid, idx_bytes, base) < g) lo = mid + 1; else hi = mid;
    }
    const int64_t beg = lo;
    hi = N;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (load_index(seg, mid, idx_bytes, base) < g + 1) lo = mid + 1; else hi = mid;
    }
    const int64_t end = lo;
    const float cnt = (float)(end - beg);
    for (int f = lig * VEC; f < D; f += G * VEC) {
        float acc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP>();
        int64_t n = beg;
        for (; n + 4 <= end; n += 4) {
            float v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) Vec<VEC>::load(x + (n + u) * D + f, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = op_apply<OP>(acc[q], v[u][q]);
        }
        for (; n < end; ++n) {
            float v[VEC];
            Vec<VEC>::load(x + n * D + f, v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = op_apply<OP>(acc[q], v[q]);
        }
        if (OP == OP_SUM && mean) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = 0.0f + (end == beg ? acc[q] : acc[q] / cnt);
        }
        Vec<VEC>::store(out + g * D + f, acc);
    }
}

// Segment boundaries of a sorted indicator, O(1) depth: node i starts every graph id in (seg[i-1], seg[i]] (empty graphs between
// two non-empty ones start — and end — where the next one starts); ids past the last node's graph start at N.
__global__ void __launch_bounds__(256) segment_bounds_kernel(const void *seg, int idx_bytes, int base, int64_t N, int64_t Gn,
                                                             int64_t *ptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    const int64_t prev = i == 0 ? -1 : load_index(seg, i - 1, idx_bytes, base);
    const int64_t cur = i == N ? Gn : load_index(seg, i, idx_bytes, base);
    for (int64_t k = max(prev + 1, (int64_t)0); k <= min(cur, Gn); ++k) ptr[k] = i;
}

// reduce_nodes with the boundaries given: one lane group per graph streams its contiguous rows, 8 in flight, sum in node order
template <int VEC, int OP>
__global__ void __launch_bounds__(256) segment_pool_ptr_kernel(const float *x, const int64_t *ptr, float *out, int D, int64_t Gn,
                                                               int log2g, int mean) {
    const int G = 1 << log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> log2g;
    if (g >= Gn) return;
    const int64_t beg = ptr[g], end = ptr[g + 1];
    const float cnt = (float)(end - beg);
    constexpr int U = 8;
    for (int f = lig * VEC; f < D; f += G * VEC) {
        float acc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP>();
        for (int64_t n = beg; n < end; n += U) {
            float v[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) Vec<VEC>::load(x + min(n + u, end - 1) * D + f, v[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (n + u < end) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[q] = op_apply<OP>(acc[q], v[u][q]);
                }
            }
        }
        if (OP == OP_SUM && mean) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = 0.0f + (end == beg ? acc[q] : acc[q] / cnt);
        }
        Vec<VEC>::store(out + g * D + f, acc);
    }
}

// ---- small elementwise helpers so that no arithmetic of the layer bodies is left to the host framework --------------
__global__ void __launch_bounds__(256) add_kernel(const float *a, const float *b, float *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
// out[n][d] = a[n][d * (Da > 1)] * b[n][d]   — `α .* feats` of global_attention_pool (GNNlib/src/layers/pool.jl:6-10) with a
// gate of one channel (broadcast over the features) or of D channels
__global__ void __launch_bounds__(256) mul_rows_kernel(const float *a, int Da, const float *b, float *out, int64_t N, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int64_t n = i / D;
    const int d = (int)(i - n * D);
    out[i] = a[n * Da + (Da > 1 ? d : 0)] * b[i];
}
// out = alpha .* x .+ y   — gin_conv's `(1 .+ ϵ) .* xi .+ m` (GNNlib/src/layers/conv.jl:250-256): product rounded, then sum
__global__ void __launch_bounds__(256) axpy_kernel(float alpha, const float *x, const float *y, float *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = alpha * x[i] + y[i];
}
// out[n][c] = act( (Σ_h y[n][h][c]) / H + bias[c] )   — `mean(x, dims = 2)` of gat_conv with concat = false
// (GNNlib/src/layers/conv.jl:143-147): heads summed in order h = 1..H, one true division, then σ.(x .+ bias)
__global__ void __launch_bounds__(256) head_mean_kernel(const float *y, const float *bias, int act, float *out,
                                                        int64_t N, int H, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int64_t n = i / C;
    const int c = (int)(i - n * C);
    const float *row = y + n * (int64_t)H * C + c;
    float acc = row[0];
    for (int h = 1; h < H; ++h) acc = acc + row[(int64_t)h * C];
    float v = acc / (float)H;
    if (bias) v = v + bias[c];
    if (act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
    out[i] = v;
}
// pullback of `mean(x, dims = 2)`: dy[n][h][c] = dz[n][c] / H for every head
__global__ void __launch_bounds__(256) head_mean_grad_kernel(const float *dz, float *dy, int64_t N, int H, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * H * C) return;
    const int64_t n = i / ((int64_t)H * C);
    const int c = (int)(i % C);
    dy[i] = dz[n * C + c] / (float)H;
}
// sum over the G lanes (a power of two <= 64) that share a row
template <int G>
__device__ __forceinline__ float lanes_sum(float v) {
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// xn = x ./ sqrt.(sum(x .^ 2, dims = 1))  (agnn_conv, GNNlib/src/layers/conv.jl:341-342): G lanes per row, the norm kept
template <int G>
__global__ void __launch_bounds__(256) row_normalize_kernel(const float *__restrict__ x, float *__restrict__ xn,
                                                            float *__restrict__ rnorm, int64_t N, int D) {
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int l = threadIdx.x & (G - 1);
    const int64_t rc = min(row, N - 1);
    const float *xr = x + rc * D;
    float ss = 0.0f;
    for (int k = l; k < D; k += G) { const float v = xr[k]; ss += v * v; }
    const float r = sqrtf(lanes_sum<G>(ss));
    if (row >= N) return;
    for (int k = l; k < D; k += G) xn[rc * D + k] = xr[k] / r;
    if (l == 0 && rnorm) rnorm[rc] = r;
}
// pullback of the row normalisation with two cotangents of xn (as query and as key of the attention):
//   dn = dq + dk,  dx = base + (dn - xn (xn . dn)) / r,   qdot[row] = qscale (xn . dq)  (the share that carries d/dβ)
template <int G>
__global__ void __launch_bounds__(256) row_normalize_grad_kernel(const float *__restrict__ dq, const float *__restrict__ dk,
                                                                 const float *__restrict__ xn, const float *__restrict__ rnorm,
                                                                 const float *__restrict__ base, float *__restrict__ dx,
                                                                 float *__restrict__ qdot, float qscale, int64_t N, int D) {
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    const int l = threadIdx.x & (G - 1);
    const int64_t rc = min(row, N - 1);
    const int64_t o = rc * D;
    float p = 0.0f, q = 0.0f;
    for (int k = l; k < D; k += G) {
        const float n = xn[o + k];
        const float a = dq ? dq[o + k] : 0.0f, b = dk ? dk[o + k] : 0.0f;
        p += n * (a + b);
        q += n * a;
    }
    p = lanes_sum<G>(p);
    q = lanes_sum<G>(q);
    if (row >= N) return;
    const float r = rnorm[rc];
    for (int k = l; k < D; k += G) {
        const float n = xn[o + k];
        const float dn = (dq ? dq[o + k] : 0.0f) + (dk ? dk[o + k] : 0.0f);
        const float v = (dn - n * p) / r;
        dx[o + k] = base ? base[o + k] + v : v;
    }
    if (l == 0 && qdot) qdot[rc] = qscale * q;
}
// Flux.GRUCell's pointwise part (the cell of gated_graph_conv, GNNlib/src/layers/conv.jl:228-232; Flux 0.16, un-vendored):
// gx = Wi m, gh = Wh h, both [N][3D] with the gates in the order r, z, candidate, b [3D] (nullable):
//   r = σ(gx_r + gh_r + b_r),  z = σ(gx_z + gh_z + b_z),  h~ = tanh(gx_n + r .* gh_n + b_n),  h' = (1 - z) .* h~ + z .* h
__global__ void __launch_bounds__(256) gru_pointwise_kernel(const float *__restrict__ gx, const float *__restrict__ gh,
                                                            const float *__restrict__ b, const float *__restrict__ h,
                                                            float *__restrict__ out, int64_t N, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int64_t n = i / D;
    const int d = (int)(i - n * D);
    const int64_t o = n * 3 * D + d;
    const float br = b ? b[d] : 0.0f, bz = b ? b[D + d] : 0.0f, bn = b ? b[2 * D + d] : 0.0f;
    const float xr = gx[o] + gh[o] + br, xz = gx[o + D] + gh[o + D] + bz;
    const float tr = expf(-fabsf(xr)), tz = expf(-fabsf(xz));
    const float r = xr >= 0.0f ? 1.0f / (1.0f + tr) : tr / (1.0f + tr);
    const float z = xz >= 0.0f ? 1.0f / (1.0f + tz) : tz / (1.0f + tz);
    const float c = tanhf(gx[o + 2 * D] + r * gh[o + 2 * D] + bn);
    out[i] = (1.0f - z) * c + z * h[i];
}
// gmm_conv's mixture weights (GNNlib/src/layers/conv.jl:379-385), statement by statement:
//   w[d][k][e] = ((e[d][e] - mu[d][k])^2) / 2;  w = w .* sigma_inv[d][k]^2;  w[k][e] = exp(sum over d)
// written once per (edge, kernel k) and repeated over the C output channels of block k: out[e][k * C + c], the (C * K, E)
// factor array `propagate(e_mul_xj, g, mean; xj = (out, K, N), e = (1, K, E))` broadcasts to.
__global__ void __launch_bounds__(256) gmm_weights_kernel(const float *__restrict__ e, const float *__restrict__ mu,
                                                          const float *__restrict__ sinv, float *__restrict__ out, int64_t E,
                                                          int ein, int K, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * K * C) return;
    const int64_t ed = i / ((int64_t)K * C);
    const int k = (int)((i / C) % K);
    float s = 0.0f;
    for (int d = 0; d < ein; ++d) {
        const float df = e[ed * ein + d] - mu[k * ein + d];   // mu, sigma_inv: Julia (ein, K) column-major = [K][ein]
        const float si = sinv[k * ein + d];
        float w = (df * df) / 2.0f;
        w = w * (si * si);
        s = s + w;
    }
    out[i] = expf(s);
}
// egnn_conv (conv.jl:465-467): sqnorm = sum(x_diff .^ 2, dims = 1); x_diff ./ (sqrt.(sqnorm) .+ eps) — rows are coordinates
// (D = 3 typically): one thread per row, features added in order
__global__ void __launch_bounds__(256) row_sqnorm_normalize_kernel(const float *__restrict__ x, float *__restrict__ sq,
                                                                   float *__restrict__ xn, float eps, int64_t N, int D) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    float s = 0.0f;
    for (int d = 0; d < D; ++d) { const float v = x[r * D + d]; s = s + v * v; }
    if (sq) sq[r] = s;
    const float den = sqrtf(s) + eps;
    if (xn) for (int d = 0; d < D; ++d) xn[r * D + d] = x[r * D + d] / den;
}
// Flux.LSTMCell's pointwise part (the cell of set2set_pool, GNNlib/src/layers/pool.jl:31-44; Flux 0.16, un-vendored):
// g = Wi x + Wh h + b is [N][4D] with the gates in the order input, forget, cell, output:
//   c' = σ(forget) .* c + σ(input) .* tanh(cell),   h' = σ(output) .* tanh(c')
__global__ void __launch_bounds__(256) lstm_pointwise_kernel(const float *__restrict__ gx, const float *__restrict__ gh,
                                                             const float *__restrict__ b, const float *__restrict__ c,
                                                             float *__restrict__ h_out, float *__restrict__ c_out, int64_t N,
                                                             int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int64_t n = i / D;
    const int d = (int)(i - n * D);
    const int64_t o = n * 4 * D + d;
    float g4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g4[q] = gx[o + q * D] + gh[o + q * D] + (b ? b[q * D + d] : 0.0f);
    auto sg = [](float x) { const float t = expf(-fabsf(x)); return x >= 0.0f ? 1.0f / (1.0f + t) : t / (1.0f + t); };
    const float cn = sg(g4[1]) * c[i] + sg(g4[0]) * tanhf(g4[2]);
    c_out[i] = cn;
    h_out[i] = sg(g4[3]) * tanhf(cn);
}
// out[n] = sum(a[n][:] .* b[n][:]) — `sum(qn .* x, dims = 1)` of set2set_pool; features added in order
__global__ void __launch_bounds__(256) rowdot_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                     float *__restrict__ out, int64_t N, int D) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.0f;
    for (int d = 0; d < D; ++d) s = s + a[n * D + d] * b[n * D + d];
    out[n] = s;
}
// flag[0] = 1 if idx[k] > idx[k+1] for some k
__global__ void __launch_bounds__(256) unsorted_kernel(const void *idx, int idx_bytes, int64_t n, int *flag) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k + 1 < n && load_index(idx, k, idx_bytes, 0) > load_index(idx, k + 1, idx_bytes, 0)) *flag = 1;
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_gather_f32(const float *x, const void *idx, int idx_bytes, int index_base, int64_t K,
                     float *out, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "gather: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "gather: index_base %d", index_base);
    if (K < 0 || D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "gather: bad size");
    if (K == 0 || D == 0) return GNNMP_OK;
    if (!x || !idx || !out) return fail(GNNMP_EINVAL, "gather: null pointer");
    const int vec = pick_vec(D, x, out);
    const int log2g = pick_log2g((D + vec - 1) / vec);
    const int64_t threads = K << log2g;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    switch (vec) {
        case 4: gather_kernel<4><<<nb, 256, 0, stream>>>(x, idx, idx_bytes, index_base, K, out, (int)D, log2g); break;
        case 2: gather_kernel<2><<<nb, 256, 0, stream>>>(x, idx, idx_bytes, index_base, K, out, (int)D, log2g); break;
        default: gather_kernel<1><<<nb, 256, 0, stream>>>(x, idx, idx_bytes, index_base, K, out, (int)D, log2g); break;
    }
    GNNMP_LAUNCH_CHECK("gather_kernel");
    return GNNMP_OK;
}

int gnnmp_edge_sub_f32(const float *xi, const float *xj, const void *s, const void *t, int idx_bytes, int index_base,
                       int64_t K, int xj_minus_xi, float *out, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "edge_sub: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "edge_sub: index_base %d", index_base);
    if (K < 0 || D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "edge_sub: bad size");
    if (K == 0 || D == 0) return GNNMP_OK;
    if (!xi || !xj || !s || !t || !out) return fail(GNNMP_EINVAL, "edge_sub: null pointer");
    int vec = pick_vec(D, xi, out);
    if ((reinterpret_cast<uintptr_t>(xj) & (4 * vec - 1)) != 0) vec = 1;
    const int log2g = pick_log2g((D + vec - 1) / vec);
    const int64_t threads = K << log2g;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    const int sw = xj_minus_xi ? 1 : 0;
    switch (vec) {
        case 4: edge_sub_kernel<4><<<nb, 256, 0, stream>>>(xi, xj, s, t, idx_bytes, index_base, K, sw, out, (int)D, log2g); break;
        case 2: edge_sub_kernel<2><<<nb, 256, 0, stream>>>(xi, xj, s, t, idx_bytes, index_base, K, sw, out, (int)D, log2g); break;
        default: edge_sub_kernel<1><<<nb, 256, 0, stream>>>(xi, xj, s, t, idx_bytes, index_base, K, sw, out, (int)D, log2g); break;
    }
    GNNMP_LAUNCH_CHECK("edge_sub_kernel");
    return GNNMP_OK;
}

int gnnmp_scatter_atomic_f32(int aggr, const float *m, const void *idx, int idx_bytes,
                             int index_base, int64_t K, float *out, int64_t D,
                             gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "scatter_atomic: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "scatter_atomic: index_base %d", index_base);
    if (aggr != GNNMP_SUM && aggr != GNNMP_MAX && aggr != GNNMP_MIN)
        return fail(GNNMP_EINVAL, "scatter_atomic: aggr must be SUM, MAX or MIN (got %d)", aggr);
    if (K < 0 || D < 0) return fail(GNNMP_EINVAL, "scatter_atomic: bad size");
    if (K == 0 || D == 0) return GNNMP_OK;
    if (!m || !idx || !out) return fail(GNNMP_EINVAL, "scatter_atomic: null pointer");
    const int64_t total = K * D;
    const unsigned nb = (unsigned)((total + 255) / 256);
    if (aggr == GNNMP_SUM)
        scatter_atomic_kernel<OP_SUM><<<nb, 256, 0, stream>>>(m, idx, idx_bytes, index_base, K, out, (int)D);
    else if (aggr == GNNMP_MAX)
        scatter_atomic_kernel<OP_MAX><<<nb, 256, 0, stream>>>(m, idx, idx_bytes, index_base, K, out, (int)D);
    else
        scatter_atomic_kernel<OP_MIN><<<nb, 256, 0, stream>>>(m, idx, idx_bytes, index_base, K, out, (int)D);
    GNNMP_LAUNCH_CHECK("scatter_atomic_kernel");
    return GNNMP_OK;
}

int gnnmp_add_f32(const float *a, const float *b, float *out, int64_t n, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) return fail(GNNMP_EINVAL, "add: negative n");
    if (n == 0) return GNNMP_OK;
    if (!a || !b || !out) return fail(GNNMP_EINVAL, "add: null pointer");
    add_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(a, b, out, n);
    GNNMP_LAUNCH_CHECK("add_kernel");
    return GNNMP_OK;
}

int gnnmp_mul_rows_f32(const float *a, int64_t Da, const float *b, float *out, int64_t N, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D <= 0 || (Da != 1 && Da != D)) return fail(GNNMP_EINVAL, "mul_rows: a must have 1 or D channels");
    if (N == 0) return GNNMP_OK;
    if (!a || !b || !out) return fail(GNNMP_EINVAL, "mul_rows: null pointer");
    mul_rows_kernel<<<(unsigned)((N * D + 255) / 256), 256, 0, stream>>>(a, (int)Da, b, out, N, (int)D);
    GNNMP_LAUNCH_CHECK("mul_rows_kernel");
    return GNNMP_OK;
}

int gnnmp_axpy_f32(float alpha, const float *x, const float *y, float *out, int64_t n, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) return fail(GNNMP_EINVAL, "axpy: negative n");
    if (n == 0) return GNNMP_OK;
    if (!x || !y || !out) return fail(GNNMP_EINVAL, "axpy: null pointer");
    axpy_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(alpha, x, y, out, n);
    GNNMP_LAUNCH_CHECK("axpy_kernel");
    return GNNMP_OK;
}

int gnnmp_head_mean_f32(const float *y, const float *bias, int act, float *out, int64_t N, int64_t H,
                        int64_t C, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || H <= 0 || C <= 0) return fail(GNNMP_EINVAL, "head_mean: bad size");
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "head_mean: bad act %d", act);
    if (N == 0) return GNNMP_OK;
    if (!y || !out) return fail(GNNMP_EINVAL, "head_mean: null pointer");
    head_mean_kernel<<<(unsigned)((N * C + 255) / 256), 256, 0, stream>>>(y, bias, act, out, N, (int)H, (int)C);
    GNNMP_LAUNCH_CHECK("head_mean_kernel");
    return GNNMP_OK;
}

int gnnmp_head_mean_grad_f32(const float *dz, float *dy, int64_t N, int64_t H, int64_t C, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || H <= 0 || C <= 0) return fail(GNNMP_EINVAL, "head_mean_grad: bad size");
    if (N == 0) return GNNMP_OK;
    if (!dz || !dy) return fail(GNNMP_EINVAL, "head_mean_grad: null pointer");
    head_mean_grad_kernel<<<(unsigned)((N * H * C + 255) / 256), 256, 0, stream>>>(dz, dy, N, (int)H, (int)C);
    GNNMP_LAUNCH_CHECK("head_mean_grad_kernel");
    return GNNMP_OK;
}

int gnnmp_gru_pointwise_f32(const float *gx, const float *gh, const float *b, const float *h, float *out, int64_t N, int64_t D,
                            gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "gru_pointwise: bad size");
    if (N == 0) return GNNMP_OK;
    if (!gx || !gh || !h || !out) return fail(GNNMP_EINVAL, "gru_pointwise: null pointer");
    gru_pointwise_kernel<<<(unsigned)((N * D + 255) / 256), 256, 0, stream>>>(gx, gh, b, h, out, N, (int)D);
    GNNMP_LAUNCH_CHECK("gru_pointwise_kernel");
    return GNNMP_OK;
}

int gnnmp_gmm_weights_f32(const float *e, const float *mu, const float *sigma_inv, float *out, int64_t E, int64_t ein, int64_t K,
                          int64_t C, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (E < 0 || ein <= 0 || K <= 0 || C <= 0 || K * C > (1 << 20)) return fail(GNNMP_EINVAL, "gmm_weights: bad size");
    if (E == 0) return GNNMP_OK;
    if (!e || !mu || !sigma_inv || !out) return fail(GNNMP_EINVAL, "gmm_weights: null pointer");
    gmm_weights_kernel<<<(unsigned)((E * K * C + 255) / 256), 256, 0, stream>>>(e, mu, sigma_inv, out, E, (int)ein, (int)K, (int)C);
    GNNMP_LAUNCH_CHECK("gmm_weights_kernel");
    return GNNMP_OK;
}

int gnnmp_row_sqnorm_normalize_f32(const float *x, float *sq, float *xn, float eps, int64_t N, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "row_sqnorm_normalize: bad size");
    if (N == 0) return GNNMP_OK;
    if (!x || (!sq && !xn)) return fail(GNNMP_EINVAL, "row_sqnorm_normalize: null pointer");
    row_sqnorm_normalize_kernel<<<(unsigned)((N + 255) / 256), 256, 0, stream>>>(x, sq, xn, eps, N, (int)D);
    GNNMP_LAUNCH_CHECK("row_sqnorm_normalize_kernel");
    return GNNMP_OK;
}

int gnnmp_lstm_pointwise_f32(const float *gx, const float *gh, const float *b, const float *c, float *h_out, float *c_out,
                             int64_t N, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "lstm_pointwise: bad size");
    if (N == 0) return GNNMP_OK;
    if (!gx || !gh || !c || !h_out || !c_out) return fail(GNNMP_EINVAL, "lstm_pointwise: null pointer");
    lstm_pointwise_kernel<<<(unsigned)((N * D + 255) / 256), 256, 0, stream>>>(gx, gh, b, c, h_out, c_out, N, (int)D);
    GNNMP_LAUNCH_CHECK("lstm_pointwise_kernel");
    return GNNMP_OK;
}

int gnnmp_rowdot_f32(const float *a, const float *b, float *out, int64_t N, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "rowdot: bad size");
    if (N == 0) return GNNMP_OK;
    if (!a || !b || !out) return fail(GNNMP_EINVAL, "rowdot: null pointer");
    rowdot_kernel<<<(unsigned)((N + 255) / 256), 256, 0, stream>>>(a, b, out, N, (int)D);
    GNNMP_LAUNCH_CHECK("rowdot_kernel");
    return GNNMP_OK;
}

static int row_group(int64_t D) {   // lanes per row: the power of two >= D, clamped to [4, 64]
    int g = 4;
    while (g < 64 && g < D) g <<= 1;
    return g;
}

int gnnmp_row_normalize_f32(const float *x, float *xn, float *rnorm, int64_t N, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "row_normalize: bad size");
    if (N == 0) return GNNMP_OK;
    if (!x || !xn) return fail(GNNMP_EINVAL, "row_normalize: null pointer");
    const int g = row_group(D);
    const unsigned grid = (unsigned)((N * g + 255) / 256);
    switch (g) {
        case 4: row_normalize_kernel<4><<<grid, 256, 0, stream>>>(x, xn, rnorm, N, (int)D); break;
        case 8: row_normalize_kernel<8><<<grid, 256, 0, stream>>>(x, xn, rnorm, N, (int)D); break;
        case 16: row_normalize_kernel<16><<<grid, 256, 0, stream>>>(x, xn, rnorm, N, (int)D); break;
        case 32: row_normalize_kernel<32><<<grid, 256, 0, stream>>>(x, xn, rnorm, N, (int)D); break;
        default: row_normalize_kernel<64><<<grid, 256, 0, stream>>>(x, xn, rnorm, N, (int)D); break;
    }
    GNNMP_LAUNCH_CHECK("row_normalize_kernel");
    return GNNMP_OK;
}

int gnnmp_row_normalize_grad_f32(const float *dq, const float *dk, const float *xn, const float *rnorm, const float *base,
                                 float *dx, float *qdot, float qscale, int64_t N, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "row_normalize_grad: bad size");
    if (N == 0) return GNNMP_OK;
    if (!xn || !rnorm || !dx || (!dq && !dk)) return fail(GNNMP_EINVAL, "row_normalize_grad: null pointer");
    const int g = row_group(D);
    const unsigned grid = (unsigned)((N * g + 255) / 256);
    switch (g) {
        case 4: row_normalize_grad_kernel<4><<<grid, 256, 0, stream>>>(dq, dk, xn, rnorm, base, dx, qdot, qscale, N, (int)D); break;
        case 8: row_normalize_grad_kernel<8><<<grid, 256, 0, stream>>>(dq, dk, xn, rnorm, base, dx, qdot, qscale, N, (int)D); break;
        case 16: row_normalize_grad_kernel<16><<<grid, 256, 0, stream>>>(dq, dk, xn, rnorm, base, dx, qdot, qscale, N, (int)D); break;
        case 32: row_normalize_grad_kernel<32><<<grid, 256, 0, stream>>>(dq, dk, xn, rnorm, base, dx, qdot, qscale, N, (int)D); break;
        default: row_normalize_grad_kernel<64><<<grid, 256, 0, stream>>>(dq, dk, xn, rnorm, base, dx, qdot, qscale, N, (int)D); break;
    }
    GNNMP_LAUNCH_CHECK("row_normalize_grad_kernel");
    return GNNMP_OK;
}

int gnnmp_is_sorted(const void *idx, int idx_bytes, int64_t n, int *result_host, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "is_sorted: idx_bytes %d", idx_bytes);
    if (!result_host || n < 0) return fail(GNNMP_EINVAL, "is_sorted: bad argument");
    *result_host = 1;
    if (n < 2) return GNNMP_OK;
    if (!idx) return fail(GNNMP_EINVAL, "is_sorted: null pointer");
    int *flag = nullptr;
    GNNMP_HIP(hipMalloc((void **)&flag, sizeof(int)));
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), stream);
    if (e == hipSuccess) {
        unsorted_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(idx, idx_bytes, n, flag);
        e = hipGetLastError();
    }
    int h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(flag);
    if (e != hipSuccess) return hip_fail(e, "is_sorted");
    *result_host = h ? 0 : 1;
    return GNNMP_OK;
}

int gnnmp_segment_pool_f32(int aggr, const float *x, const void *seg_ids, int idx_bytes,
                           int index_base, float *out, int64_t D, int64_t N, int64_t G,
                           gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "segment_pool: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "segment_pool: index_base %d", index_base);
    if (aggr < GNNMP_SUM || aggr > GNNMP_MIN) return fail(GNNMP_EINVAL, "segment_pool: bad aggr %d", aggr);
    if (N < 0 || G < 0 || D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "segment_pool: bad size");
    if (G == 0 || D == 0) return GNNMP_OK;
    if (!out || (N > 0 && (!x || !seg_ids))) return fail(GNNMP_EINVAL, "segment_pool: null pointer");
    const int vec = pick_vec(D, x, out);
    const int log2g = pick_log2g((D + vec - 1) / vec);
    const int64_t threads = G << log2g;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    const int mean = aggr == GNNMP_MEAN;
#define POOL_LAUNCH(V, O) \
    segment_pool_kernel<V, O><<<nb, 256, 0, stream>>>(x, seg_ids, idx_bytes, index_base, out, (int)D, N, G, log2g, mean)
#define POOL_OP(V)                                             \
    do {                                                       \
        if (aggr == GNNMP_MAX) POOL_LAUNCH(V, OP_MAX);         \
        else if (aggr == GNNMP_MIN) POOL_LAUNCH(V, OP_MIN);    \
        else POOL_LAUNCH(V, OP_SUM);                           \
    } while (0)
    switch (vec) {
        case 4: POOL_OP(4); break;
        case 2: POOL_OP(2); break;
        default: POOL_OP(1); break;
    }
#undef POOL_OP
#undef POOL_LAUNCH
    GNNMP_LAUNCH_CHECK("segment_pool_kernel");
    return GNNMP_OK;
}

int gnnmp_segment_bounds(const void *seg_ids, int idx_bytes, int index_base, int64_t N, int64_t G, int64_t *ptr,
                         gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "segment_bounds: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "segment_bounds: index_base %d", index_base);
    if (N < 0 || G < 0) return fail(GNNMP_EINVAL, "segment_bounds: bad size");
    if (!ptr || (N > 0 && !seg_ids)) return fail(GNNMP_EINVAL, "segment_bounds: null pointer");
    segment_bounds_kernel<<<(unsigned)((N + 1 + 255) / 256), 256, 0, stream>>>(seg_ids, idx_bytes, index_base, N, G, ptr);
    GNNMP_LAUNCH_CHECK("segment_bounds_kernel");
    return GNNMP_OK;
}

int gnnmp_segment_pool_ptr_f32(int aggr, const float *x, const int64_t *seg_ptr, float *out, int64_t D, int64_t N, int64_t G,
                               gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (aggr < GNNMP_SUM || aggr > GNNMP_MIN) return fail(GNNMP_EINVAL, "segment_pool_ptr: bad aggr %d", aggr);
    if (N < 0 || G < 0 || D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "segment_pool_ptr: bad size");
    if (G == 0 || D == 0) return GNNMP_OK;
    if (!out || !seg_ptr || (N > 0 && !x)) return fail(GNNMP_EINVAL, "segment_pool_ptr: null pointer");
    const int vec = pick_vec(D, x, out);
    const int log2g = pick_log2g((D + vec - 1) / vec);
    const int64_t threads = G << log2g;
    const unsigned nb = (unsigned)((threads + 255) / 256);
    const int mean = aggr == GNNMP_MEAN;
#define POOLP_LAUNCH(V, O) segment_pool_ptr_kernel<V, O><<<nb, 256, 0, stream>>>(x, seg_ptr, out, (int)D, G, log2g, mean)
#define POOLP_OP(V)                                             \
    do {                                                        \
        if (aggr == GNNMP_MAX) POOLP_LAUNCH(V, OP_MAX);         \
        else if (aggr == GNNMP_MIN) POOLP_LAUNCH(V, OP_MIN);    \
        else POOLP_LAUNCH(V, OP_SUM);                           \
    } while (0)
    switch (vec) {
        case 4: POOLP_OP(4); break;
        case 2: POOLP_OP(2); break;
        default: POOLP_OP(1); break;
    }
#undef POOLP_OP
#undef POOLP_LAUNCH
    GNNMP_LAUNCH_CHECK("segment_pool_ptr_kernel");
    return GNNMP_OK;
}

}  // extern "C"
