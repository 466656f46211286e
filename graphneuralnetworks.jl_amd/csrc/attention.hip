// attention.hip — softmax_edge_neighbors (GNNlib/src/utils.jl:84-97) and the GATConv attention path
// (GNNlib/src/layers/conv.jl:136-141 + gat_message :152-167) fused into one pass over the destination rows.
//
// The reference materialises (2C,H,E'), (1,H,E') x6 and (C,H,E') x3 temporaries (13x the algorithmic traffic on the
// arxiv shape).  Here the attention logit of edge j->i splits into two node-level dot products
//     logit = leakyrelu( a[0:C,h] . Wx_i[:,h]  +  a[C:2C,h] . Wx_j[:,h] )
// computed once per node (gat_node_scores_kernel, (N,H) floats each), and the edge pass reads only score_src[col] (L2
// resident) and the 4*H*C-byte source row.  Per destination row the three reference steps run in the reference's order:
// max over the row, sum of exp(l - max) in edge order, then alpha = num / den (true division), beta = alpha * Wx_j
// (rounded), accumulated in edge order.
#include "common.h"

namespace gnnmp {

int run_softmax(gnnmp_graph_t *p, const float *e, float *alpha, int64_t D, float den_add, hipStream_t stream);  // propagate.hip

__device__ __forceinline__ float leaky_relu(float x, float slope) {
    return x > 0.0f ? x : x * slope;  // NNlib.leakyrelu
}

// ---- node-level scores ---------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) gat_node_scores_kernel(const float *Wx, const float *a,
                                                              float *sdst, float *ssrc, int64_t NH,
                                                              int H, int C) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NH) return;
    const int h = (int)(t % H);
    const float *row = Wx + t * C;
    const float *ad = a + (int64_t)h * 2 * C;
    const float *as = ad + C;
    float d = 0.0f, s = 0.0f;
    for (int c = 0; c < C; c += VEC) {
        float v[VEC];
        Vec<VEC>::load(row + c, v);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            d = d + ad[c + q] * v[q];
            s = s + as[c + q] * v[q];
        }
    }
    if (sdst) sdst[t] = d;
    if (ssrc) ssrc[t] = s;
}

// ---- fused edge softmax + weighted aggregate -----------------------------------------------------
struct GatArgs {
    const uint32_t *rowptr;
    const int32_t *col;
    const int32_t *eid;
    const float *Wx;     // [n_src][D], D = H*C
    const float *sdst;   // [n_dst][H]
    const float *ssrc;   // [n_src][H]
    float *out;          // [n_dst][D]
    float *alpha_out;    // [E'][H] or null
    const float *bias;   // [D] or null  (σ.(x .+ bias), conv.jl:147, concat case)
    int act;
    const int32_t *long_rows;
    int n_long;
    int H, C, D;
    int n_rows;
    int log2g;
    float slope;
    int long_thresh;
    int cpx;
    int waves;
};

// pass 1: running max of the row's logits (per lane: the lane's head)
template <int U>
__device__ __forceinline__ float gat_pass_max(const GatArgs &a, uint32_t beg, uint32_t end, int lig, int gbase,
                                              int G, int h, float sd) {
    float mx = -__builtin_inff();
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cj = __shfl(c, gbase + min(j + u, n - 1), 64);
                s[u] = a.ssrc[(int64_t)cj * a.H + h];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j + u < n) mx = jl_max(mx, leaky_relu(sd + s[u], a.slope));
        }
    }
    return mx;
}
// pass 2: denominator, summed in edge order
template <int U>
__device__ __forceinline__ float gat_pass_den(const GatArgs &a, uint32_t beg, uint32_t end, int lig, int gbase,
                                              int G, int h, float sd, float mx) {
    float den = 0.0f;
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cj = __shfl(c, gbase + min(j + u, n - 1), 64);
                s[u] = a.ssrc[(int64_t)cj * a.H + h];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j + u < n) den = den + expf(leaky_relu(sd + s[u], a.slope) - mx);
        }
    }
    return den;
}
// pass 3: acc += (num/den) * Wx[col]
template <int VEC, int U>
__device__ __forceinline__ void gat_pass_acc(const GatArgs &a, uint32_t beg, uint32_t end, int lig, int gbase,
                                             int G, int h, int f0, bool active, float sd, float mx,
                                             float den, float acc[VEC]) {
    const bool write_alpha = a.alpha_out != nullptr && active && (f0 % a.C == 0);
    for (uint32_t base = beg; base < end; base += G) {
        const uint32_t p = base + lig;
        int c = 0, e = 0;                      // e: original edge position, an unsigned 32-bit value carried in an int
        if (p < end) {
            c = a.col[p];
            if (a.alpha_out) e = a.eid[p];
        }
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float v[U][VEC];
            float s[U];
            int ej[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src_lane = gbase + min(j + u, n - 1);
                const int cj = __shfl(c, src_lane, 64);
                ej[u] = a.alpha_out ? __shfl(e, src_lane, 64) : 0;
                s[u] = a.ssrc[(int64_t)cj * a.H + h];
                if (active && (j + u < n)) {
                    Vec<VEC>::load(a.Wx + (int64_t)cj * a.D + f0, v[u]);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[u][q] = 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < n) {
                    const float num = expf(leaky_relu(sd + s[u], a.slope) - mx);
                    const float al = num / den;
                    if (write_alpha) a.alpha_out[(int64_t)(uint32_t)ej[u] * a.H + h] = al;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float b = al * v[u][q];  // β = α .* Wxj   (conv.jl:140)
                        acc[q] = acc[q] + b;           // aggregate_neighbors(g, +, β)
                    }
                }
            }
        }
    }
}

template <int VEC>
__device__ __forceinline__ void gat_store(const GatArgs &a, int row, int f0, bool active, float acc[VEC]) {
    if (!active) return;
    if (a.bias) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] + a.bias[f0 + q];
    }
    if (a.act == GNNMP_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] < 0.0f ? 0.0f : acc[q];  // NNlib.relu = ifelse(x < 0, 0, x): NaN-preserving
    }
    Vec<VEC>::store(a.out + (int64_t)row * a.D + f0, acc);
}

template <int VEC, int U>
__global__ void __launch_bounds__(256) gat_rows_kernel(const GatArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = 1 << a.log2g;
    const int lig = lane & (G - 1);
    const int grp = lane >> a.log2g;
    const int gbase = lane - lig;
    const int rpw = 64 >> a.log2g;
    const int chunk = a.cpx ? xcd_remap(blockIdx.x, a.cpx, 1) : (int)blockIdx.x;
    const int64_t row64 = ((int64_t)chunk * a.waves + wave) * rpw + grp;
    if (row64 >= a.n_rows) return;
    const int row = (int)row64;
    const int f0 = ((int)blockIdx.y * G + lig) * VEC;
    const bool active = f0 < a.D;
    const int h = active ? f0 / a.C : 0;
    const uint32_t beg = a.rowptr[row];
    const uint32_t end = a.rowptr[row + 1];
    if (end - beg > a.long_thresh) return;
    const float sd = a.sdst[(int64_t)row * a.H + h];
    const float mx = gat_pass_max<U>(a, beg, end, lig, gbase, G, h, sd);
    const float den = gat_pass_den<U>(a, beg, end, lig, gbase, G, h, sd, mx);
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.0f;
    gat_pass_acc<VEC, U>(a, beg, end, lig, gbase, G, h, f0, active, sd, mx, den, acc);
    gat_store<VEC>(a, row, f0, active, acc);
}

// one 1024-thread workgroup per long row; fixed partition, fixed combine order (deterministic).
template <int VEC, int U>
__global__ void __launch_bounds__(1024) gat_long_rows_kernel(const GatArgs a) {
    extern __shared__ float lds[];  // [NG][G*VEC]
    const int lane = threadIdx.x & 63;
    const int G = 1 << a.log2g;
    const int lig = lane & (G - 1);
    const int gbase = lane - lig;
    const int q = threadIdx.x >> a.log2g;
    const int NG = 1024 >> a.log2g;
    const int row = a.long_rows[blockIdx.x];
    const int f0 = ((int)blockIdx.y * G + lig) * VEC;
    const bool active = f0 < a.D;
    const int h = active ? f0 / a.C : 0;
    const uint32_t beg = a.rowptr[row];
    const uint32_t end = a.rowptr[row + 1];
    const uint32_t len = end - beg;
    const uint32_t part = (len + NG - 1) / NG;
    const uint32_t pb = (uint32_t)min((uint64_t)beg + (uint64_t)q * part, (uint64_t)end);
    const uint32_t pe = (uint32_t)min((uint64_t)pb + part, (uint64_t)end);
    const float sd = a.sdst[(int64_t)row * a.H + h];

    float mx = gat_pass_max<U>(a, pb, pe, lig, gbase, G, h, sd);
    lds[q * G + lig] = mx;
    __syncthreads();
    mx = -__builtin_inff();
    for (int k = 0; k < NG; ++k) mx = jl_max(mx, lds[k * G + lig]);
    __syncthreads();

    float den = gat_pass_den<U>(a, pb, pe, lig, gbase, G, h, sd, mx);
    lds[q * G + lig] = den;
    __syncthreads();
    den = 0.0f;
    for (int k = 0; k < NG; ++k) den = den + lds[k * G + lig];
    __syncthreads();

    float acc[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
    gat_pass_acc<VEC, U>(a, pb, pe, lig, gbase, G, h, f0, active, sd, mx, den, acc);
#pragma unroll
    for (int t = 0; t < VEC; ++t) lds[(q * G + lig) * VEC + t] = acc[t];
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] = 0.0f;
        for (int k = 0; k < NG; ++k) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[t] = acc[t] + lds[(k * G + lig) * VEC + t];
        }
        gat_store<VEC>(a, row, f0, active, acc);
    }
}

template <int VEC>
static int launch_gat(GatArgs a, hipStream_t stream) {
    const int G = 1 << a.log2g;
    const int rpw = 64 / G;
    int waves = knob(KNOB_BLOCK_WAVES);
    if (waves < 1 || waves > 4) waves = 1;   // auto (see gat_fused.hip)
    a.waves = waves;
    const int rows_per_block = rpw * waves;
    const int64_t chunks = ((int64_t)a.n_rows + rows_per_block - 1) / rows_per_block;
    const int lanes_needed = (a.D + VEC - 1) / VEC;
    const int tiles = (lanes_needed + G - 1) / G;
    if (chunks > 0) {
        int64_t gx = chunks;
        a.cpx = 0;
        if (use_xcd_remap(a.n_rows, a.D, chunks)) {
            a.cpx = (int)((chunks + 7) / 8);
            gx = (int64_t)a.cpx * 8;
        }
        dim3 grid((unsigned)gx, (unsigned)tiles);
        gat_rows_kernel<VEC, 4><<<grid, 64 * waves, 0, stream>>>(a);
        GNNMP_LAUNCH_CHECK("gat_rows_kernel");
    }
    if (a.n_long > 0) {
        dim3 grid((unsigned)a.n_long, (unsigned)tiles);
        gat_long_rows_kernel<VEC, 4><<<grid, 1024, sizeof(float) * 1024 * VEC, stream>>>(a);
        GNNMP_LAUNCH_CHECK("gat_long_rows_kernel");
    }
    return GNNMP_OK;
}

// ---- standalone softmax_edge_neighbors: one thread per (destination, channel) ---------------------
__global__ void __launch_bounds__(256) bias_act_kernel(const float *x, const float *bias, int act,
                                                       float *out, int64_t total, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = x[i];
    if (bias) v = v + bias[i % D];
    if (act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
    if (act == GNNMP_ACT_SOFTPLUS) v = log1pf(expf(-fabsf(v))) + (v < 0.0f ? 0.0f : v);   // NNlib.softplus
    if (act == GNNMP_ACT_TANH) v = tanhf(v);
    if (act == GNNMP_ACT_SWISH) {                       // NNlib.swish = x * sigmoid(x)
        const float t = expf(-fabsf(v));
        v = v * (v >= 0.0f ? 1.0f / (1.0f + t) : t / (1.0f + t));
    }
    out[i] = v;
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_gat_node_scores_f32(const float *Wx, const float *a, float *score_dst, float *score_src,
                              int64_t N, int64_t H, int64_t C, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || H <= 0 || C <= 0) return fail(GNNMP_EINVAL, "gat_node_scores: bad size");
    if (N == 0) return GNNMP_OK;
    if (!Wx || !a || (!score_dst && !score_src)) return fail(GNNMP_EINVAL, "gat_node_scores: null pointer");
    const int64_t NH = N * H;
    const unsigned nb = (unsigned)((NH + 255) / 256);
    const bool v4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(Wx) & 15) == 0);
    if (v4)
        gat_node_scores_kernel<4><<<nb, 256, 0, stream>>>(Wx, a, score_dst, score_src, NH, (int)H, (int)C);
    else
        gat_node_scores_kernel<1><<<nb, 256, 0, stream>>>(Wx, a, score_dst, score_src, NH, (int)H, (int)C);
    GNNMP_LAUNCH_CHECK("gat_node_scores_kernel");
    return GNNMP_OK;
}

int gnnmp_gat_aggregate_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *score_dst,
                            const float *score_src, float negative_slope, const float *bias, int act,
                            float *out, float *alpha_out, int64_t H, int64_t C,
                            gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!plan) return fail(GNNMP_EINVAL, "gat_aggregate: null plan");
    if (H <= 0 || C <= 0 || H * C > (1 << 20)) return fail(GNNMP_EINVAL, "gat_aggregate: bad H/C");
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "gat_aggregate: bad act %d", act);
    if (plan->n_dst == 0) return GNNMP_OK;
    if (!out || !score_dst || (plan->n_total > 0 && (!Wx_src || !score_src)))
        return fail(GNNMP_EINVAL, "gat_aggregate: null pointer");
    GatArgs a;
    a.rowptr = plan->rowptr;
    a.col = plan->col;
    a.eid = plan->eid;
    a.Wx = Wx_src;
    a.sdst = score_dst;
    a.ssrc = score_src;
    a.out = out;
    a.alpha_out = alpha_out;
    a.bias = bias;
    a.act = act;
    a.long_rows = plan->long_rows;
    a.n_long = plan->n_long;
    a.H = (int)H;
    a.C = (int)C;
    a.D = (int)(H * C);
    a.n_rows = (int)plan->n_dst;
    a.slope = negative_slope;
    a.long_thresh = plan->long_thresh;
    a.cpx = 0;
    a.waves = 4;
    // the lane's VEC features must lie inside one head
    int vec = pick_vec(a.D, Wx_src, out);
    while (vec > 1 && (C % vec) != 0) vec >>= 1;
    a.log2g = pick_log2g((a.D + vec - 1) / vec);
    switch (vec) {
        case 4: return launch_gat<4>(a, stream);
        case 2: return launch_gat<2>(a, stream);
        default: return launch_gat<1>(a, stream);
    }
}

int gnnmp_bias_act_f32(const float *x, const float *bias, int act, float *out, int64_t N, int64_t D,
                       gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "bias_act: bad size");
    if (act < GNNMP_ACT_IDENTITY || act > GNNMP_ACT_SWISH) return fail(GNNMP_EINVAL, "bias_act: bad act %d", act);
    if (N == 0 || D == 0) return GNNMP_OK;
    if (!x || !out) return fail(GNNMP_EINVAL, "bias_act: null pointer");
    const int64_t total = N * D;
    bias_act_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(x, bias, act, out, total, (int)D);
    GNNMP_LAUNCH_CHECK("bias_act_kernel");
    return GNNMP_OK;
}

int gnnmp_edge_softmax_f32(gnnmp_graph_t *plan, const float *logits, float *alpha, int64_t H,
                           gnnmp_stream_t stream_) {
    if (!plan) return fail(GNNMP_EINVAL, "edge_softmax: null plan");
    if (H <= 0 || H > (1 << 20)) return fail(GNNMP_EINVAL, "edge_softmax: bad H");
    if (plan->n_dst == 0 || plan->n_total == 0) return GNNMP_OK;
    if (!logits || !alpha) return fail(GNNMP_EINVAL, "edge_softmax: null pointer");
    if (plan->self_loops) return fail(GNNMP_EINVAL, "edge_softmax: the plan must not add self loops (e has one row per edge of g)");
    return run_softmax(plan, logits, alpha, H, 0.0f, (hipStream_t)stream_);
}

int gnnmp_segment_softmax_f32(gnnmp_graph_t *plan, const float *x, float *out, int64_t D, float den_add,
                              gnnmp_stream_t stream_) {
    if (!plan) return fail(GNNMP_EINVAL, "segment_softmax: null plan");
    if (D <= 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "segment_softmax: bad D");
    if (plan->n_dst == 0 || plan->n_total == 0) return GNNMP_OK;
    if (!x || !out) return fail(GNNMP_EINVAL, "segment_softmax: null pointer");
    if (plan->self_loops) return fail(GNNMP_EINVAL, "segment_softmax: the plan must not add self loops");
    return run_softmax(plan, x, out, D, den_add, (hipStream_t)stream_);
}

}  // extern "C"
