// attn_backward.hip — pullback of the one-pass attention kernel for the GATv2 and dot-product (Transformer) logits
// (gat_fused.hip modes GNNMP_ATTN_GATV2 / GNNMP_ATTN_DOT; SURVEY.md §8f rank 1 applied to the rank-2 layers).  The GAT
// logit has its own, cheaper pullback in gat_backward.hip (its Jacobian is rank one per head).
//
//   l_ij = logit(Q_i, K_j)     α_ij = softmax_{j in N(i)} l_ij     o_i = Σ_j α_ij V_j          (per head)
// With Δ_i = dL/do_i:   g_ij = Δ_i . V_j,   D_i = Σ_j α_ij g_ij,   dl_ij = α_ij (g_ij - D_i),   dV_j = Σ_i α_ij Δ_i
//   GATV2  l = Σ_c a_c lrelu(z_c), z_c = Q_ic + K_jc, s_c = lrelu'(z_c):
//          dQ_ic = a_c Σ_j dl_ij s_c      dK_jc = a_c Σ_i dl_ij s_c      da_c = Σ_ij dl_ij lrelu(z_c)        (V = K)
//   DOT    l = Q_i . K_j / scale:   dQ_i = Σ_j dl_ij K_j / scale      dK_j = Σ_i dl_ij Q_i / scale
// α is rebuilt in registers from the forward's (m_i, den_i) statistics, as in gat_backward.hip.  dl_ij needs D_i, which is
// only known after the whole row: every Σ_j dl_ij f_j is therefore accumulated as  Σ α g f  -  D_i Σ α f  in one pass.
//   pass 1 (destination plan):  D_i, dQ_i (and the per-destination da terms), writes the line (m, 1/den, D) per (i, h)
//   pass 2 (plan of the reversed edges): per source j gathers Δ_i, Q_i and the line of every out-edge: dK_j, dV_j
// Long rows are chunked into virtual rows; partials are folded in chunk order.  No atomics.
#include <algorithm>

#include "common.h"

namespace gnnmp {

struct AttnBwdArgs {
    const uint32_t *rowptr;
    const int32_t *col;
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    const int32_t *long_rows, *long_cptr;
    int n_chunks, n_long, n_rows, long_thresh;
    const float *Q, *K, *V;   // [n_dst][D], [n_src][D], [n_src][D]
    const float *a;           // GATV2: [H][C]
    const float *dout;        // Δ [n_dst][D]
    const float *stats;       // [n_dst][H][2]
    float *line;              // [n_dst][H][4] = (m, 1/den, D, 0)
    float *dQ;                // [n_dst][D]
    float *dA;                // GATV2: per-destination Σ_j dl_ij lrelu(z_c)  [n_dst][D]
    float *dK, *dV;           // [n_src][D]  (GATV2: dK holds dK + dV, dV unused)
    float *partial;
    int H, C, D, log2g, lph, waves;
    float slope, scale;
    const int32_t *eid;       // DROP: slot -> original edge position of the plan the running pass walks
    DropArgs drop;
};

__device__ __forceinline__ float lrelu_a(float x, float slope) { return x > 0.0f ? x : x * slope; }

__device__ __forceinline__ bool attn_virtual_row(const AttnBwdArgs &a, int &v, bool &is_chunk, int &row, uint32_t &beg, uint32_t &end,
                                                 int &lig, int &gbase, int &G) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    G = 1 << a.log2g;
    lig = lane & (G - 1);
    gbase = lane - lig;
    const int grp = lane >> a.log2g;
    const int rpw = 64 >> a.log2g;
    const int64_t v64 = ((int64_t)blockIdx.x * a.waves + wave) * rpw + grp;
    if (v64 >= (int64_t)a.n_rows + a.n_chunks) return false;
    v = (int)v64;
    is_chunk = v < a.n_chunks;
    if (is_chunk) {
        row = a.chunk_row[v];
        beg = a.chunk_beg[v];
        end = a.chunk_end[v];
    } else {
        row = v - a.n_chunks;
        beg = a.rowptr[row];
        end = a.rowptr[row + 1];
        if (end - beg > a.long_thresh) return false;
    }
    return true;
}

// number of VEC-wide accumulators pass 1 keeps per lane: Σαgf and Σαf for f = s_c (+ f = lrelu(z_c) for da) | f = K_jc
template <int MODE>
struct Acc1 {
    static constexpr int N = MODE == GNNMP_ATTN_GATV2 ? 4 : 2;
};

// finalize pass 1 for one lane: S1 = D_i; acc = the N accumulators
template <int VEC, int MODE>
__device__ __forceinline__ void attn_dst_finalize(const AttnBwdArgs &a, int row, int f0, int h, float S1, float m, float rden,
                                                  bool nonempty, const float ca[VEC], float acc[][VEC]) {
    float dq[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        if (MODE == GNNMP_ATTN_GATV2) dq[q] = ca[q] * (acc[0][q] - S1 * acc[1][q]);
        if (MODE == GNNMP_ATTN_DOT) dq[q] = (acc[0][q] - S1 * acc[1][q]) / a.scale;
    }
    Vec<VEC>::store(a.dQ + (int64_t)row * a.D + f0, dq);
    if (MODE == GNNMP_ATTN_GATV2) {
        float da[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) da[q] = acc[2][q] - S1 * acc[3][q];
        Vec<VEC>::store(a.dA + (int64_t)row * a.D + f0, da);
    }
    if ((f0 % a.C) == 0) {
        float *ln = a.line + ((int64_t)row * a.H + h) * 4;
        ln[0] = m;
        ln[1] = nonempty ? rden : 0.0f;
        ln[2] = S1;
        ln[3] = 0.0f;
    }
}

template <int VEC, int U, int LPH, int MODE, bool DROP>
__global__ void __launch_bounds__(256) attn_bwd_dst_kernel(const AttnBwdArgs a) {
    constexpr int NA = Acc1<MODE>::N;
    int v, row, lig, gbase, G;
    uint32_t beg, end;
    bool is_chunk;
    if (!attn_virtual_row(a, v, is_chunk, row, beg, end, lig, gbase, G)) return;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;
    const int fc = active ? f0 : 0;
    const int h = fc / a.C;
    float ca[VEC], qi[VEC], di[VEC];
    Vec<VEC>::load(a.Q + (int64_t)row * a.D + fc, qi);
    Vec<VEC>::load(a.dout + (int64_t)row * a.D + fc, di);
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        ca[q] = 0.0f;
        qi[q] = active ? qi[q] : 0.0f;
        di[q] = active ? di[q] : 0.0f;
    }
    if (MODE == GNNMP_ATTN_GATV2 && active) {
        const float *ah = a.a + (int64_t)h * a.C + (fc - h * a.C);
#pragma unroll
        for (int q = 0; q < VEC; ++q) ca[q] = ah[q];
    }
    const float m = a.stats[((int64_t)row * a.H + h) * 2];
    const float rden = 1.0f / a.stats[((int64_t)row * a.H + h) * 2 + 1];

    float S1 = 0.0f, acc[NA][VEC];
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[k][q] = 0.0f;
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        const int ev = (DROP && p < end) ? a.eid[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float kv[U][VEC];                                   // K_j
            float vv[MODE == GNNMP_ATTN_DOT ? U : 1][VEC];      // V_j when it is a different array
            float kf[DROP ? U : 1];                             // keep_ij / (1 - p) of conv.jl:191's dropout: g_ij becomes kf g_ij
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (DROP) {
                    const uint32_t ej = (uint32_t)__shfl(ev, gbase + min(j + u, n - 1), 64);
                    kf[DROP ? u : 0] = drop_bits(a.drop.seed_lo, a.drop.seed_hi, ej, (uint32_t)h) >= a.drop.thr ? a.drop.inv : 0.0f;
                }
                const int cj = __shfl(c, gbase + min(j + u, n - 1), 64);
                Vec<VEC>::load(a.K + (int64_t)cj * a.D + fc, kv[u]);
                if (MODE == GNNMP_ATTN_DOT) Vec<VEC>::load(a.V + (int64_t)cj * a.D + fc, vv[MODE == GNNMP_ATTN_DOT ? u : 0]);
            }
            float l[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                l[u] = 0.0f;
                g[u] = 0.0f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const float vq = MODE == GNNMP_ATTN_DOT ? vv[MODE == GNNMP_ATTN_DOT ? u : 0][q] : kv[u][q];
                    g[u] = fmaf(di[q], vq, g[u]);
                    if (MODE == GNNMP_ATTN_GATV2) l[u] = fmaf(ca[q], lrelu_a(qi[q] + kv[u][q], a.slope), l[u]);
                    if (MODE == GNNMP_ATTN_DOT) l[u] = fmaf(qi[q], kv[u][q], l[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                l[u] = group_sum<LPH>(l[u], a.lph);
                g[u] = group_sum<LPH>(g[u], a.lph);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float lu = l[u];
                if (MODE == GNNMP_ATTN_DOT) lu = lu / a.scale;
                float al = expf(lu - m) * rden;
                al = (j + u < n) ? al : 0.0f;
                const float ag = DROP ? al * (kf[DROP ? u : 0] * g[u]) : al * g[u];
                S1 += ag;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    if (MODE == GNNMP_ATTN_GATV2) {
                        const float z = qi[q] + kv[u][q];
                        const float s = z > 0.0f ? 1.0f : a.slope;
                        const float lr = lrelu_a(z, a.slope);
                        acc[0][q] = fmaf(ag, s, acc[0][q]);
                        acc[1][q] = fmaf(al, s, acc[1][q]);
                        acc[2][q] = fmaf(ag, lr, acc[2][q]);
                        acc[3][q] = fmaf(al, lr, acc[3][q]);
                    }
                    if (MODE == GNNMP_ATTN_DOT) {
                        acc[0][q] = fmaf(ag, kv[u][q], acc[0][q]);
                        acc[1][q] = fmaf(al, kv[u][q], acc[1][q]);
                    }
                }
            }
        }
    }
    if (!active) return;
    if (is_chunk) {
        const int LN = a.D / VEC;
        float *pc = a.partial + (int64_t)v * ((int64_t)NA * a.D + LN);
#pragma unroll
        for (int k = 0; k < NA; ++k) Vec<VEC>::store(pc + (int64_t)k * a.D + f0, acc[k]);
        pc[(int64_t)NA * a.D + f0 / VEC] = S1;
        return;
    }
    attn_dst_finalize<VEC, MODE>(a, row, f0, h, S1, m, rden, end > beg, ca, acc);
}

template <int VEC, int MODE>
__global__ void __launch_bounds__(256) attn_bwd_dst_combine_kernel(const AttnBwdArgs a) {
    constexpr int NA = Acc1<MODE>::N;
    const int G = 1 << a.log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> a.log2g;
    if (r >= a.n_long) return;
    const int f0 = lig * VEC;
    if (f0 >= a.D) return;
    const int h = f0 / a.C;
    const int row = a.long_rows[r];
    const int LN = a.D / VEC;
    const int64_t S = (int64_t)NA * a.D + LN;
    float S1 = 0.0f, acc[NA][VEC];
#pragma unroll
    for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[k][q] = 0.0f;
    for (int c = a.long_cptr[r]; c < a.long_cptr[r + 1]; ++c) {
        const float *pc = a.partial + (int64_t)c * S;
        float t[NA][VEC];
#pragma unroll
        for (int k = 0; k < NA; ++k) Vec<VEC>::load(pc + (int64_t)k * a.D + f0, t[k]);
        S1 += pc[(int64_t)NA * a.D + f0 / VEC];
#pragma unroll
        for (int k = 0; k < NA; ++k)
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[k][q] += t[k][q];
    }
    float ca[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) ca[q] = 0.0f;
    if (MODE == GNNMP_ATTN_GATV2) {
        const float *ah = a.a + (int64_t)h * a.C + (f0 - h * a.C);
#pragma unroll
        for (int q = 0; q < VEC; ++q) ca[q] = ah[q];
    }
    const float m = a.stats[((int64_t)row * a.H + h) * 2];
    const float rden = 1.0f / a.stats[((int64_t)row * a.H + h) * 2 + 1];
    attn_dst_finalize<VEC, MODE>(a, row, f0, h, S1, m, rden, true, ca, acc);
}

template <int VEC, int MODE>
__device__ __forceinline__ void attn_src_store(const AttnBwdArgs &a, int row, int f0, float dk[VEC], float dv[VEC]) {
    if (MODE == GNNMP_ATTN_GATV2) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) dk[q] = dk[q] + dv[q];    // V is K: one gradient
        Vec<VEC>::store(a.dK + (int64_t)row * a.D + f0, dk);
    } else {
        Vec<VEC>::store(a.dK + (int64_t)row * a.D + f0, dk);
        Vec<VEC>::store(a.dV + (int64_t)row * a.D + f0, dv);
    }
}

template <int VEC, int U, int LPH, int MODE, bool DROP>
__global__ void __launch_bounds__(256) attn_bwd_src_kernel(const AttnBwdArgs a) {
    int v, row, lig, gbase, G;
    uint32_t beg, end;
    bool is_chunk;
    if (!attn_virtual_row(a, v, is_chunk, row, beg, end, lig, gbase, G)) return;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;
    const int fc = active ? f0 : 0;
    const int h = fc / a.C;
    float ca[VEC], kj[VEC], vj[VEC];
    Vec<VEC>::load(a.K + (int64_t)row * a.D + fc, kj);
    Vec<VEC>::load(a.V + (int64_t)row * a.D + fc, vj);
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        ca[q] = 0.0f;
        kj[q] = active ? kj[q] : 0.0f;
        vj[q] = active ? vj[q] : 0.0f;
    }
    if (MODE == GNNMP_ATTN_GATV2 && active) {
        const float *ah = a.a + (int64_t)h * a.C + (fc - h * a.C);
#pragma unroll
        for (int q = 0; q < VEC; ++q) ca[q] = ah[q];
    }
    float dk[VEC], dv[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) dk[q] = dv[q] = 0.0f;
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        const int ev = (DROP && p < end) ? a.eid[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float dd[U][VEC], qq[U][VEC];
            float4 ln[U];
            float kf[DROP ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (DROP) {    // (the transposed plan's slots carry the same original edge positions)
                    const uint32_t ej = (uint32_t)__shfl(ev, gbase + min(j + u, n - 1), 64);
                    kf[DROP ? u : 0] = drop_bits(a.drop.seed_lo, a.drop.seed_hi, ej, (uint32_t)h) >= a.drop.thr ? a.drop.inv : 0.0f;
                }
                const int ci = __shfl(c, gbase + min(j + u, n - 1), 64);
                Vec<VEC>::load(a.dout + (int64_t)ci * a.D + fc, dd[u]);
                Vec<VEC>::load(a.Q + (int64_t)ci * a.D + fc, qq[u]);
                ln[u] = *reinterpret_cast<const float4 *>(a.line + ((int64_t)ci * a.H + h) * 4);
            }
            float l[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                l[u] = 0.0f;
                g[u] = 0.0f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    g[u] = fmaf(dd[u][q], vj[q], g[u]);
                    if (MODE == GNNMP_ATTN_GATV2) l[u] = fmaf(ca[q], lrelu_a(qq[u][q] + kj[q], a.slope), l[u]);
                    if (MODE == GNNMP_ATTN_DOT) l[u] = fmaf(qq[u][q], kj[q], l[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                l[u] = group_sum<LPH>(l[u], a.lph);
                g[u] = group_sum<LPH>(g[u], a.lph);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float lu = l[u];
                if (MODE == GNNMP_ATTN_DOT) lu = lu / a.scale;
                float al = expf(lu - ln[u].x) * ln[u].y;
                al = (j + u < n) ? al : 0.0f;
                const float dl = al * ((DROP ? kf[DROP ? u : 0] * g[u] : g[u]) - ln[u].z);
                const float ak = DROP ? al * kf[DROP ? u : 0] : al;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    dv[q] = fmaf(ak, dd[u][q], dv[q]);
                    if (MODE == GNNMP_ATTN_GATV2) {
                        const float s = (qq[u][q] + kj[q]) > 0.0f ? 1.0f : a.slope;
                        dk[q] = fmaf(dl * s, ca[q], dk[q]);
                    }
                    if (MODE == GNNMP_ATTN_DOT) dk[q] = fmaf(dl, qq[u][q], dk[q]);
                }
            }
        }
    }
    if (!active) return;
    if (MODE == GNNMP_ATTN_DOT) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) dk[q] = dk[q] / a.scale;
    }
    if (is_chunk) {
        float *pc = a.partial + (int64_t)v * 2 * a.D;
        Vec<VEC>::store(pc + f0, dk);
        Vec<VEC>::store(pc + a.D + f0, dv);
        return;
    }
    attn_src_store<VEC, MODE>(a, row, f0, dk, dv);
}

template <int VEC, int MODE>
__global__ void __launch_bounds__(256) attn_bwd_src_combine_kernel(const AttnBwdArgs a) {
    const int G = 1 << a.log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> a.log2g;
    if (r >= a.n_long) return;
    const int f0 = lig * VEC;
    if (f0 >= a.D) return;
    const int row = a.long_rows[r];
    float dk[VEC], dv[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) dk[q] = dv[q] = 0.0f;
    for (int c = a.long_cptr[r]; c < a.long_cptr[r + 1]; ++c) {
        const float *pc = a.partial + (int64_t)c * 2 * a.D;
        float t1[VEC], t2[VEC];
        Vec<VEC>::load(pc + f0, t1);
        Vec<VEC>::load(pc + a.D + f0, t2);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            dk[q] += t1[q];
            dv[q] += t2[q];
        }
    }
    attn_src_store<VEC, MODE>(a, row, f0, dk, dv);
}

// da[h][c] = Σ_i dA[i][h*C + c]: slab partials, folded in slab order
__global__ void __launch_bounds__(256) attn_colsum_partial_kernel(const float *x, int64_t N, int D, int64_t R, float *part) {
    const int64_t r0 = (int64_t)blockIdx.x * R, r1 = min(N, r0 + R);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.0f;
        int64_t r = r0;
        for (; r + 8 <= r1; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(r + u) * D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + v[u];
        }
        for (; r < r1; ++r) acc = acc + x[r * D + d];
        part[(int64_t)blockIdx.x * D + d] = acc;
    }
}
__global__ void __launch_bounds__(256) attn_colsum_fold_kernel(const float *part, int nparts, int D, float *out) {
    __shared__ float red[256];
    const int d = blockIdx.x;
    float acc = 0.0f;
    for (int p = threadIdx.x; p < nparts; p += 256) acc = acc + part[(int64_t)p * D + d];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[d] = red[0];
}

static void fill_plan(AttnBwdArgs &g, const gnnmp_graph *p) {
    g.rowptr = p->rowptr;
    g.col = p->col;
    g.chunk_row = p->chunk_row;
    g.chunk_beg = p->chunk_beg;
    g.chunk_end = p->chunk_end;
    g.long_rows = p->long_rows;
    g.long_cptr = p->long_cptr;
    g.n_chunks = p->n_chunks;
    g.n_long = p->n_long;
    g.n_rows = (int)p->n_dst;
    g.long_thresh = p->long_thresh;
    g.partial = p->ws;
}

template <int VEC, int LPH, int MODE, bool DROP>
static int launch_attn_bwd(AttnBwdArgs g, gnnmp_graph *plan, gnnmp_graph *plan_t, float *da, hipStream_t stream) {
    const int G = 1 << g.log2g;
    const int rpw = 64 / G;
    g.waves = 1;
    fill_plan(g, plan);
    g.eid = plan->eid;
    {
        const int64_t nvirt = (int64_t)g.n_rows + g.n_chunks;
        const int64_t blocks = (nvirt + rpw - 1) / rpw;
        if (blocks > 0) {
            attn_bwd_dst_kernel<VEC, 4, LPH, MODE, DROP><<<(unsigned)blocks, 64, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("attn_bwd_dst_kernel");
        }
        if (g.n_long > 0) {
            const int64_t threads = (int64_t)g.n_long << g.log2g;
            attn_bwd_dst_combine_kernel<VEC, MODE><<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("attn_bwd_dst_combine_kernel");
        }
    }
    fill_plan(g, plan_t);
    g.eid = plan_t->eid;
    {
        const int64_t nvirt = (int64_t)g.n_rows + g.n_chunks;
        const int64_t blocks = (nvirt + rpw - 1) / rpw;
        if (blocks > 0) {
            attn_bwd_src_kernel<VEC, 4, LPH, MODE, DROP><<<(unsigned)blocks, 64, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("attn_bwd_src_kernel");
        }
        if (g.n_long > 0) {
            const int64_t threads = (int64_t)g.n_long << g.log2g;
            attn_bwd_src_combine_kernel<VEC, MODE><<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("attn_bwd_src_combine_kernel");
        }
    }
    if (MODE == GNNMP_ATTN_GATV2 && da) {
        const int64_t N = plan->n_dst;
        const int64_t R = std::max<int64_t>(256, (N + 2047) / 2048);
        const int nparts = (int)((N + R - 1) / R);
        attn_colsum_partial_kernel<<<nparts, 256, 0, stream>>>(g.dA, N, g.D, R, plan->ws);
        GNNMP_LAUNCH_CHECK("attn_colsum_partial_kernel");
        attn_colsum_fold_kernel<<<g.D, 256, 0, stream>>>(plan->ws, nparts, g.D, da);
        GNNMP_LAUNCH_CHECK("attn_colsum_fold_kernel");
    }
    return GNNMP_OK;
}

template <int MODE>
static int dispatch_attn_bwd(const AttnBwdArgs &g, int vec, int lph, gnnmp_graph *plan, gnnmp_graph *plan_t, float *da,
                             hipStream_t stream) {
    if (vec == 4) {
        switch (lph) {
            case 1: return launch_attn_bwd<4, 1, MODE, false>(g, plan, plan_t, da, stream);
            case 2: return launch_attn_bwd<4, 2, MODE, false>(g, plan, plan_t, da, stream);
            case 4: return launch_attn_bwd<4, 4, MODE, false>(g, plan, plan_t, da, stream);
            case 8: return launch_attn_bwd<4, 8, MODE, false>(g, plan, plan_t, da, stream);
            case 16: return launch_attn_bwd<4, 16, MODE, false>(g, plan, plan_t, da, stream);
            default: return launch_attn_bwd<4, 0, MODE, false>(g, plan, plan_t, da, stream);
        }
    }
    if (vec == 2) return launch_attn_bwd<2, 0, MODE, false>(g, plan, plan_t, da, stream);
    return launch_attn_bwd<1, 0, MODE, false>(g, plan, plan_t, da, stream);
}

}  // namespace gnnmp

using namespace gnnmp;

static int attn_conv_grad_impl(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, int mode, const float *Q, const float *K,
                               const float *V, const float *a, float negative_slope, float scale, float drop_p, uint64_t drop_seed,
                               const float *stats, const float *dout, float *line, float *dQ, float *dK, float *dV,
                               float *dA, float *da, int64_t H, int64_t C, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!(drop_p >= 0.0f && drop_p < 1.0f)) return fail(GNNMP_EINVAL, "attn_conv_grad: dropout probability %g outside [0, 1)", (double)drop_p);
    if (drop_p > 0.0f && mode != GNNMP_ATTN_GATV2)
        return fail(GNNMP_EUNSUPPORTED, "attn_conv_grad: attention dropout only on the GATv2 logit here (GAT: gnnmp_gat_conv_grad_drop_f32)");
    if (!plan || !plan_t) return fail(GNNMP_EINVAL, "attn_conv_grad: null plan");
    if (mode != GNNMP_ATTN_GATV2 && mode != GNNMP_ATTN_DOT)
        return fail(GNNMP_EUNSUPPORTED, "attn_conv_grad: mode %d (GAT has gnnmp_gat_conv_grad_f32; the cosine logit has no pullback yet)", mode);
    if (H <= 0 || C <= 0 || H * C > (1 << 20)) return fail(GNNMP_EINVAL, "attn_conv_grad: bad H/C");
    if (plan_t->n_dst != plan->n_src || plan_t->n_src != plan->n_dst || plan_t->n_total != plan->n_total)
        return fail(GNNMP_EINVAL, "attn_conv_grad: plan_t is not the transpose of plan");
    if (plan->n_dst == 0 && plan->n_src == 0) return GNNMP_OK;
    if (!V) V = K;
    if (mode == GNNMP_ATTN_GATV2 && V != K) return fail(GNNMP_EINVAL, "attn_conv_grad: GATV2 has V = K");
    if (!Q || !K || !stats || !dout || !line || !dQ || !dK || (mode == GNNMP_ATTN_DOT && !dV) ||
        (mode == GNNMP_ATTN_GATV2 && (!a || !dA)))
        return fail(GNNMP_EINVAL, "attn_conv_grad: null pointer");
    if ((reinterpret_cast<uintptr_t>(line) & 15) != 0) return fail(GNNMP_EINVAL, "attn_conv_grad: line must be 16-byte aligned");
    const int D = (int)(H * C);
    int vec = pick_vec(D, K, dK);
    const uintptr_t all = reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(dout) |
                          reinterpret_cast<uintptr_t>(dQ) | reinterpret_cast<uintptr_t>(dV) | reinterpret_cast<uintptr_t>(dA);
    if ((all & (4 * vec - 1)) != 0) vec = 1;
    while (vec > 1 && (C % vec) != 0) vec >>= 1;
    int lph = (int)(C / vec);
    const int lanes = D / vec;
    int log2g = 0;
    while ((1 << log2g) < lanes) ++log2g;
    if (H == 1 && lanes <= 64) lph = 1 << log2g;
    if (lanes > 64)
        return fail(GNNMP_EUNSUPPORTED, "attn_conv_grad: the feature row must fit one wave (H*C = %lld)", (long long)(H * C));
    const int NA = mode == GNNMP_ATTN_GATV2 ? 4 : 2;
    const int64_t R = std::max<int64_t>(256, (plan->n_dst + 2047) / 2048);
    const size_t colsum_need = (size_t)((plan->n_dst + R - 1) / R) * (size_t)D;
    if (int rc = ensure_workspace(plan, std::max((size_t)plan->n_chunks * ((size_t)NA * D + lanes), colsum_need))) return rc;
    if (plan_t->n_chunks > 0)
        if (int rc = ensure_workspace(plan_t, (size_t)plan_t->n_chunks * 2 * (size_t)D)) return rc;
    AttnBwdArgs g;
    g.Q = Q;
    g.K = K;
    g.V = V;
    g.a = a;
    g.dout = dout;
    g.stats = stats;
    g.line = line;
    g.dQ = dQ;
    g.dA = dA;
    g.dK = dK;
    g.dV = dV;
    g.H = (int)H;
    g.C = (int)C;
    g.D = D;
    g.log2g = log2g;
    g.lph = lph_code(lph, log2g);
    g.waves = 1;
    g.slope = negative_slope;
    g.scale = scale;
    g.eid = nullptr;
    g.drop = make_drop(drop_p, drop_seed);
    if (drop_p > 0.0f) {       // (the dropout variants walk the head butterfly with the run-time lane count: one instantiation per width)
        if (vec == 4) return launch_attn_bwd<4, 0, GNNMP_ATTN_GATV2, true>(g, plan, plan_t, da, stream);
        if (vec == 2) return launch_attn_bwd<2, 0, GNNMP_ATTN_GATV2, true>(g, plan, plan_t, da, stream);
        return launch_attn_bwd<1, 0, GNNMP_ATTN_GATV2, true>(g, plan, plan_t, da, stream);
    }
    if (mode == GNNMP_ATTN_GATV2) return dispatch_attn_bwd<GNNMP_ATTN_GATV2>(g, vec, lph, plan, plan_t, da, stream);
    return dispatch_attn_bwd<GNNMP_ATTN_DOT>(g, vec, lph, plan, plan_t, da, stream);
}

extern "C" int gnnmp_attn_conv_grad_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, int mode, const float *Q, const float *K,
                                        const float *V, const float *a, float negative_slope, float scale,
                                        const float *stats, const float *dout, float *line, float *dQ, float *dK, float *dV,
                                        float *dA, float *da, int64_t H, int64_t C, gnnmp_stream_t stream) {
    return attn_conv_grad_impl(plan, plan_t, mode, Q, K, V, a, negative_slope, scale, 0.0f, 0, stats, dout, line, dQ, dK, dV, dA, da, H, C,
                               stream);
}
extern "C" int gnnmp_attn_conv_grad_drop_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, int mode, const float *Q, const float *K,
                                             const float *V, const float *a, float negative_slope, float scale, float p,
                                             uint64_t seed, const float *stats, const float *dout, float *line, float *dQ,
                                             float *dK, float *dV, float *dA, float *da, int64_t H, int64_t C,
                                             gnnmp_stream_t stream) {
    return attn_conv_grad_impl(plan, plan_t, mode, Q, K, V, a, negative_slope, scale, p, seed, stats, dout, line, dQ, dK, dV, dA, da, H,
                               C, stream);
}
