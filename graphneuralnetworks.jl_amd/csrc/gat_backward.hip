// gat_backward.hip — adjoint of the one-pass GATConv attention path (gat_fused.hip; SURVEY.md §8f rank 1, "next").
// What Zygote differentiates in the reference (GNNlib/src/layers/conv.jl:136-141,152-167 + utils.jl:84-97), per head h:
//   z_ij = a_d.Wx_i + a_s.Wx_j     l_ij = leakyrelu(z_ij)     α_ij = softmax_{j in N(i)}(l_ij)     o_i = Σ_j α_ij Wx_j
// With Δ_i = dL/do_i [C] the chain rule collapses to
//   g_ij   = Δ_i . Wx_j                      D_i   = Σ_j α_ij g_ij                 (the softmax rrule's  Σ α dα)
//   dz_ij  = α_ij (g_ij - D_i) lrelu'(z_ij)  dsd_i = Σ_j dz_ij                     dss_j = Σ_i dz_ij
//   dWx_j  = Σ_i α_ij Δ_i  +  a_s dss_j  +  a_d dsd_j                             da_d = Σ_i dsd_i Wx_i,  da_s = Σ_j dss_j Wx_j
// The α are never stored (E' x H floats): the forward saves (m_i, den_i) per destination and head (gnnmp_gat_conv_stats_f32)
// and both kernels below rebuild α_ij = exp(l_ij - m_i) / den_i in registers, the same way the forward builds the logits.
//
//   gat_bwd_dst_kernel   destination-sorted plan (the forward's): one pass gathering Wx_j, three running sums per (i,h)
//                        S1 = Σ α g, S2 = Σ α g s, S3 = Σ α s  ->  D_i = S1,  dsd_i = S2 - S1 S3;  writes the 16-byte line
//                        (sd_i, m_i, 1/den_i, D_i) that the second kernel gathers per edge
//   gat_bwd_src_kernel   source-sorted plan (the transposed one): row j gathers Δ_i + that line for every edge that
//                        leaves j, accumulates Σ α Δ_i and dss_j, adds the two rank-one terms and stores dWx_j
//   gat_wcolsum_*        da: two weighted column sums over the nodes, deterministic slab partials
// Long rows are chunked into virtual rows exactly like the forward kernels; partials are folded in chunk order.
#include <algorithm>

#include "common.h"

namespace gnnmp {

struct GatBwdArgs {
    const uint32_t *rowptr;
    const int32_t *col;
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    const int32_t *long_rows, *long_cptr;
    int n_chunks, n_long, n_rows, long_thresh;
    const float *Wx_src;  // [n_src][D]
    const float *Wx_dst;  // [n_dst][D]
    const float *a;       // [H][2C]
    const float *dout;    // Δ [n_dst][D]
    const float *stats;   // [n_dst][H][2]  (m, den) from the forward
    float *line;          // [n_dst][H][4]  (sd, m, 1/den, D)
    float *dsd;           // [n_dst][H]
    float *dss;           // [n_src][H]
    float *dWx;           // [n_src][D]
    float *partial;
    int fold_dst;         // add a_d * dsd_j into dWx_j (non-bipartite: Wx_dst is Wx_src)
    int H, C, D, log2g, lph, waves;
    float slope;
    const int32_t *eid;   // DROP: plan slot -> original edge position (of the plan the running pass walks)
    DropArgs drop;
    // gnnmp_gat_conv_grad2_f32: what the training forward saved (gnnmp_gat_conv_train_f32) — the destination side then needs no edge pass
    const float *outk;    // [n_dst][D] the forward's output act(o + bias) with act in {identity, relu}
    const float *bias;    // [D] or null
    const float *oplus;   // [n_dst][D]
    const float *pplus;   // [n_dst][H]
};

__device__ __forceinline__ float lrelu_b(float x, float slope) { return x > 0.0f ? x : x * slope; }

// decode the virtual row of this lane group; false = nothing to do
__device__ __forceinline__ bool virtual_row(const GatBwdArgs &a, int &v, bool &is_chunk, int &row, uint32_t &beg, uint32_t &end,
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

template <int VEC, int U, int LPH, bool DROP>
__global__ void __launch_bounds__(256) gat_bwd_dst_kernel(const GatBwdArgs a) {
    int v, row, lig, gbase, G;
    uint32_t beg, end;
    bool is_chunk;
    if (!virtual_row(a, v, is_chunk, row, beg, end, lig, gbase, G)) return;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;
    const int fc = active ? f0 : 0;   // idle lanes (D/VEC not a power of two) shadow lane 0 with zero coefficients
    const int h = fc / a.C;
    float ad[VEC], as[VEC], vi[VEC], di[VEC];
    {
        const float *ah = a.a + (int64_t)h * 2 * a.C + (fc - h * a.C);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            ad[q] = active ? ah[q] : 0.0f;
            as[q] = active ? ah[a.C + q] : 0.0f;
        }
    }
    Vec<VEC>::load(a.Wx_dst + (int64_t)row * a.D + fc, vi);
    Vec<VEC>::load(a.dout + (int64_t)row * a.D + fc, di);
#pragma unroll
    for (int q = 0; q < VEC; ++q) di[q] = active ? di[q] : 0.0f;   // H = 1: idle lanes sit inside the head's butterfly
    float sd = 0.0f;
#pragma unroll
    for (int q = 0; q < VEC; ++q) sd = fmaf(ad[q], vi[q], sd);
    sd = group_sum<LPH>(sd, a.lph);
    const float m = a.stats[((int64_t)row * a.H + h) * 2];
    const float den = a.stats[((int64_t)row * a.H + h) * 2 + 1];
    const float rden = 1.0f / den;

    // branch-free body: the U loads, the 2U dot products, their butterflies and the U exponentials are independent
    // chains the scheduler interleaves; slots past the end of the row re-read the last edge and get α = 0
    float S1 = 0.0f, S2 = 0.0f, S3 = 0.0f;
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        const int ev = (DROP && p < end) ? a.eid[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float w[U][VEC];
            float kf[DROP ? U : 1];      // keep_ij / (1 - p) of conv.jl:139's dropout (common.h drop_bits): g_ij becomes kf g_ij
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cj = __shfl(c, gbase + min(j + u, n - 1), 64);
                Vec<VEC>::load(a.Wx_src + (int64_t)cj * a.D + fc, w[u]);
                if (DROP) {
                    const uint32_t ej = (uint32_t)__shfl(ev, gbase + min(j + u, n - 1), 64);
                    kf[DROP ? u : 0] = drop_bits(a.drop.seed_lo, a.drop.seed_hi, ej, (uint32_t)h) >= a.drop.thr ? a.drop.inv : 0.0f;
                }
            }
            float d[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = 0.0f;
                g[u] = 0.0f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    d[u] = fmaf(as[q], w[u][q], d[u]);
                    g[u] = fmaf(di[q], w[u][q], g[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = group_sum<LPH>(d[u], a.lph);
                g[u] = group_sum<LPH>(g[u], a.lph);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float z = sd + d[u];
                float al = expf(lrelu_b(z, a.slope) - m) * rden;
                al = (j + u < n) ? al : 0.0f;
                const float s = z > 0.0f ? 1.0f : a.slope;
                const float ag = DROP ? al * (kf[DROP ? u : 0] * g[u]) : al * g[u];
                S1 += ag;
                S2 = fmaf(ag, s, S2);
                S3 = fmaf(al, s, S3);
            }
        }
    }
    if (!active || (f0 % a.C) != 0) return;
    if (is_chunk) {
        float *pc = a.partial + ((int64_t)v * a.H + h) * 4;
        pc[0] = S1;
        pc[1] = S2;
        pc[2] = S3;
        pc[3] = sd;
        return;
    }
    float *ln = a.line + ((int64_t)row * a.H + h) * 4;
    ln[0] = sd;
    ln[1] = m;
    ln[2] = end > beg ? rden : 0.0f;
    ln[3] = S1;
    a.dsd[(int64_t)row * a.H + h] = S2 - S1 * S3;
}

// The destination side WITHOUT an edge pass (round 4).  With o_i = Σ_j α_ij Wx_j (the forward's output before bias and σ):
//   D_i = Σ_j α_ij g_ij = Δ_i . o_i                                   (g_ij = Δ_i . Wx_j is linear in Wx_j)
//   dsd_i = Σ_j α_ij (g_ij - D_i) lrelu'(z_ij),  lrelu' = slope + (1 - slope) [z_ij > 0]
//         = slope D_i + (1 - slope) Δ_i . o+_i - D_i (slope + (1 - slope) P_i) = (1 - slope) (Δ_i . o+_i - D_i P_i)
// o+_i / P_i = the same sums over the edges with z_ij > 0 only: the training forward accumulates them next to o_i (gat_fused.hip,
// ATTN_GAT_PLUS).  o_i is read back from the forward's output act(o_i + b): on entries that relu switched off Δ is zero (the caller
// passes dL/d(o + b)), elsewhere o = out - b.  One lane group per row, three row loads, three butterflies: 3.8 GB instead of the
// 5.7 ms gather of every Wx_j.
template <int VEC, int LPH>
__global__ void __launch_bounds__(256) gat_bwd_node_kernel(const GatBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = 1 << a.log2g;
    const int lig = lane & (G - 1);
    const int grp = lane >> a.log2g;
    const int rpw = 64 >> a.log2g;
    const int64_t r64 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * rpw + grp;
    if (r64 >= a.n_rows) return;
    const int row = (int)r64;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;
    const int fc = active ? f0 : 0;
    const int h = fc / a.C;
    float vi[VEC], di[VEC], ok[VEC], op[VEC];
    Vec<VEC>::load(a.Wx_dst + (int64_t)row * a.D + fc, vi);
    Vec<VEC>::load(a.dout + (int64_t)row * a.D + fc, di);
    Vec<VEC>::load(a.outk + (int64_t)row * a.D + fc, ok);
    Vec<VEC>::load(a.oplus + (int64_t)row * a.D + fc, op);
    const float *ah = a.a + (int64_t)h * 2 * a.C + (fc - h * a.C);
    float sd = 0.0f, Dv = 0.0f, Tv = 0.0f;
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        const float d = active ? di[q] : 0.0f;
        const float o = ok[q] - (a.bias ? a.bias[fc + q] : 0.0f);
        sd = fmaf(active ? ah[q] : 0.0f, vi[q], sd);
        Dv = fmaf(d, o, Dv);
        Tv = fmaf(d, op[q], Tv);
    }
    sd = group_sum<LPH>(sd, a.lph);
    Dv = group_sum<LPH>(Dv, a.lph);
    Tv = group_sum<LPH>(Tv, a.lph);
    if (!active || (f0 % a.C) != 0) return;
    const float m = a.stats[((int64_t)row * a.H + h) * 2];
    const float den = a.stats[((int64_t)row * a.H + h) * 2 + 1];
    const bool any = a.rowptr[row + 1] > a.rowptr[row];
    float *ln = a.line + ((int64_t)row * a.H + h) * 4;
    ln[0] = sd;
    ln[1] = m;
    ln[2] = any ? 1.0f / den : 0.0f;
    ln[3] = any ? Dv : 0.0f;
    a.dsd[(int64_t)row * a.H + h] = any ? (1.0f - a.slope) * (Tv - Dv * a.pplus[(int64_t)row * a.H + h]) : 0.0f;
}

__global__ void __launch_bounds__(256) gat_bwd_dst_combine_kernel(const GatBwdArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.n_long * a.H) return;
    const int r = (int)(i / a.H), h = (int)(i - (int64_t)r * a.H);
    const int row = a.long_rows[r];
    const int c0 = a.long_cptr[r], c1 = a.long_cptr[r + 1];
    float S1 = 0.0f, S2 = 0.0f, S3 = 0.0f;
    for (int c = c0; c < c1; ++c) {
        const float *pc = a.partial + ((int64_t)c * a.H + h) * 4;
        S1 += pc[0];
        S2 += pc[1];
        S3 += pc[2];
    }
    float *ln = a.line + ((int64_t)row * a.H + h) * 4;
    ln[0] = a.partial[((int64_t)c0 * a.H + h) * 4 + 3];
    ln[1] = a.stats[((int64_t)row * a.H + h) * 2];
    ln[2] = 1.0f / a.stats[((int64_t)row * a.H + h) * 2 + 1];
    ln[3] = S1;
    a.dsd[(int64_t)row * a.H + h] = S2 - S1 * S3;
}

template <int VEC>
__device__ __forceinline__ void gat_bwd_src_store(const GatBwdArgs &a, int row, int f0, int h, const float as[VEC],
                                                  const float ad[VEC], float acc[VEC], float dss) {
    float extra = 0.0f;
    if (a.fold_dst) extra = a.dsd[(int64_t)row * a.H + h];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        acc[q] = fmaf(as[q], dss, acc[q]);
        if (a.fold_dst) acc[q] = fmaf(ad[q], extra, acc[q]);
    }
    Vec<VEC>::store(a.dWx + (int64_t)row * a.D + f0, acc);
    if ((f0 % a.C) == 0) a.dss[(int64_t)row * a.H + h] = dss;
}

template <int VEC, int U, int LPH, bool DROP>
__global__ void __launch_bounds__(256) gat_bwd_src_kernel(const GatBwdArgs a) {
    int v, row, lig, gbase, G;
    uint32_t beg, end;
    bool is_chunk;
    if (!virtual_row(a, v, is_chunk, row, beg, end, lig, gbase, G)) return;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;
    const int fc = active ? f0 : 0;
    const int h = fc / a.C;
    float ad[VEC], as[VEC], wj[VEC];
    {
        const float *ah = a.a + (int64_t)h * 2 * a.C + (fc - h * a.C);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            ad[q] = active ? ah[q] : 0.0f;
            as[q] = active ? ah[a.C + q] : 0.0f;
        }
    }
    Vec<VEC>::load(a.Wx_src + (int64_t)row * a.D + fc, wj);
#pragma unroll
    for (int q = 0; q < VEC; ++q) wj[q] = active ? wj[q] : 0.0f;   // H = 1: idle lanes sit inside the head's butterfly
    float ss = 0.0f;
#pragma unroll
    for (int q = 0; q < VEC; ++q) ss = fmaf(as[q], wj[q], ss);
    ss = group_sum<LPH>(ss, a.lph);

    float acc[VEC], dss = 0.0f;
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.0f;
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        const int ev = (DROP && p < end) ? a.eid[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float dv[U][VEC];
            float4 ln[U];
            float kf[DROP ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ci = __shfl(c, gbase + min(j + u, n - 1), 64);
                Vec<VEC>::load(a.dout + (int64_t)ci * a.D + fc, dv[u]);
                ln[u] = *reinterpret_cast<const float4 *>(a.line + ((int64_t)ci * a.H + h) * 4);
                if (DROP) {    // (the transposed plan's slots carry the same original edge positions)
                    const uint32_t ej = (uint32_t)__shfl(ev, gbase + min(j + u, n - 1), 64);
                    kf[DROP ? u : 0] = drop_bits(a.drop.seed_lo, a.drop.seed_hi, ej, (uint32_t)h) >= a.drop.thr ? a.drop.inv : 0.0f;
                }
            }
            float g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = 0.0f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) g[u] = fmaf(dv[u][q], wj[q], g[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) g[u] = group_sum<LPH>(g[u], a.lph);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float z = ln[u].x + ss;
                float al = expf(lrelu_b(z, a.slope) - ln[u].y) * ln[u].z;
                al = (j + u < n) ? al : 0.0f;
                const float s = z > 0.0f ? 1.0f : a.slope;
                const float gk = DROP ? kf[DROP ? u : 0] * g[u] : g[u];
                dss = fmaf(al * (gk - ln[u].w), s, dss);
                const float ak = DROP ? al * kf[DROP ? u : 0] : al;
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = fmaf(ak, dv[u][q], acc[q]);
            }
        }
    }
    if (!active) return;
    if (is_chunk) {
        const int LN = a.D / VEC;
        float *pc = a.partial + (int64_t)v * (a.D + LN);
        Vec<VEC>::store(pc + f0, acc);
        pc[a.D + f0 / VEC] = dss;
        return;
    }
    gat_bwd_src_store<VEC>(a, row, f0, h, as, ad, acc, dss);
}

template <int VEC>
__global__ void __launch_bounds__(256) gat_bwd_src_combine_kernel(const GatBwdArgs a) {
    const int G = 1 << a.log2g;
    const int lig = threadIdx.x & (G - 1);
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> a.log2g;
    if (r >= a.n_long) return;
    const int f0 = lig * VEC;
    if (f0 >= a.D) return;
    const int h = f0 / a.C;
    const int row = a.long_rows[r];
    const int c0 = a.long_cptr[r], c1 = a.long_cptr[r + 1];
    const int LN = a.D / VEC;
    const int64_t S = a.D + LN;
    constexpr int CB = 8;
    float acc[VEC], dss = 0.0f;
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = 0.0f;
    for (int c = c0; c < c1; c += CB) {
        float sv[CB], pv[CB][VEC];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const float *pc = a.partial + (int64_t)min(c + u, c1 - 1) * S;
            sv[u] = pc[a.D + f0 / VEC];
            Vec<VEC>::load(pc + f0, pv[u]);
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (c + u < c1) {
                dss += sv[u];
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] += pv[u][q];
            }
        }
    }
    float ad[VEC], as[VEC];
    const float *ah = a.a + (int64_t)h * 2 * a.C + (f0 - h * a.C);
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        ad[q] = ah[q];
        as[q] = ah[a.C + q];
    }
    gat_bwd_src_store<VEC>(a, row, f0, h, as, ad, acc, dss);
}

// dWx_dst[i][h][c] = a_d[h][c] * dsd[i][h]   (bipartite layers only: otherwise folded into gat_bwd_src_kernel)
__global__ void __launch_bounds__(256) gat_bwd_dst_rows_kernel(const float *a, const float *dsd, float *out, int64_t n,
                                                               int H, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D = H * C;
    if (i >= n * D) return;
    const int64_t r = i / D;
    const int f = (int)(i - r * D), h = f / C;
    out[i] = a[(int64_t)h * 2 * C + (f - h * C)] * dsd[r * H + h];
}

// stage 1 of da: block b sums s[r][h] * x[r][h*C + c] over its slab of rows
__global__ void __launch_bounds__(256) gat_wcolsum_partial_kernel(const float *x, const float *s, int64_t N, int H, int C,
                                                                  int64_t R, float *part) {
    const int D = H * C;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int64_t r1 = min(N, r0 + R);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const int h = d / C;
        float acc = 0.0f;
        int64_t r = r0;
        for (; r + 8 <= r1; r += 8) {
            float xv[8], sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xv[u] = x[(r + u) * D + d];
                sv[u] = s[(r + u) * H + h];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(sv[u], xv[u], acc);
        }
        for (; r < r1; ++r) acc = fmaf(s[r * H + h], x[r * D + d], acc);
        part[(int64_t)blockIdx.x * D + d] = acc;
    }
}
// stage 2: da[h][off + c] = Σ_p part[p][h*C + c].  One block per column: thread k adds parts k, k+256, ... in order,
// then a fixed-shape tree over the 256 partial sums in LDS (deterministic; a serial walk over 2048 parts cost 0.48 ms).
__global__ void __launch_bounds__(256) gat_wcolsum_fold_kernel(const float *part, int nparts, int H, int C, int off,
                                                               float *da) {
    __shared__ float red[256];
    const int d = blockIdx.x;
    const int D = H * C;
    float acc = 0.0f;
    for (int p = threadIdx.x; p < nparts; p += 256) acc = acc + part[(int64_t)p * D + d];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int h = d / C;
        da[(int64_t)h * 2 * C + off + (d - h * C)] = red[0];
    }
}

static void fill_plan(GatBwdArgs &g, const gnnmp_graph *p) {
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

template <int VEC, int LPH, bool DROP>
static int launch_gat_bwd(GatBwdArgs g, gnnmp_graph *plan, gnnmp_graph *plan_t, float *dWx_dst, float *da,
                          hipStream_t stream) {
    const int G = 1 << g.log2g;
    const int rpw = 64 / G;
    int waves = knob(KNOB_BLOCK_WAVES);
    if (waves < 1 || waves > 4) waves = 1;
    g.waves = waves;
    const int unroll = knob(KNOB_UNROLL);
    // ---- pass 1: destinations (forward plan)
    fill_plan(g, plan);
    g.eid = plan->eid;
    if (g.oplus) {        // the training forward saved o+ / P: a node kernel, no edge pass
        const int64_t blocks = ((int64_t)g.n_rows + (int64_t)rpw * 4 - 1) / ((int64_t)rpw * 4);
        if (blocks > 0) {
            gat_bwd_node_kernel<VEC, LPH><<<(unsigned)blocks, 256, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("gat_bwd_node_kernel");
        }
    } else {
        const int64_t nvirt = (int64_t)g.n_rows + g.n_chunks;
        const int64_t blocks = (nvirt + (int64_t)rpw * waves - 1) / ((int64_t)rpw * waves);
        if (blocks > 0) {
            if (unroll == 4)
                gat_bwd_dst_kernel<VEC, 4, LPH, DROP><<<(unsigned)blocks, 64 * waves, 0, stream>>>(g);
            else
                gat_bwd_dst_kernel<VEC, 8, LPH, DROP><<<(unsigned)blocks, 64 * waves, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("gat_bwd_dst_kernel");
        }
        if (g.n_long > 0) {
            const int64_t threads = (int64_t)g.n_long * g.H;
            gat_bwd_dst_combine_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("gat_bwd_dst_combine_kernel");
        }
    }
    // ---- pass 2: sources (transposed plan)
    fill_plan(g, plan_t);
    g.eid = plan_t->eid;
    {
        const int64_t nvirt = (int64_t)g.n_rows + g.n_chunks;
        const int64_t blocks = (nvirt + (int64_t)rpw * waves - 1) / ((int64_t)rpw * waves);
        if (blocks > 0) {
            // 20 bytes of operands per lane and edge here (Δ slice + the statistics line): 4 in flight keeps 6 waves/SIMD
            // (7.0 ms on the products shape against 8.4 ms with 8 in flight at 4 waves/SIMD)
            if (unroll == 8)
                gat_bwd_src_kernel<VEC, 8, LPH, DROP><<<(unsigned)blocks, 64 * waves, 0, stream>>>(g);
            else
                gat_bwd_src_kernel<VEC, 4, LPH, DROP><<<(unsigned)blocks, 64 * waves, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("gat_bwd_src_kernel");
        }
        if (g.n_long > 0) {
            const int64_t threads = (int64_t)g.n_long << g.log2g;
            gat_bwd_src_combine_kernel<VEC><<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(g);
            GNNMP_LAUNCH_CHECK("gat_bwd_src_combine_kernel");
        }
    }
    // ---- rank-one target term for bipartite layers
    if (dWx_dst) {
        const int64_t n = (int64_t)plan->n_dst * g.D;
        gat_bwd_dst_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(g.a, g.dsd, dWx_dst, plan->n_dst, g.H, g.C);
        GNNMP_LAUNCH_CHECK("gat_bwd_dst_rows_kernel");
    }
    // ---- da (both plans' chunk partials are dead by now in stream order: reuse the forward plan's workspace)
    if (da) {
        for (int side = 0; side < 2; ++side) {
            const float *x = side == 0 ? g.Wx_dst : g.Wx_src;
            const float *s = side == 0 ? g.dsd : g.dss;
            const int64_t N = side == 0 ? plan->n_dst : plan->n_src;
            const int64_t R = std::max<int64_t>(256, (N + 2047) / 2048);
            const int nparts = (int)((N + R - 1) / R);
            if (nparts > 0) {
                gat_wcolsum_partial_kernel<<<nparts, 256, 0, stream>>>(x, s, N, g.H, g.C, R, plan->ws);
                GNNMP_LAUNCH_CHECK("gat_wcolsum_partial_kernel");
            }
            gat_wcolsum_fold_kernel<<<g.D, 256, 0, stream>>>(plan->ws, nparts, g.H, g.C, side * g.C, da);
            GNNMP_LAUNCH_CHECK("gat_wcolsum_fold_kernel");
        }
    }
    return GNNMP_OK;
}

}  // namespace gnnmp

using namespace gnnmp;

static int gat_conv_grad_impl(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, const float *Wx_src,
                              const float *Wx_dst, const float *a, float negative_slope, float drop_p, uint64_t drop_seed,
                              const float *stats, const float *dout, float *line, float *dsd, float *dss, float *dWx_src,
                              float *dWx_dst, float *da, int64_t H, int64_t C, gnnmp_stream_t stream_, const float *outk = nullptr,
                              const float *bias = nullptr, const float *oplus = nullptr, const float *pplus = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    if (oplus && (!outk || !pplus || drop_p > 0.0f)) return fail(GNNMP_EINVAL, "gat_conv_grad2: needs out, oplus and pplus of gnnmp_gat_conv_train_f32");
    if (!(drop_p >= 0.0f && drop_p < 1.0f)) return fail(GNNMP_EINVAL, "gat_conv_grad: dropout probability %g outside [0, 1)", (double)drop_p);
    if (!plan || !plan_t) return fail(GNNMP_EINVAL, "gat_conv_grad: null plan");
    if (H <= 0 || C <= 0 || H * C > (1 << 20)) return fail(GNNMP_EINVAL, "gat_conv_grad: bad H/C");
    if (plan_t->n_dst != plan->n_src || plan_t->n_src != plan->n_dst || plan_t->n_total != plan->n_total)
        return fail(GNNMP_EINVAL, "gat_conv_grad: plan_t is not the transpose of plan (%lld x %lld, %lld edges vs %lld x %lld, %lld)",
                    (long long)plan_t->n_dst, (long long)plan_t->n_src, (long long)plan_t->n_total,
                    (long long)plan->n_dst, (long long)plan->n_src, (long long)plan->n_total);
    const bool same = !Wx_dst || Wx_dst == Wx_src;
    if (same) Wx_dst = Wx_src;
    if (same && plan->n_src != plan->n_dst) return fail(GNNMP_EINVAL, "gat_conv_grad: bipartite plan needs Wx_dst");
    if (!same && !dWx_dst) return fail(GNNMP_EINVAL, "gat_conv_grad: separate Wx_dst needs dWx_dst");
    if (same && dWx_dst) return fail(GNNMP_EINVAL, "gat_conv_grad: dWx_dst given but Wx_dst is Wx_src (the target term is folded into dWx_src)");
    if (plan->n_dst == 0 && plan->n_src == 0) return GNNMP_OK;
    if (!Wx_src || !a || !stats || !dout || !line || !dsd || !dss || !dWx_src)
        return fail(GNNMP_EINVAL, "gat_conv_grad: null pointer");
    const int D = (int)(H * C);
    int vec = pick_vec(D, Wx_src, dWx_src);
    if (((reinterpret_cast<uintptr_t>(Wx_dst) | reinterpret_cast<uintptr_t>(dout)) & (4 * vec - 1)) != 0) vec = 1;
    if ((reinterpret_cast<uintptr_t>(line) & 15) != 0) return fail(GNNMP_EINVAL, "gat_conv_grad: line must be 16-byte aligned");
    while (vec > 1 && (C % vec) != 0) vec >>= 1;
    int lph = (int)(C / vec);
    const int lanes = D / vec;
    if (H == 1 && lanes <= 64) {   // a single head may spill over idle lanes: they carry zeros
        lph = 1;
        while (lph < lanes) lph <<= 1;
    }
    if (lanes > 64)
        return fail(GNNMP_EUNSUPPORTED, "gat_conv_grad: the feature row must fit one wave (H*C = %lld)", (long long)(H * C));
    // workspaces: chunk partials of each pass in that plan's workspace; the da partials reuse the forward plan's
    const int64_t Rd = std::max<int64_t>(256, (plan->n_dst + 2047) / 2048), Rs = std::max<int64_t>(256, (plan->n_src + 2047) / 2048);
    const size_t colsum_need = (size_t)std::max((plan->n_dst + Rd - 1) / Rd, (plan->n_src + Rs - 1) / Rs) * (size_t)D;
    if (int rc = ensure_workspace(plan, std::max((size_t)plan->n_chunks * (size_t)H * 4, da ? colsum_need : (size_t)0))) return rc;
    if (plan_t->n_chunks > 0)
        if (int rc = ensure_workspace(plan_t, (size_t)plan_t->n_chunks * (size_t)(D + lanes))) return rc;
    GatBwdArgs g;
    g.Wx_src = Wx_src;
    g.Wx_dst = Wx_dst;
    g.a = a;
    g.dout = dout;
    g.stats = stats;
    g.line = line;
    g.dsd = dsd;
    g.dss = dss;
    g.dWx = dWx_src;
    g.fold_dst = same ? 1 : 0;
    g.H = (int)H;
    g.C = (int)C;
    g.D = D;
    g.log2g = 0;
    while ((1 << g.log2g) < lanes) ++g.log2g;
    g.lph = lph_code(lph, g.log2g);   // odd head widths sum their lanes one by one (common.h group_sum<0>)
    g.waves = 1;
    g.slope = negative_slope;
    g.eid = nullptr;
    g.drop = make_drop(drop_p, drop_seed);
    g.outk = outk;
    g.bias = bias;
    g.oplus = oplus;
    g.pplus = pplus;
    if (oplus && ((reinterpret_cast<uintptr_t>(outk) | reinterpret_cast<uintptr_t>(oplus)) & (4 * vec - 1)) != 0)
        return fail(GNNMP_EINVAL, "gat_conv_grad2: out / oplus not aligned like Wx");
    if (drop_p > 0.0f) {   // the dropout variants walk the head butterfly with the run-time lane count (one instantiation per width)
        if (vec == 4) return launch_gat_bwd<4, 0, true>(g, plan, plan_t, dWx_dst, da, stream);
        if (vec == 2) return launch_gat_bwd<2, 0, true>(g, plan, plan_t, dWx_dst, da, stream);
        return launch_gat_bwd<1, 0, true>(g, plan, plan_t, dWx_dst, da, stream);
    }
    if (vec == 4) {   // the usual case (C a multiple of 4): compile-time lane count per head -> DPP butterflies
        switch (lph) {
            case 1: return launch_gat_bwd<4, 1, false>(g, plan, plan_t, dWx_dst, da, stream);
            case 2: return launch_gat_bwd<4, 2, false>(g, plan, plan_t, dWx_dst, da, stream);
            case 4: return launch_gat_bwd<4, 4, false>(g, plan, plan_t, dWx_dst, da, stream);
            case 8: return launch_gat_bwd<4, 8, false>(g, plan, plan_t, dWx_dst, da, stream);
            case 16: return launch_gat_bwd<4, 16, false>(g, plan, plan_t, dWx_dst, da, stream);
            default: return launch_gat_bwd<4, 0, false>(g, plan, plan_t, dWx_dst, da, stream);
        }
    }
    if (vec == 2) return launch_gat_bwd<2, 0, false>(g, plan, plan_t, dWx_dst, da, stream);
    return launch_gat_bwd<1, 0, false>(g, plan, plan_t, dWx_dst, da, stream);
}

extern "C" int gnnmp_gat_conv_grad_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, const float *Wx_src,
                                       const float *Wx_dst, const float *a, float negative_slope, const float *stats,
                                       const float *dout, float *line, float *dsd, float *dss, float *dWx_src,
                                       float *dWx_dst, float *da, int64_t H, int64_t C, gnnmp_stream_t stream) {
    return gat_conv_grad_impl(plan, plan_t, Wx_src, Wx_dst, a, negative_slope, 0.0f, 0, stats, dout, line, dsd, dss, dWx_src, dWx_dst,
                              da, H, C, stream);
}
extern "C" int gnnmp_gat_conv_grad_drop_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, const float *Wx_src,
                                            const float *Wx_dst, const float *a, float negative_slope, float p, uint64_t seed,
                                            const float *stats, const float *dout, float *line, float *dsd, float *dss,
                                            float *dWx_src, float *dWx_dst, float *da, int64_t H, int64_t C,
                                            gnnmp_stream_t stream) {
    return gat_conv_grad_impl(plan, plan_t, Wx_src, Wx_dst, a, negative_slope, p, seed, stats, dout, line, dsd, dss, dWx_src, dWx_dst,
                              da, H, C, stream);
}

/* The pullback after gnnmp_gat_conv_train_f32 (see gnnmp.h): the destination side is a node kernel on (Δ, out, o+, P), one edge pass
 * (the source side) instead of two */
extern "C" int gnnmp_gat_conv_grad2_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, const float *Wx_src, const float *Wx_dst,
                                        const float *a, float negative_slope, const float *stats, const float *out, const float *bias,
                                        const float *oplus, const float *pplus, const float *dout, float *line, float *dsd, float *dss,
                                        float *dWx_src, float *dWx_dst, float *da, int64_t H, int64_t C, gnnmp_stream_t stream) {
    if (!out || !oplus || !pplus) return fail(GNNMP_EINVAL, "gat_conv_grad2: null out / oplus / pplus");
    return gat_conv_grad_impl(plan, plan_t, Wx_src, Wx_dst, a, negative_slope, 0.0f, 0, stats, dout, line, dsd, dss, dWx_src, dWx_dst,
                              da, H, C, stream, out, bias, oplus, pplus);
}
