// gat_fused.hip — neighbourhood attention in ONE pass over the edges.  The headline instance is GATConv:
//   GNNlib/src/layers/conv.jl:136-141  apply_edges(gat_message) -> softmax_edge_neighbors -> α .* Wxj -> aggregate_neighbors(+)
//   gat_message :152-167               logα = leakyrelu( sum(a .* vcat(Wxi, Wxj), dims = 1) )
//
// What makes it one pass: the source row Wx_j (4*H*C bytes) has to be fetched for the weighted sum anyway, and the
// source half of the logit, a[C:2C,h] . Wx_j[:,h], is a function of exactly that row.  The lanes that hold head h's
// channels (C/VEC of them, adjacent) form the dot product in registers with a butterfly (DPP quad permutes / row
// mirrors, no LDS: common.h group_sum), so there is no (N,H) score array to gather (a random 64-byte sector per edge on
// top of the row) and no node pre-pass.  The neighbourhood softmax is computed online (running max m, running
// denominator, one rescale per batch of U edges), so the row's edges are visited once instead of the reference's six
// (H,E') passes + three (C,H,E') passes:
//     out_i[h,:] = ( sum_j exp(l_ij - m_i) * V_j[h,:] ) / ( sum_j exp(l_ij - m_i) )
// Algebraically the reference's  sum_j (exp(l_ij - max_i) / den_i) * V_j ; the fp32 rounding differs (<= 1e-6 rel
// measured, north_star allows 1e-5).  Callers that need the reference's exact operation order, the α coefficients, or a
// head width whose lane count is not a power of two go through the three-pass kernels of attention.hip (the GAT ABI
// entry falls back automatically).
//
// The same kernel serves the other attention layers of the reference that share the path; only
// the per-edge logit differs (template parameter MODE, enum gnnmp_attn in gnnmp.h):
//   GAT    l = leakyrelu(a_d . Q_i + a_s . K_j)                 V = K          conv.jl:152-167
//   GATV2  l = a . leakyrelu(Q_i + K_j)                         V = K          conv.jl:202-214  (gatv2_message)
//   DOT    l = (Q_i . K_j) / scale                              V separate     conv.jl:609-616  (transformer_message_uij)
//   COS    l = scale * (x_i . x_j) / (|x_i| |x_j|), one head    V = K = Q = x  conv.jl:337-352  (agnn_conv)
//
// Long rows: same chunking as propagate.hip — a chunk is a virtual row that emits (acc, m, den) partials; the combine
// kernel merges them with the usual log-sum-exp rescale.
#include "common.h"

namespace gnnmp {

struct GatFusedArgs {
    const uint32_t *rowptr;
    const int32_t *row_order;   // [n_rows] rows by decreasing length, or null: virtual row v -> destination (see common.h)
    const int32_t *col;
    const float *Wx_src;  // K [n_src][D]
    const float *Wx_val;  // V [n_src][D] (== Wx_src unless MODE = DOT)
    const float *Wx_dst;  // Q [n_dst][D]
    const float *a;       // GAT [H][2C], GATV2 [H][C], else unused
    const float *escore;  // GAT with edge features: [n_edges][H] = a_e . We_k, original edge order; else null
    const int32_t *eid;   // plan slot -> original edge position (escore lookup)
    const float *bias;    // [D] or null
    float *out;           // [n_dst][D]
    float *partial;       // [n_chunks][D + 2*D/VEC]
    float *stats;         // [n_dst][H][2] = (running max m, denominator) of every destination, or null (saved for the adjoint)
    float *oplus;         // ATTN_GAT_PLUS: [n_dst][D]  o+_i = sum over the neighbours with a POSITIVE pre-activation logit of α_ij Wx_j
    float *pplus;         // ATTN_GAT_PLUS: [n_dst][H]  P_i = sum of those α_ij            (gat_backward.hip: the pullback's dsd_i from them)
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    const int32_t *long_rows, *long_cptr;
    int n_chunks, n_long;
    int H, C, D;
    int n_rows;
    int n_src;
    int log2g;
    int off24;            // 1: row offsets fit the 24-bit multiply (see gat_online_range)
    int lph;              // lanes per head = C / VEC (power of two; the whole group when H = 1)
    int act;
    float slope;
    float scale;          // DOT: divisor (sqrt(out)); COS: multiplier (β)
    int long_thresh;
    int cpx;
    int nbc;              // leading blocks (chunk virtual rows) that are not remapped
    int waves;
    DropArgs drop;        // ATTN_GAT_DROP / ATTN_GATV2_DROP only
    // fold-in-kernel (FOLD instances of gat_fused_rows_kernel; csr_reduce.h's scheme): chunk v belongs to long row chunk_lrow[v];
    // arrive[r] counts the chunks of long row r whose partial is stored — the last one to arrive merges the row (gat_fold_row)
    const int32_t *chunk_lrow;
    uint32_t *arrive;     // [n_long][256 / G + 1]
    float *spart;         // [n_long][256 / G][partial row]
};

// internal fifth mode: GAT with the per-edge logit term (gnnmp_gat_conv_edge_f32) — a compile-time property, so the headline
// kernel carries no trace of it (as a runtime flag both logit variants were computed and selected per edge)
constexpr int ATTN_GAT_EDGE = 4;
// internal sixth mode: GAT with dropout on the attention coefficients (conv.jl:139; gnnmp_gat_conv_drop_f32): the numerator takes
// keep_ij / (1 - p) * exp(l_ij - m), the denominator and the saved statistics are those of the undropped softmax
constexpr int ATTN_GAT_DROP = 5;
constexpr int ATTN_GATV2_DROP = 6;      // the same for gatv2_conv (conv.jl:191)
// internal eighth mode: the TRAINING forward of GATConv (gnnmp_gat_conv_train_f32): next to the output and (m, den) it accumulates the
// same weighted sum restricted to the edges whose logit z_ij = a_d.Wx_i + a_s.Wx_j is positive (leakyrelu' = 1 there, slope elsewhere):
//   dsd_i = Σ_j α_ij (g_ij - D_i) lrelu'(z_ij) = (1 - slope) (Δ_i . o+_i - D_i P_i)   with  D_i = Δ_i . o_i
// so the pullback needs NO pass over the destination-sorted edges (gat_backward.hip: gat_bwd_node_kernel)
constexpr int ATTN_GAT_PLUS = 7;
__host__ __device__ constexpr bool is_gat(int mode) {
    return mode == GNNMP_ATTN_GAT || mode == ATTN_GAT_EDGE || mode == ATTN_GAT_DROP || mode == ATTN_GAT_PLUS;
}
__host__ __device__ constexpr bool is_gatv2(int mode) { return mode == GNNMP_ATTN_GATV2 || mode == ATTN_GATV2_DROP; }
__host__ __device__ constexpr bool is_drop(int mode) { return mode == ATTN_GAT_DROP || mode == ATTN_GATV2_DROP; }
__host__ __device__ constexpr bool needs_eid(int mode) { return mode == ATTN_GAT_EDGE || is_drop(mode); }

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.0f ? x : x * slope; }
// softmax_exp (common.h): PMC showed this kernel at 84 % VALU utilisation with 9 libm exponentials per batch of 8 edges a
// quarter of its instructions (and a RUNTIME fast/accurate switch had made it evaluate both)
__device__ __forceinline__ float gexp(float x) { return softmax_exp(x); }

// what a lane keeps for the whole row: its coefficient slice, its slice of Q_i, and one per-head scalar
template <int VEC>
struct LaneRow {
    float ca[VEC];  // GAT: a_s slice; GATV2: a slice
    float vi[VEC];  // Q_i slice (GATV2, DOT, COS)
    float s0;       // GAT: a_d . Q_i;  COS: |x_i|
    float am;       // 1 for lanes that own features, 0 for idle lanes
    int h;          // this lane's head
};

// One batch = U edges: U independent row loads, U dot products, their butterflies (DPP when the lane count per head LPH
// is a compile-time constant), ONE rescale of the running state to the batch maximum and U exponentials — no branch
// anywhere, so the scheduler interleaves all of it under the loads.  Slots past the end of the row re-read the last
// edge with logit -inf (weight exactly 0).
//   c / ev: this lane's source id / edge id of the G slots loaded for the current stretch of the row; j: first slot of the
//   batch within the stretch, n: slots in the stretch.
template <int VEC, int U, int LPH, int MODE, bool OFF24>
__device__ __forceinline__ void gat_batch(const GatFusedArgs &a, int c, int ev, int gbase, int j, int n, int fc,
                                          const LaneRow<VEC> &r, float &m, float &den, float acc[VEC], float &den2, float acc2[VEC]) {
    constexpr bool edge_term = MODE == ATTN_GAT_EDGE;
    constexpr bool plus = MODE == ATTN_GAT_PLUS;      // (den2 / acc2 are dead code in every other mode)
    bool pos[plus ? U : 1];
    float v[U][VEC];                                  // K_j
    float w[MODE == GNNMP_ATTN_DOT ? U : 1][VEC];     // V_j when it is a different array
    float es[U];
    // all U source ids first (one LDS round trip for the batch), then the U row loads back to back; the edge term's
    // uniform branch sits after them (between the loads it made hipcc wait for every ds_bpermute separately)
    int cj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cj[u] = __shfl(c, gbase + min(j + u, n - 1), 64);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // OFF24 (fewer than 2^24 rows of fewer than 2^24 floats, fewer than 2^32 floats in all): the row offset is
        // ONE full-rate v_mad_u32_u24 instead of a quarter-rate 64-bit multiply-add per edge
        const int64_t o = OFF24 ? (int64_t)(uint32_t)(__umul24(cj[u], a.D) + fc) : (int64_t)cj[u] * a.D + fc;
        Vec<VEC>::load(a.Wx_src + o, v[u]);
        if (MODE == GNNMP_ATTN_DOT) Vec<VEC>::load(a.Wx_val + o, w[MODE == GNNMP_ATTN_DOT ? u : 0]);
        es[u] = 0.0f;
    }
    if (edge_term) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t ej = (uint32_t)__shfl(ev, gbase + min(j + u, n - 1), 64);   // edge positions are unsigned 32-bit
            es[u] = a.escore[(int64_t)ej * a.H + r.h];
        }
    }
    constexpr bool drop = is_drop(MODE);
    float kf[drop ? U : 1];      // keep_ij / (1 - p)
    if (drop) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t ej = (uint32_t)__shfl(ev, gbase + min(j + u, n - 1), 64);
            kf[drop ? u : 0] = drop_bits(a.drop.seed_lo, a.drop.seed_hi, ej, (uint32_t)r.h) >= a.drop.thr ? a.drop.inv : 0.0f;
        }
    }
    float l[U], nn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        l[u] = 0.0f;
        nn[u] = 0.0f;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            if (is_gat(MODE)) l[u] = fmaf(r.ca[q], v[u][q], l[u]);
            if (is_gatv2(MODE)) l[u] = fmaf(r.ca[q], lrelu(r.vi[q] + v[u][q], a.slope), l[u]);
            if (MODE == GNNMP_ATTN_DOT) l[u] = fmaf(r.vi[q], v[u][q], l[u]);
            if (MODE == GNNMP_ATTN_COS) {
                l[u] = fmaf(r.vi[q], v[u][q], l[u]);
                nn[u] = fmaf(v[u][q] * r.am, v[u][q], nn[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        l[u] = group_sum<LPH>(l[u], a.lph);
        if (MODE == GNNMP_ATTN_COS) nn[u] = group_sum<LPH>(nn[u], a.lph);
    }
    float mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float lu = l[u];
        if (plus) pos[plus ? u : 0] = r.s0 + lu > 0.0f;
        if (is_gat(MODE)) lu = lrelu(edge_term ? (r.s0 + lu) + es[u] : r.s0 + lu, a.slope);
        if (MODE == GNNMP_ATTN_DOT) lu = lu / a.scale;
        if (MODE == GNNMP_ATTN_COS) lu = a.scale * (lu / (r.s0 * sqrtf(nn[u])));
        l[u] = (j + u < n) ? lu : -__builtin_inff();
        mn = fmaxf(mn, l[u]);
    }
    const float sc = gexp(m - mn);   // exp(0) = 1 when the maximum did not move, exp(-inf) = 0 the first time
    den *= sc;
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] *= sc;
    if (plus) {
        den2 *= sc;
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc2[q] *= sc;
    }
    m = mn;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float pe = gexp(l[u] - m);
        den += pe;
        const float pw = drop ? pe * kf[drop ? u : 0] : pe;
#pragma unroll
        for (int q = 0; q < VEC; ++q)
            acc[q] = fmaf(pw, MODE == GNNMP_ATTN_DOT ? w[MODE == GNNMP_ATTN_DOT ? u : 0][q] : v[u][q], acc[q]);
        if (plus) {
            const float pp = pos[plus ? u : 0] ? pe : 0.0f;      // (slots past the end: pe = exp(-inf) = 0)
            den2 += pp;
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc2[q] = fmaf(pp, v[u][q], acc2[q]);
        }
    }
}

// A row = stretches of G slots (one coalesced load of source ids per stretch), each stretch = full batches of U edges and,
// for what is left, ONE batch of the smallest width in {U/4, U/2, U} that holds it: a row of 9 edges costs 8 + 2 slots, not
// 16 (on the products shape a third of the rows are shorter than two batches: padded batches were 12 % of all issue slots).
template <int VEC, int U, int LPH, int MODE, bool OFF24>
__device__ __forceinline__ void gat_online_range(const GatFusedArgs &a, uint32_t beg, uint32_t end, int lig,
                                                 int gbase, int G, int fc, const LaneRow<VEC> &r,
                                                 float &m, float &den, float acc[VEC], float &den2, float acc2[VEC]) {
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (see csr_reduce.h: reduce_range)
        const uint32_t p = base + lig;
        const int c = p < end ? a.col[p] : 0;
        // edge features (gat_conv with dense_e): the edge's share of the logit, a_e . We_k, precomputed per edge and head,
        // is fetched by original edge position (uniform branch: absent for the headline layer)
        const int ev = (needs_eid(MODE) && p < end) ? a.eid[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        int j = 0;
        for (; j + U <= n; j += U) gat_batch<VEC, U, LPH, MODE, OFF24>(a, c, ev, gbase, j, n, fc, r, m, den, acc, den2, acc2);
        const int rem = n - j;
        if (rem > 0) {
            if (U >= 8 && rem <= U / 4)
                gat_batch<VEC, (U >= 8 ? U / 4 : U), LPH, MODE, OFF24>(a, c, ev, gbase, j, n, fc, r, m, den, acc, den2, acc2);
            else if (U >= 4 && rem <= U / 2)
                gat_batch<VEC, (U >= 4 ? U / 2 : U), LPH, MODE, OFF24>(a, c, ev, gbase, j, n, fc, r, m, den, acc, den2, acc2);
            else
                gat_batch<VEC, U, LPH, MODE, OFF24>(a, c, ev, gbase, j, n, fc, r, m, den, acc, den2, acc2);
        }
    }
}

template <int VEC>
__device__ __forceinline__ void gat_fused_store(const GatFusedArgs &a, int row, int f0, bool active,
                                                float acc[VEC]) {
    if (!active) return;
    if (a.bias) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] + a.bias[f0 + q];
    }
    if (a.act == GNNMP_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] < 0.0f ? 0.0f : acc[q];
    }
    Vec<VEC>::store(a.out + (int64_t)row * a.D + f0, acc);
}

// the end of a split row, given its merged state (active lanes only): normalise, statistics, o+ / P, epilogue + store
template <int VEC, bool PLUS>
__device__ __forceinline__ void gat_long_row_finish(const GatFusedArgs &a, int row, int f0, float acc[VEC], float Mt, float den,
                                                    float acc2[VEC], float den2) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = acc[q] / den;
    if (a.stats && (f0 % a.C) == 0) {
        float *st = a.stats + ((int64_t)row * a.H + f0 / a.C) * 2;
        st[0] = Mt;
        st[1] = den;
        if (PLUS) a.pplus[(int64_t)row * a.H + f0 / a.C] = den2 / den;
    }
    if (PLUS) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc2[q] = acc2[q] / den;
        Vec<VEC>::store(a.oplus + (int64_t)row * a.D + f0, acc2);
    }
    gat_fused_store<VEC>(a, row, f0, true, acc);
}

// gat_fused_combine_kernel's merge INSIDE the row kernel, in the combine kernel's own two levels (csr_reduce.h: long_geom) — the same
// slices, the same log-sum-exp rescales, the same fma order, so a split row comes out bit-identical to the two-kernel path.  Active lanes
// only.  Level 1: the last chunk of a slice to arrive merges the slice's chunk partials relative to the slice's maximum and stores
// (acc, M, den [, acc2, den2]) in the chunk-partial layout to the slice partials.
template <int VEC, bool PLUS>
__device__ __forceinline__ void gat_fold_slice(const GatFusedArgs &a, int s0, int s1, int f0, float *__restrict__ sp) {
    const int LN = a.D / VEC;
    const int64_t S = PLUS ? 2 * a.D + 3 * LN : a.D + 2 * LN;
    constexpr int CB = 4;
    const int li = f0 / VEC;
    float M = -__builtin_inff();
    for (int c = s0; c < s1; c += CB) {
        float mv[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) mv[u] = coh_load1(a.partial + (int64_t)min(c + u, s1 - 1) * S + a.D + li);
#pragma unroll
        for (int u = 0; u < CB; ++u) M = fmaxf(M, mv[u]);
    }
    float den = 0.0f, den2 = 0.0f, acc[VEC], acc2[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = acc2[q] = 0.0f;
    for (int c = s0; c < s1; c += CB) {
        float mv[CB], dv[CB], v[CB][VEC], dv2[PLUS ? CB : 1], v2[PLUS ? CB : 1][VEC];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const float *pc = a.partial + (int64_t)min(c + u, s1 - 1) * S;
            mv[u] = coh_load1(pc + a.D + li);
            dv[u] = coh_load1(pc + a.D + LN + li);
            coh_load<VEC>(pc + f0, v[u]);
            if (PLUS) {
                coh_load<VEC>(pc + a.D + 2 * LN + f0, v2[PLUS ? u : 0]);
                dv2[PLUS ? u : 0] = coh_load1(pc + 2 * a.D + 2 * LN + li);
            }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (c + u < s1) {
                const float sc = expf(mv[u] - M);
                den = fmaf(dv[u], sc, den);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = fmaf(v[u][q], sc, acc[q]);
                if (PLUS) {
                    den2 = fmaf(dv2[PLUS ? u : 0], sc, den2);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc2[q] = fmaf(v2[PLUS ? u : 0][q], sc, acc2[q]);
                }
            }
        }
    }
    coh_store<VEC>(sp + f0, acc);
    coh_store1(sp + a.D + li, M);
    coh_store1(sp + a.D + LN + li, den);
    if (PLUS) {
        coh_store<VEC>(sp + a.D + 2 * LN + f0, acc2);
        coh_store1(sp + 2 * a.D + 2 * LN + li, den2);
    }
}
// Level 2: the last slice of the row to finish merges the ns slice partials (slice 0 rescaled by a multiplication, the others folded in
// with an fma each — group 0 of the combine kernel) and finishes the row.
template <int VEC, bool PLUS>
__device__ __forceinline__ void gat_fold_slices(const GatFusedArgs &a, int row, int ns, int f0, const float *__restrict__ sp) {
    const int LN = a.D / VEC;
    const int64_t S = PLUS ? 2 * a.D + 3 * LN : a.D + 2 * LN;
    const int li = f0 / VEC;
    float Mt = -__builtin_inff();
    for (int k = 0; k < ns; ++k) Mt = fmaxf(Mt, coh_load1(sp + (int64_t)k * S + a.D + li));
    float den = 0.0f, den2 = 0.0f, acc[VEC], acc2[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = acc2[q] = 0.0f;
    for (int k = 0; k < ns; ++k) {
        const float *o = sp + (int64_t)k * S;
        float sa[VEC], sa2[VEC];
        coh_load<VEC>(o + f0, sa);
        const float sc = expf(coh_load1(o + a.D + li) - Mt);
        const float sd = coh_load1(o + a.D + LN + li);
        float sd2 = 0.0f;
#pragma unroll
        for (int q = 0; q < VEC; ++q) sa2[q] = 0.0f;
        if (PLUS) {
            coh_load<VEC>(o + a.D + 2 * LN + f0, sa2);
            sd2 = coh_load1(o + 2 * a.D + 2 * LN + li);
        }
        if (k == 0) {
            den = sd * sc;
            den2 = sd2 * sc;
#pragma unroll
            for (int q = 0; q < VEC; ++q) { acc[q] = sa[q] * sc; acc2[q] = sa2[q] * sc; }
        } else {
            den = fmaf(sd, sc, den);
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = fmaf(sa[q], sc, acc[q]);
            if (PLUS) {
                den2 = fmaf(sd2, sc, den2);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc2[q] = fmaf(sa2[q], sc, acc2[q]);
            }
        }
    }
    gat_long_row_finish<VEC, PLUS>(a, row, f0, acc, Mt, den, acc2, den2);
}

// A chunk's lane group has stored its partial (coh_store): count it in; true for the group that completes its slice / row
// (csr_reduce.h: chunk_arrive — no fence, coherent accesses only)
__device__ __forceinline__ bool gat_chunk_arrive(uint32_t *counter, int n, int lig, int gbase) {
    coh_publish();
    unsigned prev = 0;
    if (lig == 0) prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = (unsigned)__shfl((int)prev, gbase, 64);
    if (prev != (unsigned)(n - 1)) return false;
    if (lig == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <int VEC, int U, int LPH, int MODE, bool FOLD = false>
__global__ void __launch_bounds__(256) gat_fused_rows_kernel(const GatFusedArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = 1 << a.log2g;
    const int lig = lane & (G - 1);
    const int grp = lane >> a.log2g;
    const int gbase = lane - lig;
    const int rpw = 64 >> a.log2g;
    const int chunk = a.cpx ? xcd_remap_after(blockIdx.x, a.nbc, a.cpx) : (int)blockIdx.x;
    const int64_t v64 = ((int64_t)chunk * a.waves + wave) * rpw + grp;
    if (v64 >= (int64_t)a.n_rows + a.n_chunks) return;
    const int v = (int)v64;
    const int f0 = ((int)blockIdx.y * G + lig) * VEC;
    const bool active = f0 < a.D;
    const bool is_chunk = v < a.n_chunks;
    int row;
    uint32_t beg, end;
    if (is_chunk) {
        row = a.chunk_row[v];
        beg = a.chunk_beg[v];
        end = a.chunk_end[v];
    } else {
        row = v - a.n_chunks;
        if (a.row_order) row = a.row_order[row];
        beg = a.rowptr[row];
        end = a.rowptr[row + 1];
        if (end - beg > a.long_thresh) return;
    }
    // this lane's slice of the attention vector and of Q_i; idle lanes (D/VEC not a power of two) shadow lane 0 with zero
    // coefficients so that no load below needs a predicate
    const int fc = active ? f0 : 0;
    LaneRow<VEC> r;
    r.am = active ? 1.0f : 0.0f;
    r.s0 = 0.0f;
    r.h = fc / a.C;
    {
        const int h = fc / a.C, c0 = fc - h * a.C;
        float qi[VEC];
        Vec<VEC>::load(a.Wx_dst + (int64_t)row * a.D + fc, qi);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            r.vi[q] = active ? qi[q] : 0.0f;
            r.ca[q] = 0.0f;
        }
        if (is_gat(MODE)) {
            const float *ah = a.a + (int64_t)h * 2 * a.C + c0;
            float sd = 0.0f;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float adq = active ? ah[q] : 0.0f;
                r.ca[q] = active ? ah[a.C + q] : 0.0f;
                sd = fmaf(adq, qi[q], sd);
            }
            r.s0 = group_sum<LPH>(sd, a.lph);
        }
        if (is_gatv2(MODE)) {
            const float *ah = a.a + (int64_t)h * a.C + c0;
#pragma unroll
            for (int q = 0; q < VEC; ++q) r.ca[q] = active ? ah[q] : 0.0f;
        }
        if (MODE == GNNMP_ATTN_COS) {
            float n2 = 0.0f;
#pragma unroll
            for (int q = 0; q < VEC; ++q) n2 = fmaf(r.vi[q], r.vi[q], n2);
            r.s0 = sqrtf(group_sum<LPH>(n2, a.lph));
        }
    }

    constexpr bool plus = MODE == ATTN_GAT_PLUS;
    float m = -__builtin_inff(), den = 0.0f, den2 = 0.0f;
    float acc[VEC], acc2[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = acc2[q] = 0.0f;
    if (a.off24)
        gat_online_range<VEC, U, LPH, MODE, true>(a, beg, end, lig, gbase, G, fc, r, m, den, acc, den2, acc2);
    else
        gat_online_range<VEC, U, LPH, MODE, false>(a, beg, end, lig, gbase, G, fc, r, m, den, acc, den2, acc2);
    if (is_chunk) {
        if (active) {
            const int LN = a.D / VEC;
            float *pc = a.partial + (int64_t)v * (plus ? 2 * a.D + 3 * LN : a.D + 2 * LN);
            if (FOLD) {           // read by another workgroup of this launch: coherent stores (common.h)
                coh_store<VEC>(pc + f0, acc);
                coh_store1(pc + a.D + f0 / VEC, m);
                coh_store1(pc + a.D + LN + f0 / VEC, den);
            } else {
                Vec<VEC>::store(pc + f0, acc);
                pc[a.D + f0 / VEC] = m;
                pc[a.D + LN + f0 / VEC] = den;
            }
            if (plus) {
                Vec<VEC>::store(pc + a.D + 2 * LN + f0, acc2);
                pc[2 * a.D + 2 * LN + f0 / VEC] = den2;
            }
        }
        if (FOLD) {
            const int lr = a.chunk_lrow[v];
            const int NG = 256 >> a.log2g;
            const int c0 = a.long_cptr[lr], c1 = a.long_cptr[lr + 1];
            const int per = (c1 - c0 + NG - 1) / NG, ns = (c1 - c0 + per - 1) / per;
            const int k = (v - c0) / per;
            const int s0 = c0 + k * per, s1 = min(c1, s0 + per);
            const int LN = a.D / VEC;
            const int64_t S = plus ? 2 * a.D + 3 * LN : a.D + 2 * LN;
            uint32_t *cnt = a.arrive + (int64_t)lr * (NG + 1);
            float *sp = a.spart + (int64_t)lr * NG * S;
            if (!gat_chunk_arrive(cnt + k, s1 - s0, lig, gbase)) return;
            if (active) gat_fold_slice<VEC, plus>(a, s0, s1, f0, sp + (int64_t)k * S);
            if (!gat_chunk_arrive(cnt + NG, ns, lig, gbase)) return;
            if (active) gat_fold_slices<VEC, plus>(a, row, ns, f0, sp);
        }
        return;
    }
    if (end > beg) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] / den;
        if (plus) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc2[q] = acc2[q] / den;
            den2 = den2 / den;
        }
    }
    if (a.stats && active && (f0 % a.C) == 0) {
        float *st = a.stats + ((int64_t)row * a.H + f0 / a.C) * 2;
        st[0] = m;
        st[1] = den;
        if (plus) a.pplus[(int64_t)row * a.H + f0 / a.C] = den2;
    }
    if (plus && active) Vec<VEC>::store(a.oplus + (int64_t)row * a.D + f0, acc2);
    gat_fused_store<VEC>(a, row, f0, active, acc);
}

// one BLOCK per long row (see csr_combine_kernel): every lane group merges a contiguous slice of the row's chunk partials
// (acc, m, den) with the log-sum-exp rescale, the slice results meet in LDS, group 0 merges them in slice order.
template <int VEC, bool PLUS>
__global__ void __launch_bounds__(256) gat_fused_combine_kernel(const GatFusedArgs a) {
    constexpr int RS = PLUS ? 2 * VEC + 3 : VEC + 2;        // floats a thread parks in LDS: acc, M, den (, acc2, den2)
    __shared__ float red[256 * RS];
    const int G = 1 << a.log2g;
    const int lig = threadIdx.x & (G - 1);
    const int grp = threadIdx.x >> a.log2g;
    const int NG = 256 >> a.log2g;
    const int r = blockIdx.x;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;
    const int fc = active ? f0 : 0;
    const int row = a.long_rows[r];
    const int c0 = a.long_cptr[r], c1 = a.long_cptr[r + 1];
    const int per = (c1 - c0 + NG - 1) / NG;
    const int s0 = min(c1, c0 + grp * per), s1 = min(c1, s0 + per);
    const int LN = a.D / VEC;
    const int64_t S = PLUS ? 2 * a.D + 3 * LN : a.D + 2 * LN;
    constexpr int CB = PLUS ? 4 : 8;
    const int li = fc / VEC;
    float M = -__builtin_inff();
    for (int c = s0; c < s1; c += CB) {
        float mv[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) mv[u] = a.partial[(int64_t)min(c + u, s1 - 1) * S + a.D + li];
#pragma unroll
        for (int u = 0; u < CB; ++u) M = fmaxf(M, mv[u]);
    }
    float den = 0.0f, den2 = 0.0f, acc[VEC], acc2[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = acc2[q] = 0.0f;
    for (int c = s0; c < s1; c += CB) {
        float mv[CB], dv[CB], v[CB][VEC], dv2[PLUS ? CB : 1], v2[PLUS ? CB : 1][VEC];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const float *pc = a.partial + (int64_t)min(c + u, s1 - 1) * S;
            mv[u] = pc[a.D + li];
            dv[u] = pc[a.D + LN + li];
            Vec<VEC>::load(pc + fc, v[u]);
            if (PLUS) {
                Vec<VEC>::load(pc + a.D + 2 * LN + fc, v2[PLUS ? u : 0]);
                dv2[PLUS ? u : 0] = pc[2 * a.D + 2 * LN + li];
            }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (c + u < s1) {
                const float sc = expf(mv[u] - M);
                den = fmaf(dv[u], sc, den);
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = fmaf(v[u][q], sc, acc[q]);
                if (PLUS) {
                    den2 = fmaf(dv2[PLUS ? u : 0], sc, den2);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc2[q] = fmaf(v2[PLUS ? u : 0][q], sc, acc2[q]);
                }
            }
        }
    }
    float *mine = red + threadIdx.x * RS;
#pragma unroll
    for (int q = 0; q < VEC; ++q) mine[q] = acc[q];
    mine[VEC] = M;
    mine[VEC + 1] = den;
    if (PLUS) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) mine[VEC + 2 + q] = acc2[q];
        mine[2 * VEC + 2] = den2;
    }
    __syncthreads();
    if (grp != 0 || !active) return;
    float Mt = M;
    for (int k = 1; k < NG; ++k) {
        if (c0 + k * per >= c1) break;
        Mt = fmaxf(Mt, red[((k << a.log2g) + lig) * RS + VEC]);
    }
    {
        const float sc = expf(M - Mt);
        den *= sc;
        den2 *= sc;
#pragma unroll
        for (int q = 0; q < VEC; ++q) { acc[q] *= sc; acc2[q] *= sc; }
    }
    for (int k = 1; k < NG; ++k) {
        if (c0 + k * per >= c1) break;
        const float *o = red + ((k << a.log2g) + lig) * RS;
        const float sc = expf(o[VEC] - Mt);
        den = fmaf(o[VEC + 1], sc, den);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = fmaf(o[q], sc, acc[q]);
        if (PLUS) {
            den2 = fmaf(o[2 * VEC + 2], sc, den2);
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc2[q] = fmaf(o[VEC + 2 + q], sc, acc2[q]);
        }
    }
    gat_long_row_finish<VEC, PLUS>(a, row, f0, acc, Mt, den, acc2, den2);
}

template <int VEC, int U, int MODE, bool FOLD = false>
static void launch_rows_lph(const GatFusedArgs &a, dim3 grid, int blk, hipStream_t stream) {
    // compile-time lane count per head for the usual VEC = 4 shapes (DPP butterflies); anything else walks the xor
    // butterfly with the run-time count
    if (VEC == 4 && a.lph == 1)
        gat_fused_rows_kernel<VEC, U, 1, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
    else if (VEC == 4 && a.lph == 2)
        gat_fused_rows_kernel<VEC, U, 2, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
    else if (VEC == 4 && a.lph == 4)
        gat_fused_rows_kernel<VEC, U, 4, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
    else if (VEC == 4 && a.lph == 8)
        gat_fused_rows_kernel<VEC, U, 8, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
    else if (VEC == 4 && a.lph == 16)
        gat_fused_rows_kernel<VEC, U, 16, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
    else if (VEC == 4 && a.lph == 32)
        gat_fused_rows_kernel<VEC, U, 32, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
    else if (VEC == 4 && a.lph == 64)
        gat_fused_rows_kernel<VEC, U, 64, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
    else
        gat_fused_rows_kernel<VEC, U, 0, MODE, FOLD><<<grid, blk, 0, stream>>>(a);
}

template <int VEC, int MODE>
static int launch_gat_fused(GatFusedArgs a, hipStream_t stream) {
    const int G = 1 << a.log2g;
    const int rpw = 64 / G;
    int waves = knob(KNOB_BLOCK_WAVES);
    if (waves < 1 || waves > 4) waves = 1;   // auto: single-wave blocks (measured 5.21 vs 5.41 ms on products: finer
                                              // grained retirement for a kernel whose rows differ 100x in length)
    a.waves = waves;
    const int rows_per_block = rpw * waves;
    const int64_t nvirt = (int64_t)a.n_rows + a.n_chunks;
    const int64_t chunks = (nvirt + rows_per_block - 1) / rows_per_block;
    if (chunks > 0) {
        int64_t gx = chunks;
        a.cpx = 0;
        if (use_xcd_remap(a.n_src, a.D, chunks)) {
            a.nbc = (int)std::min<int64_t>(chunks, (a.n_chunks + rows_per_block - 1) / rows_per_block);
            a.cpx = (int)((chunks - a.nbc + 7) / 8);
            gx = (int64_t)a.nbc + (int64_t)a.cpx * 8;
        }
        dim3 grid((unsigned)gx, 1);
        const int U = knob(KNOB_UNROLL);
        const int blk = 64 * waves;
        if (MODE == GNNMP_ATTN_DOT) {
            launch_rows_lph<VEC, 4, MODE>(a, grid, blk, stream);   // two rows per edge in flight: half the batch
        } else if (is_gat(MODE) && U == 4) {
            gat_fused_rows_kernel<VEC, 4, 0, MODE><<<grid, blk, 0, stream>>>(a);
        } else if (is_gat(MODE) && U == 2) {
            gat_fused_rows_kernel<VEC, 2, 0, MODE><<<grid, blk, 0, stream>>>(a);
        } else if (MODE == GNNMP_ATTN_GAT && a.arrive && a.n_long > 0) {
            // GATConv's forward merges its split rows inside the row kernel: no second launch.  (Not the training forward: with the o+ / P
            // accumulators the merge code took the kernel from 86 to 101 registers — 5 -> 4 waves a SIMD.)
            launch_rows_lph<VEC, 8, MODE, MODE == GNNMP_ATTN_GAT>(a, grid, blk, stream);
            GNNMP_LAUNCH_CHECK("gat_fused_rows_kernel<FOLD>");
            return GNNMP_OK;
        } else {
            launch_rows_lph<VEC, 8, MODE>(a, grid, blk, stream);
        }
        GNNMP_LAUNCH_CHECK("gat_fused_rows_kernel");
    }
    if (a.n_long > 0) {
        if (MODE == ATTN_GAT_PLUS)
            gat_fused_combine_kernel<VEC, true><<<(unsigned)a.n_long, 256, 0, stream>>>(a);
        else
            gat_fused_combine_kernel<VEC, false><<<(unsigned)a.n_long, 256, 0, stream>>>(a);
        GNNMP_LAUNCH_CHECK("gat_fused_combine_kernel");
    }
    return GNNMP_OK;
}

template <int MODE>
static int launch_mode(const GatFusedArgs &g, int vec, hipStream_t stream) {
    switch (vec) {
        case 4: return launch_gat_fused<4, MODE>(g, stream);
        case 2: return launch_gat_fused<2, MODE>(g, stream);
        default: return launch_gat_fused<1, MODE>(g, stream);
    }
}

}  // namespace gnnmp

using namespace gnnmp;

static int attn_conv_impl(gnnmp_graph_t *plan, int mode, const float *Q, const float *K, const float *V, const float *a,
                          float negative_slope, float scale, const float *bias, int act, float *out, float *stats,
                          int64_t H, int64_t C, gnnmp_stream_t stream_, const float *escore = nullptr, float drop_p = 0.0f,
                          uint64_t drop_seed = 0, float *oplus = nullptr, float *pplus = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool plus = oplus != nullptr;
    if (plus && (mode != GNNMP_ATTN_GAT || !stats || !pplus || escore || drop_p > 0.0f))
        return fail(GNNMP_EINVAL, "gat_conv_train: needs stats, oplus and pplus; plain GAT logits only");
    if (!plan) return fail(GNNMP_EINVAL, "attn_conv: null plan");
    if (!(drop_p >= 0.0f && drop_p < 1.0f)) return fail(GNNMP_EINVAL, "gat_conv: dropout probability %g outside [0, 1)", (double)drop_p);
    if (drop_p > 0.0f && ((mode != GNNMP_ATTN_GAT && mode != GNNMP_ATTN_GATV2) || escore))
        return fail(GNNMP_EUNSUPPORTED, "attention dropout: only on the GAT / GATv2 logits without edge features");
    if (escore && plan->self_loops)
        return fail(GNNMP_EINVAL, "gat_conv: edge features and add_self_loops cannot be combined (GNNlib/src/layers/conv.jl:120)");
    if (mode < GNNMP_ATTN_GAT || mode > GNNMP_ATTN_COS) return fail(GNNMP_EINVAL, "attn_conv: bad mode %d", mode);
    if (H <= 0 || C <= 0 || H * C > (1 << 20)) return fail(GNNMP_EINVAL, "attn_conv: bad H/C");
    if (mode == GNNMP_ATTN_COS && H != 1) return fail(GNNMP_EINVAL, "attn_conv: the cosine logit is single-head");
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "attn_conv: bad act %d", act);
    if (plan->n_dst == 0) return GNNMP_OK;
    if (!Q) Q = K;
    if (!V) V = K;
    if (mode != GNNMP_ATTN_DOT && V != K) return fail(GNNMP_EINVAL, "attn_conv: a separate value array needs mode DOT");
    const bool needs_a = mode == GNNMP_ATTN_GAT || mode == GNNMP_ATTN_GATV2;
    if (!out || (needs_a && !a) || !Q || (plan->n_total > 0 && !K)) return fail(GNNMP_EINVAL, "attn_conv: null pointer");
    if (Q == K && plan->n_src != plan->n_dst) return fail(GNNMP_EINVAL, "attn_conv: bipartite plan needs a separate Q");
    const int D = (int)(H * C);
    int vec = pick_vec(D, K, out);
    if (((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(V)) & (4 * vec - 1)) != 0) vec = 1;
    while (vec > 1 && (C % vec) != 0) vec >>= 1;
    int lph = (int)(C / vec);
    const int lanes = D / vec;
    int log2g = 0;
    while ((1 << log2g) < lanes) ++log2g;   // one feature tile: the head butterfly needs the whole row in one group
    if (H == 1 && lanes <= 64) lph = 1 << log2g;   // a single head may spill over idle lanes: they carry zeros
    if (lanes > 64) {
        if (mode != GNNMP_ATTN_GAT || stats || escore || drop_p > 0.0f || plus)
            return fail(GNNMP_EUNSUPPORTED,
                        "attn_conv: the one-pass kernel needs a feature row that fits one wave (H*C = %lld lanes %d > 64)",
                        (long long)(H * C), lanes);
        // GAT rows wider than a wave: three-pass kernels on node scores
        const size_t need = (size_t)(plan->n_dst + plan->n_src) * (size_t)H;
        if (int rc = ensure_workspace(plan, need)) return rc;
        float *sdst = plan->ws, *ssrc = plan->ws + (size_t)plan->n_dst * (size_t)H;
        if (int rc = gnnmp_gat_node_scores_f32(Q, a, sdst, nullptr, plan->n_dst, H, C, stream_)) return rc;
        if (int rc = gnnmp_gat_node_scores_f32(K, a, nullptr, ssrc, plan->n_src, H, C, stream_)) return rc;
        return gnnmp_gat_aggregate_f32(plan, K, sdst, ssrc, negative_slope, bias, act, out, nullptr, H, C, stream_);
    }
    if (plan->n_chunks > 0) {
        if (int rc = ensure_workspace(plan, (size_t)plan->n_chunks * (size_t)(plus ? 2 * D + 3 * lanes : D + 2 * lanes))) return rc;
    }
    GatFusedArgs g;
    g.oplus = oplus;
    g.pplus = pplus;
    g.rowptr = plan->rowptr;
    g.row_order = nullptr;
    if (use_row_order(plan->n_src, D) && lanes <= 32) {   // two or more rows per wave: pair rows of equal length
        if (int rc = ensure_row_order(plan, stream)) return rc;
        g.row_order = plan->row_order;
    }
    g.col = plan->col;
    g.Wx_src = K;
    g.Wx_val = V;
    g.Wx_dst = Q;
    g.a = a;
    g.escore = escore;
    g.eid = plan->eid;
    g.bias = bias;
    g.out = out;
    g.partial = plan->ws;
    g.stats = stats;
    g.chunk_row = plan->chunk_row;
    g.chunk_beg = plan->chunk_beg;
    g.chunk_end = plan->chunk_end;
    g.long_rows = plan->long_rows;
    g.long_cptr = plan->long_cptr;
    g.n_chunks = plan->n_chunks;
    g.n_long = plan->n_long;
    g.H = (int)H;
    g.C = (int)C;
    g.D = D;
    g.n_rows = (int)plan->n_dst;
    g.n_src = (int)plan->n_src;
    g.log2g = log2g;
    g.lph = lph_code(lph, log2g);   // odd head widths (C = 7 classes, ...) sum their lanes one by one
    g.act = act;
    g.slope = negative_slope;
    g.scale = scale;
    g.long_thresh = plan->long_thresh;
    g.cpx = 0;
    g.nbc = 0;
    g.waves = 4;
    g.off24 = plan->n_src < (1 << 24) && D < (1 << 24) && (int64_t)plan->n_src * D < (1ll << 32);
    g.drop = make_drop(drop_p, drop_seed);
    g.chunk_lrow = plan->chunk_lrow;
    g.arrive = nullptr;
    g.spart = nullptr;
    if (plan->n_long > 0 && use_fold() && (mode == GNNMP_ATTN_GAT && !escore && drop_p == 0.0f && !plus)) {   // (plain GAT only)
        const size_t NG = (size_t)(256 >> log2g);
        if (int rc = ensure_arrive(plan, (size_t)plan->n_long * (NG + 1),
                                   (size_t)plan->n_long * NG * (size_t)(plus ? 2 * D + 3 * lanes : D + 2 * lanes), stream))
            return rc;
        g.arrive = plan->arrive;
        g.spart = plan->spart;
    }
    if (drop_p > 0.0f)
        return mode == GNNMP_ATTN_GATV2 ? launch_mode<ATTN_GATV2_DROP>(g, vec, stream) : launch_mode<ATTN_GAT_DROP>(g, vec, stream);
    if (plus) return launch_mode<ATTN_GAT_PLUS>(g, vec, stream);
    switch (mode) {
        case GNNMP_ATTN_GATV2: return launch_mode<GNNMP_ATTN_GATV2>(g, vec, stream);
        case GNNMP_ATTN_DOT: return launch_mode<GNNMP_ATTN_DOT>(g, vec, stream);
        case GNNMP_ATTN_COS: return launch_mode<GNNMP_ATTN_COS>(g, vec, stream);
        default: return g.escore ? launch_mode<ATTN_GAT_EDGE>(g, vec, stream) : launch_mode<GNNMP_ATTN_GAT>(g, vec, stream);
    }
}

extern "C" int gnnmp_gat_conv_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                                  float negative_slope, const float *bias, int act, float *out, int64_t H, int64_t C,
                                  gnnmp_stream_t stream) {
    return attn_conv_impl(plan, GNNMP_ATTN_GAT, Wx_dst, Wx_src, nullptr, a, negative_slope, 1.0f, bias, act, out, nullptr,
                          H, C, stream);
}
extern "C" int gnnmp_gat_conv_stats_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                                        float negative_slope, const float *bias, int act, float *out, float *stats,
                                        int64_t H, int64_t C, gnnmp_stream_t stream) {
    if (!stats) return fail(GNNMP_EINVAL, "gat_conv_stats: null stats");
    return attn_conv_impl(plan, GNNMP_ATTN_GAT, Wx_dst, Wx_src, nullptr, a, negative_slope, 1.0f, bias, act, out, stats, H,
                          C, stream);
}
/* the training forward: out, (m, den), and what lets the pullback skip its destination-side edge pass (see ATTN_GAT_PLUS above) */
extern "C" int gnnmp_gat_conv_train_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                                        float negative_slope, const float *bias, int act, float *out, float *stats, float *oplus,
                                        float *pplus, int64_t H, int64_t C, gnnmp_stream_t stream) {
    if (!stats || !oplus || !pplus) return fail(GNNMP_EINVAL, "gat_conv_train: null stats / oplus / pplus");
    return attn_conv_impl(plan, GNNMP_ATTN_GAT, Wx_dst, Wx_src, nullptr, a, negative_slope, 1.0f, bias, act, out, stats, H, C, stream,
                          nullptr, 0.0f, 0, oplus, pplus);
}
extern "C" int gnnmp_gat_conv_edge_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                                       const float *edge_score, float negative_slope, const float *bias, int act,
                                       float *out, int64_t H, int64_t C, gnnmp_stream_t stream) {
    if (!edge_score && plan && plan->n_edges > 0) return fail(GNNMP_EINVAL, "gat_conv_edge: null edge_score");
    return attn_conv_impl(plan, GNNMP_ATTN_GAT, Wx_dst, Wx_src, nullptr, a, negative_slope, 1.0f, bias, act, out, nullptr,
                          H, C, stream, edge_score);
}
extern "C" int gnnmp_gat_conv_drop_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                                       float negative_slope, float p, uint64_t seed, const float *bias, int act, float *out,
                                       float *stats, int64_t H, int64_t C, gnnmp_stream_t stream) {
    return attn_conv_impl(plan, GNNMP_ATTN_GAT, Wx_dst, Wx_src, nullptr, a, negative_slope, 1.0f, bias, act, out, stats, H, C,
                          stream, nullptr, p, seed);
}

extern "C" int gnnmp_attn_conv_drop_f32(gnnmp_graph_t *plan, int mode, const float *Q, const float *K, const float *V,
                                        const float *a, float negative_slope, float scale, float p, uint64_t seed,
                                        const float *bias, int act, float *out, float *stats, int64_t H, int64_t C,
                                        gnnmp_stream_t stream) {
    return attn_conv_impl(plan, mode, Q, K, V, a, negative_slope, scale, bias, act, out, stats, H, C, stream, nullptr, p, seed);
}

__global__ void __launch_bounds__(256) dropout_keep_kernel(DropArgs d, int64_t n, int H, uint8_t *keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * H) return;
    const int64_t e = i / H;
    keep[i] = drop_bits(d.seed_lo, d.seed_hi, (uint32_t)e, (uint32_t)(i - e * H)) >= d.thr ? 1 : 0;
}
extern "C" int gnnmp_dropout_keep_u8(uint64_t seed, float p, int64_t n_edges, int64_t H, uint8_t *keep, gnnmp_stream_t stream_) {
    if (!(p >= 0.0f && p < 1.0f)) return fail(GNNMP_EINVAL, "dropout_keep: probability %g outside [0, 1)", (double)p);
    if (n_edges < 0 || H <= 0 || n_edges >= ((int64_t)1 << 32)) return fail(GNNMP_EINVAL, "dropout_keep: bad size");
    if (n_edges == 0) return GNNMP_OK;
    if (!keep) return fail(GNNMP_EINVAL, "dropout_keep: null pointer");
    const int64_t n = n_edges * H;
    dropout_keep_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>(make_drop(p, seed), n_edges, (int)H, keep);
    GNNMP_LAUNCH_CHECK("dropout_keep_kernel");
    return GNNMP_OK;
}
extern "C" int gnnmp_attn_conv_f32(gnnmp_graph_t *plan, int mode, const float *Q, const float *K, const float *V,
                                   const float *a, float negative_slope, float scale, const float *bias, int act,
                                   float *out, float *stats, int64_t H, int64_t C, gnnmp_stream_t stream) {
    return attn_conv_impl(plan, mode, Q, K, V, a, negative_slope, scale, bias, act, out, stats, H, C, stream);
}
