// csr_reduce.h — the per-destination row walk shared by the propagate kernels (propagate.hip) and the fused
// aggregate-then-transform kernel (fused_conv.hip): arguments, message functors and `reduce_range`, the loop that keeps U
// 16-byte row loads in flight per lane group and folds them in ORIGINAL edge order (GNNlib/src/msgpass.jl:71-79,145-149;
// NNlib's CPU scatter loop order).  See propagate.hip's header for the mapping.
#pragma once
#include "common.h"

namespace gnnmp {

struct ReduceArgs {
    const uint32_t *rowptr;
    const int32_t *row_order; // [n_rows] rows by decreasing length, or null (common.h: gnnmp_graph::row_order)
    const int32_t *idx;      // per slot: source row of x to read (plan->col, or plan->eid for scatter)
    const int32_t *eid;      // per slot: original edge position (weights lookup); unused unless w
    const float *x;          // [n_src][D]
    const float *w;          // [n_edges] original order, nullable
    const float *emat;       // [n_edges][D] original order: per-edge, per-feature factor (e_mul_xj with a matrix e)
    const float *rowsub;     // [n_dst][D]: the message is exp(x - rowsub[row]) (softmax numerator, utils.jl:94)
    const float *rowden;     // [n_dst][D]: softmax_write_kernel divides by it
    float den_add;           // softmax_edges adds eps(T) to the denominator (utils.jl:71)
    const float *gate_i;     // [n_dst][D] GATED = 1: message = sigmoid(gate_i[row] + x[j][0:D]) .* x[j][D:2D]  (x rows are 2D wide)
                             // [n_dst][2D] GATED = 2 (cg_message, conv.jl:326-333): sigmoid(gate_i[row][0:D] + x[j][0:D] + e[0:D]) .*
                             //   act(gate_i[row][D:2D] + x[j][D:2D] + e[D:2D]), e = emat row (2D wide) of the edge, optional
    int gated;               // 0 | 1 | 2
    int act;                 // GATED = 2: dense_s's activation (gnnmp_act)
    const float *ss;         // [n_src] nullable
    const float *w_slot;     // [E'] slot order, nullable (takes precedence over w)
    const float *ss_slot;    // [E'] slot order, nullable (takes precedence over ss)
    const float *sd;         // [n_dst] nullable
    float *out;              // [n_dst][D]
    const float *bias;       // [D] or null: added to the finished row (after the destination scaling) ...
    int bias_relu;           // ... and then relu, if set: the layer epilogue σ.(x .+ b) of gcn_conv's W-first branch (conv.jl:36-40,71)
    const float *addend;     // [n_dst][D] or null: out = row + addend[row] — the other summand of graph_conv's pullback
                             //   Δx = Δz W_root + Aᵀ(Δz W_agg) (conv.jl:102-108), added where the row is finished instead of in a pass of its own
    const float *mask_y;     // [n_dst][D] or null: out = mask_y[row] > 0 ? out : 0 — relu' of the layer below (its stored output), ditto
    float *partial;          // [n_chunks][D]
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    const int32_t *long_rows, *long_cptr;
    int n_chunks;
    int n_long;
    int D;
    int n_rows;
    int n_src;               // rows of x (XCD-remap heuristic)
    uint32_t n_edges;        // weights exist for eid < n_edges; others are 1 (edge positions are unsigned 32-bit)
    int log2g;
    int mean;
    int long_thresh;
    int cpx;                 // logical blocks per XCD (grid.x = nbc + 8*cpx) ; 0 = no remap
    int nbc;                 // leading blocks (chunk virtual rows) that are not remapped
    int waves;               // waves per block
    int compact_long;        // csr_combine_kernel writes long row r to out[r] (a compact [n_long][D] buffer), not out[row]
    // fold-in-kernel (FOLD instances of csr_rows_kernel, see long_geom below): chunk v belongs to long row chunk_lrow[v]; arrival
    // counters per slice of the row's chunks and per row — the LAST one to arrive folds, and resets the counter for the next launch
    const int32_t *chunk_lrow;
    uint32_t *arrive;        // [n_long][tiles][256 / G + 1]: per slice, then the row's own
    float *spart;            // [n_long][256 / G][D] slice partials
};

// The gated functors evaluate their activations once per edge and feature (6e9 times on the products shape): with libm's
// expf / log1pf / tanhf the kernel is ALU-bound at 4x its HBM time.  These use the hardware transcendentals (v_exp_f32,
// v_log_f32, v_rcp_f32: 1 ulp each) on the same formulas; measured deviation from the libm forms < 3e-7 relative, far
// inside the 1e-5 parity bound of the layers that use them.
__device__ __forceinline__ float hw_exp_neg_abs(float x) {   // exp(-|x|) in (0, 1]
    return __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
}
// NNlib.sigmoid: t = exp(-abs(x)); ifelse(x >= 0, inv(1 + t), t / (1 + t))
__device__ __forceinline__ float nn_sigmoid(float x) {
    const float t = hw_exp_neg_abs(x);
    const float r = __builtin_amdgcn_rcpf(1.0f + t);
    return x >= 0.0f ? r : t * r;
}
// NNlib.softplus: log1p(exp(-abs(x))) + relu(x)
__device__ __forceinline__ float nn_softplus(float x) {
    const float t = hw_exp_neg_abs(x);
    // log1p(t): log(1 + t) loses the low bits of a small t, the series t - t^2/2 + t^3/3 does not (t < 2^-8: error < t^4/4)
    const float l = t < 0.00390625f ? t * (1.0f - t * (0.5f - t * 0.333333343f))
                                    : __builtin_amdgcn_logf(1.0f + t) * 0.693147180559945309f;
    return l + (x < 0.0f ? 0.0f : x);
}
// tanh: odd series near zero (where 1 - 2 / (1 + e^2x) cancels), the exponential form elsewhere
__device__ __forceinline__ float nn_tanh(float x) {
    const float ax = fabsf(x);
    if (ax < 0.125f) {
        const float x2 = x * x;
        return x * (1.0f - x2 * (0.333333343f - x2 * (0.133333340f - x2 * 0.0539682545f)));
    }
    const float e = __builtin_amdgcn_exp2f(-2.0f * ax * 1.44269504088896341f);   // e^(-2|x|)
    const float r = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return x < 0.0f ? -r : r;
}

// dense_s's σ of CGConv (constructor argument `act`, GraphNeuralNetworks/src/layers/conv.jl:925-930)
__device__ __forceinline__ float cg_act(float x, int act) {
    switch (act) {
        case GNNMP_ACT_RELU: return x < 0.0f ? 0.0f : x;
        case GNNMP_ACT_SOFTPLUS: return nn_softplus(x);
        case GNNMP_ACT_TANH: return nn_tanh(x);
        default: return x;
    }
}

// reduce slots [beg, end) of one destination into acc[VEC]; all lanes of the group call this together.
template <int VEC, int OP, bool SCALED, int U, bool EMAT = false, bool EXPSUB = false, int GATED = 0>
__device__ __forceinline__ void reduce_range(const ReduceArgs &a, uint32_t beg, uint32_t end, int lig,
                                             int gbase, int G, int f0, bool active,
                                             float acc[VEC], int row = 0) {
    float sub[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) sub[q] = 0.0f;
    if (EXPSUB && active) Vec<VEC>::load(a.rowsub + (int64_t)row * a.D + f0, sub);
    float sub2[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) sub2[q] = 0.0f;
    if (GATED == 1 && active) Vec<VEC>::load(a.gate_i + (int64_t)row * a.D + f0, sub);   // Ax_i slice
    if (GATED == 2 && active) {
        Vec<VEC>::load(a.gate_i + (int64_t)row * 2 * a.D + f0, sub);                     // dense_f's share of x_i
        Vec<VEC>::load(a.gate_i + (int64_t)row * 2 * a.D + a.D + f0, sub2);              // dense_s's share of x_i
    }
    const int64_t ldx = GATED ? 2 * (int64_t)a.D : (int64_t)a.D;
    // Slots are UNSIGNED 32-bit (a plan holds fewer than 2^32 - 65536 of them): the walk costs what it cost with int32 slots.
    // (A 64-bit rowptr was tried first: 64-bit loop counters cost the VALU-sensitive attention kernel 7 % on the products shape,
    // and with per-row base pointers instead the row kernels went from 78 to 90 VGPRs — 6 -> 5 waves per SIMD, arxiv shape +4 %.)
    for (uint32_t base = beg; base < end; base += G) {
        const uint32_t p = base + lig;
        uint32_t c = 0, ev = 0;     // row ids: sources (< 2^31) or, for _scatter, edge positions (< 2^32)
        float wv = 1.0f, sv = 1.0f;
        if (p < end) {
            c = (uint32_t)a.idx[p];
            if (EMAT) ev = (uint32_t)a.eid[p];
            if (SCALED) {
                if (a.w_slot) {
                    wv = a.w_slot[p];
                } else if (a.w) {
                    const uint32_t e = (uint32_t)a.eid[p];
                    if (e < a.n_edges) wv = a.w[e];
                }
                if (a.ss_slot)
                    sv = a.ss_slot[p];
                else if (a.ss)
                    sv = a.ss[c];
            }
        }
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float v[U][VEC];
            float gb[GATED ? U : 1][VEC];
            float em[EMAT ? U : 1][VEC];
            float em2[(EMAT && GATED == 2) ? U : 1][VEC];
            float wj[U], sj[U];
            // every cross-lane broadcast of the batch first (one LDS round trip), then the row loads back to back
            uint32_t cjs[U], ejs[EMAT ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = min(j + u, n - 1);
                cjs[u] = (uint32_t)__shfl((int)c, gbase + jj, 64);
                if (EMAT) ejs[EMAT ? u : 0] = (uint32_t)__shfl((int)ev, gbase + jj, 64);
                if (SCALED) {
                    wj[u] = __shfl(wv, gbase + jj, 64);
                    sj[u] = __shfl(sv, gbase + jj, 64);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t cj = cjs[u];
                if (EMAT) {
                    // e .* xj with e (D, E'): the edge's own row of factors, by original edge position; self loops the
                    // plan added carry no features and weigh 1
                    const uint32_t ej = ejs[EMAT ? u : 0];
                    if (GATED == 2) {   // the edge's share of both pre-activations (additive: absent = 0)
                        if (active && (j + u < n) && ej < a.n_edges) {
                            Vec<VEC>::load(a.emat + (int64_t)ej * 2 * a.D + f0, em[EMAT ? u : 0]);
                            Vec<VEC>::load(a.emat + (int64_t)ej * 2 * a.D + a.D + f0, em2[(EMAT && GATED == 2) ? u : 0]);
                        } else {
#pragma unroll
                            for (int q = 0; q < VEC; ++q) em[EMAT ? u : 0][q] = em2[(EMAT && GATED == 2) ? u : 0][q] = 0.0f;
                        }
                    } else if (active && (j + u < n) && ej < a.n_edges) {
                        Vec<VEC>::load(a.emat + (int64_t)ej * a.D + f0, em[EMAT ? u : 0]);
                    } else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) em[EMAT ? u : 0][q] = 1.0f;
                    }
                }
                if (active && (j + u < n)) {
                    if (GATED) {
                        Vec<VEC>::load(a.x + (int64_t)cj * ldx + f0, gb[GATED ? u : 0]);          // Bx_j
                        Vec<VEC>::load(a.x + (int64_t)cj * ldx + a.D + f0, v[u]);                 // Vx_j
                    } else {
                        Vec<VEC>::load(a.x + (int64_t)cj * ldx + f0, v[u]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[u][q] = 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < n) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        float t = v[u][q];
                        if (SCALED) {
                            t = t * sj[u];  // xj .* cout'   (GNNlib/src/layers/conv.jl:59), rounded
                            t = wj[u] * t;  // w .* xj        (GNNlib/src/msgpass.jl:203-208), rounded
                        }
                        if (EMAT && GATED != 2) t = em[EMAT ? u : 0][q] * t;  // e .* xj (GNNlib/src/msgpass.jl:187-191)
                        if (EXPSUB) t = expf(t - sub[q]);        // num = exp.(e .- max_) (GNNlib/src/utils.jl:94)
                        if (GATED == 1) t = nn_sigmoid(sub[q] + gb[GATED ? u : 0][q]) * t;   // sigmoid.(Ax_i .+ Bx_j) .* Vx_j (conv.jl:291)
                        if (GATED == 2) {   // dense_f(z) .* dense_s(z), z = vcat(xi, xj, e): the K-sum in z's order (conv.jl:326-333)
                            float f = sub[q] + gb[GATED ? u : 0][q], g = sub2[q] + t;
                            if (EMAT) {
                                f = f + em[EMAT ? u : 0][q];
                                g = g + em2[(EMAT && GATED == 2) ? u : 0][q];
                            }
                            t = nn_sigmoid(f) * cg_act(g, a.act);
                        }
                        acc[q] = op_apply<OP>(acc[q], t);
                    }
                }
            }
        }
    }
}

// mean division and the destination scaling of one finished row (no store)
template <int VEC, int OP>
__device__ __forceinline__ void finalize_row(const ReduceArgs &a, int row, uint32_t len, float acc[VEC]) {
    if (OP == OP_SUM && a.mean) {
        // NNlib scatter(mean): dst = 0 .+ safe_div.(sum, count); count == 0 keeps the sum (0)
        const float cnt = (float)len;
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = 0.0f + (len == 0 ? acc[q] : acc[q] / cnt);
    }
    if (a.sd) {
        const float s = a.sd[row];  // x .* cin'  (GNNlib/src/layers/conv.jl:67)
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] * s;
    }
}
// ... and the store to out[out_row] (out_row = row, or the compact slot of a split row: ReduceArgs::compact_long)
template <int VEC, int OP>
__device__ __forceinline__ void finalize_store(const ReduceArgs &a, int row, uint32_t len, int f0,
                                               bool active, float acc[VEC], int out_row = -1) {
    finalize_row<VEC, OP>(a, row, len, acc);
    if (a.bias && active) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] + a.bias[f0 + q];
    }
    if (a.bias_relu) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = acc[q] < 0.0f ? 0.0f : acc[q];
    }
    if (a.addend && active) {
        float ad[VEC];
        Vec<VEC>::load(a.addend + (int64_t)row * a.D + f0, ad);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = ad[q] + acc[q];
    }
    if (a.mask_y && active) {
        float my[VEC];
        Vec<VEC>::load(a.mask_y + (int64_t)row * a.D + f0, my);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = my[q] > 0.0f ? acc[q] : 0.0f;
    }
    if (active) Vec<VEC>::store(a.out + (int64_t)(out_row < 0 ? row : out_row) * a.D + f0, acc);
}

// The fold of csr_combine_kernel (propagate.hip) INSIDE the row kernel, by whoever arrives last — in the combine kernel's own two levels,
// so a row comes out bit-identical to the two-kernel path and no lane group walks a hub's 200 partials alone:
//   level 1  the chunks of long row r fall into 256 / G contiguous SLICES of `per` chunks (csr_combine_kernel's lane groups); the last chunk
//            of a slice to store its partial folds the slice from the identity, in chunk order, into a slice partial (fold_slice);
//   level 2  the last slice of the row to finish folds the slice partials in slice order and runs the row's epilogue (fold_slices).
// A 12 800-edge hub (200 chunks, G = 32): 8 slices of 25 partials folded by 8 different lane groups + 8 slice partials, instead of one
// chain of 200.  The loads are 4 deep (the row kernel's register count — its occupancy — is set by its main loop, not by this tail).
struct LongGeom {
    int c0, c1, per, ns;      // chunk range of the row, chunks per slice, non-empty slices
};
__device__ __forceinline__ LongGeom long_geom(const int32_t *long_cptr, int r, int log2g) {
    LongGeom g;
    const int NG = 256 >> log2g;
    g.c0 = long_cptr[r];
    g.c1 = long_cptr[r + 1];
    g.per = (g.c1 - g.c0 + NG - 1) / NG;
    g.ns = (g.c1 - g.c0 + g.per - 1) / g.per;
    return g;
}
template <int VEC, int OP>
__device__ __forceinline__ void fold_rows(const float *__restrict__ base, int64_t stride, int n, int f0, bool active, float acc[VEC],
                                          bool from_identity) {
    // acc = (from_identity ? identity : row 0) folded with the following rows in order; rows are `stride` floats apart
    constexpr int CB = 4;
    int c = 0;
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP>();
    if (!active) return;
    if (!from_identity) {
        coh_load<VEC>(base + f0, acc);
        c = 1;
    }
    for (; c < n; c += CB) {
        float v[CB][VEC];
#pragma unroll
        for (int u = 0; u < CB; ++u) coh_load<VEC>(base + (int64_t)min(c + u, n - 1) * stride + f0, v[u]);
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (c + u < n) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = op_apply<OP>(acc[q], v[u][q]);
            }
        }
    }
}

// A chunk's lane group has stored its partial (coh_store): count it in; true for the group that completes its slice / row (it then folds,
// reading through coh_load).  No fence: see common.h coh_store.
__device__ __forceinline__ bool chunk_arrive(uint32_t *counter, int n, int lig, int gbase) {
    coh_publish();
    unsigned prev = 0;
    if (lig == 0) prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prev = (unsigned)__shfl((int)prev, gbase, 64);
    if (prev != (unsigned)(n - 1)) return false;
    // ready for the plan's next launch (every participant has arrived: nobody touches the counter again in this one)
    if (lig == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

}  // namespace gnnmp
