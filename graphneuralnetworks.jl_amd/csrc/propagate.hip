// propagate.hip — the fused gather -> message -> per-destination reduce of GNNlib's propagate
// (GNNlib/src/msgpass.jl:71-79), i.e. apply_edges (:121-129: _gather(xj, s), message fn) followed by
// aggregate_neighbors (:145-149: _scatter(aggr, m, t, n)), without ever materialising the (D, E') message
// array.  Also the reference's SpMM fast path `xj * adjacency_matrix(g)` (:215-238): same kernel.
//
// HBM-bound (0.33-0.48 flop/B): the design goal is to keep every lane's 16-byte row loads in flight.
//   - one GROUP of G = 2^k lanes owns one destination row; lane l of the group owns VEC consecutive features
//     (G*VEC >= D; D=128 -> 32 lanes x float4, two rows per wave; D=100 -> 25 of 32 lanes active).
//   - the row's source ids (and the per-edge factors, when given in slot order) are loaded coalesced, G at a time,
//     one per lane, and broadcast inside the group with ds_bpermute (__shfl) — the CDNA cross-lane path, no LDS
//     allocation, no barrier.
//   - U row loads are issued back to back before the first add; the adds then run in ORIGINAL edge order,
//     one fp32 accumulator chain per feature => bit-identical to NNlib's CPU scatter loop, no atomics.
//   - rows longer than the plan's threshold are cut into balanced chunks (plan.hip); a chunk is processed by the SAME
//     kernel as a virtual row whose result goes to a partial buffer, and a small second kernel folds each long row's
//     partials in chunk order.  The dominant kernel therefore touches every edge exactly once and has no tail.
//   - block -> row-chunk mapping is XCD-aware (contiguous destination ranges per XCD / L2).
#include "csr_reduce.h"

namespace gnnmp {

// virtual rows: [0, n_chunks) are chunks of long rows (raw partials), [n_chunks, n_chunks + n_rows) ordinary rows.
// FOLD: no second kernel for the split rows — the last chunk of a long row to arrive folds the row's partials itself, in the order of
// csr_combine_kernel (csr_reduce.h: chunk_arrive, fold_long_row).  One launch and one launch seam less per call: 5.6 us + the seam of a
// 107 us arxiv-shaped propagate.
template <int VEC, int OP, bool SCALED, int U, bool EMAT = false, bool EXPSUB = false, int GATED = 0, bool FOLD = false>
__global__ void __launch_bounds__(256) csr_rows_kernel(const ReduceArgs a) {
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
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP>();
    if (v < a.n_chunks) {
        reduce_range<VEC, OP, SCALED, U, EMAT, EXPSUB, GATED>(a, a.chunk_beg[v], a.chunk_end[v], lig, gbase, G, f0, active, acc,
                                                              (EXPSUB || GATED) ? a.chunk_row[v] : 0);
        if (active) {
            if (FOLD) coh_store<VEC>(a.partial + (int64_t)v * a.D + f0, acc);      // read by another workgroup of this launch
            else Vec<VEC>::store(a.partial + (int64_t)v * a.D + f0, acc);
        }
        if (FOLD) {
            const int r = a.chunk_lrow[v];
            const LongGeom lg = long_geom(a.long_cptr, r, a.log2g);
            const int NG = 256 >> a.log2g;
            const int k = (v - lg.c0) / lg.per;
            const int s0 = lg.c0 + k * lg.per, s1 = min(lg.c1, s0 + lg.per);
            uint32_t *cnt = a.arrive + ((int64_t)r * gridDim.y + blockIdx.y) * (NG + 1);
            if (!chunk_arrive(cnt + k, s1 - s0, lig, gbase)) return;
            fold_rows<VEC, OP>(a.partial + (int64_t)s0 * a.D, a.D, s1 - s0, f0, active, acc, true);
            float *sp = a.spart + (int64_t)r * NG * a.D;
            if (active) coh_store<VEC>(sp + (int64_t)k * a.D + f0, acc);
            if (!chunk_arrive(cnt + NG, lg.ns, lig, gbase)) return;
            fold_rows<VEC, OP>(sp, a.D, lg.ns, f0, active, acc, false);
            const int lrow = a.long_rows[r];
            finalize_store<VEC, OP>(a, lrow, a.rowptr[lrow + 1] - a.rowptr[lrow], f0, active, acc, a.compact_long ? r : -1);
        }
        return;
    }
    int row = v - a.n_chunks;
    if (a.row_order) row = a.row_order[row];
    const uint32_t beg = a.rowptr[row];
    const uint32_t end = a.rowptr[row + 1];
    if (end - beg > (uint32_t)a.long_thresh) return;  // split row: its chunks are virtual rows, folded by csr_combine_kernel
    reduce_range<VEC, OP, SCALED, U, EMAT, EXPSUB, GATED>(a, beg, end, lig, gbase, G, f0, active, acc, row);
    finalize_store<VEC, OP>(a, row, end - beg, f0, active, acc);
}

// softmax_edge_neighbors, last step (GNNlib/src/utils.jl:96): alpha[k] = exp(e[k] - max_[t_k]) / den[t_k] for every edge of
// the (virtual) row, written back in ORIGINAL edge order.  idx = the plan's eid (rows of e / alpha).
template <int VEC, int U>
__global__ void __launch_bounds__(256) softmax_write_kernel(const ReduceArgs a) {
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
    const int f0 = ((int)blockIdx.y * G + lig) * VEC;
    const bool active = f0 < a.D;
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
        if (end - beg > a.long_thresh) return;
    }
    float mx[VEC], den[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) mx[q] = den[q] = 1.0f;
    if (active) {
        Vec<VEC>::load(a.rowsub + (int64_t)row * a.D + f0, mx);
        Vec<VEC>::load(a.rowden + (int64_t)row * a.D + f0, den);
    }
    if (a.den_add != 0.0f) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) den[q] = den[q] + a.den_add;
    }
    float *outp = a.out;
    for (uint32_t base = beg; base < end; base += G) {   // slots are unsigned 32-bit (csr_reduce.h)
        const uint32_t p = base + lig;
        const int c = p < end ? a.idx[p] : 0;
        const int n = (int)min((uint32_t)G, end - base);
        for (int j = 0; j < n; j += U) {
            float x[U][VEC];
            uint32_t cj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cj[u] = (uint32_t)__shfl(c, gbase + min(j + u, n - 1), 64);
                if (active) Vec<VEC>::load(a.x + (int64_t)cj[u] * a.D + f0, x[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (active && (j + u < n)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) x[u][q] = expf(x[u][q] - mx[q]) / den[q];
                    Vec<VEC>::store(outp + (int64_t)cj[u] * a.D + f0, x[u]);
                }
            }
        }
    }
}

// one BLOCK per long row: its 256 / G lane groups fold contiguous slices of the row's chunk partials (in chunk order, 8
// loads in flight), the slice results meet in LDS and group 0 folds them in slice order, then the usual epilogue.  (One
// lane group per row walked a 12 800-edge hub's 200 chunks alone: 14-40 us of a 250 us arxiv layer.)
template <int VEC, int OP>
__global__ void __launch_bounds__(256) csr_combine_kernel(const ReduceArgs a) {
    __shared__ float red[256 * VEC];
    const int G = 1 << a.log2g;
    const int lig = threadIdx.x & (G - 1);
    const int grp = threadIdx.x >> a.log2g;
    const int NG = 256 >> a.log2g;
    const int r = blockIdx.x;
    const int f0 = ((int)blockIdx.y * G + lig) * VEC;
    const bool active = f0 < a.D;
    const int row = a.long_rows[r];
    const int c0 = a.long_cptr[r], c1 = a.long_cptr[r + 1];
    const int per = (c1 - c0 + NG - 1) / NG;
    const int s0 = min(c1, c0 + grp * per), s1 = min(c1, s0 + per);
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP>();
    if (active) {
        constexpr int CB = 8;
        for (int c = s0; c < s1; c += CB) {
            float v[CB][VEC];
#pragma unroll
            for (int u = 0; u < CB; ++u) Vec<VEC>::load(a.partial + (int64_t)min(c + u, s1 - 1) * a.D + f0, v[u]);
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                if (c + u < s1) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[q] = op_apply<OP>(acc[q], v[u][q]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) red[threadIdx.x * VEC + q] = acc[q];
    __syncthreads();
    if (grp != 0) return;
    for (int k = 1; k < NG; ++k) {
        if (c0 + k * per >= c1) break;               // empty slices hold the identity: nothing to fold
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = op_apply<OP>(acc[q], red[((k << a.log2g) + lig) * VEC + q]);
    }
    finalize_store<VEC, OP>(a, row, a.rowptr[row + 1] - a.rowptr[row], f0, active, acc, a.compact_long ? r : -1);
}

// nn_conv's propagate (GNNlib/src/layers/conv.jl:260-273): the message of edge k is W_k x_j with W_k = reshape(nn(e_k), out, in)
// — an (out, in) matrix PER EDGE, read once, by original edge position (`we`, Julia column-major: element (o, c) at o + out * c).
// One group of G >= out lanes per destination (virtual) row, lane o owns output feature o: for every in-channel the out
// weights of the edge are one contiguous, coalesced load and x_j[c] is a broadcast.  Edges in original order, products
// rounded then added (the reference runs a batched gemm: tolerance, not bits).  Chunks of long rows leave raw partials for
// csr_combine_kernel like every other row kernel.
template <int OP>
__global__ void __launch_bounds__(256) nn_rows_kernel(const ReduceArgs a, const float *__restrict__ we, int Din) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int G = 1 << a.log2g;
    const int lig = lane & (G - 1);
    const int grp = lane >> a.log2g;
    const int rpw = 64 >> a.log2g;
    const int64_t v64 = ((int64_t)blockIdx.x * a.waves + wave) * rpw + grp;
    if (v64 >= (int64_t)a.n_rows + a.n_chunks) return;
    const int v = (int)v64;
    const int o = (int)blockIdx.y * G + lig;
    const bool active = o < a.D;
    int row = 0;
    uint32_t beg, end;
    const bool is_chunk = v < a.n_chunks;
    if (is_chunk) {
        beg = a.chunk_beg[v];
        end = a.chunk_end[v];
    } else {
        row = v - a.n_chunks;
        beg = a.rowptr[row];
        end = a.rowptr[row + 1];
        if (end - beg > a.long_thresh) return;
    }
    float acc = op_identity<OP>();
    const int oc = active ? o : 0;
    // EB edges at a time: their EB x 2 in-channel loads are independent (a row walks its edges alone — with one edge in flight
    // the kernel ran at a third of the HBM rate); messages are still formed per edge and folded in edge order.  Slots past
    // the end re-read the last edge and are not folded.
    constexpr int EB = 4;
    for (uint32_t p0 = beg; p0 < end; p0 += EB) {
        const float *wk[EB], *xr[EB];
        bool ok[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            const uint32_t p = min(p0 + u, end - 1);
            const int cj = a.idx[p];
            const uint32_t ej = (uint32_t)a.eid[p];
            ok[u] = p0 + u < end && ej < a.n_edges;          // a self loop the plan added: no edge features, no message
            wk[u] = we + (int64_t)(a.n_edges ? min(ej, a.n_edges - 1) : 0u) * a.D * Din + oc;
            xr[u] = a.x + (int64_t)cj * Din;
        }
        float m[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) m[u] = 0.0f;
        int c = 0;
        for (; c + 2 <= Din; c += 2) {
            float w0[EB], w1[EB], x0[EB], x1[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                w0[u] = wk[u][(int64_t)c * a.D];
                w1[u] = wk[u][(int64_t)(c + 1) * a.D];
                x0[u] = xr[u][c];
                x1[u] = xr[u][c + 1];
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                m[u] = m[u] + w0[u] * x0[u];
                m[u] = m[u] + w1[u] * x1[u];
            }
        }
        if (c < Din) {
#pragma unroll
            for (int u = 0; u < EB; ++u) m[u] = m[u] + wk[u][(int64_t)c * a.D] * xr[u][c];
        }
#pragma unroll
        for (int u = 0; u < EB; ++u)
            if (ok[u]) acc = op_apply<OP>(acc, m[u]);
    }
    if (is_chunk) {
        if (active) a.partial[(int64_t)v * a.D + o] = acc;
        return;
    }
    float av[1] = {acc};
    finalize_store<1, OP>(a, row, end - beg, o, active, av);
}

template <int VEC, int OP, bool SCALED, int U, bool EMAT = false, bool EXPSUB = false, int GATED = 0>
static int launch_reduce(const ReduceArgs &a0, hipStream_t stream) {
    ReduceArgs a = a0;
    const int G = 1 << a.log2g;
    const int rpw = 64 / G;
    int waves = knob(KNOB_BLOCK_WAVES);
    if (waves < 1 || waves > 4) waves = 4;
    a.waves = waves;
    const int rows_per_block = rpw * waves;
    const int64_t nvirt = (int64_t)a.n_rows + a.n_chunks;
    const int64_t chunks = (nvirt + rows_per_block - 1) / rows_per_block;
    const int lanes_needed = (a.D + VEC - 1) / VEC;
    const int tiles = (lanes_needed + G - 1) / G;
    if (chunks > 0) {
        int64_t gx = chunks;
        a.cpx = 0;
        if (use_xcd_remap(a.n_src, a.D, chunks)) {
            a.nbc = (int)std::min<int64_t>(chunks, (a.n_chunks + rows_per_block - 1) / rows_per_block);
            a.cpx = (int)((chunks - a.nbc + 7) / 8);
            gx = (int64_t)a.nbc + (int64_t)a.cpx * 8;
        }
        dim3 grid((unsigned)gx, (unsigned)tiles);
        // the plain propagate / scatter instances fold their split rows themselves (a.arrive set by run_reduce when it may)
        constexpr bool CAN_FOLD = !EMAT && !EXPSUB && GATED == 0 && U == 8;
        if (CAN_FOLD && a.arrive && a.n_long > 0) {
            csr_rows_kernel<VEC, OP, SCALED, U, EMAT, EXPSUB, GATED, CAN_FOLD><<<grid, 64 * waves, 0, stream>>>(a);
            GNNMP_LAUNCH_CHECK("csr_rows_kernel<FOLD>");
            return GNNMP_OK;
        }
        csr_rows_kernel<VEC, OP, SCALED, U, EMAT, EXPSUB, GATED><<<grid, 64 * waves, 0, stream>>>(a);
        GNNMP_LAUNCH_CHECK("csr_rows_kernel");
    }
    if (a.n_long > 0) {
        dim3 grid((unsigned)a.n_long, (unsigned)tiles);
        csr_combine_kernel<VEC, OP><<<grid, 256, 0, stream>>>(a);
        GNNMP_LAUNCH_CHECK("csr_combine_kernel");
    }
    return GNNMP_OK;
}

template <int VEC, int OP, bool SCALED>
static int dispatch_u(const ReduceArgs &a, hipStream_t s) {
    switch (knob(KNOB_UNROLL)) {
        case 2: return launch_reduce<VEC, OP, SCALED, 2>(a, s);
        case 4: return launch_reduce<VEC, OP, SCALED, 4>(a, s);
        default: return launch_reduce<VEC, OP, SCALED, 8>(a, s);
    }
}
template <int VEC, int OP>
static int dispatch_scaled(const ReduceArgs &a, bool scaled, hipStream_t s) {
    return scaled ? dispatch_u<VEC, OP, true>(a, s) : dispatch_u<VEC, OP, false>(a, s);
}
template <int VEC>
static int dispatch_op(const ReduceArgs &a, int op, bool scaled, hipStream_t s) {
    if (a.rowsub) return launch_reduce<VEC, OP_SUM, false, 8, false, true>(a, s);   // softmax denominator
    if (a.gate_i && a.gated == 2) {   // cg_conv aggregates with + (conv.jl:313)
        return a.emat ? launch_reduce<VEC, OP_SUM, false, 2, true, false, 2>(a, s)
                      : launch_reduce<VEC, OP_SUM, false, 4, false, false, 2>(a, s);
    }
    if (a.gate_i) {   // two rows per edge in flight: half the batch
        switch (op) {
            case OP_SUM: return launch_reduce<VEC, OP_SUM, false, 4, false, false, 1>(a, s);
            case OP_MAX: return launch_reduce<VEC, OP_MAX, false, 4, false, false, 1>(a, s);
            default: return launch_reduce<VEC, OP_MIN, false, 4, false, false, 1>(a, s);
        }
    }
    if (a.emat) {   // two rows per edge in flight: half the batch
        switch (op) {
            case OP_SUM: return launch_reduce<VEC, OP_SUM, false, 4, true>(a, s);
            case OP_MAX: return launch_reduce<VEC, OP_MAX, false, 4, true>(a, s);
            default: return launch_reduce<VEC, OP_MIN, false, 4, true>(a, s);
        }
    }
    switch (op) {
        case OP_SUM: return dispatch_scaled<VEC, OP_SUM>(a, scaled, s);
        case OP_MAX: return dispatch_scaled<VEC, OP_MAX>(a, scaled, s);
        default: return dispatch_scaled<VEC, OP_MIN>(a, scaled, s);
    }
}

// shared by propagate (idx = col) and scatter (idx = eid)
int run_reduce(gnnmp_graph_t *p, const int32_t *idx, int aggr, const float *x, const float *w,
               const float *ss, const float *w_slot, const float *ss_slot, const float *sd, float *out,
               int64_t D, hipStream_t stream, const float *emat = nullptr, const float *rowsub = nullptr,
               const float *gate_i = nullptr, int gated = 1, int act = 0, int long_only = 0, const float *bias = nullptr,
               int bias_relu = 0, const float *addend = nullptr, const float *mask_y = nullptr) {
    // long_only: reduce ONLY the split rows (their chunk virtual rows + the combine) and write row long_rows[r], finalised, to
    // out[r] — a compact [n_long][D] buffer the fused kernel reads instead of walking those rows (the caller sized the
    // workspace and passes out inside it)
    if (p->n_dst == 0 || D == 0) return GNNMP_OK;
    if (long_only && p->n_long == 0) return GNNMP_OK;
    if (p->n_chunks > 0 && !long_only) {
        if (int rc = ensure_workspace(p, (size_t)p->n_chunks * (size_t)D)) return rc;
    }
    ReduceArgs a;
    a.compact_long = long_only;
    a.rowptr = p->rowptr;
    a.row_order = nullptr;
    a.idx = idx;
    a.eid = p->eid;
    a.x = x;
    a.w = w;
    a.emat = emat;
    a.rowsub = rowsub;
    a.rowden = nullptr;
    a.den_add = 0.0f;
    a.gate_i = gate_i;
    a.gated = gate_i ? gated : 0;
    a.act = act;
    a.ss = ss;
    a.w_slot = w_slot;
    a.ss_slot = ss_slot;
    a.sd = sd;
    a.out = out;
    a.bias = bias;
    a.bias_relu = bias_relu;
    a.addend = addend;
    a.mask_y = mask_y;
    a.partial = p->ws;
    a.chunk_row = p->chunk_row;
    a.chunk_beg = p->chunk_beg;
    a.chunk_end = p->chunk_end;
    a.long_rows = p->long_rows;
    a.long_cptr = p->long_cptr;
    a.n_chunks = p->n_chunks;
    a.n_long = p->n_long;
    a.D = (int)D;
    a.n_rows = long_only ? 0 : (int)p->n_dst;
    a.n_src = (int)p->n_src;
    a.n_edges = (uint32_t)p->n_edges;
    a.mean = (aggr == GNNMP_MEAN);
    a.long_thresh = p->long_thresh;
    a.cpx = 0;
    a.nbc = 0;
    a.waves = 4;
    a.chunk_lrow = p->chunk_lrow;
    a.arrive = nullptr;
    int vec = pick_vec(D, x, out);
    if (addend && (reinterpret_cast<uintptr_t>(addend) & (4 * vec - 1)) != 0) vec = 1;
    if (mask_y && (reinterpret_cast<uintptr_t>(mask_y) & (4 * vec - 1)) != 0) vec = 1;
    if (emat && (reinterpret_cast<uintptr_t>(emat) & (4 * vec - 1)) != 0) vec = 1;
    if (gate_i && (reinterpret_cast<uintptr_t>(gate_i) & (4 * vec - 1)) != 0) vec = 1;
    a.log2g = pick_log2g((D + vec - 1) / vec);
    // >= 2 rows per wave: pair rows of equal length — but only when an output row is whole 128-byte lines: out of index order,
    // rows of 400 bytes (D = 100) leave every line half written by one wave and finished by another, measured 4.75 -> 5.11 ms
    if (!long_only && use_row_order(p->n_src, D) && a.log2g <= 5 && idx == p->col && (D & 31) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 127) == 0) {
        if (int rc = ensure_row_order(p, stream)) return rc;
        a.row_order = p->row_order;
    }
    const int op = (aggr == GNNMP_MAX) ? OP_MAX : (aggr == GNNMP_MIN ? OP_MIN : OP_SUM);
    const bool scaled = w || ss || w_slot || ss_slot;
    a.spart = nullptr;
    if (p->n_long > 0 && use_fold()) {      // split rows folded inside the row kernel: arrival counters per (long row, feature tile, slice)
        const int G = 1 << a.log2g, NG = 256 >> a.log2g;
        const size_t tiles = (size_t)(((D + vec - 1) / vec + G - 1) / G);
        if (int rc = ensure_arrive(p, (size_t)p->n_long * tiles * (size_t)(NG + 1), (size_t)p->n_long * (size_t)NG * (size_t)D, stream)) return rc;
        a.arrive = p->arrive;
        a.spart = p->spart;
    }
    switch (vec) {
        case 4: return dispatch_op<4>(a, op, scaled, stream);
        case 2: return dispatch_op<2>(a, op, scaled, stream);
        default: return dispatch_op<1>(a, op, scaled, stream);
    }
}

// softmax over the rows of a plan in the reference's three steps (utils.jl:84-97 / :49-72): max_ = scatter(max, e, t),
// den = scatter(+, exp.(e .- max_[t]), t) (edge order), alpha = num ./ den.  Every step is the balanced row-group walk
// (chunked long rows), so a 17 000-edge hub costs what 17 000 edges cost, not a serial tail.
int softmax_rows_try(gnnmp_graph_t *p, const float *e, float *alpha, int64_t D, float den_add, float *partial, float *mx, float *den,
                     hipStream_t stream);  // softmax_rows.hip

int run_softmax(gnnmp_graph_t *p, const float *e, float *alpha, int64_t D, float den_add, hipStream_t stream) {
    if (p->n_dst == 0 || p->n_total == 0 || D == 0) return GNNMP_OK;
    const size_t nd = (size_t)p->n_dst * (size_t)D, pc = (size_t)p->n_chunks * (size_t)D;
    if (int rc = ensure_workspace(p, pc + 2 * nd)) return rc;
    float *mx = p->ws + pc, *den = mx + nd;
    // narrow rows: one pass over the rows the plan does not split, a wave per chunk on those it does (softmax_rows.hip)
    const int rc1 = softmax_rows_try(p, e, alpha, D, den_add, p->ws, mx, den, stream);
    if (rc1 != 1) return rc1;
    if (int rc = run_reduce(p, p->eid, GNNMP_MAX, e, nullptr, nullptr, nullptr, nullptr, nullptr, mx, D, stream)) return rc;
    if (int rc = run_reduce(p, p->eid, GNNMP_SUM, e, nullptr, nullptr, nullptr, nullptr, nullptr, den, D, stream, nullptr, mx))
        return rc;
    ReduceArgs a = {};
    a.rowptr = p->rowptr;
    a.idx = p->eid;
    a.x = e;
    a.out = alpha;
    a.rowsub = mx;
    a.rowden = den;
    a.den_add = den_add;
    a.chunk_row = p->chunk_row;
    a.chunk_beg = p->chunk_beg;
    a.chunk_end = p->chunk_end;
    a.n_chunks = p->n_chunks;
    a.D = (int)D;
    a.n_rows = (int)p->n_dst;
    a.long_thresh = p->long_thresh;
    a.waves = 4;
    int vec = pick_vec(D, e, alpha);
    a.log2g = pick_log2g((D + vec - 1) / vec);
    const int G = 1 << a.log2g;
    const int tiles = (int)(((D + vec - 1) / vec + G - 1) / G);
    const int64_t nvirt = (int64_t)a.n_rows + a.n_chunks;
    const int64_t rows_per_block = (int64_t)(64 / G) * a.waves;
    dim3 grid((unsigned)((nvirt + rows_per_block - 1) / rows_per_block), (unsigned)tiles);
    switch (vec) {
        case 4: softmax_write_kernel<4, 8><<<grid, 64 * a.waves, 0, stream>>>(a); break;
        case 2: softmax_write_kernel<2, 8><<<grid, 64 * a.waves, 0, stream>>>(a); break;
        default: softmax_write_kernel<1, 8><<<grid, 64 * a.waves, 0, stream>>>(a); break;
    }
    GNNMP_LAUNCH_CHECK("softmax_write_kernel");
    return GNNMP_OK;
}

// fold plan->ws ([n_chunks][D] partial sums written by another kernel in the same virtual-row layout) into out's long rows
int run_combine(gnnmp_graph_t *p, float *out, int64_t D, int aggr, hipStream_t stream) {
    if (p->n_long == 0) return GNNMP_OK;
    ReduceArgs a = {};
    a.rowptr = p->rowptr;
    a.out = out;
    a.partial = p->ws;
    a.long_rows = p->long_rows;
    a.long_cptr = p->long_cptr;
    a.n_long = p->n_long;
    a.D = (int)D;
    const int vec = pick_vec(D, p->ws, out);
    a.log2g = pick_log2g((D + vec - 1) / vec);
    const int G = 1 << a.log2g;
    const int tiles = (int)(((D + vec - 1) / vec + G - 1) / G);
    dim3 grid((unsigned)a.n_long, (unsigned)tiles);
    if (aggr == GNNMP_MAX) {
        switch (vec) {
            case 4: csr_combine_kernel<4, OP_MAX><<<grid, 256, 0, stream>>>(a); break;
            case 2: csr_combine_kernel<2, OP_MAX><<<grid, 256, 0, stream>>>(a); break;
            default: csr_combine_kernel<1, OP_MAX><<<grid, 256, 0, stream>>>(a); break;
        }
    } else {
        switch (vec) {
            case 4: csr_combine_kernel<4, OP_SUM><<<grid, 256, 0, stream>>>(a); break;
            case 2: csr_combine_kernel<2, OP_SUM><<<grid, 256, 0, stream>>>(a); break;
            default: csr_combine_kernel<1, OP_SUM><<<grid, 256, 0, stream>>>(a); break;
        }
    }
    GNNMP_LAUNCH_CHECK("csr_combine_kernel");
    return GNNMP_OK;
}
int run_combine_sum(gnnmp_graph_t *p, float *out, int64_t D, hipStream_t stream) { return run_combine(p, out, D, GNNMP_SUM, stream); }

// ---- degree / norm ------------------------------------------------------------------------------
__global__ void degree_count_kernel(const uint32_t *rowptr, int64_t n, float *deg) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) deg[i] = 0.0f + (float)(rowptr[i + 1] - rowptr[i]);
}
// weighted in-degree: one thread per destination, weights added in original edge order
// (zeros(T,N) .+ scatter(+, w, t) — GNNGraphs/src/query.jl:359-369)
__global__ void degree_weighted_kernel(const uint32_t *rowptr, const int32_t *eid, const float *w,
                                       int64_t n, uint32_t n_edges, float *deg, int long_thresh) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t beg = rowptr[i], end = rowptr[i + 1];
    if (end - beg > long_thresh) return;   // split rows: degree_long_kernel (a 17 000-edge hub made this thread a 5.6 ms tail)
    float acc = 0.0f;
    int64_t p = beg;
    for (; p + 8 <= end; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t e = (uint32_t)eid[p + u];
            v[u] = e < n_edges ? w[e] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; p < end; ++p) {
        const uint32_t e = (uint32_t)eid[p];
        acc = acc + (e < n_edges ? w[e] : 1.0f);
    }
    deg[i] = 0.0f + acc;
}
// weighted in-degree of the split rows: one block per long row, threads stride over its slots, fixed-shape LDS tree
__global__ void __launch_bounds__(256) degree_long_kernel(const uint32_t *rowptr, const int32_t *eid, const float *w,
                                                          const int32_t *long_rows, uint32_t n_edges, float *deg) {
    __shared__ float red[256];
    const int row = long_rows[blockIdx.x];
    const int64_t beg = rowptr[row], end = rowptr[row + 1];
    float acc = 0.0f;
    for (int64_t p = beg + (int)threadIdx.x; p < end; p += 256) {
        const uint32_t e = (uint32_t)eid[p];
        acc = acc + (e < n_edges ? w[e] : 1.0f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) deg[row] = 0.0f + red[0];
}
__global__ void inv_sqrt_kernel(const float *deg, float *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // plain sqrtf and '/': hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt makes both IEEE-rounded
    // (HIP's __fsqrt_rn maps to the native, 1-ulp sqrt: measured 13 % mismatches vs the CPU).
    if (i < n) out[i] = 1.0f / sqrtf(deg[i]);
}

}  // namespace gnnmp

using namespace gnnmp;

static int check_aggr(int aggr, const char *who) {
    if (aggr < GNNMP_SUM || aggr > GNNMP_MIN) return fail(GNNMP_EINVAL, "%s: bad aggr %d", who, aggr);
    return GNNMP_OK;
}

extern "C" {

int gnnmp_propagate_f32(gnnmp_graph_t *plan, int msg, int aggr, const float *xj, const float *w,
                        const float *scale_src, const float *scale_dst, float *out, int64_t D,
                        gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "propagate: null plan");
    if (int rc = check_aggr(aggr, "propagate")) return rc;
    if (msg != GNNMP_COPY_XJ && msg != GNNMP_W_MUL_XJ) return fail(GNNMP_EINVAL, "propagate: bad msg %d", msg);
    if (D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "propagate: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || (!xj && plan->n_total > 0)))
        return fail(GNNMP_EINVAL, "propagate: null xj/out");
    if (msg == GNNMP_W_MUL_XJ && !w && plan->n_edges > 0)
        return fail(GNNMP_EINVAL, "propagate: W_MUL_XJ needs w");
    if (msg == GNNMP_COPY_XJ) w = nullptr;
    return run_reduce(plan, plan->col, aggr, xj, w, scale_src, nullptr, nullptr, scale_dst, out, D,
                      (hipStream_t)stream);
}

int gnnmp_propagate_emul_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *e, float *out, int64_t D,
                             gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "propagate_emul: null plan");
    if (int rc = check_aggr(aggr, "propagate_emul")) return rc;
    if (D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "propagate_emul: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || (!xj && plan->n_total > 0) || (!e && plan->n_edges > 0)))
        return fail(GNNMP_EINVAL, "propagate_emul: null xj/e/out");
    if (!e) return run_reduce(plan, plan->col, aggr, xj, nullptr, nullptr, nullptr, nullptr, nullptr, out, D, (hipStream_t)stream);
    return run_reduce(plan, plan->col, aggr, xj, nullptr, nullptr, nullptr, nullptr, nullptr, out, D, (hipStream_t)stream, e);
}

int gnnmp_propagate_gated_f32(gnnmp_graph_t *plan, int aggr, const float *gate_i, const float *bv_j, float *out, int64_t D,
                              gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "propagate_gated: null plan");
    if (int rc = check_aggr(aggr, "propagate_gated")) return rc;
    if (D < 0 || D > (1 << 19)) return fail(GNNMP_EINVAL, "propagate_gated: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || !gate_i || (!bv_j && plan->n_total > 0)))
        return fail(GNNMP_EINVAL, "propagate_gated: null pointer");
    return run_reduce(plan, plan->col, aggr, bv_j, nullptr, nullptr, nullptr, nullptr, nullptr, out, D, (hipStream_t)stream,
                      nullptr, nullptr, gate_i);
}

int gnnmp_propagate_cg_f32(gnnmp_graph_t *plan, const float *fs_i, const float *fs_j, const float *fs_e, int act, float *out,
                           int64_t D, gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "propagate_cg: null plan");
    if (act < GNNMP_ACT_IDENTITY || act > GNNMP_ACT_TANH) return fail(GNNMP_EINVAL, "propagate_cg: bad act %d", act);
    if (D < 0 || D > (1 << 19)) return fail(GNNMP_EINVAL, "propagate_cg: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || !fs_i || (!fs_j && plan->n_total > 0)))
        return fail(GNNMP_EINVAL, "propagate_cg: null pointer");
    return run_reduce(plan, plan->col, GNNMP_SUM, fs_j, nullptr, nullptr, nullptr, nullptr, nullptr, out, D, (hipStream_t)stream,
                      fs_e, nullptr, fs_i, 2, act);
}

int gnnmp_propagate_nn_f32(gnnmp_graph_t *p, int aggr, const float *xj, const float *we, float *out, int64_t Din, int64_t Dout,
                           gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "propagate_nn: null plan");
    if (int rc = check_aggr(aggr, "propagate_nn")) return rc;
    if (Din <= 0 || Dout <= 0 || Din > (1 << 16) || Dout > (1 << 16)) return fail(GNNMP_EINVAL, "propagate_nn: bad size");
    if (p->n_dst == 0) return GNNMP_OK;
    if (!out || ((!xj || !we) && p->n_total > 0)) return fail(GNNMP_EINVAL, "propagate_nn: null pointer");
    if (p->n_chunks > 0) {
        if (int rc = ensure_workspace(p, (size_t)p->n_chunks * (size_t)Dout)) return rc;
    }
    ReduceArgs a = {};
    a.rowptr = p->rowptr;
    a.idx = p->col;
    a.eid = p->eid;
    a.x = xj;
    a.out = out;
    a.partial = p->ws;
    a.chunk_row = p->chunk_row;
    a.chunk_beg = p->chunk_beg;
    a.chunk_end = p->chunk_end;
    a.long_rows = p->long_rows;
    a.long_cptr = p->long_cptr;
    a.n_chunks = p->n_chunks;
    a.n_long = p->n_long;
    a.D = (int)Dout;
    a.n_rows = (int)p->n_dst;
    a.n_src = (int)p->n_src;
    a.n_edges = (uint32_t)p->n_edges;
    a.mean = (aggr == GNNMP_MEAN);
    a.long_thresh = p->long_thresh;
    a.waves = 4;
    a.log2g = pick_log2g(Dout);
    const int G = 1 << a.log2g;
    const int tiles = (int)((Dout + G - 1) / G);
    const int64_t nvirt = (int64_t)a.n_rows + a.n_chunks;
    const int64_t rows_per_block = (int64_t)(64 / G) * a.waves;
    dim3 grid((unsigned)((nvirt + rows_per_block - 1) / rows_per_block), (unsigned)tiles);
    const int op = (aggr == GNNMP_MAX) ? OP_MAX : (aggr == GNNMP_MIN ? OP_MIN : OP_SUM);
    switch (op) {
        case OP_SUM: nn_rows_kernel<OP_SUM><<<grid, 64 * a.waves, 0, stream>>>(a, we, (int)Din); break;
        case OP_MAX: nn_rows_kernel<OP_MAX><<<grid, 64 * a.waves, 0, stream>>>(a, we, (int)Din); break;
        default: nn_rows_kernel<OP_MIN><<<grid, 64 * a.waves, 0, stream>>>(a, we, (int)Din); break;
    }
    GNNMP_LAUNCH_CHECK("nn_rows_kernel");
    if (a.n_long > 0) {
        dim3 cg((unsigned)a.n_long, (unsigned)tiles);
        switch (op) {
            case OP_SUM: csr_combine_kernel<1, OP_SUM><<<cg, 256, 0, stream>>>(a); break;
            case OP_MAX: csr_combine_kernel<1, OP_MAX><<<cg, 256, 0, stream>>>(a); break;
            default: csr_combine_kernel<1, OP_MIN><<<cg, 256, 0, stream>>>(a); break;
        }
        GNNMP_LAUNCH_CHECK("csr_combine_kernel");
    }
    return GNNMP_OK;
}

int gnnmp_propagate_slots_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *w_slot,
                              const float *ss_slot, const float *scale_dst, float *out, int64_t D,
                              gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "propagate_slots: null plan");
    if (int rc = check_aggr(aggr, "propagate_slots")) return rc;
    if (D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "propagate_slots: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || (!xj && plan->n_total > 0)))
        return fail(GNNMP_EINVAL, "propagate_slots: null xj/out");
    return run_reduce(plan, plan->col, aggr, xj, nullptr, nullptr, w_slot, ss_slot, scale_dst, out, D,
                      (hipStream_t)stream);
}

int gnnmp_propagate_slots_act_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *w_slot, const float *ss_slot,
                                  const float *scale_dst, const float *bias, int act, float *out, int64_t D,
                                  gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "propagate_slots_act: null plan");
    if (int rc = check_aggr(aggr, "propagate_slots_act")) return rc;
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "propagate_slots_act: bad act %d", act);
    if (D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "propagate_slots_act: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || (!xj && plan->n_total > 0)))
        return fail(GNNMP_EINVAL, "propagate_slots_act: null xj/out");
    return run_reduce(plan, plan->col, aggr, xj, nullptr, nullptr, w_slot, ss_slot, scale_dst, out, D, (hipStream_t)stream, nullptr,
                      nullptr, nullptr, 1, 0, 0, bias, act == GNNMP_ACT_RELU ? 1 : 0);
}

/* out = act'( addend + aggregate ): the tail of graph_conv's / sage_conv's pullback w.r.t. x in the row kernel's epilogue (gnnmp.h) */
int gnnmp_propagate_add_mask_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *scale_dst, const float *addend,
                                 const float *mask_y, float *out, int64_t D, gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "propagate_add_mask: null plan");
    if (aggr != GNNMP_SUM && aggr != GNNMP_MEAN) return fail(GNNMP_EINVAL, "propagate_add_mask: aggr must be + or mean (got %d)", aggr);
    if (D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "propagate_add_mask: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || (!xj && plan->n_total > 0)))
        return fail(GNNMP_EINVAL, "propagate_add_mask: null xj/out");
    return run_reduce(plan, plan->col, aggr, xj, nullptr, nullptr, nullptr, nullptr, scale_dst, out, D, (hipStream_t)stream, nullptr,
                      nullptr, nullptr, 1, 0, 0, nullptr, 0, addend, mask_y);
}

int gnnmp_scatter_f32(gnnmp_graph_t *plan, int aggr, const float *m, float *out, int64_t D,
                      gnnmp_stream_t stream) {
    if (!plan) return fail(GNNMP_EINVAL, "scatter: null plan");
    if (int rc = check_aggr(aggr, "scatter")) return rc;
    if (D < 0 || D > (1 << 20)) return fail(GNNMP_EINVAL, "scatter: bad D %lld", (long long)D);
    if (plan->n_dst > 0 && D > 0 && (!out || (!m && plan->n_total > 0)))
        return fail(GNNMP_EINVAL, "scatter: null m/out");
    return run_reduce(plan, plan->eid, aggr, m, nullptr, nullptr, nullptr, nullptr, nullptr, out, D,
                      (hipStream_t)stream);
}

int gnnmp_degree_f32(gnnmp_graph_t *plan, const float *w, float *deg, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!plan) return fail(GNNMP_EINVAL, "degree: null plan");
    if (plan->n_dst == 0) return GNNMP_OK;
    if (!deg) return fail(GNNMP_EINVAL, "degree: null output");
    const unsigned nb = (unsigned)((plan->n_dst + 255) / 256);
    if (w) {
        degree_weighted_kernel<<<nb, 256, 0, stream>>>(plan->rowptr, plan->eid, w, plan->n_dst,
                                                        (uint32_t)plan->n_edges, deg, plan->long_thresh);
        if (plan->n_long > 0)
            degree_long_kernel<<<(unsigned)plan->n_long, 256, 0, stream>>>(plan->rowptr, plan->eid, w, plan->long_rows,
                                                                            (uint32_t)plan->n_edges, deg);
    } else
        degree_count_kernel<<<nb, 256, 0, stream>>>(plan->rowptr, plan->n_dst, deg);
    GNNMP_LAUNCH_CHECK("degree kernel");
    return GNNMP_OK;
}

int gnnmp_inv_sqrt_f32(const float *deg, float *out, int64_t n, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) return fail(GNNMP_EINVAL, "inv_sqrt: negative n");
    if (n == 0) return GNNMP_OK;
    if (!deg || !out) return fail(GNNMP_EINVAL, "inv_sqrt: null pointer");
    inv_sqrt_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(deg, out, n);
    GNNMP_LAUNCH_CHECK("inv_sqrt_kernel");
    return GNNMP_OK;
}

}  // extern "C"
