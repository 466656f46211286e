// fused_conv.hip — aggregate-then-transform in ONE kernel: the layer bodies whose dense product FOLLOWS the aggregation
//   gcn_conv   (Dout >= Din)  σ.(W * (cin .* Σ_j cout_j x_j) .+ b)        GNNlib/src/layers/conv.jl:59-71
//   graph_conv                σ.(W1 * x_i .+ W2 * aggr_j x_j .+ b)         conv.jl:102-108
//   sage_conv                 σ.(W * vcat(x_i, aggr_j x_j) .+ b)           conv.jl:277-283
// without the (N, D) aggregate ever going to HBM: round 1 wrote it (4 N D bytes) and the dense kernel read it back.
//
// One persistent block per CU; W^T (both segments) lives in LDS as the operand image of mfma16.h.  Each WAVE takes 16-row
// tiles — the first 7/8 dealt statically, the last 1/8 from a device-wide ticket (rows differ 100x in length: a purely
// static split leaves a tail; a purely ticketed one is bound by the atomic) — and for each tile
//   1. walks its 16 destination rows exactly like csr_rows_kernel (csr_reduce.h: one lane group per row, U = 8 sixteen-byte
//      row loads in flight, adds in original edge order => the aggregate is BIT-IDENTICAL to the unfused kernel's) and puts
//      the finished rows into a wave-private 16 x D LDS tile (padded stride: the later ds_read_b128 are conflict-free);
//      rows the plan splits (> long_thresh edges) are not walked: their aggregates were reduced beforehand by the chunked
//      row kernel + combine into a compact buffer and are copied in;
//   2. runs the tile through the 16x16x4 fp32 MFMA core: B operands by ds_read_b128 from the tile (and, for the root term
//      W1 * x_i, by 16-byte loads of x_i straight from HBM), A operands from the W image; bias + activation on the
//      accumulators; one 16-byte store per accumulator to the output row.
// No workgroup barrier after the image is built; the only cross-lane hand-off is inside a wave (LDS tile).  The kernel stays
// bound by the row gather (cache-line requests: DESIGN.md section 5): the MFMA phase of a tile (175 MFMAs = 5 600 cycles at
// 100 => 100) is ~5 % of the tile's gather time and overlaps with the other waves' gathers.  Used by default only where it
// removes an HBM round trip (aggregate >= 128 MiB) and there is no root term: see gnnmp_fused_conv_f32 below.
#include <algorithm>

#include "csr_reduce.h"
#include "mfma16.h"

namespace gnnmp {

struct FusedArgs {
    ReduceArgs r;             // the aggregation: rowptr / idx / x / scalings / D / n_rows / long_thresh / log2g (out unused)
    const float *agg_long;    // [n_long][D] finalised aggregates of the split rows (r.long_rows sorted ascending)
    float *agg_out;           // optional [n_rows][D]: also write the aggregate (parity tests of the pre-GEMM value)
    const float *xi;          // optional root features [n_rows][D1] (first segment of the contraction)
    int D1;
    const float *W[2];        // [0] root, [1] aggregate
    int64_t sj[2], sk[2];
    const float *bias;
    int act;
    float *out;               // [n_rows][Dout]
    int Dout;
    int stride;               // floats per LDS tile row (== 8 mod 16)
    int waves;
    uint32_t *ticket;         // next dynamically dealt tile (zeroed by the host before every launch)
};

// slot of `row` in the sorted list of split rows (it is there: the caller checked the row's length)
__device__ __forceinline__ int long_slot(const int32_t *__restrict__ long_rows, int n_long, int row) {
    int lo = 0, hi = n_long - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (long_rows[mid] < row) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int NCB, int KQA, int OP, bool SCALED>
__global__ void __launch_bounds__(1024) fused_conv_kernel(const FusedArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int DP = NCB * 16;
    constexpr int VEC = 4, U = 8;
    const ReduceArgs &r = a.r;
    f32x4 *img = reinterpret_cast<f32x4 *>(lds_raw);
    const int tid = threadIdx.x, nthreads = a.waves * 64;
    const int D = KQA >= 0 ? 4 * KQA : r.D;
    const int rows0 = a.xi ? t16_img_rows(a.D1) : 0;
    const int rows1 = t16_img_rows(D);
    f32x4 *bias4 = img + (rows0 + rows1) * DP;
    float *tiles = reinterpret_cast<float *>(bias4 + DP / 4);
    const int ncols = min(DP, a.Dout);
    if (a.xi) t16_fill_image(img, 0, DP, a.W[0], a.sj[0], a.sk[0], a.D1, 0, ncols, tid, nthreads);
    t16_fill_image(img, rows0, DP, a.W[1], a.sj[1], a.sk[1], D, 0, ncols, tid, nthreads);
    for (int i = tid; i < DP / 4; i += nthreads) {
        f32x4 b = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * i + q < ncols) b[q] = a.bias[4 * i + q];
        }
        bias4[i] = b;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    float *tile = tiles + (size_t)wave * 16 * a.stride;
    // gather-phase lane roles (csr_rows_kernel's): group of G lanes per row, lane lig owns features f0 .. f0 + 3
    const int G = 1 << r.log2g;
    const int lig = lane & (G - 1), grp = lane >> r.log2g, gbase = lane - lig, rpw = 64 >> r.log2g;
    const int f0 = lig * VEC;
    const bool active = f0 < D;
    // MFMA-phase lane roles: node n of the tile, k-slot q
    const int n = lane & 15, q = lane >> 4;
    const int ntiles = (r.n_rows + 15) >> 4;
    const bool has_bias = a.bias != nullptr;

    // Tile hand-out: the first 7/8 of the tiles are dealt statically (wave w takes w, w + W, w + 2 W, ...: no traffic), the last
    // 1/8 by a device-wide ticket so that the waves that drew short rows finish the job.  (All tiles by ticket: one word takes
    // ~88 atomics per microsecond, MI355X_MICROARCH.md "dequeue" — 10 600 tiles of the arxiv shape alone cost 120 us.)
    const int total_waves = (int)gridDim.x * a.waves;
    const int wave_global = (int)blockIdx.x * a.waves + wave;
    const int n_static = (int)(((int64_t)ntiles * 7 / 8) / total_waves) * total_waves;
    int next_static = wave_global;
    for (;;) {
        int t;
        if (next_static < n_static) {
            t = next_static;
            next_static += total_waves;
        } else {
            t = 0;
            if (lane == 0) t = n_static + (int)atomicAdd(a.ticket, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
        }
        if (t >= ntiles) break;
        const int row0 = t * 16;
        // the tile's 17 row pointers in one coalesced load (lane l holds rowptr[row0 + l]); each row then takes its bounds from
        // two lanes instead of starting with a dependent load of its own (not with a row order: those rows are not adjacent)
        const uint32_t rp = (!r.row_order && lane <= 16) ? r.rowptr[min(row0 + lane, r.n_rows)] : 0u;
        // ---- 1. the 16 rows of the tile, rpw at a time ----
        for (int rr0 = 0; rr0 < 16; rr0 += rpw) {
            const int rr = rr0 + grp;
            // every lane of the wave takes part in the two exchanges (a lane group past the tile's 16th row reads lane 16)
            const uint32_t rb = (uint32_t)__shfl((int)rp, min(rr, 16), 64), re = (uint32_t)__shfl((int)rp, min(rr + 1, 16), 64);
            if (rr >= 16) continue;
            int row = row0 + rr;
            if (r.row_order && row < r.n_rows) row = r.row_order[row];
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = op_identity<OP>();
            if (row < r.n_rows) {
                const uint32_t beg = r.row_order ? r.rowptr[row] : rb, end = r.row_order ? r.rowptr[row + 1] : re;
                if (end - beg > r.long_thresh) {
                    if (active) Vec<VEC>::load(a.agg_long + (int64_t)long_slot(r.long_rows, r.n_long, row) * D + f0, acc);
                } else {
                    reduce_range<VEC, OP, SCALED, U>(r, beg, end, lig, gbase, G, f0, active, acc, row);
                    finalize_row<VEC, OP>(r, row, end - beg, acc);
                }
                if (a.agg_out && active) Vec<VEC>::store(a.agg_out + (int64_t)row * D + f0, acc);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
            }
            if (active) *reinterpret_cast<float4 *>(tile + rr * a.stride + f0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- 2. contraction + epilogue ----
        f32x4 acc2[NCB];
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc2[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const bool row_ok = row0 + n < r.n_rows;
        int row = min(row0 + n, r.n_rows - 1);
        if (r.row_order) row = r.row_order[row];
        if (a.xi) {
            const float *xr = a.xi + (int64_t)row * a.D1;
            t16_segment_rt<NCB>(acc2, img, 0, a.D1 >> 2, n, q, [&](int kcol) { return *reinterpret_cast<const float4 *>(xr + kcol); });
        }
        {
            const float *tr = tile + n * a.stride;
            auto ld = [&](int kcol) { return *reinterpret_cast<const float4 *>(tr + kcol); };
            if constexpr (KQA >= 0) {
                constexpr int MAXB = (KQA + 3) / 4;
                float4 xv[MAXB];
                t16_load<MAXB, KQA>(xv, q, ld);
                t16_compute<NCB, MAXB, KQA>(acc2, img, rows0, n, q, xv);
            } else {
                t16_segment_rt<NCB>(acc2, img, rows0, D >> 2, n, q, ld);
            }
        }
        t16_store<NCB>(acc2, bias4, has_bias, a.act, a.out + (int64_t)row * a.Dout, row_ok, ncols, q);
        // the tile is rewritten by the next gather: program order within the wave keeps the reads above before those writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int NCB, int KQA, int OP, bool SCALED>
static int launch_fused(FusedArgs &a, size_t img_bytes, hipStream_t stream) {
    const size_t tile_bytes = (size_t)16 * a.stride * sizeof(float);
    const size_t budget = 160 * 1024;
    int waves = (int)std::min<size_t>(16, (budget - img_bytes) / tile_bytes);
    if (waves < 4) return 1;                        // the image leaves no room for enough gathering waves: not fused
    const int kw = knob(KNOB_FUSED_WAVES);
    if (kw >= 1 && kw <= waves) waves = kw;
    a.waves = waves;
    GNNMP_LDS_OPTIN("fused_conv_kernel", &fused_conv_kernel<NCB, KQA, OP, SCALED>);
    const int ntiles = (a.r.n_rows + 15) / 16;
    const int gx = std::min(device_cus(), (ntiles + waves - 1) / waves);
    // the ticket starts every launch at zero: an 8-byte memset node on the same stream (a kernel that re-armed it itself would
    // leave it dirty after a failed launch and the next one would silently skip tiles)
    GNNMP_HIP(hipMemsetAsync(a.ticket, 0, 2 * sizeof(uint32_t), stream));
    fused_conv_kernel<NCB, KQA, OP, SCALED><<<gx, 64 * waves, img_bytes + (size_t)waves * tile_bytes, stream>>>(a);
    GNNMP_LAUNCH_CHECK("fused_conv_kernel");
    return GNNMP_OK;
}

template <int NCB, int KQA>
static int dispatch_fused(FusedArgs &a, int op, bool scaled, size_t img_bytes, hipStream_t stream) {
    switch (op) {
        case OP_SUM:
            return scaled ? launch_fused<NCB, KQA, OP_SUM, true>(a, img_bytes, stream)
                          : launch_fused<NCB, KQA, OP_SUM, false>(a, img_bytes, stream);
        case OP_MAX: return launch_fused<NCB, KQA, OP_MAX, false>(a, img_bytes, stream);
        default: return launch_fused<NCB, KQA, OP_MIN, false>(a, img_bytes, stream);
    }
}

int run_reduce(gnnmp_graph_t *p, const int32_t *idx, int aggr, const float *x, const float *w, const float *ss,
               const float *w_slot, const float *ss_slot, const float *sd, float *out, int64_t D, hipStream_t stream,
               const float *emat, const float *rowsub, const float *gate_i, int gated, int act, int long_only, const float *bias,
               int bias_relu, const float *addend, const float *mask_y);   // propagate.hip

}  // namespace gnnmp

using namespace gnnmp;

extern "C" int gnnmp_fused_conv_f32(gnnmp_graph_t *p, int aggr, const float *xj, const float *w, const float *scale_src,
                                    const float *w_slot, const float *ss_slot, const float *scale_dst, int64_t D,
                                    const float *xi, int64_t D1, const float *W_root, int64_t ldw_root, const float *W_agg,
                                    int64_t ldw_agg, int w_layout, const float *bias, int act, float *out, int64_t Dout,
                                    float *agg_out, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "fused_conv: null plan");
    if (aggr < GNNMP_SUM || aggr > GNNMP_MIN) return fail(GNNMP_EINVAL, "fused_conv: bad aggr %d", aggr);
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "fused_conv: bad act %d", act);
    if (w_layout != 0 && w_layout != 1) return fail(GNNMP_EINVAL, "fused_conv: bad w_layout %d", w_layout);
    if (D <= 0 || Dout <= 0 || D1 < 0) return fail(GNNMP_EINVAL, "fused_conv: bad size");
    if (p->n_dst == 0) return GNNMP_OK;
    if (!out || !W_agg || (!xj && p->n_total > 0) || (D1 > 0 && (!xi || !W_root)))
        return fail(GNNMP_EINVAL, "fused_conv: null pointer");
    // shapes this kernel takes; anything else: GNNMP_EUNSUPPORTED, the caller runs propagate + dense
    const bool scaled = w || scale_src || w_slot || ss_slot;
    const int op = (aggr == GNNMP_MAX) ? OP_MAX : (aggr == GNNMP_MIN ? OP_MIN : OP_SUM);
    if ((D & 3) || D > 128 || (D1 & 3) || D1 > 128 || (Dout & 3) || Dout > 128 || p->n_dst < 16 || (scaled && op != OP_SUM) ||
        (reinterpret_cast<uintptr_t>(xj) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
        (xi && (reinterpret_cast<uintptr_t>(xi) & 15)) || (agg_out && (reinterpret_cast<uintptr_t>(agg_out) & 15)) ||
        knob(KNOB_FUSED_WAVES) < 0)
        return fail(GNNMP_EUNSUPPORTED, "fused_conv: shape not fused (D=%lld D1=%lld Dout=%lld)", (long long)D, (long long)D1,
                    (long long)Dout);
    // Fusing removes the aggregate's HBM round trip (4 N D bytes written, then read by the dense kernel).  When the aggregate
    // fits the 256 MiB Infinity Cache beside the features there is no HBM round trip to remove, and the unfused pair wins on
    // occupancy (no LDS-resident W image => three times the gathering waves per CU): measured arxiv shape 178 vs 187 us,
    // config 5 316 vs 369 us; products shape 5.32 -> 4.67 ms the other way.  knob 14 > 0 forces the fused kernel.
    if (knob(KNOB_FUSED_WAVES) == 0 && (int64_t)p->n_dst * D * (int64_t)sizeof(float) < ((int64_t)128 << 20))
        return fail(GNNMP_EUNSUPPORTED, "fused_conv: aggregate fits the Infinity Cache, unfused path is faster");
    // With a root term (graph_conv / sage_conv) a tile's MFMA phase doubles while the gathering waves stay capped at 16 per CU
    // by the LDS image: measured SAGEConv(100 => 128) on the products shape 5.95 ms fused vs 5.79 ms unfused.  Not by default.
    if (knob(KNOB_FUSED_WAVES) == 0 && D1 > 0)
        return fail(GNNMP_EUNSUPPORTED, "fused_conv: root term, unfused path is faster");
    // split rows first: chunk partials + combine into the compact buffer behind the partials
    const size_t pc = (size_t)p->n_chunks * (size_t)D, pl = (size_t)p->n_long * (size_t)D;
    if (int rc = ensure_workspace(p, pc + pl + 4)) return rc;
    if (int rc = ensure_ticket(p, stream)) return rc;
    float *agg_long = p->ws + ((pc + 3) & ~(size_t)3);
    if (p->n_long > 0) {
        if (int rc = run_reduce(p, p->col, aggr, xj, w, scale_src, w_slot, ss_slot, scale_dst, agg_long, D, stream, nullptr,
                                nullptr, nullptr, 1, 0, 1, nullptr, 0, nullptr, nullptr))
            return rc;
    }
    FusedArgs a = {};
    ReduceArgs &r = a.r;
    r.rowptr = p->rowptr;
    r.idx = p->col;
    r.eid = p->eid;
    r.x = xj;
    r.w = w;
    r.ss = scale_src;
    r.w_slot = w_slot;
    r.ss_slot = ss_slot;
    r.sd = scale_dst;
    r.long_rows = p->long_rows;
    r.n_long = p->n_long;
    r.D = (int)D;
    r.n_rows = (int)p->n_dst;
    r.n_src = (int)p->n_src;
    r.n_edges = (uint32_t)p->n_edges;
    r.mean = (aggr == GNNMP_MEAN);
    r.long_thresh = p->long_thresh;
    r.log2g = pick_log2g((D + 3) / 4);
    // >= 2 rows per wave: the rows of a wave should be equally long (only when output rows are whole 128-byte lines, see
    // run_reduce in propagate.hip: measured 4.98 -> 6.02 ms with 400-byte rows)
    if (use_row_order(p->n_src, D) && r.log2g <= 5 && (Dout & 31) == 0 && (reinterpret_cast<uintptr_t>(out) & 127) == 0) {
        if (int rc = ensure_row_order(p, stream)) return rc;
        r.row_order = p->row_order;
    }
    a.agg_long = agg_long;
    a.agg_out = agg_out;
    a.xi = D1 > 0 ? xi : nullptr;
    a.D1 = (int)D1;
    a.W[0] = W_root; a.W[1] = W_agg;
    a.sj[0] = w_layout == 0 ? ldw_root : 1; a.sk[0] = w_layout == 0 ? 1 : ldw_root;
    a.sj[1] = w_layout == 0 ? ldw_agg : 1; a.sk[1] = w_layout == 0 ? 1 : ldw_agg;
    a.bias = bias;
    a.act = act;
    a.out = out;
    a.Dout = (int)Dout;
    a.stride = (int)(((D + 15) & ~15) + 8);         // == 8 mod 16: the tile's ds_read_b128 are bank-conflict free
    if (a.stride - 16 >= D) a.stride -= 16;
    a.ticket = p->ticket;
    const int cb = (int)((Dout + 15) / 16);
    const int ncb = cb <= 2 ? 2 : (cb <= 4 ? 4 : (cb <= 6 ? 6 : (cb == 7 ? 7 : 8)));
    const size_t img_rows = (size_t)(a.xi ? t16_img_rows((int)D1) : 0) + (size_t)t16_img_rows((int)D);
    const size_t img_bytes = img_rows * (size_t)ncb * 16 * 16 + (size_t)ncb * 16 * 4;
    int rc;
    const int kqa = (int)D / 4;
    switch (ncb) {
        case 2: rc = dispatch_fused<2, -1>(a, op, scaled, img_bytes, stream); break;
        case 4: rc = dispatch_fused<4, -1>(a, op, scaled, img_bytes, stream); break;
        case 6: rc = dispatch_fused<6, -1>(a, op, scaled, img_bytes, stream); break;
        case 7: rc = kqa == 25 ? dispatch_fused<7, 25>(a, op, scaled, img_bytes, stream)
                               : dispatch_fused<7, -1>(a, op, scaled, img_bytes, stream); break;
        default: rc = kqa == 32 ? dispatch_fused<8, 32>(a, op, scaled, img_bytes, stream)
                                : (kqa == 25 ? dispatch_fused<8, 25>(a, op, scaled, img_bytes, stream)
                                             : dispatch_fused<8, -1>(a, op, scaled, img_bytes, stream)); break;
    }
    if (rc == 1) return fail(GNNMP_EUNSUPPORTED, "fused_conv: W image leaves no LDS for the row tiles");
    return rc;
}
