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
// bound by the row gather (cache-line requests: DESIGN.md §5, LABNOTES.md §5): the MFMA phase of a tile (175 MFMAs = 5 600 cycles at
// 100 => 100) is ~5 % of the tile's gather time and overlaps with the other waves' gathers.  Used by default only where it
// removes an HBM round trip (aggregate >= 128 MiB) and there is no root term: see gnnmp_fused_conv_f32 below.
#include <algorithm>

#include "csr_reduce.h"
#include "mfma16.h"
#include "msplit.h"

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


// The gather phase of a persistent wave (rounds 6): ROWS consecutive destination rows starting at row0 into an LDS tile, by EIGHT lane
// groups of 8 lanes, each on its own row and pulling the tile's next row when its own is done.  A lane owns the 16-byte pieces at columns
// 4 gl + 32 k, k = 0 .. ceil(D / 32) - 1, of its group's row; up to four edges of each of the 8 rows are in flight, and no group waits for
// a longer neighbour.  (The walk the row kernels use — a group of 32 lanes per row, two rows a wave — leaves a persistent wave with only
// two rows' loads in flight, and most rows are short: fused_cat_kernel's first version ran 6.8 ms gather-bound with it; with this one its
// gather phase alone takes 4.4-4.5 ms on the products shape at 8-12 waves a CU — the fetch ceiling csr_rows_kernel reaches with 24.)
// Adds are in ORIGINAL edge order per row from the identity, products rounded separately: the same bits as reduce_range / csr_rows_kernel.
// rp: lane l <= ROWS holds rowptr[row0 + l].  Rows the plan splits are copied from agg_long; rows past n_rows become zero rows.
template <int D, int ROWS, int OP, bool SCALED>
__device__ __forceinline__ void gather_rows8(const ReduceArgs &r, const float *__restrict__ agg_long, float *__restrict__ agg_out, int row0,
                                             uint32_t rp, float *tile, int stride, int lane) {
    constexpr int GL = 8, NT = (D + 31) / 32;           // lanes per group, 16-byte pieces per lane
    const int gl = lane & (GL - 1), gb8 = lane - gl, g8 = lane >> 3;
    bool pact[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) pact[k] = 4 * gl + 32 * k < D;
    float acc[NT][4];
    int cur = g8;                                       // tile-relative row of this group (>= ROWS = none left)
    int next_rr = 8;                                    // wave-uniform: next row of the tile nobody has taken
    uint32_t p = 0, pend = 0, pbeg = 0;
    bool is_long = false;
    // start row rr in the lanes with `mine` set (a whole group at a time); EVERY lane of the wave executes the two exchanges — a
    // ds_bpermute under a divergent branch would read the row pointers from lanes that are switched off
    auto take = [&](int rr, bool mine) {
        const uint32_t rb = (uint32_t)__shfl((int)rp, min(rr, ROWS), 64), re = (uint32_t)__shfl((int)rp, min(rr + 1, ROWS), 64);
        if (!mine) return;
        cur = rr;
        is_long = false;
        p = pend = pbeg = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[k][q] = op_identity<OP>();
        if (rr >= ROWS) return;
        const int row = row0 + rr;
        if (row >= r.n_rows) {                           // past the last row: a zero row (never stored)
#pragma unroll
            for (int k = 0; k < NT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[k][q] = 0.0f;
            is_long = true;                              // (= "finished as it stands")
        } else if (re - rb > (uint32_t)r.long_thresh) {  // split row: reduced beforehand by the chunk pass, copied in
            const float *src = agg_long + (int64_t)long_slot(r.long_rows, r.n_long, row) * D;
#pragma unroll
            for (int k = 0; k < NT; ++k)
                if (pact[k]) Vec<4>::load(src + 4 * gl + 32 * k, acc[k]);
            is_long = true;
        } else {
            p = pbeg = rb; pend = re;
        }
    };
    take(cur, cur < ROWS);
    for (;;) {
        if (__builtin_amdgcn_ballot_w64(cur < ROWS) == 0) break;
        // up to 8 slots of this group's row: their source ids (and factors) in one coalesced load, then two batches of four row fetches
        const uint32_t left = pend - p;
        const int nb = (int)min(left, (uint32_t)GL);
        const bool have = (uint32_t)gl < left;
        const uint32_t cidx = have ? (uint32_t)r.idx[p + gl] : 0u;
        float wv = 1.0f, sv = 1.0f;
        if (SCALED && have) {
            if (r.w_slot) {
                wv = r.w_slot[p + gl];
            } else if (r.w) {
                const uint32_t e = (uint32_t)r.eid[p + gl];
                if (e < r.n_edges) wv = r.w[e];
            }
            if (r.ss_slot) sv = r.ss_slot[p + gl];
            else if (r.ss) sv = r.ss[cidx];
        }
#pragma unroll
        for (int j0 = 0; j0 < GL; j0 += 4) {
            if (__builtin_amdgcn_ballot_w64(j0 < nb) == 0) break;
            uint32_t cj[4];
            float wj[4], sj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cj[u] = (uint32_t)__shfl((int)cidx, gb8 + j0 + u, 64);
                if (SCALED) {
                    wj[u] = __shfl(wv, gb8 + j0 + u, 64);
                    sj[u] = __shfl(sv, gb8 + j0 + u, 64);
                }
            }
            float v[4][NT][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    if (pact[k] && j0 + u < nb) Vec<4>::load(r.x + (int64_t)cj[u] * D + 4 * gl + 32 * k, v[u][k]);
                    else v[u][k][0] = v[u][k][1] = v[u][k][2] = v[u][k][3] = 0.0f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < nb) {
#pragma unroll
                    for (int k = 0; k < NT; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float t = v[u][k][q];
                            if (SCALED) {
                                t = t * sj[u];          // xj .* cout'   (conv.jl:59), rounded
                                t = wj[u] * t;          // w .* xj       (msgpass.jl:203-208), rounded
                            }
                            acc[k][q] = op_apply<OP>(acc[k][q], t);
                        }
                }
        }
        p += (uint32_t)nb;
        const bool fin = cur < ROWS && p >= pend;
        if (fin) {
            const int row = row0 + cur;
            if (!is_long) {
#pragma unroll
                for (int k = 0; k < NT; ++k) finalize_row<4, OP>(r, row, pend - pbeg, acc[k]);
            }
#pragma unroll
            for (int k = 0; k < NT; ++k)
                if (pact[k]) {
                    if (agg_out && row < r.n_rows) Vec<4>::store(agg_out + (int64_t)row * D + 4 * gl + 32 * k, acc[k]);
                    *reinterpret_cast<float4 *>(tile + cur * stride + 4 * gl + 32 * k) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
                }
        }
        // the groups that finished take the next rows of the tile, in group order
        const unsigned long long done = __builtin_amdgcn_ballot_w64(fin && gl == 0);
        if (done) {
            const int rank = __builtin_popcountll(done & ((1ull << gb8) - 1ull));
            const int nr = min(next_rr + rank, ROWS);
            next_rr += __builtin_popcountll(done);
            take(fin ? nr : cur, fin);
        }
    }
}

// (Round 6 also tried gather_rows8 as THIS kernel's gather phase — 16-row tiles, 12 waves: products GCN layer 5.42 ms against 4.96 with the
// walk below on the same box, arxiv 179 against 186 us and 150 us unfused (profiles/r06_fused_gather_ab.txt).  With only 16 rows a tile
// eight groups get two rows each and idle at every tile's end; 32-row tiles do not fit beside the W image at more than 8 waves.  Removed.)
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


// =====================================================================================================================================
// fused_cat_kernel (round 6) — sage_conv / graph_conv with a ROOT term and 256 outputs in one kernel (BASELINE.json config 4:
// SAGEConv(100 => 256) on the products shape):   σ.(W * vcat(x_i, aggr_j x_j) .+ b)   GNNlib/src/layers/conv.jl:277-283,102-108
// The two-kernel path writes the (N, 100) aggregate (0.98 GB) and reads it back in the contraction; here it never leaves the CU.
// fused_conv_kernel above cannot take this layer: 200 x 256 weights are 205 KB in fp32 and 307 KB as three bf16 planes — more than LDS.
//   * persistent block per CU, 12 waves; each WAVE takes 32-row tiles (7/8 static, 1/8 by ticket, as above) and
//     1. walks the 32 destination rows exactly like csr_rows_kernel (reduce_range: the aggregate is BIT-IDENTICAL to gnnmp_propagate_f32)
//        into a wave-private 32 x K1 fp32 LDS tile (stride K1 = 100 floats: 100 = 36 mod 64 banks, the later ds_read_b128 of 16 rows are
//        conflict-free without padding — 12.8 KB a wave, 12 waves = 154 KB);
//     2. contracts [x_i | m_i] (x_i straight from HBM, 16 bytes per lane and k-piece; m_i from the tile) with W on the bf16 matrix core in
//        the exact three-plane split of msplit.h — six v_mfma_f32_32x32x16_bf16 per product — in two passes of 128 output columns
//        (64 accumulator registers).  W is NOT resident anywhere near the wave: its three planes, pre-split once per call into MFMA operand
//        order (fused_cat_wimg_kernel: 312 KB), are streamed from L2 — one coalesced 1 KB load per (k-block, 32 columns, plane), a ring
//        of three units in flight ahead of the MFMAs.  Every wave of the chip re-reads the same 312 KB per tile: L2 hits (the image is
//        touched every few hundred ns, the gather stream turns an L2 over every ~4 us), ~5 TB/s of L2 -> L1 traffic beside the gather.
//        The node rows are the MFMA's A operand and W its B operand, so a lane ends up with ONE output column of 16 nodes: a store
//        instruction writes two whole 128-byte lines of the output, no LDS staging.
//   * non-finite operands (the -Inf of an empty max aggregation, Inf / NaN features) turn a tile's accumulators NaN (msplit.h): that
//     half-tile is recomputed by plain fp32 fma loops (cold, out of line).
// The m-part of the contraction (k-blocks from the LDS tile) runs first: the x_i pieces of the first three k-blocks are requested at the
// start of the phase and arrive behind it.
//
// MEASURED (MI355X, products shape, SAGEConv(100 => 256, relu; mean), one box, tools/experiments/sage_fused_ab.py): the kernel is CORRECT
// (tests/test_fused_conv.py: aggregate bit-identical, output 8e-7 of the two-kernel path) and LOSES — 6.85 ms (12 waves) / 7.42 ms (8 waves)
// against 5.77-5.82 ms for csr_rows_kernel + dense_wreg_kernel — so it is off by default (knob 14 > 0 runs it).  Ablations (knob 13), ms:
//                                    12 waves    8 waves
//     whole kernel                     6.85        7.42
//     no contraction (gather + tile)   4.55        4.42      <- the gather phase alone is AT the fetch ceiling (csr_rows_kernel: 4.50),
//     no gather (contraction + store)  2.69        3.08         with a third of csr_rows_kernel's waves: eight 8-lane groups per wave, each
//     neither                          0.30        0.26         pulling the tile's next row when done (the first version walked two rows at a
//     no stores                        6.43        6.98         time like csr_rows_kernel: 6.8 / 7.9 ms, gather-bound)
// The phases ADD UP.  A wave-tile streams the whole 312 KB image through its CU's vector-memory path: 2 496 line requests per 32 rows next
// to the gather's ~3 400 (840 edges x 4 lines) — L2 hits, but they queue where the gather's misses queue, and the gather is bound by exactly
// that per-CU request capacity (LABNOTES.md: the line-request wall; profiles/r06_mall_sweep.txt: the rate follows the latency, 58 G
// lines/s from the Infinity Cache, 50 from HBM).  A fused SAGE layer therefore needs W resident on the CU — 307 KB of planes do not fit
// 160 KB of LDS, and in registers (dense_wreg's 156 per wave, eight waves) they leave the same eight waves no room to gather.  What fusing
// could save at best is the aggregate's round trip: ~200 of ~4 000 line requests per 32 rows (5 %), 0.25 ms.
constexpr int FC2_ROWS = 32;

struct FusedCatArgs {
    ReduceArgs r;
    const float *agg_long;
    float *agg_out;
    const float *xi;          // [n_rows][K0]
    const u32x4 *wimg;        // [NH][NKB][4][3][64] operand units (fused_cat_wimg_kernel)
    WCat w;                   // the raw weights (exact path only)
    const float *bias;
    int act;
    float *out;               // [n_rows][Dout]
    int Dout, NH;
    uint32_t *ticket;
    int dbg;                  // knob 13 (experiments only): 1 = skip the contraction, 2 = skip the gather, 4 = skip the stores
};

// W (Dout x (K0 + K1), two blocks) -> the three bf16 planes of every (128-column half, k-block, 32-column block) in MFMA operand order:
// lane (f, h) of a unit holds W(128 half + 32 cb + f, 16 kb + 8 (e >> 2) + 4 h + (e & 3)), e = 0..7 (msplit.h's element order)
template <int K0, int K1>
__global__ void __launch_bounds__(64) fused_cat_wimg_kernel(const WCat w, int Dout, u32x4 *img) {
    constexpr int NKB = (K0 + K1 + 15) / 16, KCAT = K0 + K1;
    const int t = blockIdx.x, lane = threadIdx.x;
    const int cb = t & 3, kb = (t >> 2) % NKB, half = (t >> 2) / NKB;
    const int f = lane & 31, h = lane >> 5;
    const int j = 128 * half + 32 * cb + f;
    float4 q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = 16 * kb + 4 * h + 8 * u;
        q[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (c < KCAT && j < Dout) q[u] = make_float4(wcat_at(w, j, c), wcat_at(w, j, c + 1), wcat_at(w, j, c + 2), wcat_at(w, j, c + 3));
    }
    const Split8 s8 = split8(q[0], q[1]);
    img[((size_t)t * 3 + 0) * 64 + lane] = s8.p0;
    img[((size_t)t * 3 + 1) * 64 + lane] = s8.p1;
    img[((size_t)t * 3 + 2) * 64 + lane] = s8.p2;
}

// the exact path of one half-tile (cold): lane (j, h) recomputes its 4 x 16 outputs with fp32 fma chains in c order and stores them
__device__ __forceinline__ void fused_cat_exact(const WCat w, const float *xi, const float *tile, int K0, int K1, int col0, int row0,
                                             int n_rows, const float *bias, int act, float *out, int Dout, int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
        const int cb = i >> 4, r = i & 15;
        const int col = col0 + 32 * cb + j;
        const int node = 8 * (r >> 2) + 4 * h + (r & 3);
        const int row = row0 + node;
        if (row >= n_rows || col >= Dout) continue;
        float s = 0.0f;
#pragma unroll 1
        for (int c = 0; c < K0; ++c) s = fmaf(wcat_at(w, col, c), xi[(int64_t)row * K0 + c], s);
#pragma unroll 1
        for (int c = 0; c < K1; ++c) s = fmaf(wcat_at(w, col, K0 + c), tile[node * K1 + c], s);
        float v = s + (bias ? bias[col] : 0.0f);
        if (act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
        out[(int64_t)row * Dout + col] = v;
    }
}

// NW waves a block (12 = what LDS holds tiles for: 3 a SIMD, 168 registers; 8 = 2 a SIMD, 256 registers), RD = W units / x_i k-blocks in
// flight, U = row loads in flight per lane group in the gather phase
template <int K0, int K1, int OP, bool SCALED, int NW, int RD, int U>
__global__ void __launch_bounds__(64 * NW) fused_cat_kernel(const FusedCatArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int FC2_WAVES = NW;
    constexpr int NKB = (K0 + K1 + 15) / 16;
    constexpr int KBX = K0 / 16;                      // k-blocks that lie entirely in x_i
    static_assert((K0 & 3) == 0 && (K1 & 3) == 0 && K1 <= 128, "segments are multiples of 4; the aggregate row fits one lane group");
    const ReduceArgs &r = a.r;
    float *tiles = reinterpret_cast<float *>(lds_raw);                       // [FC2_WAVES][32][K1]
    float *biasl = tiles + FC2_WAVES * FC2_ROWS * K1;                        // [Dout]
    const int tid = threadIdx.x;
    for (int i = tid; i < a.Dout; i += 64 * FC2_WAVES) biasl[i] = a.bias ? a.bias[i] : 0.0f;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    float *tile = tiles + wave * FC2_ROWS * K1;
    const int n = lane & 31, h = lane >> 5;
    const int ntiles = (r.n_rows + FC2_ROWS - 1) / FC2_ROWS;

    const int total_waves = (int)gridDim.x * FC2_WAVES;
    const int wave_global = (int)blockIdx.x * FC2_WAVES + wave;
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
        const int row0 = t * FC2_ROWS;
        const uint32_t rp = lane <= FC2_ROWS ? r.rowptr[min(row0 + lane, r.n_rows)] : 0u;
        // ---- 1. the 32 rows of the tile: eight lane groups of 8 lanes, each pulling the tile's next row when done (gather_rows8) ----
        if (!(a.dbg & 2)) gather_rows8<K1, FC2_ROWS, OP, SCALED>(r, a.agg_long, a.agg_out, row0, rp, tile, K1, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- 2. [x_i | m_i] * W^T, 128 columns at a time ----
        const int rowc = min(row0 + n, r.n_rows - 1);
        const float *xr = a.xi + (int64_t)rowc * K0;
        const float *tr = tile + n * K1;
        // piece p of k-block kb (p = 0, 1: positions 16 kb + 4 h + 8 p .. + 3 of the concatenated row): from x_i or from the tile
        auto piece_x = [&](int kb, int p) -> float4 {          // (only called where the piece can lie in x_i; clamped: W holds zeros past KCAT)
            const int c = min(16 * kb + 4 * h + 8 * p, K0 - 4);
            return *reinterpret_cast<const float4 *>(xr + c);
        };
        auto piece_m = [&](int kb, int p) -> float4 {
            const int c = min(max(16 * kb + 4 * h + 8 * p - K0, 0), K1 - 4);
            return *reinterpret_cast<const float4 *>(tr + c);
        };
        const bool full_tile = row0 + FC2_ROWS <= r.n_rows;
#pragma unroll 1
        for (int half = 0; half < ((a.dbg & 1) ? 0 : a.NH); ++half) {
            f32x16 acc2[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc2[cb][q] = 0.0f;
            const u32x4 *wp = a.wimg + (size_t)half * NKB * 4 * 3 * 64 + lane;
            // k-blocks in the order: those that touch the tile first (kb = NKB - 1 .. KBX), then the pure x_i blocks (KBX - 1 .. 0)
            float4 xq[RD][2];                                  // x_i pieces of up to RD k-blocks in flight
#pragma unroll
            for (int i = 0; i < RD; ++i)
                if (KBX - 1 - i >= 0) { xq[i][0] = piece_x(KBX - 1 - i, 0); xq[i][1] = piece_x(KBX - 1 - i, 1); }
            SplitA ring[RD];
            auto load_unit = [&](int step) -> SplitA {          // step = position in the visiting order * 4 + cb
                const int ord = step >> 2, cb = step & 3;
                const int kb = NKB - 1 - ord;
                const u32x4 *u = wp + (size_t)((kb * 4 + cb) * 3) * 64;
                SplitA s;
                s.w0 = u[0]; s.w1 = u[64]; s.w2 = u[128];
                return s;
            };
#pragma unroll
            for (int i = 0; i + 1 < RD; ++i) ring[i] = load_unit(i);
#pragma unroll
            for (int ord = 0; ord < NKB; ++ord) {
                const int kb = NKB - 1 - ord;
                float4 q0, q1;
                if (kb < KBX) {                                 // pure x_i block: from the in-flight ring, then refill the slot three blocks on
                    const int slot = (KBX - 1 - kb) % RD;
                    q0 = xq[slot][0]; q1 = xq[slot][1];
                    if (kb - RD >= 0) { xq[slot][0] = piece_x(kb - RD, 0); xq[slot][1] = piece_x(kb - RD, 1); }
                } else if (16 * kb >= K0) {                     // pure tile block
                    q0 = piece_m(kb, 0); q1 = piece_m(kb, 1);
                } else {                                        // the block that straddles K0: each 4-piece lies wholly on one side
                    const float4 x0 = piece_x(kb, 0), x1 = piece_x(kb, 1), m0 = piece_m(kb, 0), m1 = piece_m(kb, 1);
                    const bool in0 = 16 * kb + 4 * h < K0, in1 = 16 * kb + 4 * h + 8 < K0;
                    q0 = in0 ? x0 : m0; q1 = in1 ? x1 : m1;
                }
                const Split8 b = split8(q0, q1);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const int step = ord * 4 + cb;
                    if (step + RD - 1 < NKB * 4) ring[(step + RD - 1) % RD] = load_unit(step + RD - 1);
                    const SplitA &wv = ring[step % RD];
                    // the six plane products (smallest first), node rows as A, weights as B: lane (j, h) gets column j of nodes 8 a + 4 h + b
                    f32x16 c = acc2[cb];
                    c = mfma32(b.p0, wv.w2, c);
                    c = mfma32(b.p2, wv.w0, c);
                    c = mfma32(b.p1, wv.w1, c);
                    c = mfma32(b.p0, wv.w1, c);
                    c = mfma32(b.p1, wv.w0, c);
                    c = mfma32(b.p0, wv.w0, c);
                    acc2[cb] = c;
                }
            }
            const int col0 = 128 * half;
            if (split_any_nan<4>(acc2)) {
                fused_cat_exact(a.w, a.xi, tile, K0, K1, col0, row0, r.n_rows, a.bias, a.act, a.out, a.Dout, lane);
                continue;
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int col = col0 + 32 * cb + n;
                const float bv = biasl[col];
                float *op = a.out + ((int64_t)row0 + 4 * h) * a.Dout + col;      // lane (j, h): column j, nodes 4 h + (8 a + b)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int nu = 8 * (q >> 2) + (q & 3);                         // wave-uniform part of the node number
                    float v = acc2[cb][q] + bv;
                    if (a.act == GNNMP_ACT_RELU) v = fmaxf(v, 0.0f);      // (finite here: NaN tiles took the exact path)
                    if ((full_tile || row0 + nu + 4 * h < r.n_rows) && !(a.dbg & 4)) op[(int64_t)nu * a.Dout] = v;
                }
            }
        }
        // the tile is rewritten by the next gather: program order within the wave keeps the reads above before those writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int K0, int K1, int OP, bool SCALED, int NW, int RD, int U>
static int launch_fused_cat_v(FusedCatArgs &a, u32x4 *wimg, hipStream_t stream) {
    constexpr int NKB = (K0 + K1 + 15) / 16, FC2_WAVES = NW;
    const size_t lds = (size_t)FC2_WAVES * FC2_ROWS * K1 * sizeof(float) + (size_t)a.Dout * sizeof(float);
    if (lds > 160 * 1024) return 1;
    fused_cat_wimg_kernel<K0, K1><<<a.NH * NKB * 4, 64, 0, stream>>>(a.w, a.Dout, wimg);
    GNNMP_LAUNCH_CHECK("fused_cat_wimg_kernel");
    a.wimg = wimg;
    GNNMP_LDS_OPTIN("fused_cat_kernel", &fused_cat_kernel<K0, K1, OP, SCALED, NW, RD, U>);
    const int ntiles = (a.r.n_rows + FC2_ROWS - 1) / FC2_ROWS;
    const int gx = std::min(device_cus(), (ntiles + FC2_WAVES - 1) / FC2_WAVES);
    GNNMP_HIP(hipMemsetAsync(a.ticket, 0, 2 * sizeof(uint32_t), stream));
    fused_cat_kernel<K0, K1, OP, SCALED, NW, RD, U><<<gx, 64 * FC2_WAVES, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("fused_cat_kernel");
    return GNNMP_OK;
}
// knob 14 picks the variant (A/B runs): 8 = 8 waves a block (256 registers, three W units in flight), anything else = 12 waves (168
// registers, two units in flight)
template <int K0, int K1, int OP, bool SCALED>
static int launch_fused_cat(FusedCatArgs &a, u32x4 *wimg, hipStream_t stream) {
    if (knob(KNOB_FUSED_WAVES) == 8) return launch_fused_cat_v<K0, K1, OP, SCALED, 8, 3, 8>(a, wimg, stream);
    return launch_fused_cat_v<K0, K1, OP, SCALED, 12, 2, 8>(a, wimg, stream);
}

template <int K0, int K1>
static int dispatch_fused_cat(FusedCatArgs &a, int op, bool scaled, u32x4 *wimg, hipStream_t stream) {
    if (scaled) return 1;      // (sage_conv / graph_conv never scale: not instantiated)
    switch (op) {
        case OP_SUM: return launch_fused_cat<K0, K1, OP_SUM, false>(a, wimg, stream);
        case OP_MAX: return launch_fused_cat<K0, K1, OP_MAX, false>(a, wimg, stream);
        default: return launch_fused_cat<K0, K1, OP_MIN, false>(a, wimg, stream);
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
    // sage_conv / graph_conv with a root term and a multiple of 128 outputs: fused_cat_kernel (W streamed from L2 as pre-split bf16 planes)
    if (D1 > 0 && Dout > 128) {
        const bool shape_ok = D == 100 && D1 == 100 && (Dout & 127) == 0 && Dout <= 512 && w_layout == 0 && p->n_dst >= FC2_ROWS && !scaled && !(reinterpret_cast<uintptr_t>(xj) & 15) && !(reinterpret_cast<uintptr_t>(xi) & 15) &&
                              !(reinterpret_cast<uintptr_t>(out) & 3) && !(agg_out && (reinterpret_cast<uintptr_t>(agg_out) & 15)) &&
                              (ldw_root & 3) == 0 && (ldw_agg & 3) == 0 && knob(KNOB_FUSED_WAVES) >= 0;
        if (!shape_ok)
            return fail(GNNMP_EUNSUPPORTED, "fused_conv: shape not fused (D=%lld D1=%lld Dout=%lld)", (long long)D, (long long)D1, (long long)Dout);
        // NOT the default (knob 14 > 0 runs it): measured on the products shape it LOSES to csr_rows_kernel + dense_wreg_kernel, 6.85 ms
        // against 5.77-5.82 (tools/experiments/sage_fused_ab.py, profiles/r06_sage_fused_ab.txt; see the kernel's header for why)
        if (knob(KNOB_FUSED_WAVES) == 0)
            return fail(GNNMP_EUNSUPPORTED, "fused_conv: root term with > 128 outputs, the two-kernel path is faster (knob 14 > 0 forces fused_cat_kernel)");
        const int NH = (int)(Dout / 128);
        const size_t pc = (size_t)p->n_chunks * (size_t)D, pl = (size_t)p->n_long * (size_t)D;
        const size_t wimg_at = (((pc + 3) & ~(size_t)3) + pl + 255) & ~(size_t)255;        // 1 KB aligned
        const size_t wimg_floats = (size_t)NH * 13 * 4 * 3 * 64 * 4;
        if (int rc = ensure_workspace(p, wimg_at + wimg_floats + 4)) return rc;
        if (int rc = ensure_ticket(p, stream)) return rc;
        float *agg_long = p->ws + ((pc + 3) & ~(size_t)3);
        if (p->n_long > 0) {
            if (int rc = run_reduce(p, p->col, aggr, xj, w, scale_src, w_slot, ss_slot, scale_dst, agg_long, D, stream, nullptr,
                                    nullptr, nullptr, 1, 0, 1, nullptr, 0, nullptr, nullptr))
                return rc;
        }
        FusedCatArgs a = {};
        ReduceArgs &r = a.r;
        r.rowptr = p->rowptr; r.idx = p->col; r.eid = p->eid; r.x = xj; r.w = w; r.ss = scale_src; r.w_slot = w_slot; r.ss_slot = ss_slot;
        r.sd = scale_dst; r.long_rows = p->long_rows; r.n_long = p->n_long; r.D = (int)D; r.n_rows = (int)p->n_dst; r.n_src = (int)p->n_src;
        r.n_edges = (uint32_t)p->n_edges; r.mean = (aggr == GNNMP_MEAN); r.long_thresh = p->long_thresh;
        r.log2g = pick_log2g((D + 3) / 4);
        if (r.log2g != 5) return fail(GNNMP_EUNSUPPORTED, "fused_conv: lane-group width forced by a knob");
        a.agg_long = agg_long; a.agg_out = agg_out; a.xi = xi;
        a.w.W[0] = W_root; a.w.W[1] = W_agg; a.w.K[0] = (int)D1; a.w.K[1] = (int)D;
        a.w.sj[0] = ldw_root; a.w.sk[0] = 1; a.w.sj[1] = ldw_agg; a.w.sk[1] = 1;
        a.bias = bias; a.act = act; a.out = out; a.Dout = (int)Dout; a.NH = NH; a.ticket = p->ticket;
        a.dbg = knob(KNOB_T16_DEBUG);
        const int rc = dispatch_fused_cat<100, 100>(a, op, scaled, reinterpret_cast<u32x4 *>(p->ws + wimg_at), stream);
        if (rc == 1) return fail(GNNMP_EUNSUPPORTED, "fused_conv: no LDS for the row tiles");
        return rc;
    }
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
