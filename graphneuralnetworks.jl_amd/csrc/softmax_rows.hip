// softmax_rows.hip — softmax_edge_neighbors (GNNlib/src/utils.jl:84-97) / softmax_nodes / softmax_edges (:49-72) for NARROW rows
// (a handful of heads or channels per edge) in ONE pass over the edge rows.
//
// The reference's three steps — max_ = scatter(max, e, t); den = scatter(+, exp.(e .- max_[t]), t); num ./ den[t] — read every
// edge row three times, and with H = 8 a row is 32 bytes somewhere in a matrix in ORIGINAL edge order: each read is a whole
// 128-byte line request, a wave's rows are out of every cache long before the next step comes back to them, and a lane group
// of H / 4 lanes keeps only that many row loads in flight (propagate.hip's row walk is built for wide rows).  The three-step
// kernels took 8.4 ms on the products shape at H = 8.  Here a WAVE owns up to 64 consecutive destinations and walks them in
// batches that fit its private LDS region:
//   A  (edge-major)  gather the batch's edge rows once into LDS: 64 / G edges per load instruction, 8 loads in flight per lane
//   B  (row-major)   one lane group per destination folds its row's max from LDS
//   C  (edge-major)  num = exp(e - max[row]) in place
//   D  (row-major)   den = the row's sum in ORIGINAL edge order (the order NNlib's scatter loop adds in)
//   E  (edge-major)  alpha = num / den[row], stored to the edge's row of the output
// Same operations on the same values in the same order as the three steps: results are bit-identical to them (tested), at
// two line requests per edge instead of four.  A row longer than a batch (but not split by the plan) is swept three times by
// its wave alone, the running max / sum carried across its sub-batches in edge order.  Rows the plan splits (> long_thresh)
// go chunk by chunk through softmax_chunk_kernel — a wave per chunk, edge-parallel loads — with the same partials and the
// same combine kernels as the three-step path.
#include "csr_reduce.h"

namespace gnnmp {

struct SmxArgs {
    const uint32_t *rowptr;
    const int32_t *eid;
    const float *e;
    float *alpha;
    int D, n_rows, log2g, long_thresh;
    int cap;          // slots per batch (cap << log2g = 512: 8 items per lane)
    int waves;        // waves per block
    int rw;           // destinations per wave (<= 64)
    int wave_bytes;   // LDS per wave
    float den_add;
    // the plan's split rows (softmax_chunk_kernel)
    const int32_t *chunk_row;
    const uint32_t *chunk_beg, *chunk_end;
    int n_chunks;
    float *partial;       // [n_chunks][D]
    const float *mx;      // [n_rows][D] (split rows only)
    const float *den;     // [n_rows][D] (split rows only)
};

__device__ __forceinline__ void smx_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct SmxBatch {
    int r, nb;           // first row (wave-relative), rows; nb = 0: none left
    uint32_t sb;         // first slot (slots are unsigned 32-bit)
    int ns;              // slots (nb = 1 and ns > cap: a row longer than a batch; never more than long_thresh)
};

constexpr int SMX_IT = 8;   // items (edge, lane-of-edge) per lane and batch

// The max of a softmax needs no Julia `max` (common.h: jl_max, ~8 instructions): the sign of a zero maximum cancels in
// e - max, and a NaN in the row makes every weight of the row NaN whether the maximum carries it (the reference) or skips it
// (v_max_f32) — exp(NaN - m) poisons the denominator either way.  One instruction.
template <int OP>
__device__ __forceinline__ float smx_apply(float x, float y) {
    return OP == OP_MAX ? fmaxf(x, y) : x + y;
}

// fold LDS rows [st, st + len) into acc, 8 reads in flight; every lane of the group calls it
template <int VEC, int OP>
__device__ __forceinline__ void smx_fold(const float *vals, int Dp, int f0, int st, int len, float (&acc)[VEC]) {
    for (int t = 0; t < len; t += 8) {
        float v[8][VEC];
#pragma unroll
        for (int u = 0; u < 8; ++u) Vec<VEC>::load(vals + (size_t)(st + min(t + u, len - 1)) * Dp + f0, v[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t + u < len) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = smx_apply<OP>(acc[q], v[u][q]);
            }
        }
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) softmax_rows_lds_kernel(const SmxArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int IT = SMX_IT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = 1 << a.log2g, Dp = G * VEC, RB = 64 >> a.log2g;
    unsigned char *base = smem + (size_t)wave * a.wave_bytes;
    float *vals = reinterpret_cast<float *>(base);           // [cap][Dp]
    float *rmx = vals + (size_t)a.cap * Dp;                   // [RB][Dp]
    float *rden = rmx + RB * Dp;                              // [RB][Dp]
    uint32_t *wrp = reinterpret_cast<uint32_t *>(rden + RB * Dp);   // [rw + 1] the wave's row pointers (68 reserved)
    int *seid = reinterpret_cast<int *>(wrp + 68);            // [cap] original edge position of a slot (unsigned 32-bit)
    unsigned char *srow = reinterpret_cast<unsigned char *>(seid + a.cap);   // [cap] batch-relative row of a slot
    const int64_t r0l = ((int64_t)blockIdx.x * a.waves + wave) * a.rw;
    if (r0l >= a.n_rows) return;
    const int r0 = (int)r0l, nr = min(a.rw, a.n_rows - r0);
    wrp[lane] = a.rowptr[r0 + min(lane, nr)];
    if (lane == 0) wrp[64] = a.rowptr[r0 + min(64, nr)];
    smx_wave_sync();
    const int lig = lane & (G - 1), grp = lane >> a.log2g;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;

    // rows [r, r + nb): as many as fit one batch — one lane group each (RB), cap slots in all, none of them split (a split
    // row ends the batch before it and is stepped over: softmax_chunk_kernel has it); a row longer than cap comes alone
    auto find = [&](int r) -> SmxBatch {
        while (r < nr) {
            const uint32_t sb = wrp[r];
            const int ci = min(r + lane + 1, nr);
            const uint32_t hi = wrp[ci];
            const bool ok = lane < RB && r + lane + 1 <= nr && hi - sb <= (uint32_t)a.cap && hi - wrp[ci - 1] <= (uint32_t)a.long_thresh;
            const unsigned long long m = __ballot(ok);
            const int nb = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
            if (nb > 0) return SmxBatch{r, nb, sb, (int)(wrp[r + nb] - sb)};
            const uint32_t len0 = wrp[r + 1] - sb;
            if (len0 <= (uint32_t)a.long_thresh) return SmxBatch{r, 1, sb, (int)len0};
            ++r;
        }
        return SmxBatch{nr, 0, 0, 0};
    };
    auto load_eids = [&](uint32_t sb, int ns, int (&c)[IT]) {
        const int nitems = min(ns, a.cap) << a.log2g;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = k * 64 + lane;
            c[k] = i < nitems ? a.eid[sb + (i >> a.log2g)] : 0;
        }
    };
    auto load_rows = [&](int ns, const int (&c)[IT], float (&v)[IT][VEC]) {
        const int nitems = min(ns, a.cap) << a.log2g;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = k * 64 + lane;
            if (i < nitems && active) Vec<VEC>::load(a.e + (int64_t)(uint32_t)c[k] * a.D + f0, v[k]);
        }
    };
    // registers -> LDS (rows and edge positions of one batch)
    auto stash = [&](int ns, const int (&c)[IT], const float (&v)[IT][VEC]) {
        const int nitems = ns << a.log2g;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = k * 64 + lane;
            if (i < nitems) {
                if (active) Vec<VEC>::store(vals + (size_t)(i >> a.log2g) * Dp + f0, v[k]);
                if (lig == 0) seid[i >> a.log2g] = c[k];
            }
        }
    };
    // C: num = exp.(e .- max_[t])   (utils.jl:94)
    auto phase_exp = [&](int nitems) {
#pragma unroll 2
        for (int i = lane; i < nitems; i += 64) {
            if (active) {
                const int s = i >> a.log2g;
                const int j = srow[s];
                float v[VEC], mq[VEC];
                Vec<VEC>::load(vals + (size_t)s * Dp + f0, v);
                Vec<VEC>::load(rmx + j * Dp + f0, mq);
#pragma unroll
                for (int q = 0; q < VEC; ++q) v[q] = expf(v[q] - mq[q]);
                Vec<VEC>::store(vals + (size_t)s * Dp + f0, v);
            }
        }
    };
    // E: num ./ den[t], back in original edge order
    auto phase_write = [&](int nitems) {
#pragma unroll 4
        for (int i = lane; i < nitems; i += 64) {
            if (active) {
                const int s = i >> a.log2g;
                const int j = srow[s];
                const int c = seid[s];
                float v[VEC], dq[VEC];
                Vec<VEC>::load(vals + (size_t)s * Dp + f0, v);
                Vec<VEC>::load(rden + j * Dp + f0, dq);
#pragma unroll
                for (int q = 0; q < VEC; ++q) v[q] = v[q] / dq[q];
                Vec<VEC>::store(a.alpha + (int64_t)(uint32_t)c * a.D + f0, v);
            }
        }
    };

    // Software pipeline over the wave's batches: while batch k is folded out of LDS, the edge rows of batch k + 1 are in
    // flight into registers and the edge positions of batch k + 2 behind them.
    SmxBatch cur = find(0);
    if (cur.nb == 0) return;
    int ca[IT], cb[IT];
    float rows[IT][VEC];
    load_eids(cur.sb, cur.ns, ca);
    SmxBatch nxt = find(cur.r + cur.nb);
    load_rows(cur.ns, ca, rows);
    load_eids(nxt.sb, nxt.ns, cb);
    while (true) {
        const int r = cur.r, nb = cur.nb;
        const uint32_t sb = cur.sb;
        const bool big = cur.ns > a.cap;
        const int ns = min(cur.ns, a.cap);
        // A: the batch's edge rows (and positions), out of the registers they arrived in; then the next batch's rows and the
        // positions of the one after it
        stash(ns, ca, rows);
        if (nxt.nb) load_rows(nxt.ns, cb, rows);
#pragma unroll
        for (int k = 0; k < IT; ++k) ca[k] = cb[k];
        const SmxBatch nn = nxt.nb ? find(nxt.r + nxt.nb) : nxt;
        if (nn.nb) load_eids(nn.sb, nn.ns, cb);
        // slot -> row of the batch: the number of row starts at or before the slot
        {
            int rid[IT];
#pragma unroll
            for (int k = 0; k < IT; ++k) rid[k] = 0;
            for (int j = 1; j < nb; ++j) {
                const int st = (int)(wrp[r + j] - sb);   // same address in every lane: an LDS broadcast
#pragma unroll
                for (int k = 0; k < IT; ++k) rid[k] += (lane + 64 * k >= st) ? 1 : 0;
            }
#pragma unroll
            for (int k = 0; k < IT; ++k)
                if (lane + 64 * k < ns) srow[lane + 64 * k] = (unsigned char)rid[k];
        }
        smx_wave_sync();
        const bool rowlane = grp < nb && active;
        if (!big) {
            const int st = rowlane ? (int)(wrp[r + grp] - sb) : 0;
            const int len = rowlane ? (int)(wrp[r + grp + 1] - wrp[r + grp]) : 0;
            const int nitems = ns << a.log2g;
            // B: max_ of every row of the batch
            float acc[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP_MAX>();
            smx_fold<VEC, OP_MAX>(vals, Dp, f0, st, len, acc);
            if (rowlane) Vec<VEC>::store(rmx + grp * Dp + f0, acc);
            smx_wave_sync();
            phase_exp(nitems);
            smx_wave_sync();
            // D: den = scatter(+, num, t): the row's sum in edge order
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP_SUM>();
            smx_fold<VEC, OP_SUM>(vals, Dp, f0, st, len, acc);
            if (a.den_add != 0.0f) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = acc[q] + a.den_add;
            }
            if (rowlane) Vec<VEC>::store(rden + grp * Dp + f0, acc);
            smx_wave_sync();
            phase_write(nitems);
            smx_wave_sync();
        } else {
            // one row of cur.ns > cap slots (srow is all zero): three sweeps over its sub-batches, the first one already in LDS.
            // Lane group 0 carries the running max / sum; the sub-batches after the first are gathered on the spot.
            const int total = cur.ns;
            auto gather = [&](int off) {   // slots [off, off + cap) of the row -> LDS; returns the count
                const int n = min(a.cap, total - off);
                int c[IT];
                float v[IT][VEC];
                load_eids(sb + off, n, c);
                load_rows(n, c, v);
                stash(n, c, v);
                smx_wave_sync();
                return n;
            };
            float acc[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP_MAX>();
            for (int off = 0; off < total; off += a.cap) {
                const int n = off == 0 ? ns : gather(off);
                smx_fold<VEC, OP_MAX>(vals, Dp, f0, 0, rowlane ? n : 0, acc);
                smx_wave_sync();
            }
            if (rowlane) Vec<VEC>::store(rmx + f0, acc);
            smx_wave_sync();
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = op_identity<OP_SUM>();
            for (int off = 0; off < total; off += a.cap) {
                const int n = gather(off);
                phase_exp(n << a.log2g);
                smx_wave_sync();
                smx_fold<VEC, OP_SUM>(vals, Dp, f0, 0, rowlane ? n : 0, acc);
                smx_wave_sync();
            }
            if (a.den_add != 0.0f) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = acc[q] + a.den_add;
            }
            if (rowlane) Vec<VEC>::store(rden + f0, acc);
            smx_wave_sync();
            for (int off = 0; off < total; off += a.cap) {
                const int n = gather(off);
                phase_exp(n << a.log2g);
                smx_wave_sync();
                phase_write(n << a.log2g);
                smx_wave_sync();
            }
        }
        if (nxt.nb == 0) break;
        cur = nxt;
        nxt = nn;
    }
}

// ---- the plan's split rows: one wave per chunk ------------------------------------------------
// MODE 0: partial[v] = max of the chunk's rows (any order: max is exact)           -> csr_combine_kernel<MAX>  -> mx[row]
// MODE 1: partial[v] = sum over the chunk, in edge order, of exp(e - mx[row])      -> csr_combine_kernel<SUM>  -> den[row]
// MODE 2: alpha = exp(e - mx[row]) / (den[row] + den_add)
// The partials and their fold are those of the three-step kernels (propagate.hip), so the split rows come out bit-identical
// to them; what changes is that the chunk's up to long_thresh row loads are issued by 64 lanes, eight deep, instead of by
// one lane group of H / 4 lanes.
template <int VEC, int MODE>
__global__ void __launch_bounds__(256) softmax_chunk_kernel(const SmxArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int IT = SMX_IT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = 1 << a.log2g, Dp = G * VEC;
    const int v = (int)blockIdx.x * a.waves + wave;
    if (v >= a.n_chunks) return;
    const int row = a.chunk_row[v];
    const uint32_t beg = a.chunk_beg[v], end = a.chunk_end[v];
    const int lig = lane & (G - 1), grp = lane >> a.log2g;
    const int f0 = lig * VEC;
    const bool active = f0 < a.D;
    float *vals = reinterpret_cast<float *>(smem + (MODE == 1 ? (size_t)wave * a.wave_bytes : 0));
    float rm[VEC], rd[VEC], acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        rm[q] = 0.0f;
        rd[q] = 1.0f;
        acc[q] = MODE == 0 ? op_identity<OP_MAX>() : op_identity<OP_SUM>();
    }
    if (MODE >= 1 && active) Vec<VEC>::load(a.mx + (int64_t)row * a.D + f0, rm);
    if (MODE == 2 && active) {
        Vec<VEC>::load(a.den + (int64_t)row * a.D + f0, rd);
        if (a.den_add != 0.0f) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) rd[q] = rd[q] + a.den_add;
        }
    }
    for (uint32_t off = beg; off < end; off += a.cap) {
        const int n = (int)min((uint32_t)a.cap, end - off);
        const int nitems = n << a.log2g;
        int c[IT];
        float x[IT][VEC];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = k * 64 + lane;
            c[k] = i < nitems ? a.eid[off + (i >> a.log2g)] : 0;
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = k * 64 + lane;
            if (i < nitems && active) Vec<VEC>::load(a.e + (int64_t)(uint32_t)c[k] * a.D + f0, x[k]);
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = k * 64 + lane;
            if (i < nitems && active) {
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[q] = fmaxf(acc[q], x[k][q]);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) x[k][q] = expf(x[k][q] - rm[q]);
                    if (MODE == 1) {
                        Vec<VEC>::store(vals + (size_t)(i >> a.log2g) * Dp + f0, x[k]);
                    } else {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) x[k][q] = x[k][q] / rd[q];
                        Vec<VEC>::store(a.alpha + (int64_t)(uint32_t)c[k] * a.D + f0, x[k]);
                    }
                }
            }
        }
        if (MODE == 1) {
            smx_wave_sync();
            smx_fold<VEC, OP_SUM>(vals, Dp, f0, 0, (grp == 0 && active) ? n : 0, acc);   // edge order, one lane group
            smx_wave_sync();
        }
    }
    if (MODE == 0) {
        // the lanes that hold the same features (same lane-of-edge) meet: xor butterfly over the group index
        for (int d = G; d < 64; d <<= 1) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = fmaxf(acc[q], __shfl_xor(acc[q], d, 64));
        }
    }
    if (MODE <= 1 && grp == 0 && active) Vec<VEC>::store(a.partial + (int64_t)v * a.D + f0, acc);
}

template <int VEC, int MODE>
static int launch_chunks(const SmxArgs &a0, hipStream_t stream) {
    SmxArgs a = a0;
    a.waves = 4;
    a.wave_bytes = a.cap * (1 << a.log2g) * VEC * 4;
    const size_t lds = MODE == 1 ? (size_t)a.waves * a.wave_bytes : 0;
    const unsigned blocks = (unsigned)((a.n_chunks + a.waves - 1) / a.waves);
    softmax_chunk_kernel<VEC, MODE><<<blocks, 64 * a.waves, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("softmax_chunk_kernel");
    return GNNMP_OK;
}

template <int VEC>
static int launch_smx(const SmxArgs &a, hipStream_t stream) {
    const size_t lds = (size_t)a.waves * a.wave_bytes;
    GNNMP_LDS_OPTIN("softmax_rows_lds_kernel", &softmax_rows_lds_kernel<VEC>);
    const int64_t wave_rows = ((int64_t)a.n_rows + a.rw - 1) / a.rw;
    const int64_t blocks = (wave_rows + a.waves - 1) / a.waves;
    softmax_rows_lds_kernel<VEC><<<(unsigned)blocks, 64 * a.waves, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("softmax_rows_lds_kernel");
    return GNNMP_OK;
}

int run_combine(gnnmp_graph_t *p, float *out, int64_t D, int op, hipStream_t stream);   // propagate.hip

// softmax over the rows of a plan for narrow rows.  Returns GNNMP_OK if it ran, 1 if the row width is not one these kernels
// take (the caller runs the three-step kernels instead).  partial: [n_chunks][D]; mx, den: [n_dst][D] (touched on split rows
// only) — the caller's workspace.
int softmax_rows_try(gnnmp_graph_t *p, const float *e, float *alpha, int64_t D, float den_add, float *partial, float *mx,
                     float *den, hipStream_t stream) {
    if (knob(KNOB_SOFTMAX_ROWS) < 0) return 1;
    if (D > 256) return 1;
    SmxArgs a = {};
    const int vec = pick_vec(D, e, alpha);
    const int lanes = (int)((D + vec - 1) / vec);
    if (lanes > 64) return 1;
    a.log2g = 0;
    while ((1 << a.log2g) < lanes) ++a.log2g;   // not pick_log2g: the batch layout needs every lane of a row in ONE group
    const int G = 1 << a.log2g, Dp = G * vec, RB = 64 / G;
    a.cap = SMX_IT * 64 / G;
    if (a.cap < 64) return 1;                   // more than 8 lanes per row (H > 32): whole lines per row, the row walk of
                                                // propagate.hip is the tool (H = 64: 14.0 ms here, 13.2 ms there; H = 32: 6.6 / 8.5)
    a.rowptr = p->rowptr;
    a.eid = p->eid;
    a.e = e;
    a.alpha = alpha;
    a.D = (int)D;
    a.n_rows = (int)p->n_dst;
    a.long_thresh = p->long_thresh;
    a.den_add = den_add;
    const size_t bytes = ((size_t)a.cap * Dp + 2 * (size_t)RB * Dp + 68 + (size_t)a.cap) * 4 + (size_t)a.cap;
    a.wave_bytes = (int)((bytes + 15) & ~(size_t)15);
    a.waves = 4;
    // destinations per wave: 64, fewer on small inputs so that every CU has waves to run
    a.rw = 64;
    while (a.rw > 8 && ((int64_t)a.n_rows + a.rw - 1) / a.rw < 16 * (int64_t)device_cus()) a.rw >>= 1;
    int rc;
    switch (vec) {
        case 4: rc = launch_smx<4>(a, stream); break;
        case 2: rc = launch_smx<2>(a, stream); break;
        default: rc = launch_smx<1>(a, stream); break;
    }
    if (rc != GNNMP_OK || p->n_chunks == 0) return rc;
    a.chunk_row = p->chunk_row;
    a.chunk_beg = p->chunk_beg;
    a.chunk_end = p->chunk_end;
    a.n_chunks = p->n_chunks;
    a.partial = partial;
    a.mx = mx;
    a.den = den;
#define SMX_CHUNKS(MODE)                                                     \
    switch (vec) {                                                           \
        case 4: rc = launch_chunks<4, MODE>(a, stream); break;               \
        case 2: rc = launch_chunks<2, MODE>(a, stream); break;               \
        default: rc = launch_chunks<1, MODE>(a, stream); break;              \
    }                                                                        \
    if (rc != GNNMP_OK) return rc;
    SMX_CHUNKS(0)
    if ((rc = run_combine(p, mx, D, GNNMP_MAX, stream)) != GNNMP_OK) return rc;
    SMX_CHUNKS(1)
    if ((rc = run_combine(p, den, D, GNNMP_SUM, stream)) != GNNMP_OK) return rc;
    SMX_CHUNKS(2)
#undef SMX_CHUNKS
    return GNNMP_OK;
}

}  // namespace gnnmp
