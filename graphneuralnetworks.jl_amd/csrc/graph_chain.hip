// graph_chain.hip — BASELINE.json config 5 as ONE graph-parallel kernel:
//   GNNChain(GraphConv(D0 => D1, σ), ..., GraphConv(D_{L-1} => D_L, σ), GlobalPool(mean | +), Dense(D_L => nout))
//   (examples/graph_classification_tudataset.jl:79-82; layer bodies GNNlib/src/layers/conv.jl:102-108, layers/pool.jl:3-5)
// on a batched GNNGraph (MLUtils.batch, GNNGraphs/src/transform.jl:682-709: block-diagonal, member graphs contiguous).
//
// Round 2 ran this chain as six launches with every intermediate through HBM (0.30 ms, 0.36 of its roofline).  Here a BLOCK owns a
// contiguous run of whole member graphs (~ N / #CU rows: 960 at G = 8192) and walks the chain in phases separated by block
// barriers; nothing but x, the edge index and the (G, nout) result crosses HBM on the way:
//   layer l, part p   every wave draws 32-row tiles of the block's rows from an LDS ticket and runs them through the split-bf16
//                     MFMA core (msplit.h) against the layer's W planes in LDS.  The B operand of the contraction is formed on the
//                     fly: the ROOT positions are the row itself (two 16-byte loads per k-block), the AGGREGATE positions are
//                     sum_j h[j] over the row's in-neighbours, gathered in ORIGINAL edge order from the previous layer's
//                     output (the adds of NNlib.scatter(+): the aggregate has the bits of gnnmp_propagate_f32's; `mean` divides
//                     by the count).  A layer whose 2 Din x 128 image does not fit LDS (Din = 128: 192 KB of planes) runs as two
//                     parts — W_root * x first, the pre-activation parked in the output buffer, then W_agg * m on top of it.
//                     bias + σ on the accumulators; layer l < L writes its rows to a scratch matrix that stays in the block's
//                     L2 slice (the next layer's gathers read other rows of the same graphs: same block, same CU, after a
//                     barrier — no inter-block traffic at all); the last layer never writes its rows: each wave folds its tile
//                     into z[row] = W_head * h[row] (nout <= 8 numbers per row — Dense and mean / + pooling commute).
//   pool              one thread per (graph, output) sums z over the graph's rows in row order, divides by the node count
//                     (mean), adds the head's bias: nout floats per graph leave the kernel.
// Tiles ignore graph boundaries (rows of one block are contiguous, whole graphs): no MFMA work is spent on padding except the last
// tile of a block.  Sizes: Din, Dout multiples of 4, Dout <= 128, nout <= 8, aggr in {+, mean}, pool in {+, mean}; anything else
// returns GNNMP_EUNSUPPORTED and the caller runs the chain layer by layer.
#include <algorithm>

#include "msplit.h"

namespace gnnmp {

constexpr int CHAIN_MAX_LAYERS = 4;
constexpr int CHAIN_MAX_PHASES = 2 * CHAIN_MAX_LAYERS;
constexpr int CHAIN_DP = 128, CHAIN_NCB = 4;
constexpr int CHAIN_THREADS = 512;   // 8 waves, 256 VGPRs: no spill in the tile loop (a spilled register reloaded there is a scratch load:
                                     // vmcnt(0), i.e. waiting for every store in flight)

struct ChainLayer {
    const float *W_root, *W_agg, *bias;
    int64_t sj, sk;     // W(j, k) of W_root / W_agg at [j * sj + k * sk]: C row-major [Dout][Din] (sj = Din, sk = 1) or Julia's (Dout, Din)
                        // column-major as stored (sj = 1, sk = Dout)
    int Din, Dout, act;
    int parts;          // 1: root and aggregate positions in one pass; 2: one pass each (the pre-activation is parked in between)
};
struct ChainArgs {
    const uint32_t *rowptr;
    const int32_t *col;
    const int64_t *seg_ptr;   // [G + 1] first row of each member graph
    int G, N;
    const float *x;           // [N][L[0].Din]
    ChainLayer L[CHAIN_MAX_LAYERS];
    int n_layers;
    int mean_aggr, pool_mean;
    const float *W_head;      // W_head(o, c) at [o * hsj + c * hsk]
    int64_t hsj, hsk;
    const float *b_head;
    int nout;
    float *h[2];              // scratch: layer l writes h[l & 1]
    int ldh_buf[2];           // row stride of h[0] / h[1]: ONE stride per buffer (the widest layer it holds), so that a row's bytes belong to
                              // the block that owns the row in every layer — blocks are not synchronised with each other
    float *z;                 // [N][nout]
    float *out;               // [G][nout]
    int waves;
};

struct ChainCtl {
    int r0, r1, g0, g1;
    int ticket[CHAIN_MAX_PHASES];
    int bad[CHAIN_MAX_PHASES];   // some tile of the pass met a non-finite operand (NaN accumulators): the pass is re-scanned
};

// what a lane knows about its row's in-neighbours: the first four source rows as pointers into the previous layer's output
struct Nbrs {
    const float *p[4];
    uint32_t beg;
    int deg;
};

// B-operand pieces in flight for one k-block: [piece u][neighbour j] (root positions use [u][0] only)
struct Fetch {
    float4 v[2][4];
};

// mode of one pass over the tiles
template <int DIN_, bool ROOT_, bool AGG_>
struct Pass {
    static constexpr int DIN = DIN_;     // compile-time Din, or 0: run time
    static constexpr bool ROOT = ROOT_, AGG = AGG_;
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 add4(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sel4(bool c, const float4 a, const float4 b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// One 32-row tile of one pass.  src: the layer's input rows (row stride din); hdst: the layer's output buffer (row stride dout).
template <class P, bool FULLCOLS>
__device__ __forceinline__ void chain_tile(const ChainArgs &a, const ChainLayer &ly, int part, bool last_layer,
                                           const float *__restrict__ src, int src_ld, float *__restrict__ hdst, int dst_ld,
                                           const u32x4 *__restrict__ img, const float4 *__restrict__ bias4,
                                           const float *__restrict__ head, int r0, int r1, int t, int n, int h, SplitA &cur,
                                           bool &any_bad) {
    constexpr int NCB = CHAIN_NCB, DP = CHAIN_DP;
    const int din = P::DIN > 0 ? P::DIN : ly.Din;
    const int dout = ly.Dout;
    const int kroot = P::ROOT ? din : 0;
    const int kcat = kroot + (P::AGG ? din : 0);
    const int nkb = split_nkb(kcat);
    const int units = nkb * 2 * DP;
    const int row = r0 + 32 * t + n;
    const int rowc = min(row, r1 - 1);
    const float *xr = src + (int64_t)rowc * src_ld;
    Nbrs nb = {};
    int degmax = 0, degmin = 0;
    if (P::AGG) {
        nb.beg = a.rowptr[rowc];
        nb.deg = (int)(a.rowptr[rowc + 1] - nb.beg);     // (rows past r1 are copies of row r1 - 1, neighbours included)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = j < nb.deg ? a.col[nb.beg + j] : rowc;
            nb.p[j] = src + (int64_t)m * src_ld;
        }
        degmax = wave_max(nb.deg);
        degmin = wave_min(nb.deg);
    }
    // the loads of k-block kb: root positions = the row itself, aggregate positions = the first four neighbours' rows
    auto issue = [&](int kb, Fetch &f) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = 16 * kb + 4 * h + 8 * u;
            if (P::ROOT && (!P::AGG || 16 * kb < kroot)) {           // (root and aggregate k-blocks never mix: Din % 16 == 0 when both)
                f.v[u][0] = ld4(xr + min(c, kroot - 4));
            } else {
                const int cc = min(c - kroot, din - 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) f.v[u][j] = ld4(nb.p[j] + cc);
            }
        }
    };
    auto finish = [&](int kb, const Fetch &f, float4 (&q)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (P::ROOT && (!P::AGG || 16 * kb < kroot)) {
                q[u] = f.v[u][0];
            } else {
                // NNlib.scatter(+): dst = 0, then dst += src for the edges in order
                float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (degmin >= 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = add4(s, f.v[u][j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = sel4(j < nb.deg, add4(s, f.v[u][j]), s);
                }
                if (degmax > 4) {
                    // rows with more than four in-neighbours: the rest one at a time (an inline loop, NOT a call: a call inside the
                    // k-block loop makes hipcc drain vmcnt around it on every path)
                    const int cc = min(16 * kb + 4 * h + 8 * u - kroot, din - 4);
#pragma unroll 1
                    for (int j = 4; j < degmax; ++j) {
                        const bool ok = j < nb.deg;
                        const int m = ok ? a.col[nb.beg + j] : rowc;    // (never col[beg] of a row without edges: for the last rows that is col[E])
                        const float4 v = ld4(src + (int64_t)m * src_ld + cc);
                        s = sel4(ok, add4(s, v), s);
                    }
                }
                if (a.mean_aggr && nb.deg > 0) {   // NNlib scatter(mean): 0 .+ sum ./ count
                    const float cnt = (float)nb.deg;
                    s = make_float4(0.0f + s.x / cnt, 0.0f + s.y / cnt, 0.0f + s.z / cnt, 0.0f + s.w / cnt);
                }
                q[u] = s;
            }
        }
    };

    f32x16 acc[NCB];
    float *out_row = hdst + (int64_t)rowc * dst_ld;
    if (part == 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
    } else {
        // the pre-activation the root pass parked in the output buffer
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int colq = 32 * cb + 8 * q4 + 4 * h;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (colq < dout) v = ld4(out_row + colq);
                acc[cb][4 * q4] = v.x; acc[cb][4 * q4 + 1] = v.y; acc[cb][4 * q4 + 2] = v.z; acc[cb][4 * q4 + 3] = v.w;
            }
    }
    Fetch fa, fb;
    float4 q[2];
    issue(0, fa);
    finish(0, fa, q);
    if constexpr (P::DIN > 0) {
        constexpr int NKB = P::DIN > 0 ? ((P::ROOT ? P::DIN : 0) + (P::AGG ? P::DIN : 0) + 15) / 16 : 1;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + 1 < NKB) issue(kb + 1, fb);
            const Split8 b = split8(q[0], q[1]);
            split_kblock<NCB>(acc, img, units, DP, kb, kb + 1 < NKB ? kb + 1 : 0, n, h, b, cur);
            if (kb + 1 < NKB) finish(kb + 1, fb, q);
        }
    } else {
        for (int kb = 0; kb < nkb; ++kb) {
            const bool more = kb + 1 < nkb;
            issue(more ? kb + 1 : kb, fb);
            const Split8 b = split8(q[0], q[1]);
            split_kblock<NCB>(acc, img, units, DP, kb, more ? kb + 1 : 0, n, h, b, cur);
            finish(more ? kb + 1 : kb, fb, q);
        }
    }

    // Epilogue: NO branch that depends on the data, no predicated store (dense_split.hip explains why: hipcc sizes the vmcnt waits of the
    // next tile for the path with the fewest stores).  Rows past the block's end hold the operands of row r1 - 1 and store its values
    // again; a tile with NaN accumulators (non-finite operand: msplit.h) is stored as it is and only remembered — chain_redo.
    any_bad |= split_any_nan<NCB>(acc);
    const bool final_part = part + 1 == ly.parts;
    if (!final_part) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int colq = 32 * cb + 8 * q4 + 4 * h;
                const float4 v = make_float4(acc[cb][4 * q4], acc[cb][4 * q4 + 1], acc[cb][4 * q4 + 2], acc[cb][4 * q4 + 3]);
                if (FULLCOLS || colq < dout) *reinterpret_cast<float4 *>(out_row + colq) = v;
            }
    } else if (!last_layer) {
        if (FULLCOLS)
            split_store_all<NCB, true>(acc, bias4, ly.act, out_row, dout, h);
        else
            split_store_all<NCB, false>(acc, bias4, ly.act, out_row, dout, h);
    } else {
        // z[row][o] = sum_col W_head[o][col] * h[row][col]: this lane's 16 NCB columns, then the row's two lanes together
        float zo[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) zo[o] = 0.0f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int colq = 32 * cb + 8 * q4 + 4 * h;
                const float4 v = split_out4(acc[cb], q4, bias4[colq >> 2], ly.act);
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    if (o < a.nout) {
                        const float4 wv = *reinterpret_cast<const float4 *>(head + o * DP + colq);
                        zo[o] = fmaf(wv.x, v.x, fmaf(wv.y, v.y, fmaf(wv.z, v.z, fmaf(wv.w, v.w, zo[o]))));
                    }
                }
            }
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            if (o < a.nout) {
                const float tot = zo[o] + __shfl_xor(zo[o], 32, 64);
                a.z[(int64_t)rowc * a.nout + o] = tot;     // both lanes of a row (and the lanes past r1) store the same value
            }
        }
    }
}

// The exact path of a pass: re-scan the block's tiles, and redo those whose stored rows hold a NaN with plain fp32 fma loops
// (Inf * w = +-Inf, -Inf of an empty max... the reference product's behaviour).  Cold code: runs only when a tile raised ctl->bad.
__device__ __noinline__ void chain_redo(const ChainArgs a, const ChainLayer ly, const WCat w, int part, bool last_layer, bool has_root,
                                        const float *src, int src_ld, float *hdst, int dst_ld, const float *head, int r0, int r1, int wave,
                                        int nwaves, int lane) {
    const int n = lane & 31, h = lane >> 5;
    const int din = ly.Din, dout = ly.Dout;
    const int kroot = has_root ? din : 0;
    const bool final_part = part + 1 == ly.parts;
    const int ntile = (r1 - r0 + 31) >> 5;
    for (int t = wave; t < ntile; t += nwaves) {
        const int row = r0 + 32 * t + n;
        const bool row_ok = row < r1;
        const int rowc = min(row, r1 - 1);
        float *out_row = hdst + (int64_t)rowc * dst_ld;
        bool bad = false;
        if (final_part && last_layer) {
            for (int o = 0; o < a.nout; ++o) { const float zv = a.z[(int64_t)rowc * a.nout + o]; bad |= zv != zv; }
        } else {
            for (int c = h; c < dout; c += 2) { const float v = out_row[c]; bad |= v != v; }
        }
        if (__builtin_amdgcn_ballot_w64(bad) == 0) continue;
        // (a NaN of the ROOT pass's pre-activation is recomputed there; here it is an operand like any other: the second pass adds to it)
        const float *xr = src + (int64_t)rowc * src_ld;
        const float *bias = ly.bias;
        const int act = ly.act, mean = a.mean_aggr;
        const uint32_t *rowptr = a.rowptr;
        const int32_t *colidx = a.col;
        if (part == 1) {   // the parked pre-activation of this tile may itself be the NaN: redo the root pass's product first
            WCat wr = w;
            wr.W[0] = ly.W_root;
            split_exact_tile(CHAIN_NCB, wr, 0, dout, h, [=](int c) { return xr[c]; },
                             [=](int colq, float sv) { if (row_ok) out_row[colq] = sv; });
        }
        split_exact_tile(CHAIN_NCB, w, 0, dout, h,
                         [=](int c) {
                             if (c < kroot) return xr[c];
                             const int cc = c - kroot;
                             const uint32_t beg = rowptr[rowc], end = rowptr[rowc + 1];
                             float s = 0.0f;
                             for (uint32_t p = beg; p < end; ++p) s = s + src[(int64_t)colidx[p] * src_ld + cc];
                             if (mean && end > beg) s = 0.0f + s / (float)(end - beg);
                             return s;
                         },
                         [=](int colq, float sv) {
                             float v = (part == 0 ? 0.0f : out_row[colq]) + sv;
                             if (final_part) {
                                 v = v + (bias ? bias[colq] : 0.0f);
                                 if (act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
                             }
                             if (row_ok) out_row[colq] = v;
                         });
        if (final_part && last_layer && h == 0 && row_ok) {
            for (int o = 0; o < a.nout; ++o) {
                float s = 0.0f;
                for (int c = 0; c < dout; ++c) s = fmaf(head[o * CHAIN_DP + c], out_row[c], s);
                a.z[(int64_t)row * a.nout + o] = s;
            }
        }
    }
}

template <class P, bool FULLCOLS>
__device__ __forceinline__ void chain_pass(const ChainArgs &a, const ChainLayer &ly, const WCat &w, int part, bool last_layer,
                                           const float *src, int src_ld, float *hdst, int dst_ld, u32x4 *img, float4 *bias4,
                                           const float *head, ChainCtl *ctl, int phase, int tid, int nthreads) {
    constexpr int DP = CHAIN_DP;
    const int kcat = w.K[0] + w.K[1];
    const int nkb = split_nkb(kcat);
    __syncthreads();   // the previous pass is complete: its image is free, its rows are written
    split_fill_image(img, nkb, DP, w, 0, ly.Dout, tid, nthreads);
    split_fill_bias(bias4, DP, ly.bias, 0, ly.Dout, tid, nthreads);
    __syncthreads();
    const int lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int r0 = ctl->r0, r1 = ctl->r1;
    const int ntile = (r1 - r0 + 31) >> 5;
    SplitA cur = split_read_a(img + h * DP + n, nkb * 2 * DP);
    bool any_bad = false;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&ctl->ticket[phase], 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntile) break;
        chain_tile<P, FULLCOLS>(a, ly, part, last_layer, src, src_ld, hdst, dst_ld, img, bias4, head, r0, r1, t, n, h, cur, any_bad);
    }
    if (any_bad && lane == 0) ctl->bad[phase] = 1;
    __syncthreads();
    if (ctl->bad[phase])
        chain_redo(a, ly, w, part, last_layer, P::ROOT, src, src_ld, hdst, dst_ld, head, r0, r1, tid >> 6, nthreads >> 6, lane);
}

template <bool FULLCOLS>
__global__ void __launch_bounds__(CHAIN_THREADS) graph_chain_kernel(const ChainArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int DP = CHAIN_DP;
    // LDS: [control 128 B][head 8 x 128 floats][bias 128 floats][image]
    ChainCtl *ctl = reinterpret_cast<ChainCtl *>(lds_raw);
    float *head = reinterpret_cast<float *>(lds_raw + 128);
    float4 *bias4 = reinterpret_cast<float4 *>(head + 8 * DP);
    u32x4 *img = reinterpret_cast<u32x4 *>(bias4 + DP / 4);
    const int tid = threadIdx.x, nthreads = a.waves * 64;
    if (tid == 0) {
        // this block's member graphs: those whose first row falls in [b N / B, (b + 1) N / B)
        auto first_graph = [&](int64_t target) {
            int lo = 0, hi = a.G;       // first g in [0, G] with seg_ptr[g] >= target
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (a.seg_ptr[mid] < target) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        const int64_t B = gridDim.x, b = blockIdx.x;
        const int g0 = b == 0 ? 0 : first_graph(b * (int64_t)a.N / B);
        const int g1 = b + 1 == B ? a.G : first_graph((b + 1) * (int64_t)a.N / B);
        ctl->g0 = g0; ctl->g1 = g1;
        ctl->r0 = (int)a.seg_ptr[g0]; ctl->r1 = (int)a.seg_ptr[g1];
        for (int i = 0; i < CHAIN_MAX_PHASES; ++i) ctl->ticket[i] = ctl->bad[i] = 0;
    }
    const int dl = a.L[a.n_layers - 1].Dout;
    for (int i = tid; i < 8 * DP; i += nthreads) {
        const int o = i / DP, c = i - o * DP;
        head[i] = (o < a.nout && c < dl) ? a.W_head[(int64_t)o * a.hsj + (int64_t)c * a.hsk] : 0.0f;
    }
    __syncthreads();
    if (ctl->r1 > ctl->r0) {   // (block-uniform: a block without rows only takes part in nothing)
        int phase = 0;
        for (int l = 0; l < a.n_layers; ++l) {
            const ChainLayer &ly = a.L[l];
            const float *src = l == 0 ? a.x : a.h[(l - 1) & 1];
            float *hdst = a.h[l & 1];
            const int src_ld = l == 0 ? ly.Din : a.ldh_buf[(l - 1) & 1], dst_ld = a.ldh_buf[l & 1];
            const bool last = l + 1 == a.n_layers;
            WCat w;
            w.W[0] = ly.W_root; w.sj[0] = ly.sj; w.sk[0] = ly.sk; w.K[0] = ly.Din;
            w.W[1] = ly.W_agg; w.sj[1] = ly.sj; w.sk[1] = ly.sk; w.K[1] = ly.Din;
            if (ly.parts == 1) {
                if (ly.Din == 16)
                    chain_pass<Pass<16, true, true>, FULLCOLS>(a, ly, w, 0, last, src, src_ld, hdst, dst_ld, img, bias4, head, ctl, phase++, tid, nthreads);
                else
                    chain_pass<Pass<0, true, true>, FULLCOLS>(a, ly, w, 0, last, src, src_ld, hdst, dst_ld, img, bias4, head, ctl, phase++, tid, nthreads);
            } else {
                WCat wr = w, wa = w;
                wr.K[1] = 0;
                wa.W[0] = ly.W_agg; wa.K[1] = 0;
                if (ly.Din == 128) {
                    chain_pass<Pass<128, true, false>, FULLCOLS>(a, ly, wr, 0, last, src, src_ld, hdst, dst_ld, img, bias4, head, ctl, phase++, tid, nthreads);
                    chain_pass<Pass<128, false, true>, FULLCOLS>(a, ly, wa, 1, last, src, src_ld, hdst, dst_ld, img, bias4, head, ctl, phase++, tid, nthreads);
                } else {
                    chain_pass<Pass<0, true, false>, FULLCOLS>(a, ly, wr, 0, last, src, src_ld, hdst, dst_ld, img, bias4, head, ctl, phase++, tid, nthreads);
                    chain_pass<Pass<0, false, true>, FULLCOLS>(a, ly, wa, 1, last, src, src_ld, hdst, dst_ld, img, bias4, head, ctl, phase++, tid, nthreads);
                }
            }
        }
    }
    __syncthreads();   // every z of the block's rows is written
    // GlobalPool + the head's bias: reduce_nodes(aggr, g, x) = scatter(aggr, x, graph_indicator) in node order (utils.jl:12-16)
    const int g0 = ctl->g0, ng = ctl->g1 - ctl->g0;
    for (int i = tid; i < ng * a.nout; i += nthreads) {
        const int g = g0 + i / a.nout, o = i % a.nout;
        const int64_t rb = a.seg_ptr[g], re = a.seg_ptr[g + 1];
        float s = 0.0f;
        for (int64_t r = rb; r < re; ++r) s = s + a.z[r * a.nout + o];
        if (a.pool_mean && re > rb) s = 0.0f + s / (float)(re - rb);
        a.out[(int64_t)g * a.nout + o] = s + (a.b_head ? a.b_head[o] : 0.0f);
    }
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" int64_t gnnmp_graphconv_chain_scratch_floats(int64_t N, int n_layers, const int64_t *dims, int64_t nout) {
    if (N < 0 || n_layers < 1 || n_layers > CHAIN_MAX_LAYERS || !dims || nout < 1) return -1;
    int64_t d0 = 0, d1 = 0;     // widths of the layers written to h[0] / h[1]
    for (int l = 0; l < n_layers; ++l) {
        int64_t &d = (l & 1) ? d1 : d0;
        d = std::max(d, dims[l + 1]);
    }
    return N * (d0 + d1 + nout) + 16;
}

namespace gnnmp {
int graph_chain2_try(gnnmp_graph_t *p, const gnnmp_chain_jobs_t *J, const int64_t *seg_ptr, int64_t G, const float *x, int n_layers,
                     const int64_t *dims, const float *const *W_root, const float *const *W_agg, const float *const *bias,
                     const int *act, int w_layout, int aggr, int pool_aggr, const float *W_head, const float *b_head, int64_t nout,
                     float *out, hipStream_t stream);   // graph_chain2.hip
}

extern "C" int gnnmp_graphconv_chain_f32(gnnmp_graph_t *p, const gnnmp_chain_jobs_t *jobs, const int64_t *seg_ptr, int64_t G, const float *x, int n_layers,
                                         const int64_t *dims, const float *const *W_root, const float *const *W_agg,
                                         const float *const *bias, const int *act, int w_layout, int aggr, int pool_aggr,
                                         const float *W_head,
                                         const float *b_head, int64_t nout, float *scratch, float *out, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "graphconv_chain: null plan");
    if (n_layers < 1 || !dims || !W_root || !W_agg || !act) return fail(GNNMP_EINVAL, "graphconv_chain: bad layer list");
    if (G < 0 || nout < 1) return fail(GNNMP_EINVAL, "graphconv_chain: bad size");
    if (w_layout != 0 && w_layout != 1) return fail(GNNMP_EINVAL, "graphconv_chain: bad w_layout %d", w_layout);
    if (aggr < GNNMP_SUM || aggr > GNNMP_MIN || pool_aggr < GNNMP_SUM || pool_aggr > GNNMP_MIN)
        return fail(GNNMP_EINVAL, "graphconv_chain: bad aggr");
    if (p->n_src != p->n_dst) return fail(GNNMP_EINVAL, "graphconv_chain: the plan is not a square graph");
    if (G == 0) return GNNMP_OK;
    if (!seg_ptr || !x || !W_head || !scratch || !out) return fail(GNNMP_EINVAL, "graphconv_chain: null pointer");
    // the envelope of the fused kernel; outside it the caller runs the chain layer by layer
    if (knob(KNOB_CHAIN) < 0 || n_layers > CHAIN_MAX_LAYERS || nout > 8 || aggr > GNNMP_MEAN || pool_aggr > GNNMP_MEAN ||
        p->n_dst >= (int64_t)1 << 31 || p->n_dst < 32)
        return fail(GNNMP_EUNSUPPORTED, "graphconv_chain: outside the fused kernel's envelope");
    for (int l = 0; l < n_layers; ++l)
        if (!W_root[l] || !W_agg[l] || (act[l] != GNNMP_ACT_IDENTITY && act[l] != GNNMP_ACT_RELU))
            return fail(GNNMP_EUNSUPPORTED, "graphconv_chain: layer %d outside the fused kernels' envelope", l);
    {
        // two layers 16 => 128 => 128 on member graphs of at most 64 nodes: the wave-per-graph kernel, nothing through memory
        const int rc = graph_chain2_try(p, jobs, seg_ptr, G, x, n_layers, dims, W_root, W_agg, bias, act, w_layout, aggr, pool_aggr, W_head,
                                        b_head, nout, out, stream);
        if (rc != 1) return rc;
    }
    ChainArgs a = {};
    a.rowptr = p->rowptr;
    a.col = p->col;
    a.seg_ptr = seg_ptr;
    a.G = (int)G;
    a.N = (int)p->n_dst;
    a.x = x;
    a.n_layers = n_layers;
    a.mean_aggr = aggr == GNNMP_MEAN;
    a.pool_mean = pool_aggr == GNNMP_MEAN;
    size_t img_max = 0;
    int64_t d0 = 0, d1 = 0;
    bool fullcols = true;
    for (int l = 0; l < n_layers; ++l) {
        const int64_t din = dims[l], dout = dims[l + 1];
        // (a STORED layer — every layer but the last — narrower than 8 columns: split_store_all re-stores the lane's first piece at
        // column 4 h for pieces past ncols, which lies outside a 4-column row)
        if (din < 4 || (din & 3) || dout < 4 || (dout & 3) || dout > CHAIN_DP || din > 512 || !W_root[l] || !W_agg[l] ||
            (l + 1 < n_layers && dout < 8) || (act[l] != GNNMP_ACT_IDENTITY && act[l] != GNNMP_ACT_RELU))
            return fail(GNNMP_EUNSUPPORTED, "graphconv_chain: layer %d (%lld => %lld) outside the fused kernel's envelope", l,
                        (long long)din, (long long)dout);
        ChainLayer &ly = a.L[l];
        ly.W_root = W_root[l]; ly.W_agg = W_agg[l]; ly.bias = bias ? bias[l] : nullptr;
        ly.sj = w_layout ? 1 : din;
        ly.sk = w_layout ? dout : 1;
        ly.Din = (int)din; ly.Dout = (int)dout; ly.act = act[l];
        const size_t budget = 160 * 1024 - 128 - 8 * CHAIN_DP * 4 - CHAIN_DP * 4 - 256;
        ly.parts = ((din & 15) == 0 && split_img_bytes((int)(2 * din), CHAIN_DP) <= budget) ? 1 : 2;
        const size_t need = split_img_bytes((int)(ly.parts == 1 ? 2 * din : din), CHAIN_DP);
        if (need > budget) return fail(GNNMP_EUNSUPPORTED, "graphconv_chain: layer %d's weight planes do not fit LDS", l);
        img_max = std::max(img_max, need);
        int64_t &d = (l & 1) ? d1 : d0;
        d = std::max(d, dout);
        fullcols = fullcols && dout == CHAIN_DP;
    }
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(scratch) & 15))
        return fail(GNNMP_EUNSUPPORTED, "graphconv_chain: unaligned feature matrix");
    a.W_head = W_head;
    a.hsj = w_layout ? 1 : dims[n_layers];
    a.hsk = w_layout ? nout : 1;
    a.b_head = b_head;
    a.nout = (int)nout;
    a.h[0] = scratch;
    a.ldh_buf[0] = (int)d0;
    a.ldh_buf[1] = (int)d1;
    a.h[1] = scratch + (((int64_t)a.N * d0 + 3) & ~(int64_t)3);
    a.z = a.h[1] + (((int64_t)a.N * d1 + 3) & ~(int64_t)3);
    a.out = out;
    const int cus = device_cus();
    // one block per CU; fewer when the batch is small (at least ~4 tiles of 32 rows per block)
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cus, a.N / 128));
    a.waves = CHAIN_THREADS / 64;
    const size_t lds = 128 + 8 * CHAIN_DP * 4 + CHAIN_DP * 4 + img_max;
    GNNMP_LDS_OPTIN("graph_chain_kernel", &graph_chain_kernel<true>);
    GNNMP_LDS_OPTIN("graph_chain_kernel", &graph_chain_kernel<false>);
    if (fullcols)
        graph_chain_kernel<true><<<blocks, CHAIN_THREADS, lds, stream>>>(a);
    else
        graph_chain_kernel<false><<<blocks, CHAIN_THREADS, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("graph_chain_kernel");
    return GNNMP_OK;
}
