// dense_split.hip — the dense feature contractions of the layer bodies on the split-bf16 core of msplit.h:
//   weight * x                     gcn_conv   GNNlib/src/layers/conv.jl:36-40,70
//   W1 * xi .+ W2 * m              graph_conv conv.jl:106
//   W * vcat(xi, m)                sage_conv  conv.jl:281   (the vcat is the CONCATENATED contraction index, never built)
//   dense_x(x)                     gat_conv   conv.jl:136
// One persistent block per CU; the three bf16 planes of W^T (one column tile of DP = 32 NCB outputs, both segments) live in LDS;
// each WAVE owns 32-node tiles: per 16-position k-block two 16-byte loads per lane straight from the row-major feature
// matrices (a ring of PF k-blocks in flight, running ahead into the next tile), ~45 VALU to split them into planes, and
// 6 NCB v_mfma_f32_32x32x16_bf16 whose A operands come from the image by ds_read_b128 one column block ahead; bias +
// activation on the accumulators, one 16-byte store per four outputs.  No barrier after the image.
// K of a segment: any multiple of 4 (rows are 16-byte aligned); the image must fit LDS: (K1 + K2 rounded to 16) * DP * 6 bytes
// <= ~158 KB — DP = 128 up to K1 + K2 = 208 (SAGEConv 100 + 100), 64 up to 416, 32 up to 832; beyond that dense.hip's kernels.
#include <algorithm>

#include "msplit.h"

namespace gnnmp {

struct SplitArgs {
    const float *x[2];
    WCat w;
    const float *bias;
    int act;
    float *out;
    int64_t N;
    int Dout;
    int waves;
    int nkb;                // k-blocks of 16 positions of the concatenated contraction index
};

// this lane's two operand loads of k-block kb: positions c = 16 kb + 4 h + 8 u of the concatenated row.  Positions past the end
// re-read the row's last float4 (the image holds zeros there: finite garbage contributes nothing, and a non-finite value sends
// the tile to the exact path, where it belongs anyway) — no select on loaded values, no guarded load.
template <bool TWO>
struct RowPtr {
    const float *p0, *p1;   // p0 + c indexes segment 1; p1 + c (c >= K0) indexes segment 2
};
template <bool TWO>
__device__ __forceinline__ RowPtr<TWO> row_ptr(const SplitArgs &a, int64_t row) {
    RowPtr<TWO> r;
    r.p0 = a.x[0] + row * a.w.K[0];
    r.p1 = TWO ? a.x[1] + row * a.w.K[1] - a.w.K[0] : nullptr;
    return r;
}
// kb, u compile-time at the unrolled call sites, h in {0, 1}: with K0 and kcat known at compile time too (K0C > 0) the clamp and the
// segment choice fold away everywhere except in the one k-block that straddles the two segments
template <bool TWO>
__device__ __forceinline__ float4 load_q(const RowPtr<TWO> &r, int c, int K0, int kcat) {
    const int ce = min(c, kcat - 4);
    const float *p = r.p0;
    if (TWO) p = ce < K0 ? r.p0 : r.p1;
    return *reinterpret_cast<const float4 *>(p + ce);
}

// VAR: experiment variants, compile-time only (a run-time switch inside the tile loop costs registers, and a spilled register
// reloaded in the loop is a scratch load = an s_waitcnt vmcnt(0) = waiting for every store in flight).  Production = 0.
//   1 = ring slots of ONE k-block (default: pairs of k-blocks = one 128-byte line of each row, loaded in a burst of four instructions)
//   2 = no peeled first tile          4 = 768 / 1024-thread blocks (default 512: 256 VGPRs)      8 = predicated stores (first version)
//   16 = no stores at all             32 = no split arithmetic (wrong numbers)                    64 = no A-operand reads
//   128 = every tile reads rows 0..31 (operands from L1)
//   256 = (not an experiment) every column tile is full, Dout % DP == 0: no column test in the epilogue
//   4096 = (not an experiment) stores through a 4 KB LDS stage per wave (whole 128-byte lines): taken when the stage fits beside the image
template <int NCB, int VAR>
constexpr int split_threads() { return (VAR & 4) ? (NCB >= 4 ? 768 : 1024) : 512; }

// K0C > 0: both segment lengths known at compile time (the shapes of the configs): k-blocks fully unrolled, a ring of operand loads
// in flight that runs ahead into the next tile.  K0C = 0: any K (a.nkb k-blocks, K1C > 0 = two segments), one k-block ahead.
template <int NCB, int K0C, int K1C, int VAR>
__global__ void __launch_bounds__((split_threads<NCB, VAR>())) dense_split_kernel(const SplitArgs a) {
    constexpr int NKB = K0C > 0 ? (K0C + K1C + 15) / 16 : 0;
    constexpr bool TWO = K1C > 0;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int DP = NCB * 32;
    constexpr bool RT = NKB <= 0;
    constexpr bool PEEL = !RT && !(VAR & 2);
    // ring slot = G k-blocks (G = 2: the four loads of a slot cover one 128-byte line of each of the tile's rows and are issued back to
    // back)
    constexpr int G = (RT || (VAR & 1)) ? 1 : 2;
    constexpr int NSL = RT ? 1 : (NKB + G - 1) / G;
    // ring depth in pair slots: 3 (measured at 2.4 M x 100 => 128: 2 slots 642 us, 3 slots 560 us, the whole tile 612 us); experiment
    // bits 1024 = 2 slots, 2048 = the whole tile
    constexpr int PFW = (VAR & 2048) ? NSL : ((VAR & 1024) ? 2 : 3);
    constexpr int PF = RT ? 1 : (G == 2 ? (NSL < PFW ? NSL : PFW) : (NSL < 4 ? NSL : 4));
    const int nkb = RT ? a.nkb : NKB;
    const int units = nkb * 2 * DP;
    u32x4 *img = reinterpret_cast<u32x4 *>(lds_raw);
    float4 *bias4 = reinterpret_cast<float4 *>(img + 3 * units);
    unsigned char *stages = reinterpret_cast<unsigned char *>(bias4 + DP / 4);     // (VAR & 4096) [waves][4096]
    const int tid = threadIdx.x, nthreads = a.waves * 64;
    const int n0 = (int)blockIdx.y * DP;
    const int ncols = min(DP, a.Dout - n0);
    split_fill_image(img, nkb, DP, a.w, n0, ncols, tid, nthreads);
    split_fill_bias(bias4, DP, a.bias, n0, ncols, tid, nthreads);
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int K0 = RT ? a.w.K[0] : K0C, kcat = RT ? a.w.K[0] + a.w.K[1] : K0C + K1C;
    const int ntiles = (int)((a.N + 31) >> 5);
    const int stride = (int)gridDim.x * a.waves;
    const int nlast = (int)(a.N - 1);
    // wave-major hand-out: round r gives tile r * stride + wave * gridDim.x + block (the partial last round spreads over every CU)
    int tile = wave * (int)gridDim.x + (int)blockIdx.x;
    if (tile >= ntiles) return;
    auto row_of = [&](int t) { return (VAR & 128) ? n : min(t * 32 + n, nlast); };
    RowPtr<TWO> rp = row_ptr<TWO>(a, row_of(tile));
    float4 ring[PF][2 * G];
    // the loads of ring slot `sl` of the tile whose rows `src` points at
    auto load_slot = [&](float4 (&dst)[2 * G], const RowPtr<TWO> &src, int sl) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int kb = sl * G + g;
            if (RT || kb < NKB) {
                dst[2 * g] = load_q<TWO>(src, 16 * kb + 4 * h, K0, kcat);
                dst[2 * g + 1] = load_q<TWO>(src, 16 * kb + 4 * h + 8, K0, kcat);
            }
        }
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) load_slot(ring[j], rp, j);
    SplitA cur = split_read_a(img + h * DP + n, units);
    bool any_bad = false;
    auto mac_kblock = [&](f32x16 (&acc)[NCB], int kb, int kb_next, const float4 q0, const float4 q1) {
        Split8 b = split8(q0, q1);
        if constexpr (VAR & 32) {
            b.p0 = u32x4{__float_as_uint(q0.x), __float_as_uint(q0.y), __float_as_uint(q0.z), __float_as_uint(q0.w)};
            b.p1 = u32x4{__float_as_uint(q1.x), __float_as_uint(q1.y), __float_as_uint(q1.z), __float_as_uint(q1.w)};
            b.p2 = b.p0;
        }
        if constexpr (VAR & 64) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = split_mac(acc[cb], cur, b);
        } else {
            split_kblock<NCB>(acc, img, units, DP, kb, kb_next, n, h, b, cur);
        }
    };
    // One tile.  The FIRST tile of a wave is a peeled copy of the loop body (PEEL): gfx9 has ONE counter (vmcnt) for loads and stores,
    // and hipcc's wait insertion merges the loop header's predecessors conservatively — the preheader holds only the ring's loads,
    // the back edge the ring's loads followed by 16 stores; with the peeled copy both end in the same sequence.
    auto do_tile = [&](int t) {
        const int row = t * 32 + n;
        const RowPtr<TWO> rpn = row_ptr<TWO>(a, row_of(min(t + stride, ntiles - 1)));
        f32x16 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
        if constexpr (RT) {
            for (int kb = 0; kb < nkb; ++kb) {
                const float4 q0 = ring[0][0], q1 = ring[0][1];
                const bool last = kb + 1 == nkb;
                const int kt = last ? 0 : kb + 1;
                load_slot(ring[0], last ? rpn : rp, kt);
                mac_kblock(acc, kb, kt, q0, q1);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
                float4 q[2 * G];
#pragma unroll
                for (int i = 0; i < 2 * G; ++i) q[i] = ring[sl % PF][i];
                // refill the slot with slot sl + PF of this tile, or the matching one of the next
                const int T = sl + PF;
                load_slot(ring[sl % PF], T < NSL ? rp : rpn, T < NSL ? T : T - NSL);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int kb = sl * G + g;
                    if (kb < NKB) mac_kblock(acc, kb, kb + 1 < NKB ? kb + 1 : 0, q[2 * g], q[2 * g + 1]);
                }
            }
            // ring slot j now holds the next tile's slot (j - NSL) mod PF: put slot j back into ring slot j
            if constexpr (NSL % PF != 0) {
                float4 tq[PF][2 * G];
#pragma unroll
                for (int j = 0; j < PF; ++j)
#pragma unroll
                    for (int i = 0; i < 2 * G; ++i) tq[j][i] = ring[j][i];
#pragma unroll
                for (int j = 0; j < PF; ++j)
#pragma unroll
                    for (int i = 0; i < 2 * G; ++i) ring[j][i] = tq[(j + NSL) % PF][i];
            }
        }
        float *out_row = a.out + (int64_t)min(row, nlast) * a.Dout + n0;
        // A tile that met a non-finite operand has NaN accumulators (msplit.h).  It is stored like any other and only REMEMBERED here:
        // the exact recomputation is a call, and a call inside this loop makes hipcc size the loop's vmcnt waits for the path through
        // it — on which no store follows the ring's loads — so that every tile would start by waiting for the previous tile's 16 stores
        // to complete (vmcnt(11) instead of vmcnt(27): measured 240 us of 610 at 2.4 M x 100 => 128).  Bad tiles are redone below.
        if constexpr (!(VAR & 512)) any_bad |= split_any_nan<NCB>(acc);
        if constexpr (VAR & 16) {
        } else if constexpr (VAR & 8) {
            split_store<NCB>(acc, bias4, a.act, out_row, row <= nlast, ncols, h);
        } else if constexpr (VAR & 4096) {
            split_store_rows<NCB, (VAR & 256) != 0>(acc, bias4, a.act, stages + wave * 4096, a.out + n0, a.Dout, t * 32, nlast, ncols, lane);
        } else {
            split_store_all<NCB, (VAR & 256) != 0>(acc, bias4, a.act, out_row, ncols, h);
        }
        rp = rpn;
    };
    const int tile0 = tile;
    if constexpr (PEEL) {
        do_tile(tile);
        tile += stride;
    }
    for (; tile < ntiles; tile += stride) do_tile(tile);
    if (!any_bad) return;
    // Second pass (only waves that saw a NaN accumulator): the tiles whose STORED rows hold a NaN are recomputed with plain fp32 fma
    // loops — Inf * w = +-Inf, the -Inf of an empty max-aggregation, NaN propagation: the reference product's behaviour.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's stores have landed
    for (int t = tile0; t < ntiles; t += stride) {
        const int row = t * 32 + n;
        const int rowc = min(row, nlast);
        float *out_row = a.out + (int64_t)rowc * a.Dout + n0;
        bool bad = false;
        for (int c = 4 * h; c < ncols; c += 8) {
            const float4 v = *reinterpret_cast<const float4 *>(out_row + c);
            bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
        }
        if (__builtin_amdgcn_ballot_w64(bad) == 0) continue;
        const RowPtr<TWO> rq = row_ptr<TWO>(a, (VAR & 128) ? n : rowc);
        const float *bias = a.bias;
        const int act = a.act;
        const bool row_ok = row <= nlast;
        split_exact_tile(NCB, a.w, n0, ncols, h,
                         [=](int c) { return c < K0 ? rq.p0[c] : rq.p1[c]; },
                         [=](int col, float sv) {
                             float v = sv + (bias ? bias[n0 + col] : 0.0f);
                             if (act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
                             if (row_ok) out_row[col] = v;
                         });
    }
}

template <int NCB, int K0C, int K1C, int VAR>
static int launch_split_var(const SplitArgs &a0, hipStream_t stream) {
    SplitArgs a = a0;
    constexpr int DP = NCB * 32;
    const size_t lds = split_img_bytes(a.nkb * 16, DP) + (size_t)DP * 4 + ((VAR & 4096) ? (size_t)(split_threads<NCB, VAR>() / 64) * 4096 : 0);
    GNNMP_LDS_OPTIN("dense_split_kernel", &dense_split_kernel<NCB, K0C, K1C, VAR>);
    const int cus = device_cus();
    const int64_t ntiles = (a.N + 31) / 32;
    constexpr int max_waves = split_threads<NCB, VAR>() / 64;
    int waves = (int)std::min<int64_t>(max_waves, std::max<int64_t>(4, (ntiles + cus - 1) / cus));
    if (ntiles < (int64_t)cus * max_waves * 8) {
        // few tiles per wave (arxiv shape: 5 292 tiles, 2.6 per wave at 8 waves a block): the last round of the wave-major hand-out is
        // partly empty — pick the wave count whose rounds are fullest (arxiv: 7 waves -> 2.95 tiles per wave, 98 % instead of 86 %)
        double best = -1.0;
        for (int w = max_waves; w >= 4; --w) {
            const int64_t slots = (int64_t)cus * w;
            const double eff = (double)ntiles / (double)(((ntiles + slots - 1) / slots) * slots);
            if (eff > best + 0.02) { best = eff; waves = w; }
        }
    }
    const int kw = knob(KNOB_DENSE_T16_WAVES);
    if (kw >= 1 && kw <= max_waves) waves = kw;
    a.waves = waves;
    // Several column tiles (Dout > DP: SAGEConv's 256 columns are two): a block fills a CU (LDS), so with `cus` blocks per column tile
    // the tiles ran one after the other and x came from HBM once per column tile.  cus / ny blocks per column tile instead: blocks
    // (b, 0), (b, 1), ... walk the same row tiles at the same time and — linear block ids b, b + gx, ... with gx a multiple of 8 — on
    // the same XCD, so every read of x after the first is an L2 hit (knob 19 bit 4 = the old grid, for A/B runs).
    const int ny = (a.Dout + DP - 1) / DP;
    int64_t bx = cus;
    if (ny > 1 && !(knob(KNOB_VARIANT) & 16)) bx = std::max<int64_t>(8, (int64_t)(cus / ny) & ~(int64_t)7);
    const int64_t gx = std::min<int64_t>(bx, (ntiles + waves - 1) / waves);
    dim3 grid((unsigned)gx, (unsigned)ny);
    dense_split_kernel<NCB, K0C, K1C, VAR><<<grid, 64 * waves, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("dense_split_kernel");
    return GNNMP_OK;
}
template <int NCB, int K0C, int K1C>
static int launch_split(const SplitArgs &a, hipStream_t stream) {
#ifdef GNNMP_SPLIT_EXPERIMENTS
    if constexpr ((K0C == 100 && K1C == 0) || (K0C == 128 && K1C == 128)) {   // knob 13 selects a variant, on two shapes only (build time)
        switch (knob(KNOB_T16_DEBUG)) {
#define V(X) case X: return launch_split_var<NCB, K0C, K1C, X>(a, stream);
            V(257) V(258) V(260) V(1280) V(2304) V(264) V(272) V(320) V(384) V(400) V(768)
#undef V
            default: break;
        }
    }
#endif
    // stores through the per-wave LDS stage (whole 128-byte lines) when 8 stages fit beside the image (knob 19 bit 5 = never, for A/B runs)
    const bool staged = split_img_bytes(a.nkb * 16, NCB * 32) + (size_t)NCB * 32 * 4 + 8 * 4096 <= 160 * 1024 && !(knob(KNOB_VARIANT) & 32);
    if (a.Dout % (NCB * 32) == 0)
        return staged ? launch_split_var<NCB, K0C, K1C, 256 | 4096>(a, stream) : launch_split_var<NCB, K0C, K1C, 256>(a, stream);
    return staged ? launch_split_var<NCB, K0C, K1C, 4096>(a, stream) : launch_split_var<NCB, K0C, K1C, 0>(a, stream);
}

int dense_wreg_try(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2, const float *W2, int64_t D2, int64_t ldw2,
                   int w_layout, const float *bias, int act, float *out, int64_t N, int64_t Dout, hipStream_t stream);   // dense_wreg.hip

// Returns GNNMP_OK if it launched, 1 if the shape is not one this kernel takes (the caller falls back to the fp32-MFMA kernels).
int dense_split_try(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2, const float *W2, int64_t D2,
                    int64_t ldw2, int w_layout, const float *bias, int act, float *out, int64_t N, int64_t Dout,
                    hipStream_t stream) {
    if (knob(KNOB_DENSE_GENERIC) != 0 || knob(KNOB_DENSE_SPLIT) < 0) return 1;
    const bool two = D2 > 0;
    if ((D1 & 3) || (D2 & 3) || (Dout & 3) || Dout < 4 || N < 32) return 1;
    {
        // 256 outputs (SAGEConv(100 => 256)): W in registers, x through LDS once (dense_wreg.hip)
        const int rc = dense_wreg_try(x1, W1, D1, ldw1, x2, W2, D2, ldw2, w_layout, bias, act, out, N, Dout, stream);
        if (rc != 1) return rc;
    }
    // column blocks are 32 wide: a Dout that pads by more than a tenth (100 -> 128) costs more MFMA work and a select per stored piece
    // than the fp32 16x16x4 kernel's 16-wide blocks (measured 2.4 M x 100 => 100: 625 us here, 549 us there)
    if (((Dout + 31) & ~(int64_t)31) * 10 > Dout * 11) return 1;
    if ((reinterpret_cast<uintptr_t>(x1) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return 1;
    if (two && (reinterpret_cast<uintptr_t>(x2) & 15)) return 1;
    const int kcat = (int)(D1 + D2), nkb = split_nkb(kcat);
    SplitArgs a;
    a.x[0] = x1; a.x[1] = x2;
    a.w.W[0] = W1; a.w.W[1] = two ? W2 : W1;
    a.w.K[0] = (int)D1; a.w.K[1] = (int)D2;
    a.w.sj[0] = w_layout == 0 ? ldw1 : 1; a.w.sk[0] = w_layout == 0 ? 1 : ldw1;
    a.w.sj[1] = w_layout == 0 ? ldw2 : 1; a.w.sk[1] = w_layout == 0 ? 1 : ldw2;
    a.bias = bias;
    a.act = act;
    a.out = out;
    a.N = N;
    a.Dout = (int)Dout;
    a.waves = 12;
    a.nkb = nkb;
    // the widest column tile whose image fits: fewer passes over x.  K known at compile time for the shapes of the configs.
    const size_t budget = 160 * 1024 - 1024;
    if (Dout > 64 && split_img_bytes(kcat, 128) <= budget) {
        if (two) {
            if (D1 == 16 && D2 == 16) return launch_split<4, 16, 16>(a, stream);     // GraphConv 16 + 16 => 128
            if (D1 == 100 && D2 == 100) return launch_split<4, 100, 100>(a, stream);   // SAGEConv 100 + 100 => 256 (two column tiles)
            return launch_split<4, 0, 1>(a, stream);
        }
        if (D1 == 100) return launch_split<4, 100, 0>(a, stream);        // 100 => 100 | 128 (GCNConv, GATConv dense_x: products)
        if (D1 == 128) return launch_split<4, 128, 0>(a, stream);        // 128 => 128 (arxiv)
        return launch_split<4, 0, 0>(a, stream);
    }
    if (split_img_bytes(kcat, 64) <= budget) {
        if (two) {
            if (D1 == 128 && D2 == 128) return launch_split<2, 128, 128>(a, stream);   // GraphConv 128 + 128 => 128 (two column tiles)
            return launch_split<2, 0, 1>(a, stream);
        }
        return launch_split<2, 0, 0>(a, stream);
    }
    return 1;
}

}  // namespace gnnmp
