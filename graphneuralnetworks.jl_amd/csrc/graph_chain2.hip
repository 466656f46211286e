// graph_chain2.hip — BASELINE.json config 5 with NOTHING but the features, the edge index and the logits crossing the chip's memory:
//   GNNChain(GraphConv(16 => 128, σ1), GraphConv(128 => 128, σ2), GlobalPool(mean | +), Dense(128 => nout))
//   (examples/graph_classification_tudataset.jl:79-82; layer body GNNlib/src/layers/conv.jl:102-108; pooling layers/pool.jl:3-5)
// graph_chain.hip (the general kernel) keeps layer outputs in a block-local scratch matrix; at G = 8192 that scratch (32 blocks x ~1 MB
// per XCD) falls out of the 4 MB L2 and the kernel runs at the fabric's speed (0.215 ms).  Here a WAVE owns whole member graphs:
//   * a job = up to 64 rows = whole member graphs packed best-fit-decreasing on the host side of gnnmp_chain_jobs_create (94 % full
//     at n ~ U{20..40}); the wave runs it as two 32-row MFMA tiles;
//   * layer 1, 32 features at a time: 12 split-bf16 MFMAs per tile against the W1 planes in LDS; its B operand (x_i, and sum_j x_j
//     gathered from the 64-byte feature rows in original edge order) is formed once per job;
//   * those 32 features ARE the next layer's root operand: the 32x32 accumulator layout (lane = node, registers = 8 a + 4 h + b) is the
//     B-operand layout of msplit.h for k-blocks 2 cb and 2 cb + 1 — bias, σ1, split into planes, MFMA, no memory in between;
//   * the aggregate operand sum_j h1_j: the 16 features of a k-block go to a 4 KB wave-private LDS stage (64 rows x 64 bytes, 16-byte
//     slots XOR-swizzled by the row), every lane sums its row's in-neighbours from there in original edge order (member graphs are
//     whole inside a job: a neighbour's slot is this row's slot + the difference of the node ids) — the adds of NNlib.scatter(+);
//   * layer 2's 128 outputs are two 64-column slabs handled by DIFFERENT blocks (blockIdx.y): W1's planes (24 KB) + one slab of
//     [W_root | W_agg] (96 KB) + 8 stages fit LDS for the whole launch — no image is ever swapped, there is no block barrier after the
//     set-up and no block-level hand-off; each slab block recomputes the cheap layer 1 (48 of 240 MFMAs per tile);
//   * bias, σ2, z = W_head[:, slab] * h2 on the accumulators, per-graph sum over the job's slots in row order (through the stage),
//     mean: this slab's half of each logit, stored once; a one-block finish kernel adds the two halves and the head's bias (no
//     floating-point atomic anywhere, no memset: two launches per call).
// A job that meets a non-finite operand (NaN accumulators, msplit.h) is set aside and recomputed with plain fp32 loops by the finish
// kernel (a call or the inlined loops inside the main kernel cost its job loop ~100 spilled registers).
// Envelope: exactly two layers 16 => 128 => 128, nout <= 8, aggr and pool in {+, mean}, member graphs of at most 64 nodes; everything
// else is graph_chain.hip's.
#include <algorithm>
#include <vector>

#include "msplit.h"

struct gnnmp_chain_jobs {
    int32_t *rows = nullptr;   // [njobs][64] global row of each slot, -1 = empty
    int32_t *gid = nullptr;    // [njobs][64] member graph of each slot, -1 = empty
    int njobs = 0;
    int64_t G = 0, N = 0;
    int64_t max_graph = 0;     // largest member graph (> 64: no jobs, the general kernel runs)
    double fill = 0.0;         // rows / (32 x tiles): MFMA work spent on real rows
    int has_empty = 0;         // some member graph has no node (its logits are the head's bias: left to the general kernel)
    int32_t *bad = nullptr;    // [3 + 2 njobs] count of set-aside jobs (reset by the finish kernel), two spare words, then (job * 2 + slab) of
                               // the jobs that met a non-finite operand in the running call
    float *part = nullptr;     // [2][G][8] the two slabs' halves of every logit (each word written once per call: no atomics, no memset)
};

namespace gnnmp {

constexpr int C2_MAX_WAVES = 8;
constexpr int C2_D0 = 16, C2_D1 = 128, C2_D2 = 128, C2_SLAB = 64;
constexpr int C2_UNITS1 = 2 * 2 * C2_D1;          // 16-byte units per plane of the W1 image (K = 32)
constexpr int C2_UNITS2 = 16 * 2 * C2_SLAB;       // ... of one slab of [W2_root | W2_agg] (K = 256)
constexpr int C2_STAGE_BYTES = 4096;

struct Chain2Args {
    const uint32_t *rowptr;
    const int32_t *col;
    const int32_t *job_rows, *job_gid;
    int njobs;
    const int64_t *seg_ptr;
    const float *x;
    const float *W1r, *W1a, *b1, *W2r, *W2a, *b2, *Wh, *bh;
    int lay;            // 0: weights C row-major [Dout][Din]; 1: Julia (Dout, Din) column-major as stored, W(j, k) at [k * Dout + j]
    int nout, act1, act2, mean_aggr, pool_mean;
    float *out;
    float *part;        // [2][G][nout]
    int G;
    int32_t *bad;       // [0] = count, [3..] = job * 2 + slab
};

__device__ __forceinline__ float4 c2_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 c2_add4(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 c2_sel4(bool c, const float4 a, const float4 b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
__device__ __forceinline__ void c2_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// byte offset of 16-byte slot `pos` (0..3) of stage row `row`: rows 4 apart share banks, the XOR spreads them
__device__ __forceinline__ int c2_stage_off(int row, int pos) { return row * 64 + ((pos ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float c2_act(float v, int act) { return (act == GNNMP_ACT_RELU && v < 0.0f) ? 0.0f : v; }

// The exact path of a job: z of every slot's row with fp32 loops (this lane = slot `lane`).  Runs in graph_chain2_finish_kernel only: a call
// (or the inlined loops) inside the main kernel costs its tile loop ~100 spilled registers.
__device__ void chain2_exact_job(const Chain2Args &a, int slab, const int32_t *jr, int lane, float *zst) {
    const int row = jr[lane];
    if (row < 0) return;
    auto h1_of = [&](int r, float *h) {     // relu(W1r x_r + W1a aggr_k x_k + b1)
        float xr[C2_D0], xa[C2_D0];
        const uint32_t beg = a.rowptr[r], end = a.rowptr[r + 1];
        for (int c = 0; c < C2_D0; ++c) {
            xr[c] = a.x[(int64_t)r * C2_D0 + c];
            float s = 0.0f;
            for (uint32_t p = beg; p < end; ++p) s = s + a.x[(int64_t)a.col[p] * C2_D0 + c];
            if (a.mean_aggr && end > beg) s = 0.0f + s / (float)(end - beg);
            xa[c] = s;
        }
        for (int f = 0; f < C2_D1; ++f) {
            float s = 0.0f;
            for (int c = 0; c < C2_D0; ++c) s = fmaf(a.W1r[a.lay ? c * C2_D1 + f : f * C2_D0 + c], xr[c], s);
            for (int c = 0; c < C2_D0; ++c) s = fmaf(a.W1a[a.lay ? c * C2_D1 + f : f * C2_D0 + c], xa[c], s);
            h[f] = c2_act(s + (a.b1 ? a.b1[f] : 0.0f), a.act1);
        }
    };
    float hi[C2_D1], hs[C2_D1], hj[C2_D1];
    h1_of(row, hi);
    for (int f = 0; f < C2_D1; ++f) hs[f] = 0.0f;
    const uint32_t beg = a.rowptr[row], end = a.rowptr[row + 1];
    for (uint32_t p = beg; p < end; ++p) {
        h1_of(a.col[p], hj);
        for (int f = 0; f < C2_D1; ++f) hs[f] = hs[f] + hj[f];
    }
    if (a.mean_aggr && end > beg)
        for (int f = 0; f < C2_D1; ++f) hs[f] = 0.0f + hs[f] / (float)(end - beg);
    float z[8];
    for (int o = 0; o < 8; ++o) z[o] = 0.0f;
    for (int c = 0; c < C2_SLAB; ++c) {
        const int f2 = C2_SLAB * slab + c;
        float s = 0.0f;
        for (int f = 0; f < C2_D1; ++f) s = fmaf(a.W2r[a.lay ? f * C2_D2 + f2 : f2 * C2_D1 + f], hi[f], s);
        for (int f = 0; f < C2_D1; ++f) s = fmaf(a.W2a[a.lay ? f * C2_D2 + f2 : f2 * C2_D1 + f], hs[f], s);
        const float v = c2_act(s + (a.b2 ? a.b2[f2] : 0.0f), a.act2);
        for (int o = 0; o < a.nout; ++o) z[o] = fmaf(a.Wh[a.lay ? f2 * a.nout + o : o * C2_D2 + f2], v, z[o]);
    }
    for (int o = 0; o < a.nout; ++o) zst[lane * 8 + o] = z[o];
}

// per-graph pooling of a job's z rows (stage: zst [64][8], gst [64] = member graph of each slot): this slab's half of every logit
__device__ __forceinline__ void c2_pool(const Chain2Args &a, int slab, int gid, int lane, const float *zst, const int *gst) {
    if (gid >= 0 && (lane == 0 || gst[lane - 1] != gid)) {
        // reduce_nodes(aggr, g, x) = scatter(aggr, x, graph_indicator) in node order (utils.jl:12-16); Dense and + / mean commute
        const int cnt = (int)(a.seg_ptr[gid + 1] - a.seg_ptr[gid]);
        for (int o = 0; o < a.nout; ++o) {
            float s = 0.0f;
            for (int t = 0; t < cnt; ++t) s = s + zst[(lane + t) * 8 + o];
            if (a.pool_mean && cnt > 0) s = 0.0f + s / (float)cnt;
            a.part[((int64_t)slab * a.G + gid) * a.nout + o] = s;      // this slab's half of the logit: summed by the finish kernel
        }
    }
}

// The second (and last) launch of a call, ONE block: the jobs the main kernel set aside (normally none) are redone with fp32 loops,
// then out[g][o] = half of slab 0 + half of slab 1 + b_head[o] — plain loads and stores in a fixed order: no floating-point atomic, no
// memset of the result.  It also re-arms the set-aside counter for the next call.
__global__ void __launch_bounds__(1024) graph_chain2_finish_kernel(const Chain2Args a) {
    __shared__ float zst[16][64 * 8];
    __shared__ int gst[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbad = a.bad[0];
    for (int i = wave; i < nbad; i += 16) {
        const int job = a.bad[3 + i] >> 1, slab = a.bad[3 + i] & 1;
        const int gid = a.job_gid[(int64_t)job * 64 + lane];
        gst[wave][lane] = gid;
        chain2_exact_job(a, slab, a.job_rows + (int64_t)job * 64, lane, zst[wave]);
        c2_wave_sync();
        c2_pool(a, slab, gid, lane, zst[wave], gst[wave]);
    }
    __syncthreads();       // (block scope: the halves written above are read below by other waves of this block)
    const int n = a.G * a.nout;
    for (int i = threadIdx.x; i < n; i += 1024)
        a.out[i] = (a.part[i] + a.part[(int64_t)n + i]) + (a.bh ? a.bh[i % a.nout] : 0.0f);
    if (threadIdx.x == 0) a.bad[0] = 0;
}

template <int THREADS, bool SB>
__global__ void __launch_bounds__(THREADS) graph_chain2_kernel(const Chain2Args a) {
    constexpr int C2_THREADS = THREADS, C2_WAVES = THREADS / 64;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    u32x4 *img1 = reinterpret_cast<u32x4 *>(lds_raw);
    u32x4 *img2 = img1 + 3 * C2_UNITS1;
    float4 *bias1 = reinterpret_cast<float4 *>(img2 + 3 * C2_UNITS2);     // [32]
    float4 *bias2 = bias1 + C2_D1 / 4;                                      // [16]
    float *head = reinterpret_cast<float *>(bias2 + C2_SLAB / 4);           // [8][64]
    unsigned char *stages = reinterpret_cast<unsigned char *>(head + 8 * C2_SLAB);
    int *ticket = reinterpret_cast<int *>(stages + C2_WAVES * C2_STAGE_BYTES);
    if (threadIdx.x == 0) *ticket = 0;
    const int tid = threadIdx.x;
    const int slab = blockIdx.y, n0 = C2_SLAB * slab;
    {
        WCat w1;
        w1.W[0] = a.W1r; w1.W[1] = a.W1a; w1.K[0] = w1.K[1] = C2_D0;
        w1.sj[0] = w1.sj[1] = a.lay ? 1 : C2_D0; w1.sk[0] = w1.sk[1] = a.lay ? C2_D1 : 1;
        split_fill_image(img1, 2, C2_D1, w1, 0, C2_D1, tid, C2_THREADS);
        WCat w2;
        w2.W[0] = a.W2r; w2.W[1] = a.W2a; w2.K[0] = w2.K[1] = C2_D1;
        w2.sj[0] = w2.sj[1] = a.lay ? 1 : C2_D1; w2.sk[0] = w2.sk[1] = a.lay ? C2_D2 : 1;
        split_fill_image(img2, 16, C2_SLAB, w2, n0, C2_SLAB, tid, C2_THREADS);
        split_fill_bias(bias1, C2_D1, a.b1, 0, C2_D1, tid, C2_THREADS);
        split_fill_bias(bias2, C2_SLAB, a.b2, n0, C2_SLAB, tid, C2_THREADS);
        for (int i = tid; i < 8 * C2_SLAB; i += C2_THREADS) {
            const int o = i / C2_SLAB, c = i - o * C2_SLAB;
            head[i] = o < a.nout ? a.Wh[a.lay ? (n0 + c) * a.nout + o : o * C2_D2 + n0 + c] : 0.0f;
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    unsigned char *stage = stages + wave * C2_STAGE_BYTES;
    float *zst = reinterpret_cast<float *>(stage);                 // [64][8] after the layers
    int *gst = reinterpret_cast<int *>(stage + 2048);              // [64]
    const u32x4 *a1 = img1 + h * C2_D1 + n;                        // + (2 kb) * 128 + 32 cb, planes C2_UNITS1 apart
    const u32x4 *a2 = img2 + h * C2_SLAB + n;                      // + (2 kb) * 64 + 32 cb2, planes C2_UNITS2 apart

    // Jobs: a contiguous share per block, dealt to its waves by an LDS ticket (a wave has only ~4 jobs: a static deal leaves some waves a
    // fifth; a device-wide ticket was tried — 8 200 atomics on two words cost more than they balanced, 198 -> 227 us)
    const int jb0 = (int)(((int64_t)a.njobs * blockIdx.x) / gridDim.x), jb1 = (int)(((int64_t)a.njobs * (blockIdx.x + 1)) / gridDim.x);
    for (;;) {
        int job = 0;
        if (lane == 0) job = jb0 + atomicAdd(ticket, 1);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= jb1) break;
        const int32_t *jr = a.job_rows + (int64_t)job * 64;
        const int r_first = __builtin_amdgcn_readfirstlane(jr[0]);
        const bool two = __builtin_amdgcn_readfirstlane(jr[32]) >= 0;
        const int nt = two ? 2 : 1;
        // ---- the job's rows, neighbours and layer-1 operands ----------------------------------------------------------------------
        int slot[2], deg[2];
        uint32_t locp[2][2];   // the first four neighbours' stage byte offsets at 16-byte position h, 16 bits each (position 2 + h is ^ 32)
        float4 xq[2][4];   // layer-1 operand of each tile, fp32: [0..1] the row itself, [2..3] the sum of its neighbours (split into planes
                           // per 32-feature block: 16 registers a tile instead of 24 — the planes spilled)
        int degmax = 0, degmin = 1 << 30;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            if (T < nt) {
                const int rid = jr[32 * T + n];
                const bool valid = rid >= 0;
                const int rowc = valid ? rid : r_first;            // empty slots mirror slot 0 (finite operands, results unused)
                slot[T] = valid ? 32 * T + n : 0;
                const uint32_t beg = a.rowptr[rowc];
                deg[T] = (int)(a.rowptr[rowc + 1] - beg);
                const float *xr = a.x + (int64_t)rowc * C2_D0;
                xq[T][0] = c2_ld4(xr + 4 * h);
                xq[T][1] = c2_ld4(xr + 8 + 4 * h);
                float4 s0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), s1 = s0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = j < deg[T];
                    const int m = ok ? a.col[beg + j] : rowc;
                    const uint32_t off = (uint32_t)c2_stage_off(slot[T] + (m - rowc), h);   // the neighbour's stage row: same member graph, rows in order
                    locp[T][j >> 1] = (j & 1) ? (locp[T][j >> 1] | (off << 16)) : off;
                    const float *xm = a.x + (int64_t)m * C2_D0;
                    s0 = c2_sel4(ok, c2_add4(s0, c2_ld4(xm + 4 * h)), s0);
                    s1 = c2_sel4(ok, c2_add4(s1, c2_ld4(xm + 8 + 4 * h)), s1);
                }
#pragma unroll 1
                for (int j = 4; j < deg[T]; ++j) {
                    const float *xm = a.x + (int64_t)a.col[beg + j] * C2_D0;
                    s0 = c2_add4(s0, c2_ld4(xm + 4 * h));
                    s1 = c2_add4(s1, c2_ld4(xm + 8 + 4 * h));
                }
                if (a.mean_aggr && deg[T] > 0) {
                    const float cnt = (float)deg[T];
                    s0 = make_float4(0.0f + s0.x / cnt, 0.0f + s0.y / cnt, 0.0f + s0.z / cnt, 0.0f + s0.w / cnt);
                    s1 = make_float4(0.0f + s1.x / cnt, 0.0f + s1.y / cnt, 0.0f + s1.z / cnt, 0.0f + s1.w / cnt);
                }
                xq[T][2] = s0;
                xq[T][3] = s1;
                degmax = max(degmax, deg[T]);
                degmin = min(degmin, deg[T]);
            }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            degmax = max(degmax, __shfl_xor(degmax, o, 64));
            degmin = min(degmin, __shfl_xor(degmin, o, 64));
        }

        bool bad = false;
        f32x16 out[2][2];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[T][c][r] = 0.0f;

#pragma unroll 1
        for (int cb = 0; cb < 4; ++cb) {
            // ---- layer 1, features 32 cb .. 32 cb + 31 of both tiles -------------------------------------------------------------
            f32x16 hb[2];
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int r = 0; r < 16; ++r) hb[T][r] = 0.0f;
            {
                const SplitA w0 = split_read_a(a1 + 32 * cb, C2_UNITS1);                 // k-block 0: the rows themselves
                const SplitA w1 = split_read_a(a1 + 2 * C2_D1 + 32 * cb, C2_UNITS1);     // k-block 1: sum of the neighbours
                hb[0] = split_mac(hb[0], w0, split8(xq[0][0], xq[0][1]));
                if (two) hb[1] = split_mac(hb[1], w0, split8(xq[1][0], xq[1][1]));
                hb[0] = split_mac(hb[0], w1, split8(xq[0][2], xq[0][3]));
                if (two) hb[1] = split_mac(hb[1], w1, split8(xq[1][2], xq[1][3]));
            }
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += hb[T][r];
                bad |= (s != s);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 v = split_out4(hb[T], q4, bias1[8 * cb + 2 * q4 + h], a.act1);
                    hb[T][4 * q4] = v.x; hb[T][4 * q4 + 1] = v.y; hb[T][4 * q4 + 2] = v.z; hb[T][4 * q4 + 3] = v.w;
                }
            }
            // ---- layer 2, k-blocks 2 cb and 2 cb + 1: root straight from hb, aggregate through the stage -----------------------------
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kb = 2 * cb + s;
                if (SB) __builtin_amdgcn_sched_barrier(0);   // (keep hipcc from hoisting every A-operand read of the block to its top: 96 registers)
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    if (T < nt) {
                        const float4 q0 = make_float4(hb[T][8 * s], hb[T][8 * s + 1], hb[T][8 * s + 2], hb[T][8 * s + 3]);
                        const float4 q1 = make_float4(hb[T][8 * s + 4], hb[T][8 * s + 5], hb[T][8 * s + 6], hb[T][8 * s + 7]);
                        if (slot[T] == 32 * T + n) {               // (mirror lanes of empty slots do not write)
                            *reinterpret_cast<float4 *>(stage + c2_stage_off(slot[T], h)) = q0;
                            *reinterpret_cast<float4 *>(stage + c2_stage_off(slot[T], 2 + h)) = q1;
                        }
                        const Split8 b = split8(q0, q1);
#pragma unroll
                        for (int c = 0; c < 2; ++c) {              // one tile, one column block at a time: 12 + 12 operand registers live
                            const SplitA wr = split_read_a(a2 + (2 * kb) * C2_SLAB + 32 * c, C2_UNITS2);
                            out[T][c] = split_mac(out[T][c], wr, b);
                        }
                        if (SB) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                c2_wave_sync();
                if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    if (T < nt) {
                        // NNlib.scatter(+): dst = 0, then dst += src for the edges in order
                        float4 s0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), s1 = s0;
                        if (degmin >= 4) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                s0 = c2_add4(s0, *reinterpret_cast<const float4 *>(stage + ((locp[T][j >> 1] >> (16 * (j & 1))) & 0xffff)));
                                s1 = c2_add4(s1, *reinterpret_cast<const float4 *>(stage + (((locp[T][j >> 1] >> (16 * (j & 1))) & 0xffff) ^ 32)));
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool ok = j < deg[T];
                                s0 = c2_sel4(ok, c2_add4(s0, *reinterpret_cast<const float4 *>(stage + ((locp[T][j >> 1] >> (16 * (j & 1))) & 0xffff))), s0);
                                s1 = c2_sel4(ok, c2_add4(s1, *reinterpret_cast<const float4 *>(stage + (((locp[T][j >> 1] >> (16 * (j & 1))) & 0xffff) ^ 32))), s1);
                            }
                        }
                        if (degmax > 4) {
                            const int rowc = slot[T] == 32 * T + n ? jr[32 * T + n] : r_first;
                            const uint32_t beg = a.rowptr[rowc];
#pragma unroll 1
                            for (int j = 4; j < deg[T]; ++j) {
                                const int lj = slot[T] + (a.col[beg + j] - rowc);
                                s0 = c2_add4(s0, *reinterpret_cast<const float4 *>(stage + c2_stage_off(lj, h)));
                                s1 = c2_add4(s1, *reinterpret_cast<const float4 *>(stage + c2_stage_off(lj, 2 + h)));
                            }
                        }
                        if (a.mean_aggr && deg[T] > 0) {
                            const float cnt = (float)deg[T];
                            s0 = make_float4(0.0f + s0.x / cnt, 0.0f + s0.y / cnt, 0.0f + s0.z / cnt, 0.0f + s0.w / cnt);
                            s1 = make_float4(0.0f + s1.x / cnt, 0.0f + s1.y / cnt, 0.0f + s1.z / cnt, 0.0f + s1.w / cnt);
                        }
                        const Split8 b = split8(s0, s1);
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const SplitA wa = split_read_a(a2 + (2 * (8 + kb)) * C2_SLAB + 32 * c, C2_UNITS2);
                            out[T][c] = split_mac(out[T][c], wa, b);
                        }
                        if (SB) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                c2_wave_sync();   // the stage is rewritten by the next k-block
            }
        }
        // ---- σ2, z = W_head[:, slab] * h2, per-graph pooling ---------------------------------------------------------------------------
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            if (T < nt) {
                float zo[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) zo[o] = 0.0f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float s = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += out[T][c][r];
                    bad |= (s != s);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int colq = 32 * c + 8 * q4 + 4 * h;
                        const float4 v = split_out4(out[T][c], q4, bias2[colq >> 2], a.act2);
#pragma unroll
                        for (int o = 0; o < 8; ++o) {
                            if (o < a.nout) {
                                const float4 wv = *reinterpret_cast<const float4 *>(head + o * C2_SLAB + colq);
                                zo[o] = fmaf(wv.x, v.x, fmaf(wv.y, v.y, fmaf(wv.z, v.z, fmaf(wv.w, v.w, zo[o]))));
                            }
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const float tot = zo[o] + __shfl_xor(zo[o], 32, 64);
                    if (h == 0 && slot[T] == 32 * T + n) zst[(32 * T + n) * 8 + o] = tot;
                }
            }
        }
        const int gid = a.job_gid[(int64_t)job * 64 + lane];
        gst[lane] = gid;
        c2_wave_sync();
        if (__builtin_amdgcn_ballot_w64(bad) != 0) {
            // a non-finite operand somewhere in the job (NaN accumulators): nothing of it is written here; graph_chain2_finish_kernel redoes it
            if (lane == 0) a.bad[3 + atomicAdd(a.bad, 1)] = job * 2 + slab;
        } else {
            c2_pool(a, slab, gid, lane, zst, gst);
        }
        c2_wave_sync();   // the stage is reused by the next job
    }
}

}  // namespace gnnmp

using namespace gnnmp;

// Pack the member graphs of a batch (MLUtils.batch: contiguous row ranges seg_ptr[k] .. seg_ptr[k + 1]) into wave jobs of at most 64
// rows, best-fit decreasing.  Graph prep like gnnmp_plan_create: once per batched graph, synchronises the stream.
extern "C" int gnnmp_chain_jobs_create(gnnmp_chain_jobs_t **out, const int64_t *seg_ptr, int64_t G, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out || G < 0 || (G > 0 && !seg_ptr)) return fail(GNNMP_EINVAL, "chain_jobs_create: bad argument");
    *out = nullptr;
    std::vector<int64_t> sp((size_t)G + 1, 0);
    if (G > 0) {
        GNNMP_HIP(hipMemcpyAsync(sp.data(), seg_ptr, sizeof(int64_t) * (size_t)(G + 1), hipMemcpyDeviceToHost, stream));
        GNNMP_HIP(hipStreamSynchronize(stream));
    }
    gnnmp_chain_jobs *J = new gnnmp_chain_jobs();
    J->G = G;
    J->N = G > 0 ? sp[(size_t)G] - sp[0] : 0;
    for (int64_t g = 0; g < G; ++g) J->max_graph = std::max(J->max_graph, sp[(size_t)g + 1] - sp[(size_t)g]);
    if (J->max_graph > 64 || J->N >= ((int64_t)1 << 31) || G == 0) {   // no jobs: the caller's chain runs on the general kernel
        *out = J;
        return GNNMP_OK;
    }
    // best fit decreasing: graphs by decreasing size (ties by id), each into the fullest job that still takes it
    std::vector<int32_t> order((size_t)G);
    for (int64_t g = 0; g < G; ++g) order[(size_t)g] = (int32_t)g;
    std::stable_sort(order.begin(), order.end(), [&](int32_t p, int32_t q) {
        return sp[(size_t)p + 1] - sp[(size_t)p] > sp[(size_t)q + 1] - sp[(size_t)q];
    });
    std::vector<std::vector<int32_t>> members;             // graphs of each job
    std::vector<int> room;                                   // free slots of each job
    std::vector<std::vector<int32_t>> by_room(65);          // jobs with exactly r free slots (stack)
    for (int32_t g : order) {
        const int sz = (int)(sp[(size_t)g + 1] - sp[(size_t)g]);
        if (sz == 0) { J->has_empty = 1; continue; }
        int r = sz;
        while (r <= 64 && by_room[(size_t)r].empty()) ++r;
        int j;
        if (r <= 64) {
            j = by_room[(size_t)r].back();
            by_room[(size_t)r].pop_back();
        } else {
            j = (int)members.size();
            members.emplace_back();
            room.push_back(64);
        }
        members[(size_t)j].push_back(g);
        room[(size_t)j] -= sz;
        by_room[(size_t)room[(size_t)j]].push_back(j);
    }
    const int njobs = (int)members.size();
    std::vector<int32_t> rows((size_t)njobs * 64, -1), gid((size_t)njobs * 64, -1);
    int64_t tiles = 0;
    for (int j = 0; j < njobs; ++j) {
        std::sort(members[(size_t)j].begin(), members[(size_t)j].end());   // rows of a job in node order
        int s = 0;
        for (int32_t g : members[(size_t)j])
            for (int64_t r = sp[(size_t)g]; r < sp[(size_t)g + 1]; ++r, ++s) {
                rows[(size_t)j * 64 + (size_t)s] = (int32_t)r;
                gid[(size_t)j * 64 + (size_t)s] = g;
            }
        tiles += s > 32 ? 2 : 1;
    }
    J->njobs = njobs;
    J->fill = tiles > 0 ? (double)J->N / (32.0 * (double)tiles) : 0.0;
    if (njobs > 0) {
        const size_t bytes = (size_t)njobs * 64 * sizeof(int32_t);
        if (hipMalloc(&J->rows, bytes) != hipSuccess || hipMalloc(&J->gid, bytes) != hipSuccess) {
            if (J->rows) (void)hipFree(J->rows);
            delete J;
            return fail(GNNMP_EALLOC, "chain_jobs_create: hipMalloc failed");
        }
        if (hipMalloc(&J->bad, sizeof(int32_t) * (size_t)(3 + 2 * njobs)) != hipSuccess) {
            (void)hipFree(J->rows); (void)hipFree(J->gid);
            delete J;
            return fail(GNNMP_EALLOC, "chain_jobs_create: hipMalloc failed");
        }
        if (hipMalloc(&J->part, sizeof(float) * (size_t)2 * (size_t)G * 8) != hipSuccess) {
            (void)hipFree(J->rows); (void)hipFree(J->gid); (void)hipFree(J->bad);
            delete J;
            return fail(GNNMP_EALLOC, "chain_jobs_create: hipMalloc failed");
        }
        GNNMP_HIP(hipMemsetAsync(J->bad, 0, 3 * sizeof(int32_t), stream));
        GNNMP_HIP(hipMemcpyAsync(J->rows, rows.data(), bytes, hipMemcpyHostToDevice, stream));
        GNNMP_HIP(hipMemcpyAsync(J->gid, gid.data(), bytes, hipMemcpyHostToDevice, stream));
        GNNMP_HIP(hipStreamSynchronize(stream));
    }
    *out = J;
    return GNNMP_OK;
}

extern "C" int gnnmp_chain_jobs_destroy(gnnmp_chain_jobs_t *J) {
    if (!J) return GNNMP_OK;
    if (J->rows) (void)hipFree(J->rows);
    if (J->gid) (void)hipFree(J->gid);
    if (J->bad) (void)hipFree(J->bad);
    if (J->part) (void)hipFree(J->part);
    delete J;
    return GNNMP_OK;
}

/* info[0] = jobs, info[1] = member graphs, info[2] = rows, info[3] = largest member graph, info[4] = per-mille of the MFMA tiles' rows
 * that are real rows */
extern "C" int gnnmp_chain_jobs_info(const gnnmp_chain_jobs_t *J, int64_t *info) {
    if (!J || !info) return fail(GNNMP_EINVAL, "chain_jobs_info: null pointer");
    info[0] = J->njobs; info[1] = J->G; info[2] = J->N; info[3] = J->max_graph; info[4] = (int64_t)(J->fill * 1000.0 + 0.5);
    return GNNMP_OK;
}

namespace gnnmp {
// Returns GNNMP_OK if it launched, 1 if the chain / the batch is outside this kernel's envelope (graph_chain.hip's kernel runs).
int graph_chain2_try(gnnmp_graph_t *p, const gnnmp_chain_jobs_t *J, const int64_t *seg_ptr, int64_t G, const float *x, int n_layers,
                     const int64_t *dims, const float *const *W_root, const float *const *W_agg, const float *const *bias,
                     const int *act, int w_layout, int aggr, int pool_aggr, const float *W_head, const float *b_head, int64_t nout,
                     float *out, hipStream_t stream) {
    if (!J || J->njobs <= 0 || J->has_empty || J->G != G || J->N != p->n_dst) return 1;
    if (n_layers != 2 || dims[0] != C2_D0 || dims[1] != C2_D1 || dims[2] != C2_D2 || nout > 8) return 1;
    if (knob(KNOB_CHAIN) == 1) return 1;    // 1 = the general kernel only (A/B runs)
    if ((reinterpret_cast<uintptr_t>(x) & 15)) return 1;
    Chain2Args a = {};
    a.rowptr = p->rowptr;
    a.col = p->col;
    a.job_rows = J->rows;
    a.job_gid = J->gid;
    a.njobs = J->njobs;
    a.seg_ptr = seg_ptr;
    a.x = x;
    a.W1r = W_root[0]; a.W1a = W_agg[0]; a.b1 = bias ? bias[0] : nullptr;
    a.W2r = W_root[1]; a.W2a = W_agg[1]; a.b2 = bias ? bias[1] : nullptr;
    a.Wh = W_head; a.bh = b_head;
    a.nout = (int)nout;
    a.lay = w_layout;
    a.act1 = act[0]; a.act2 = act[1];
    a.mean_aggr = aggr == GNNMP_MEAN;
    a.pool_mean = pool_aggr == GNNMP_MEAN;
    a.out = out;
    a.bad = J->bad;
    a.part = J->part;
    a.G = (int)G;
    // knob 13 (experiments): low 2 bits = waves per block 0: 8, 1: 4 (512 VGPRs a wave), 2: 6; bit 2 = scheduling barriers around the A-operand reads
    const int kv = knob(KNOB_T16_DEBUG);
    const int waves = (kv & 3) == 1 ? 4 : ((kv & 3) == 2 ? 6 : 8);
    const bool sb = (kv & 4) != 0;        // default: none (measured 191 -> 176 us at G = 8192: hipcc schedules the block better on its own)
    const size_t lds = (size_t)3 * C2_UNITS1 * 16 + (size_t)3 * C2_UNITS2 * 16 + C2_D1 * 4 + C2_SLAB * 4 + 8 * C2_SLAB * 4 +
                       (size_t)C2_MAX_WAVES * C2_STAGE_BYTES + 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipSuccess;
#define C2_ATTR(T, S) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&graph_chain2_kernel<T, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        C2_ATTR(512, true) C2_ATTR(512, false) C2_ATTR(256, true) C2_ATTR(256, false) C2_ATTR(384, true) C2_ATTR(384, false)
#undef C2_ATTR
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(graph_chain2_kernel)");
        attr_set = true;
    }
    const int cus = device_cus();
    const int gx = std::max(1, std::min(cus / 2, (a.njobs + waves - 1) / waves));
    const dim3 grid((unsigned)gx, 2);
    if (waves == 8 && sb) graph_chain2_kernel<512, true><<<grid, 512, lds, stream>>>(a);
    else if (waves == 8) graph_chain2_kernel<512, false><<<grid, 512, lds, stream>>>(a);
    else if (waves == 4 && sb) graph_chain2_kernel<256, true><<<grid, 256, lds, stream>>>(a);
    else if (waves == 4) graph_chain2_kernel<256, false><<<grid, 256, lds, stream>>>(a);
    else if (sb) graph_chain2_kernel<384, true><<<grid, 384, lds, stream>>>(a);
    else graph_chain2_kernel<384, false><<<grid, 384, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("graph_chain2_kernel");
    graph_chain2_finish_kernel<<<1, 1024, 0, stream>>>(a);
    GNNMP_LAUNCH_CHECK("graph_chain2_finish_kernel");
    return GNNMP_OK;
}
}  // namespace gnnmp
