// graph_chain2.hip — BASELINE.json config 5 with NOTHING but the features, the edge index, 8 bytes of z per row and the logits crossing
// the chip's memory:
//   GNNChain(GraphConv(16 => 128, σ1), GraphConv(128 => 128, σ2), GlobalPool(mean | +), Dense(128 => nout))
//   (examples/graph_classification_tudataset.jl:79-82; layer body GNNlib/src/layers/conv.jl:102-108; pooling layers/pool.jl:3-5)
// graph_chain.hip (the general kernel) keeps layer outputs in a block-local scratch matrix; at G = 8192 that scratch (32 blocks x ~1 MB
// per XCD) falls out of the 4 MB L2 and the kernel runs at the fabric's speed (0.215 ms).  Here whole member graphs stay inside a CU:
//   * a job = up to 64 rows = whole member graphs packed best-fit-decreasing on the host side of gnnmp_chain_jobs_create (98 % full
//     at n ~ U{20..40}); a PAIR of waves runs it, one 32-row MFMA tile each (one wave holding both tiles needs 256 registers and
//     spills; a tile a wave fits 168: three waves a SIMD);
//   * layer 1, 32 features at a time: 12 split-bf16 MFMAs against the W1 planes in LDS; its B operand (x_i, and sum_j x_j gathered
//     from the 64-byte feature rows in original edge order) is formed once per job; the job-table / rowptr / col entries of the NEXT
//     job are fetched during this job's K loop (three dependent loads, one per pass);
//   * those 32 features ARE the next layer's operand: the 32x32 accumulator layout (lane = node, registers = 8 a + 4 h + b) is the
//     B-operand layout of msplit.h for k-blocks 2 cb and 2 cb + 1 — bias, σ1, split into planes, MFMA, no memory in between;
//   * the aggregation of layer 2 runs AFTER its product: t_i = W2_agg h1_i is accumulated next to W2_root h1_i (one B operand, one
//     split, for both), and sum_j t_j — the same sum as W2_agg sum_j h1_j in exact arithmetic — is formed over this block's 64 columns:
//     8 columns at a time through the two 2 KB halves of the pair's LDS stage, every lane adding its row's in-neighbours in original
//     edge order (member graphs are whole inside a job: a neighbour's slot is this row's slot + the difference of the node ids).  The
//     two waves order their stage accesses with a pair of LDS event counters (see PairCtl): no block barrier after the set-up;
//   * layer 2's 128 outputs are two 64-column slabs handled by DIFFERENT blocks (blockIdx.y): W1's planes (24 KB) + one slab of
//     [W_root | W_agg] (96 KB) + 6 stages fit LDS for the whole launch — no image is ever swapped; each slab block recomputes the cheap
//     layer 1 (48 of 240 MFMAs per tile);
//   * bias, σ2 and z = W_head[:, slab] * h2 on the accumulators; the row's z (nout floats per slab) goes to memory, and a second small
//     launch pools the rows of every member graph in node order and adds the two slabs' halves and the head's bias (no
//     floating-point atomic anywhere, no memset: two launches per call).
// A tile that meets a non-finite operand (NaN accumulators, msplit.h) sets its job aside; the finish kernel recomputes it with plain
// fp32 loops (a call or the inlined loops inside the main kernel cost its job loop ~100 spilled registers).
// Envelope: exactly two layers 16 => 128 => 128, nout <= 8, aggr and pool in {+, mean}, member graphs of at most 64 nodes; everything
// else is graph_chain.hip's.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "chain_pack.h"
#include "msplit.h"
#include "pool.h"

struct gnnmp_chain_jobs {
    int32_t *tab = nullptr;    // rows [njobs][64]: global row of each slot, -1 = empty
    int njobs = 0;             // host-packed handles: the number of jobs; device-packed ones: -1 (it lives in hdr[0])
    int njobs_cap = 0;         // rows of `tab` that exist (device-packed: an upper bound of any best-fit packing, min(G, N / 32 + 2))
    int64_t G = 0, N = 0;
    int64_t max_graph = 0;     // largest member graph (> 64: no jobs, the general kernel runs)
    double fill = 0.0;         // rows / (32 x tiles): MFMA work spent on real rows
    int has_empty = 0;         // some member graph has no node (its logits are the head's bias: left to the general kernel)
    int32_t *bad = nullptr;    // [3 + 4 njobs]: [0], [1] the counters of set-aside jobs of even / odd calls (each re-armed by the other parity's
                               // finish kernel), [3..] (job * 2 + slab) of the tiles that met a non-finite operand in the running call
    int32_t *hdr = nullptr;    // device-packed handles: [0] jobs, [1] 32-row tiles, [2] poison (the sizes contradict what the caller announced:
                               // the finish kernel then writes NaN logits), [3] largest member graph seen, [4] member graphs without a node
    float *zrows = nullptr;    // [2][N][8] the two slabs' z of every row (each word written once per call: no atomics, no memset)
    std::atomic<unsigned> calls{0};   // launches so far: picks the parity of the set-aside counters (a handle is used on ONE stream at a time)
    void *block = nullptr;     // the ONE device allocation behind tab, bad, hdr and zrows
    size_t block_bytes = 0;
};

namespace gnnmp {

constexpr int C2_D0 = 16, C2_D1 = 128, C2_D2 = 128, C2_SLAB = 64;
constexpr int C2_UNITS1 = 2 * 2 * C2_D1;          // 16-byte units per plane of the W1 image (K = 32)
constexpr int C2_UNITS2 = 16 * 2 * C2_SLAB;       // ... of one slab of [W2_root | W2_agg] (K = 256)
constexpr int C2_STAGE_BYTES = 4096;          // per pair of waves

struct Chain2Args {
    const uint32_t *rowptr;
    const int32_t *col;
    const int32_t *job_rows;
    int njobs;
    const int32_t *hdr;   // device-packed jobs: hdr[0] = the number of jobs (njobs above is then only the bound the grid was sized for), hdr[2] = poison
    const int64_t *seg_ptr;
    const float *x;
    const float *W1r, *W1a, *b1, *W2r, *W2a, *b2, *Wh, *bh;
    int lay;            // 0: weights C row-major [Dout][Din]; 1: Julia (Dout, Din) column-major as stored, W(j, k) at [k * Dout + j]
    int nout, act1, act2, mean_aggr, pool_mean;
    float *out;
    float *zrows;       // [2][N][nout] z = W_head[:, slab] * h2 of every row: pooled by the finish kernel
    int G, N;
    long long *trace;   // (GNNMP_CHAIN_TRACE builds: cycle stamps of block (0, 0))
    int32_t *bad_count, *bad_reset, *bad_list;     // this call's counter of set-aside jobs, the next call's (re-armed), job * 2 + slab
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
__device__ __forceinline__ float c2_act(float v, int act) { return (act == GNNMP_ACT_RELU && v < 0.0f) ? 0.0f : v; }

// The exact path of a job: z of every slot's row with fp32 loops (this lane = slot `lane`).  Runs in graph_chain2_finish_kernel only: a call
// (or the inlined loops) inside the main kernel costs its tile loop ~100 spilled registers.
__device__ void chain2_exact_job(const Chain2Args &a, int slab, const int32_t *jr, int lane) {
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
    for (int o = 0; o < a.nout; ++o) a.zrows[((int64_t)slab * a.N + row) * a.nout + o] = z[o];
}

// reduce_nodes(aggr, g, x) = scatter(aggr, x, graph_indicator) in node order (utils.jl:12-16); Dense and + / mean commute: the logits of
// member graph g are the pooled z rows of its nodes.  One lane per (graph, slab): the rows of a member graph are contiguous, the sum runs
// in node order; the two slabs' halves meet through one shuffle.  No floating-point atomic, no memset of the result.
__device__ __forceinline__ void c2_pool_graph(const Chain2Args &a, int g, int slab, int lane_in_wave) {
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.0f;
    int cnt = 0;
    if (g < a.G) {
        const int64_t r0 = a.seg_ptr[g], r1 = a.seg_ptr[g + 1];
        cnt = (int)(r1 - r0);
        const float *z = a.zrows + ((int64_t)slab * a.N + r0) * a.nout;
        if (a.nout == 2) {
            const float2 *z2 = reinterpret_cast<const float2 *>(z);
            for (int t = 0; t < cnt; t += 16) {          // sixteen loads in flight (past the end: the last row again, not added), the adds in row order
                float2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = z2[min(t + u, cnt - 1)];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (t + u < cnt) { acc[0] = acc[0] + v[u].x; acc[1] = acc[1] + v[u].y; }
                }
            }
        } else {
            for (int t = 0; t < cnt; ++t)
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o < a.nout) acc[o] = acc[o] + z[(int64_t)t * a.nout + o];
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (o < a.nout) {
            float s = acc[o];
            if (a.pool_mean && cnt > 0) s = 0.0f + s / (float)cnt;
            const float other = __shfl_xor(s, 1, 64);            // the other slab's half (lanes 2 k, 2 k + 1)
            if (slab == 0 && g < a.G) a.out[(int64_t)g * a.nout + o] = (s + other) + (a.bh ? a.bh[o] : 0.0f);
        }
    }
    (void)lane_in_wave;
}

// The second (and last) launch of a call: the pooling.  If the main kernel set jobs aside (non-finite operands: normally none), block 0
// alone first redoes them with fp32 loops and then pools every graph; the other blocks leave.  The counter of set-aside jobs alternates
// between two words from call to call: this launch reads one (nobody writes it now) and re-arms the other for the next call.
__global__ void __launch_bounds__(256) graph_chain2_finish_kernel(const Chain2Args a) {
    if (a.hdr && a.hdr[2] != 0) {
        // the device packing found member graphs its caller had excluded (more than 64 nodes / none): no job ran.  Loud, not silent: NaN
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)a.G * a.nout; i += (int64_t)gridDim.x * 256)
            a.out[i] = __builtin_nanf("");
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.bad_reset = 0;
        return;
    }
    const int nbad = *a.bad_count;
    if (nbad == 0) {
        const int i = blockIdx.x * 256 + threadIdx.x;                // (the grid covers 2 G lanes)
        c2_pool_graph(a, i >> 1, i & 1, threadIdx.x & 63);
    } else if (blockIdx.x == 0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int i = wave; i < nbad; i += 4)
            chain2_exact_job(a, a.bad_list[i] & 1, a.job_rows + (int64_t)(a.bad_list[i] >> 1) * 64, lane);
        __threadfence_block();
        __syncthreads();       // (block scope: the rows written above are read below by other waves of this block)
        for (int i = threadIdx.x; i < 2 * ((a.G + 127) / 128) * 128; i += 256) c2_pool_graph(a, i >> 1, i & 1, threadIdx.x & 63);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.bad_reset = 0;
}

// ---- wave PAIRS: one 32-row tile a wave --------------------------------------------------------------------------------------------------
// The two tiles of a job belong to two waves (2 p, 2 p + 1: different SIMDs) that share the job's 4 KB stage.  What they exchange through
// it — t of the other tile's rows for the neighbour sums — is ordered by a pair of LDS event counters: each
// wave publishes the number of events it has passed (release), the other spins on it (acquire); both are resident in the same block,
// so the spin cannot starve its partner.  The event numbers are the same in both waves at every point of the program.
struct PairCtl {
    int *mine;            // this wave's event counter (LDS)
    int *theirs;          // the partner's
    int ev;               // events passed so far: identical in both waves at every event
};
// LDS executes a wave's instructions in issue order, so "my stage writes (or reads), then my counter" needs no s_waitcnt in between: the
// counter store is relaxed behind a wavefront-scope fence (compiler ordering only).  A release at workgroup scope would drain lgkmcnt
// before every counter store: two more LDS round trips (~400 cycles each under this kernel's load) per round.
__device__ __forceinline__ void pair_arrive(PairCtl &p) {
    p.ev = __builtin_amdgcn_readfirstlane(p.ev) + 1;      // (kept in a scalar register: hipcc otherwise carries it per lane and spills it)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __hip_atomic_store(p.mine, p.ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pair_wait(const PairCtl &p, int ev_) {
    const int ev = __builtin_amdgcn_readfirstlane(ev_);
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p.theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < ev)
        __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// what a lane needs of a job before any feature is touched: fetched for the NEXT job while this one is in its K loop (three dependent
// global loads: job table -> rowptr -> col)
struct JobPre {
    int rid;            // this lane's row (-1: empty slot)
    int first;          // the job's first row (uniform)
    int r32;            // row of slot 32 (uniform; < 0: the job has one tile)
    uint32_t beg;
    int deg;
    int m[4];           // the first four in-neighbours (the row itself past the degree)
};

#ifdef GNNMP_CHAIN_TRACE
#define C2_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && kjob < 8) a.trace[(wave * 8 + kjob) * 8 + (i)] = clock64(); } while (0)
#else
#define C2_STAMP(i) do { } while (0)
#endif

template <int THREADS>
__global__ void __launch_bounds__(THREADS) graph_chain2_kernel(const Chain2Args a) {
    constexpr int WAVES = THREADS / 64, PAIRS = WAVES / 2;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    u32x4 *img1 = reinterpret_cast<u32x4 *>(lds_raw);
    u32x4 *img2 = img1 + 3 * C2_UNITS1;
    float4 *bias1 = reinterpret_cast<float4 *>(img2 + 3 * C2_UNITS2);     // [32]
    float4 *bias2 = bias1 + C2_D1 / 4;                                      // [16]
    float *head = reinterpret_cast<float *>(bias2 + C2_SLAB / 4);           // [8][64]
    unsigned char *stages = reinterpret_cast<unsigned char *>(head + 8 * C2_SLAB);
    int *events = reinterpret_cast<int *>(stages + PAIRS * C2_STAGE_BYTES);  // per wave: event counter
    const int tid = threadIdx.x;
    if (tid < WAVES) events[tid] = 0;
    const int slab = blockIdx.y, n0 = C2_SLAB * slab;
    {
        WCat w1;
        w1.W[0] = a.W1r; w1.W[1] = a.W1a; w1.K[0] = w1.K[1] = C2_D0;
        w1.sj[0] = w1.sj[1] = a.lay ? 1 : C2_D0; w1.sk[0] = w1.sk[1] = a.lay ? C2_D1 : 1;
        split_fill_image(img1, 2, C2_D1, w1, 0, C2_D1, tid, THREADS);
        WCat w2;
        w2.W[0] = a.W2r; w2.W[1] = a.W2a; w2.K[0] = w2.K[1] = C2_D1;
        w2.sj[0] = w2.sj[1] = a.lay ? 1 : C2_D1; w2.sk[0] = w2.sk[1] = a.lay ? C2_D2 : 1;
        split_fill_image(img2, 16, C2_SLAB, w2, n0, C2_SLAB, tid, THREADS);
        split_fill_bias(bias1, C2_D1, a.b1, 0, C2_D1, tid, THREADS);
        split_fill_bias(bias2, C2_SLAB, a.b2, n0, C2_SLAB, tid, THREADS);
        for (int i = tid; i < 8 * C2_SLAB; i += THREADS) {
            const int o = i / C2_SLAB, c = i - o * C2_SLAB;
            head[i] = o < a.nout ? a.Wh[a.lay ? (n0 + c) * a.nout + o : o * C2_D2 + n0 + c] : 0.0f;
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int T = wave & 1, pr = wave >> 1;                        // this wave's tile of the pair's jobs
    unsigned char *stage = stages + pr * C2_STAGE_BYTES;
    const u32x4 *a1 = img1 + h * C2_D1 + n;                        // + (2 kb) * 128 + 32 cb, planes C2_UNITS1 apart
    const u32x4 *a2 = img2 + h * C2_SLAB + n;                      // + (2 kb) * 64 + 32 cb2, planes C2_UNITS2 apart
    PairCtl pc = {events + wave, events + (wave ^ 1), 0};

    // Jobs: a contiguous share per block, dealt to its pairs round-robin (the jobs are packed to the same size: a ticket balances nothing,
    // and a known next job is what lets its table entries be fetched ahead)
    const int njobs = a.hdr ? __builtin_amdgcn_readfirstlane(a.hdr[0]) : a.njobs;
    const int jb0 = (int)(((int64_t)njobs * blockIdx.x) / gridDim.x), jb1 = (int)(((int64_t)njobs * (blockIdx.x + 1)) / gridDim.x);
    auto pre_a = [&](int job, JobPre &p) {
        const int32_t *q = a.job_rows + (int64_t)job * 64;
        p.rid = q[32 * T + n];
        p.first = q[0];
        p.r32 = q[32];
    };
    auto pre_b = [&](JobPre &p) {
        const int rowc = p.rid >= 0 ? p.rid : p.first;
        p.beg = a.rowptr[rowc];
        p.deg = (int)(a.rowptr[rowc + 1] - p.beg);
    };
    auto pre_c = [&](JobPre &p) {
        const int rowc = p.rid >= 0 ? p.rid : p.first;
#pragma unroll
        for (int j = 0; j < 4; ++j) p.m[j] = j < p.deg ? a.col[p.beg + j] : rowc;
    };
    JobPre cur = {}, nxt = {};
    int ev_stage = 0;          // the partner's event after which the pair's stage is free again (its last read of the previous job)
    int job = jb0 + pr;
    if (job < jb1) { pre_a(job, cur); pre_b(cur); pre_c(cur); }
    for (int kjob = 0; job < jb1; job += PAIRS, ++kjob) {
        C2_STAMP(0);
        const int job_next = min(job + PAIRS, jb1 - 1);           // (the last job of the share is fetched again: always a valid entry)
        const bool two = __builtin_amdgcn_readfirstlane(cur.r32) >= 0;
        if (two || T == 0) {       // (a job of at most 32 rows: the even wave alone, no event but the last)
            // ---- this tile's rows, neighbours and layer-1 operands -------------------------------------------------------------------
            const bool valid = cur.rid >= 0;
            const int rowc = valid ? cur.rid : __builtin_amdgcn_readfirstlane(cur.first);   // empty slots mirror slot 0 (finite operands, results unused)
            const int slot = valid ? 32 * T + n : 0;
            const uint32_t beg = cur.beg;
            const int deg = cur.deg;
            uint32_t locp[2];      // the first four neighbours' stage byte offsets (32-byte rows, 16-byte half h), 16 bits each
            float4 xq[4];          // [0..1] the row itself, [2..3] the sum of its neighbours
            {
                const float *xr = a.x + (int64_t)rowc * C2_D0;
                xq[0] = c2_ld4(xr + 4 * h);
                xq[1] = c2_ld4(xr + 8 + 4 * h);
                float4 s0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), s1 = s0;
                locp[0] = locp[1] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = j < deg;
                    const int m = cur.m[j];
                    const uint32_t off = (uint32_t)((slot + (m - rowc)) * 32 + 16 * h);   // same member graph, rows in order
                    locp[j >> 1] |= off << (16 * (j & 1));
                    const float *xm = a.x + (int64_t)m * C2_D0;
                    s0 = c2_sel4(ok, c2_add4(s0, c2_ld4(xm + 4 * h)), s0);
                    s1 = c2_sel4(ok, c2_add4(s1, c2_ld4(xm + 8 + 4 * h)), s1);
                }
#pragma unroll 1
                for (int j = 4; j < deg; ++j) {
                    const float *xm = a.x + (int64_t)a.col[beg + j] * C2_D0;
                    s0 = c2_add4(s0, c2_ld4(xm + 4 * h));
                    s1 = c2_add4(s1, c2_ld4(xm + 8 + 4 * h));
                }
                if (a.mean_aggr && deg > 0) {
                    const float cnt = (float)deg;
                    s0 = make_float4(0.0f + s0.x / cnt, 0.0f + s0.y / cnt, 0.0f + s0.z / cnt, 0.0f + s0.w / cnt);
                    s1 = make_float4(0.0f + s1.x / cnt, 0.0f + s1.y / cnt, 0.0f + s1.z / cnt, 0.0f + s1.w / cnt);
                }
                xq[2] = s0;
                xq[3] = s1;
            }
            int degmax = deg, degmin = deg;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                degmax = max(degmax, __shfl_xor(degmax, o, 64));
                degmin = min(degmin, __shfl_xor(degmin, o, 64));
            }
            C2_STAMP(1);
            bool bad = false;
            f32x16 out[2], tacc[2];      // W2_root h1_i (the result), t_i = W2_agg h1_i (summed over the in-neighbours afterwards)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) { out[c][r] = 0.0f; tacc[c][r] = 0.0f; }

#pragma unroll 1
            for (int cb = 0; cb < 4; ++cb) {
                // the next job's table entries, one dependent load per pass (consumed a pass later: their latency is covered)
                if (cb == 0) pre_a(job_next, nxt);
                else if (cb == 1) pre_b(nxt);
                else if (cb == 2) pre_c(nxt);
                // ---- layer 1, features 32 cb .. 32 cb + 31 of this tile ----------------------------------------------------------------
                f32x16 hb;
#pragma unroll
                for (int r = 0; r < 16; ++r) hb[r] = 0.0f;
                hb = split_mac(hb, split_read_a(a1 + 32 * cb, C2_UNITS1), split8(xq[0], xq[1]));                 // the rows themselves
                hb = split_mac(hb, split_read_a(a1 + 2 * C2_D1 + 32 * cb, C2_UNITS1), split8(xq[2], xq[3]));     // sum of the neighbours
                {
                    float s = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += hb[r];
                    bad |= (s != s);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4 v = split_out4(hb, q4, bias1[8 * cb + 2 * q4 + h], a.act1);
                        hb[4 * q4] = v.x; hb[4 * q4 + 1] = v.y; hb[4 * q4 + 2] = v.z; hb[4 * q4 + 3] = v.w;
                    }
                }
                // ---- layer 2, k-blocks 2 cb and 2 cb + 1: both products of h1_i straight from hb (the 32x32 accumulator layout IS the
                // B-operand layout of msplit.h) -------------------------------------------------------------------------------------------
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int kb = 2 * cb + s;
                    const Split8 b = split8(make_float4(hb[8 * s], hb[8 * s + 1], hb[8 * s + 2], hb[8 * s + 3]),
                                            make_float4(hb[8 * s + 4], hb[8 * s + 5], hb[8 * s + 6], hb[8 * s + 7]));
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        out[c] = split_mac(out[c], split_read_a(a2 + (2 * kb) * C2_SLAB + 32 * c, C2_UNITS2), b);
                        tacc[c] = split_mac(tacc[c], split_read_a(a2 + (2 * (8 + kb)) * C2_SLAB + 32 * c, C2_UNITS2), b);
                    }
                }
            }
            {
                float s = 0.0f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += tacc[c][r];
                bad |= (s != s);
            }
            C2_STAMP(2);
            // ---- sum_j t_j: 8 columns at a time through the two 2 KB halves of the pair's stage (rows of 32 bytes).  Chunk k + 1 is written
            // before chunk k is read: a wave waits for its partner's write of a chunk a whole round after that write was due --------------
            // events of the rounds, counted from here: W0 W1 R0 W2 R1 ... W7 R6 R7
            const int eb = __builtin_amdgcn_readfirstlane(pc.ev);
            // (an offset of zero hipcc cannot see through: it otherwise forms the LDS addresses of this phase once per launch, outside the
            // job loop, keeps each in a register of its own and spills them — a scratch reload, i.e. a memory round trip, per round)
            int opq;
            asm volatile("v_mov_b32 %0, 0" : "=v"(opq));
            unsigned char *stg = stage + opq;
            auto ev_w = [&](int k) { return eb + (k == 0 ? 1 : 2 * k); };
            auto ev_r = [&](int k) { return eb + (k == 7 ? 16 : 2 * k + 3); };
            auto put = [&](int k) {
                if (valid)           // (mirror lanes of empty slots do not write)
                    *reinterpret_cast<float4 *>(stg + 2048 * (k & 1) + slot * 32 + 16 * h) =
                        make_float4(tacc[k >> 2][4 * (k & 3)], tacc[k >> 2][4 * (k & 3) + 1], tacc[k >> 2][4 * (k & 3) + 2], tacc[k >> 2][4 * (k & 3) + 3]);
                if (two) pair_arrive(pc);
            };
            // (an event of a whole K loop ago: falls through.  Also on a ONE-tile job that follows a two-tile job: the partner may still be reading
            // chunk 6 / 7 of the previous job from the half put(0) writes; its counter is monotonic, so the wait cannot deadlock)
            if (ev_stage) pair_wait(pc, ev_stage);
            put(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k + 1 < 8) {
                    if (two && k >= 1) pair_wait(pc, ev_r(k - 1));      // the partner has read chunk k - 1: its half is free
                    put(k + 1);
                }
                // (W(k) precedes R(k - 1) in the partner's program: after the wait above only rounds 0 and 7 have to ask for the write)
                if (two) { if (k == 0 || k == 7) pair_wait(pc, ev_w(k)); } else c2_wave_sync();
                const unsigned char *sb = stg + 2048 * (k & 1);
                // NNlib.scatter(+): dst = 0, then dst += src for the edges in order
                float4 s0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (degmin >= 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        s0 = c2_add4(s0, *reinterpret_cast<const float4 *>(sb + ((locp[j >> 1] >> (16 * (j & 1))) & 0xffff)));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        s0 = c2_sel4(j < deg, c2_add4(s0, *reinterpret_cast<const float4 *>(sb + ((locp[j >> 1] >> (16 * (j & 1))) & 0xffff))), s0);
                }
                if (degmax > 4) {
#pragma unroll 1
                    for (int j = 4; j < deg; ++j)
                        s0 = c2_add4(s0, *reinterpret_cast<const float4 *>(sb + (slot + (a.col[beg + j] - rowc)) * 32 + 16 * h));
                }
                if (two) pair_arrive(pc); else c2_wave_sync();     // (release: the reads above have completed)
                if (a.mean_aggr && deg > 0) {
                    const float cnt = (float)deg;
                    s0 = make_float4(0.0f + s0.x / cnt, 0.0f + s0.y / cnt, 0.0f + s0.z / cnt, 0.0f + s0.w / cnt);
                }
                out[k >> 2][4 * (k & 3)] += s0.x; out[k >> 2][4 * (k & 3) + 1] += s0.y;
                out[k >> 2][4 * (k & 3) + 2] += s0.z; out[k >> 2][4 * (k & 3) + 3] += s0.w;
            }
            C2_STAMP(3);
            // ---- σ2, z = W_head[:, slab] * h2 ------------------------------------------------------------------------------------------
            // (the same opaque zero: ~20 addresses here, nine of them were scratch reloads)
            const float4 *bias2o = bias2 + opq + h;
            const float *heado = head + 4 * (opq + h);
            float zo[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) zo[o] = 0.0f;
            {
                float s = 0.0f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += out[c][r];
                bad |= (s != s);
            }
            // three LDS round trips instead of twenty-four: the eight bias pieces together, then per output the eight pieces of its head
            // row together (one uniform branch per output, not one per piece)
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = bias2o[(32 * (i >> 2) + 8 * (i & 3)) >> 2];      // (+ 4 h: inside bias2o)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = split_out4(out[i >> 2], i & 3, v[i], a.act2);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                if (o < a.nout) {
                    int oo;
                    asm volatile("v_mov_b32 %0, 0" : "=v"(oo));         // (keeps hipcc from starting every output's reads at once)
                    const float *hr = heado + 4 * oo + o * C2_SLAB;
                    float4 hw[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) hw[i] = *reinterpret_cast<const float4 *>(hr + 32 * (i >> 2) + 8 * (i & 3));
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        zo[o] = fmaf(hw[i].x, v[i].x, fmaf(hw[i].y, v[i].y, fmaf(hw[i].z, v[i].z, fmaf(hw[i].w, v[i].w, zo[o]))));
                }
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) zo[o] = zo[o] + __shfl_xor(zo[o], 32, 64);
            C2_STAMP(4);
            // a non-finite operand somewhere in this tile (NaN accumulators): the finish kernel redoes the whole job with fp32 loops (both
            // waves of a pair may report it: the job is then redone twice, with the same result)
            if (__builtin_amdgcn_ballot_w64(bad) != 0) {
                if (lane == 0) a.bad_list[atomicAdd(a.bad_count, 1)] = job * 2 + slab;
            }
            if (h == 0 && valid) {
                // this slab's z of the row: pooled over the member graph, in node order, by the finish kernel
                float *zr = a.zrows + ((int64_t)slab * a.N + rowc) * a.nout;
                if (a.nout == 2) {
                    *reinterpret_cast<float2 *>(zr) = make_float2(zo[0], zo[1]);
                } else {
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (o < a.nout) zr[o] = zo[o];
                }
            }
            C2_STAMP(5);
            if (two) ev_stage = ev_r(7);
        } else {
            // the odd wave of a one-tile job: nothing to compute, but the next job's entries are still due
            pre_a(job_next, nxt); pre_b(nxt); pre_c(nxt);
        }
        C2_STAMP(6);
        cur = nxt;
    }
}

}  // namespace gnnmp

using namespace gnnmp;

// ---- the job packing on the device (chain_pack.h): ONE block; no copy of the sizes to the host, no synchronisation ------------------------
namespace gnnmp {
constexpr int PK_THREADS = 1024, PK_WAVES = PK_THREADS / 64;

struct PackArgs {
    const int64_t *seg_ptr;
    int G;
    int njobs_cap;        // rows of tab that exist
    int32_t *tab;         // [njobs_cap][64]
    int32_t *sorted;      // [G] scratch: member graphs by decreasing size (ties by id)
    int32_t *hdr;         // out: [0] jobs, [1] tiles, [2] poison, [3] largest graph, [4] empty graphs, [5] graphs packed; [16..] phase stamps
    int32_t *bad;         // [3] the set-aside counters of the chain kernel (zeroed here)
    PackState *gst;       // the records and class lists for chain_expand_kernel (global copy of the packing block's LDS state)
};

// chain_pack.h's table accessor on the device: entry i (0..63) lives in lane i of one VGPR, entry 64 in a second (uniform) one; the wave
// runs pack_records_t in lockstep, every lane holding the same scalars
struct PackLaneArr {
    int32_t lo, hi;
    int lane;
    __device__ __forceinline__ int32_t get(int i) const {
        const int u = __builtin_amdgcn_readfirstlane(i);
        return u == PACK_ROWS ? hi : __builtin_amdgcn_readlane(lo, u);
    }
    __device__ __forceinline__ void set(int i, int32_t x) {
        const int u = __builtin_amdgcn_readfirstlane(i);
        if (u == PACK_ROWS) hi = x; else lo = lane == u ? x : lo;
    }
    __device__ __forceinline__ bool writer() const { return lane == 0; }
};

constexpr int PK_PER = 16;     // graphs per lane whose sizes stay in registers between the two passes of the counting sort: 16 waves x 64
                               // lanes x 16 = 16 384 member graphs; a larger batch reads its sizes twice

__global__ void __launch_bounds__(PK_THREADS) chain_pack_kernel(const PackArgs a) {
    __shared__ PackState st;
    __shared__ int32_t cnt[PACK_ROWS + 2], start[PACK_ROWS + 2], work[PACK_ROWS + 2];
    __shared__ int32_t wcnt[PK_WAVES][PACK_ROWS + 1];       // per wave: graphs of each size in its range, then the running output position
    __shared__ int32_t flags[2];                             // largest size seen, graphs without a node
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // phase stamps (100 MHz wall clock) in hdr[16..]: tools/experiments/pack_phases.py
    long long *stamp = reinterpret_cast<long long *>(a.hdr + 16);
#define PK_STAMP(i) do { if (tid == 0) stamp[i] = (long long)wall_clock64(); } while (0)
    PK_STAMP(0);
    for (int i = tid; i < PK_WAVES * (PACK_ROWS + 1); i += PK_THREADS) (&wcnt[0][0])[i] = 0;
    if (tid < 2) flags[tid] = 0;
    if (tid < 3) a.bad[tid] = 0;
    __syncthreads();
    // every wave owns a contiguous range of member graphs (whole steps of 64), PK_PER steps at a time: all loads of a chunk are issued
    // before the first is used (a lone block is bound by memory round trips, not by bytes)
    const int per = ((a.G + PK_WAVES - 1) / PK_WAVES + 63) & ~63;
    const int g0 = wave * per, g1 = min(a.G, g0 + per);
    const int nchunks = (per / 64 + PK_PER - 1) / PK_PER;
    int32_t R0[PK_PER], SZ[PK_PER];      // first row and size of this lane's graphs of the chunk (SZ = -1: no graph)
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < PK_PER; ++it) {
            const int g = g0 + (chunk * PK_PER + it) * 64 + lane;
            const bool v = g < g1;
            const int64_t b0 = v ? a.seg_ptr[g] : 0, b1 = v ? a.seg_ptr[g + 1] : 0;
            const int64_t d = b1 - b0;
            R0[it] = (int32_t)b0;
            SZ[it] = !v ? -1 : (d < 0 || d > 0x7ffffff0 ? 0x7ffffff0 : (int32_t)d);
        }
    };
    int mx = 0, empties = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        load_chunk(chunk);
#pragma unroll
        for (int it = 0; it < PK_PER; ++it) {
            const int sz = SZ[it];
            if (sz >= 0) { mx = max(mx, sz); empties += sz == 0; }
            if (sz >= 1 && sz <= PACK_ROWS) atomicAdd(&wcnt[wave][sz], 1);
        }
    }
    if (mx > 0) atomicMax(&flags[0], mx);
    if (empties) atomicAdd(&flags[1], empties);
    __syncthreads();
    PK_STAMP(1);
    if (tid >= 1 && tid <= PACK_ROWS) {
        int c = 0;
        for (int w = 0; w < PK_WAVES; ++w) c += wcnt[w][tid];
        cnt[tid] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        for (int s = PACK_ROWS; s >= 1; --s) { start[s] = pos; pos += cnt[s]; }
        start[0] = pos;                                       // = the number of graphs that are packed
    }
    __syncthreads();
    if (tid >= 1 && tid <= PACK_ROWS) {                        // counts -> first output position of (wave, size)
        int pos = start[tid];
        for (int w = 0; w < PK_WAVES; ++w) { const int c = wcnt[w][tid]; wcnt[w][tid] = pos; pos += c; }
    }
    __syncthreads();
    PK_STAMP(2);
    // stable counting sort: position = first position of (wave, size) + rank among the equal sizes of this step's lower lanes.  What is
    // sorted is the graph's FIRST ROW (its size is the record's): nothing else of a graph is needed to place it.
    {
        volatile int32_t *wpos = wcnt[wave];
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if (nchunks > 1) load_chunk(chunk);
#pragma unroll
            for (int it = 0; it < PK_PER; ++it) {
                if ((chunk * PK_PER + it) * 64 >= per) break;          // (uniform)
                const int sz = SZ[it];
                const bool ok = sz >= 1 && sz <= PACK_ROWS;
                unsigned long long eq = __builtin_amdgcn_ballot_w64(ok);
#pragma unroll
                for (int b = 0; b < 7; ++b) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64((sz >> b) & 1);
                    eq &= ((sz >> b) & 1) ? m : ~m;
                }
                if (ok) {
                    const int rank = __popcll(eq & ((1ull << lane) - 1ull));
                    const int base = wpos[sz];
                    a.sorted[base + rank] = R0[it];
                    if ((eq >> lane) == 1ull) wpos[sz] = base + __popcll(eq);      // the highest lane of the group moves the cursor
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
    PK_STAMP(3);
    if (wave == 0) {          // (the job table was set to -1 by a memset node before this launch)
        PackLaneArr c = {cnt[lane], cnt[PACK_ROWS], lane}, s0 = {start[lane], start[PACK_ROWS], lane};
        PackLaneArr room = {0, 0, lane}, exits = {0, 0, lane}, entries = {0, 0, lane};
        pack_records_t(c, s0, room, exits, entries, &st);
    }
    __threadfence();
    __syncthreads();
    PK_STAMP(4);
    if (tid <= PACK_ROWS) pack_class_count(&st, tid, &work[tid]);
    __syncthreads();
    if (tid == 0) {
        st.cls_start[0] = 0;
        for (int d = 0; d <= PACK_ROWS; ++d) st.cls_start[d + 1] = st.cls_start[d] + work[d];
    }
    __syncthreads();
    if (tid <= PACK_ROWS) pack_class_fill(&st, tid);
    __syncthreads();
    PK_STAMP(5);
    const int nitems = start[0];
    const bool poison = flags[0] > PACK_ROWS || flags[1] > 0 || st.njobs > a.njobs_cap;
    {
        // the records and the class lists go to memory for chain_expand_kernel (many blocks write the job table: one block's store
        // instructions alone took 58 us)
        const int nrec = st.nrec;
        int32_t *src = reinterpret_cast<int32_t *>(&st.rec[0]), *dst = reinterpret_cast<int32_t *>(&a.gst->rec[0]);
        for (int i = tid; i < nrec * (int)(sizeof(PackRec) / 4); i += PK_THREADS) dst[i] = src[i];
        for (int i = tid; i < nrec; i += PK_THREADS) a.gst->cls_list[i] = st.cls_list[i];
        if (tid < PACK_ROWS + 2) a.gst->cls_start[tid] = st.cls_start[tid];
        if (tid == 0) { a.gst->nrec = nrec; a.gst->njobs = st.njobs; a.gst->tiles = st.tiles; }
    }
    __syncthreads();
    PK_STAMP(6);
#undef PK_STAMP
    if (tid == 0) {
        a.hdr[0] = poison ? 0 : st.njobs;
        a.hdr[1] = st.tiles;
        a.hdr[2] = poison ? 1 : 0;
        a.hdr[3] = flags[0];
        a.hdr[4] = flags[1];
        a.hdr[5] = nitems;
    }
}

// The job table itself: a lane per packed graph finds (job, first slot) from the records (chain_pack.h: pack_place), then the WAVE writes
// each graph's rows with one store (lane i = row i of the graph).  Many blocks: the table is ~250 000 rows.
constexpr int PX_THREADS = 256;
__global__ void __launch_bounds__(PX_THREADS) chain_expand_kernel(const PackArgs a) {
    __shared__ PackState st;
    if (a.hdr[2] != 0) return;             // poisoned: nothing to place
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrec = a.gst->nrec;
    {
        const int32_t *src = reinterpret_cast<const int32_t *>(&a.gst->rec[0]);
        int32_t *dst = reinterpret_cast<int32_t *>(&st.rec[0]);
        for (int i = tid; i < nrec * (int)(sizeof(PackRec) / 4); i += PX_THREADS) dst[i] = src[i];
        for (int i = tid; i < nrec; i += PX_THREADS) st.cls_list[i] = a.gst->cls_list[i];
        if (tid < PACK_ROWS + 2) st.cls_start[tid] = a.gst->cls_start[tid];
        if (tid == 0) st.nrec = nrec;
    }
    __syncthreads();
    const int nitems = a.hdr[5];
    const int ngroups = (nitems + 63) / 64;
    for (int gq = blockIdx.x * (PX_THREADS / 64) + wave; gq < ngroups; gq += gridDim.x * (PX_THREADS / 64)) {
        const int p = gq * 64 + lane;
        int32_t job = 0, slot = 0, sz = 0;
        const int32_t r0 = p < nitems ? a.sorted[p] : 0;
        if (p < nitems) pack_place(&st, p, &job, &slot, &sz);
        const int dst = job * PACK_ROWS + slot;
        const int n_here = min(64, nitems - gq * 64);
        for (int i = 0; i < n_here; ++i) {
            const int d = __builtin_amdgcn_readlane(dst, i), r = __builtin_amdgcn_readlane(r0, i), n = __builtin_amdgcn_readlane(sz, i);
            if (lane < n) a.tab[(int64_t)d + lane] = r + lane;
        }
    }
}
}  // namespace gnnmp

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
    // best fit decreasing: graphs by decreasing size (ties by id: a counting sort, sizes are 1..64), each into the fullest job that
    // still takes it.  Flat arrays only — this runs once per BATCH of a training loop.
    std::vector<int32_t> order;
    order.reserve((size_t)G);
    {
        std::vector<int32_t> start(66, 0);
        for (int64_t g = 0; g < G; ++g) ++start[(size_t)(64 - (sp[(size_t)g + 1] - sp[(size_t)g])) + 1];    // bucket 0 = size 64
        for (int b = 0; b < 65; ++b) start[(size_t)b + 1] += start[(size_t)b];
        order.resize((size_t)G);
        for (int64_t g = 0; g < G; ++g) order[(size_t)start[(size_t)(64 - (sp[(size_t)g + 1] - sp[(size_t)g]))]++] = (int32_t)g;
    }
    std::vector<int32_t> job_of((size_t)G, -1);
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
            j = (int)room.size();
            room.push_back(64);
        }
        job_of[(size_t)g] = j;
        room[(size_t)j] -= sz;
        by_room[(size_t)room[(size_t)j]].push_back(j);
    }
    const int njobs = (int)room.size();
    std::vector<int32_t> tab((size_t)njobs * 64, -1);
    {
        std::vector<int> cursor((size_t)njobs, 0);          // member graphs in ascending id: the rows of a job come out in node order
        for (int64_t g = 0; g < G; ++g) {
            const int j = job_of[(size_t)g];
            if (j < 0) continue;
            int c = cursor[(size_t)j];
            for (int64_t r = sp[(size_t)g]; r < sp[(size_t)g + 1]; ++r, ++c) tab[(size_t)j * 64 + (size_t)c] = (int32_t)r;
            cursor[(size_t)j] = c;
        }
    }
    int64_t tiles = 0;
    for (int j = 0; j < njobs; ++j) tiles += (64 - room[(size_t)j]) > 32 ? 2 : 1;
    J->njobs = njobs;
    J->fill = tiles > 0 ? (double)J->N / (32.0 * (double)tiles) : 0.0;
    if (njobs > 0) {
        // one device block for the job table, the set-aside list and the z rows; a training loop builds a handle per batch, so freed
        // blocks are parked and handed out again (jobs_block_take) instead of three hipMalloc + three hipFree (each a device-wide
        // synchronisation) per batch.  (tools/experiments/batched_prep.py: the create is ~0.37 ms at G = 8192 — the copy of seg_ptr to the host,
        // the packing, the upload of the table; 0.8 ms while the packing sorted and kept a vector per job.)
        const size_t bytes = tab.size() * sizeof(int32_t);
        const size_t off_bad = (bytes + 255) & ~(size_t)255;
        const size_t off_z = (off_bad + sizeof(int32_t) * (size_t)(3 + 4 * njobs) + 255) & ~(size_t)255;
        const size_t total = off_z + sizeof(float) * (size_t)2 * (size_t)J->N * 8;
        if (!pool_take(&J->block, &J->block_bytes, total, stream)) {
            delete J;
            return fail(GNNMP_EALLOC, "chain_jobs_create: hipMalloc failed");
        }
        unsigned char *base = static_cast<unsigned char *>(J->block);
        J->tab = reinterpret_cast<int32_t *>(base);
        J->bad = reinterpret_cast<int32_t *>(base + off_bad);
        J->zrows = reinterpret_cast<float *>(base + off_z);
        J->njobs_cap = njobs;
        hipError_t e = hipMemsetAsync(J->bad, 0, 3 * sizeof(int32_t), stream);
        if (e == hipSuccess) e = hipMemcpyAsync(J->tab, tab.data(), bytes, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {                 // (the block goes back, the handle does not leak)
            pool_park(J->block, J->block_bytes, stream, false);
            delete J;
            return hip_fail(e, "chain_jobs_create: upload of the job table");
        }
    }
    *out = J;
    return GNNMP_OK;
}

extern "C" int gnnmp_chain_jobs_destroy(gnnmp_chain_jobs_t *J) {
    if (!J) return GNNMP_OK;
    pool_park(J->block, J->block_bytes, nullptr, false);
    delete J;
    return GNNMP_OK;
}

/* stream-ordered destroy: the handle's device block goes back to the pool behind the work enqueued on `stream` so far (gnnmp.h) */
extern "C" int gnnmp_chain_jobs_release(gnnmp_chain_jobs_t *J, gnnmp_stream_t stream) {
    if (!J) return GNNMP_OK;
    pool_park(J->block, J->block_bytes, (hipStream_t)stream, true);
    delete J;
    return GNNMP_OK;
}

/* The same packing ON THE DEVICE (chain_pack.h, chain_pack_kernel): one launch of one block, no copy to the host, no synchronisation.
 * The caller says what it knows on the host — every GNNGraph carries num_nodes as a host integer (GNNGraphs/src/gnngraph.jl:108-117): n_rows
 * (the batch's nodes), max_graph (its largest member), has_empty (a member without nodes).  max_graph > 64 or has_empty: no jobs (the
 * general kernel runs), exactly like gnnmp_chain_jobs_create.  Sizes that contradict the announcement poison the handle: the chain then
 * writes NaN logits (and gnnmp_chain_jobs_info, which synchronises, reports it). */
extern "C" int gnnmp_chain_jobs_pack(gnnmp_chain_jobs_t **out, const int64_t *seg_ptr, int64_t G, int64_t n_rows, int64_t max_graph,
                                     int has_empty, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out || G < 0 || n_rows < 0 || max_graph < 0 || (G > 0 && !seg_ptr)) return fail(GNNMP_EINVAL, "chain_jobs_pack: bad argument");
    *out = nullptr;
    gnnmp_chain_jobs *J = new gnnmp_chain_jobs();
    J->G = G;
    J->N = n_rows;
    J->max_graph = max_graph;
    J->has_empty = has_empty ? 1 : 0;
    if (max_graph > 64 || has_empty || n_rows >= ((int64_t)1 << 31) || G >= ((int64_t)1 << 31) || G == 0 || n_rows == 0) {
        *out = J;                                    // no jobs: the caller's chain runs on the general kernel
        return GNNMP_OK;
    }
    // any best fit leaves at most one job half empty: jobs <= 2 N / 64 + 1; and every job holds a graph
    const int64_t cap = std::min<int64_t>(G, n_rows / 32 + 2);
    const size_t b_tab = (size_t)cap * PACK_ROWS * sizeof(int32_t);
    const size_t off_bad = (b_tab + 255) & ~(size_t)255;
    const size_t off_hdr = (off_bad + sizeof(int32_t) * (size_t)(3 + 4 * cap) + 255) & ~(size_t)255;
    const size_t off_sorted = off_hdr + 256;
    const size_t off_pst = (off_sorted + sizeof(int32_t) * (size_t)G + 255) & ~(size_t)255;
    const size_t off_z = (off_pst + sizeof(PackState) + 255) & ~(size_t)255;
    const size_t total = off_z + sizeof(float) * (size_t)2 * (size_t)n_rows * 8;
    if (!pool_take(&J->block, &J->block_bytes, total, stream)) {
        delete J;
        return fail(GNNMP_EALLOC, "chain_jobs_pack: hipMalloc of %zu bytes failed", total);
    }
    unsigned char *base = static_cast<unsigned char *>(J->block);
    J->tab = reinterpret_cast<int32_t *>(base);
    J->bad = reinterpret_cast<int32_t *>(base + off_bad);
    J->hdr = reinterpret_cast<int32_t *>(base + off_hdr);
    J->zrows = reinterpret_cast<float *>(base + off_z);
    J->njobs = -1;
    J->njobs_cap = (int)cap;
    PackArgs a = {};
    a.seg_ptr = seg_ptr;
    a.G = (int)G;
    a.njobs_cap = (int)cap;
    a.tab = J->tab;
    a.sorted = reinterpret_cast<int32_t *>(base + off_sorted);
    a.hdr = J->hdr;
    a.bad = J->bad;
    a.gst = reinterpret_cast<PackState *>(base + off_pst);
    hipError_t e = hipMemsetAsync(J->tab, 0xff, b_tab, stream);        // every slot empty (-1): a full-chip fill, not one block's stores
    if (e == hipSuccess) {
        chain_pack_kernel<<<1, PK_THREADS, 0, stream>>>(a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(128, (G + 255) / 256));
        chain_expand_kernel<<<blocks, PX_THREADS, 0, stream>>>(a);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        pool_park(J->block, J->block_bytes, stream, true);
        delete J;
        return hip_fail(e, "chain_pack_kernel");
    }
    *out = J;
    return GNNMP_OK;
}

/* the job table as the chain kernel sees it: tab_out[cap_rows][64] int32 (device), hdr_out[8] int32 (device; host-packed handles: [0] = jobs,
 * rest 0).  Tests and debugging: asynchronous copies on `stream`. */
extern "C" int gnnmp_chain_jobs_export(const gnnmp_chain_jobs_t *J, int32_t *tab_out, int64_t cap_rows, int32_t *hdr_out,
                                       gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!J || cap_rows < 0) return fail(GNNMP_EINVAL, "chain_jobs_export: bad argument");
    const int64_t rows = std::min<int64_t>(cap_rows, J->njobs_cap);
    if (tab_out && rows > 0)
        GNNMP_HIP(hipMemcpyAsync(tab_out, J->tab, sizeof(int32_t) * PACK_ROWS * (size_t)rows, hipMemcpyDeviceToDevice, stream));
    if (hdr_out) {
        if (J->hdr) {
            GNNMP_HIP(hipMemcpyAsync(hdr_out, J->hdr, sizeof(int32_t) * 32, hipMemcpyDeviceToDevice, stream));
        } else {
            const int32_t h[32] = {J->njobs, 0, 0, (int32_t)J->max_graph, J->has_empty};
            GNNMP_HIP(hipMemcpyAsync(hdr_out, h, sizeof(h), hipMemcpyHostToDevice, stream));
            GNNMP_HIP(hipStreamSynchronize(stream));
        }
    }
    return GNNMP_OK;
}

/* info[0] = jobs, info[1] = member graphs, info[2] = rows, info[3] = largest member graph, info[4] = per-mille of the MFMA tiles' rows
 * that are real rows */
extern "C" int gnnmp_chain_jobs_info(const gnnmp_chain_jobs_t *J, int64_t *info) {
    if (!J || !info) return fail(GNNMP_EINVAL, "chain_jobs_info: null pointer");
    int64_t njobs = J->njobs;
    double fill = J->fill;
    if (J->hdr) {          // packed on the device: read the header back (synchronises the device: a query, not part of a step)
        int32_t h[8] = {0};
        GNNMP_HIP(hipDeviceSynchronize());
        GNNMP_HIP(hipMemcpy(h, J->hdr, sizeof(h), hipMemcpyDeviceToHost));
        if (h[2] != 0)
            return fail(GNNMP_EINVAL, "chain_jobs_pack: the batch holds a member graph of %d nodes / %d empty member graphs, its caller "
                                      "announced max_graph = %lld, has_empty = %d", h[3], h[4], (long long)J->max_graph, J->has_empty);
        njobs = h[0];
        fill = h[1] > 0 ? (double)J->N / (32.0 * (double)h[1]) : 0.0;
    }
    info[0] = njobs; info[1] = J->G; info[2] = J->N; info[3] = J->max_graph; info[4] = (int64_t)(fill * 1000.0 + 0.5);
    return GNNMP_OK;
}

#ifdef GNNMP_CHAIN_TRACE
static long long *g_chain_trace = nullptr;
extern "C" int gnnmp_chain_trace(long long *host) {     // 16 waves x 8 jobs x 8 stamps
    if (!g_chain_trace) return 1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(host, g_chain_trace, sizeof(long long) * 16 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

namespace gnnmp {
// Returns GNNMP_OK if it launched, 1 if the chain / the batch is outside this kernel's envelope (graph_chain.hip's kernel runs).
int graph_chain2_try(gnnmp_graph_t *p, const gnnmp_chain_jobs_t *J, const int64_t *seg_ptr, int64_t G, const float *x, int n_layers,
                     const int64_t *dims, const float *const *W_root, const float *const *W_agg, const float *const *bias,
                     const int *act, int w_layout, int aggr, int pool_aggr, const float *W_head, const float *b_head, int64_t nout,
                     float *out, hipStream_t stream) {
    if (!J || J->njobs_cap <= 0 || J->has_empty || J->G != G || J->N != p->n_dst) return 1;
    if (n_layers != 2 || dims[0] != C2_D0 || dims[1] != C2_D1 || dims[2] != C2_D2 || nout > 8) return 1;
    if (knob(KNOB_CHAIN) == 1) return 1;    // 1 = the general kernel only (A/B runs)
    if ((reinterpret_cast<uintptr_t>(x) & 15)) return 1;
    Chain2Args a = {};
    a.rowptr = p->rowptr;
    a.col = p->col;
    a.job_rows = J->tab;
    a.njobs = J->hdr ? J->njobs_cap : J->njobs;      // (device-packed: the bound the grid is sized for; the kernel reads hdr[0])
    a.hdr = J->hdr;
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
    const unsigned parity = const_cast<gnnmp_chain_jobs_t *>(J)->calls.fetch_add(1u, std::memory_order_relaxed) & 1u;
    a.bad_count = J->bad + parity;
    a.bad_reset = J->bad + (parity ^ 1u);
    a.bad_list = J->bad + 3;
    a.zrows = J->zrows;
    a.N = (int)J->N;
#ifdef GNNMP_CHAIN_TRACE
    if (!g_chain_trace) { (void)hipMalloc(&g_chain_trace, sizeof(long long) * 16 * 8 * 8); (void)hipMemset(g_chain_trace, 0, sizeof(long long) * 16 * 8 * 8); }
    a.trace = g_chain_trace;
#endif
    a.G = (int)G;
    // knob 19 (A/B runs): 1 = 8 waves a block (4 pairs, up to 256 registers a wave) instead of 12 (6 pairs, 168 registers)
    const int waves = (knob(KNOB_VARIANT) & 3) == 1 ? 8 : 12;
    const size_t lds = (size_t)3 * C2_UNITS1 * 16 + (size_t)3 * C2_UNITS2 * 16 + C2_D1 * 4 + C2_SLAB * 4 + 8 * C2_SLAB * 4 +
                       (size_t)(waves / 2) * C2_STAGE_BYTES + 4 * 2 * (size_t)waves;
    GNNMP_LDS_OPTIN("graph_chain2_kernel<768>", &graph_chain2_kernel<768>);
    GNNMP_LDS_OPTIN("graph_chain2_kernel<512>", &graph_chain2_kernel<512>);
    const int cus = device_cus();
    const int gx = std::max(1, std::min(cus / 2, (a.njobs + waves / 2 - 1) / (waves / 2)));     // one block a CU: half of them per slab
    const dim3 grid((unsigned)gx, 2);
    if (waves == 12) graph_chain2_kernel<768><<<grid, 768, lds, stream>>>(a);
    else graph_chain2_kernel<512><<<grid, 512, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("graph_chain2_kernel");
    graph_chain2_finish_kernel<<<(unsigned)((2 * G + 255) / 256), 256, 0, stream>>>(a);      // a lane per (member graph, slab)
    GNNMP_LAUNCH_CHECK("graph_chain2_finish_kernel");
    return GNNMP_OK;
}
}  // namespace gnnmp
