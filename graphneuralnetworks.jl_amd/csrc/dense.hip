// dense.hip — the dense feature contractions of the four layer bodies, the only MFMA work on the path:
//   weight * x                      GNNlib/src/layers/conv.jl:39,69   (gcn_conv)
//   weight1 * xi .+ weight2 * m     :106                              (graph_conv)
//   weight * vcat(xi, m)            :281                              (sage_conv; never materialises the vcat)
//   dense_x(x)                      :127                              (gat_conv)
// plus `.+ bias` and σ fused in the epilogue (:71,:107,:147,:281).
//
// Shape: tall-skinny, out[N][Dout] with N ~ 1e5..1e7 and K, Dout ~ 16..256.  fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain in k order; 157 TF peak = the fp32 vector rate, but it leaves the VALU
// free and needs one VGPR per operand).  Two kernels:
//   dense_wlds_kernel  (default) W^T resident in LDS for the lifetime of a persistent block, wave-private x staging, no
//                      workgroup barrier in the main loop — used whenever W^T for a 128-column tile plus >= 4 wave
//                      regions fit the 160 KB LDS (all layer shapes of the bench configs);
//   dense_mfma_kernel  the K-chunked fallback (128 x 128 block tile, x and W^T chunks of 32 k-values through LDS with
//                      two barriers per chunk) for large K (e.g. Cora's 1433 input features) and tiny N.
// Both use padded / odd leading dimensions so that the ds_read_b32 of either MFMA operand is bank-conflict free.
#include <algorithm>

#include "common.h"

namespace gnnmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, KC = 32;
constexpr int XS_LD = KC + 1;   // bank = (row + k) % 32
constexpr int WS_LD = BN + 1;   // bank = (k + j) % 32

struct DenseArgs {
    const float *x[2];
    const float *W[2];
    int K[2];
    int ldw[2];
    int nseg;
    int w_layout;  // 0: W[Dout][K] (row j contiguous in k), 1: W[K][Dout] (Julia column-major (Dout, K))
    const float *bias;
    int act;
    float *out;
    int64_t N;
    int Dout;
};

__global__ void __launch_bounds__(256) dense_mfma_kernel(const DenseArgs a) {
    __shared__ float xs[BM * XS_LD];
    __shared__ float ws[KC * WS_LD];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int ncols = min(BN, a.Dout - n0);
    const int ntiles = (ncols + 31) >> 5;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    for (int seg = 0; seg < a.nseg; ++seg) {
        const float *__restrict__ x = a.x[seg];
        const float *__restrict__ W = a.W[seg];
        const int K = a.K[seg];
        const int ldw = a.ldw[seg];
        const bool xvec = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        for (int k0 = 0; k0 < K; k0 += KC) {
            // ---- stage x[m0 : m0+128][k0 : k0+32] ----
            if (xvec) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = t + 256 * r;
                    const int row = idx >> 3;
                    const int c4 = (idx & 7) * 4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m0 + row < a.N && k0 + c4 < K)
                        v = *reinterpret_cast<const float4 *>(x + (m0 + row) * K + k0 + c4);
                    float *d = xs + row * XS_LD + c4;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
            } else {
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int idx = t + 256 * r;
                    const int row = idx >> 5;
                    const int c = idx & 31;
                    float v = 0.f;
                    if (m0 + row < a.N && k0 + c < K) v = x[(m0 + row) * K + k0 + c];
                    xs[row * XS_LD + c] = v;
                }
            }
            // ---- stage W^T chunk: ws[k][j] = W(j = n0 + j, k = k0 + k) ----
            if (a.w_layout == 0) {
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int idx = t + 256 * r;
                    const int k = idx & 31;
                    const int j = idx >> 5;
                    float v = 0.f;
                    if (j < ncols && k0 + k < K) v = W[(int64_t)(n0 + j) * ldw + k0 + k];
                    ws[k * WS_LD + j] = v;
                }
            } else {
#pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int idx = t + 256 * r;
                    const int j = idx & 127;
                    const int k = idx >> 7;
                    float v = 0.f;
                    if (j < ncols && k0 + k < K) v = W[(int64_t)(k0 + k) * ldw + n0 + j];
                    ws[k * WS_LD + j] = v;
                }
            }
            __syncthreads();
            // ---- 16 k-steps of 2 ----
            const float *xa = xs + (wave * 32 + (lane & 31)) * XS_LD + (lane >> 5);
            const float *wb = ws + (lane >> 5) * WS_LD + (lane & 31);
#pragma unroll 4
            for (int kk = 0; kk < KC; kk += 2) {
                const float av = xa[kk];
                const float *wk = wb + kk * WS_LD;
                if (ntiles > 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wk[0], acc[0], 0, 0, 0);
                if (ntiles > 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wk[32], acc[1], 0, 0, 0);
                if (ntiles > 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wk[64], acc[2], 0, 0, 0);
                if (ntiles > 3) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wk[96], acc[3], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
    const int colb = lane & 31;
    const int rowb = 4 * (lane >> 5);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        if (nt >= ntiles) break;
        const int col = n0 + nt * 32 + colb;
        if (col >= a.Dout) continue;
        const float b = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + rowb;
            if (row < a.N) {
                float v = acc[nt][r];
                if (a.bias) v = v + b;
                if (a.act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;  // NNlib.relu = ifelse(x < 0, 0, x) (NaN-preserving)
                a.out[row * a.Dout + col] = v;
            }
        }
    }
}

// ---- W-resident variant ------------------------------------------------------------------------------------------
// For the layer shapes of the hot path (K, Dout <= a few hundred) the whole W^T fits in the CU's 160 KB LDS.  Each
// persistent block loads it ONCE; after that no workgroup barrier is needed: every wave streams its own 32-row tiles of
// x (a contiguous 32*K*4-byte block of HBM: perfectly coalesced 16-byte loads) through a wave-private LDS region laid out
// for the MFMA A operand, runs K/2 k-steps of NT back-to-back v_mfma_f32_32x32x2_f32 (NT independent accumulator
// chains keep the matrix pipe issuing every 64 cycles), and writes the finished 32 x ncols tile back through the same
// region as whole rows (one contiguous block when the tile spans all output columns).  With two waves per SIMD one
// wave's staging hides under the other's MFMAs.
struct DenseWArgs {
    DenseArgs d;
    int n0;        // first output column of this launch's column tile (blockIdx.y adds NT*32)
    int waves;     // waves per block
    int xld;       // leading dimension of the wave-private x image (odd)
    int old_;      // leading dimension of the wave-private output image (multiple of 4)
    int skew;      // s_sleep(127) repetitions for waves 4-7 before their first tile (0 = none)
    int token;     // 1 = serialise the k-loops of the two waves of a SIMD with an LDS token
    int prefetch;  // 1 = cross-tile register prefetch of the next x block (one-segment, whole-K staging)
    int tp;        // output column tiles per epilogue pass
    int ks;        // columns of x staged per k-chunk (multiple of 4; = K rounded up when the whole tile fits)
    int region;    // floats per wave region
    int ktot_pad;  // rows of W^T image
};

template <int NT>
__global__ void __launch_bounds__(512) dense_wlds_kernel(const DenseWArgs w) {
    extern __shared__ __align__(16) float lds[];
    const DenseArgs &a = w.d;
    constexpr int WLD = NT * 32 + 1;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int nthreads = w.waves * 64;
    const int n0 = w.n0 + (int)blockIdx.y * NT * 32;
    const int ncols = min(NT * 32, a.Dout - n0);
    float *Wt = lds;
    float *reg = lds + (size_t)w.ktot_pad * WLD + (size_t)wave * w.region;
    // one matrix-pipe token per SIMD (waves i and i + 4 of an 8-wave block share a SIMD), after the wave regions
    int *tok = reinterpret_cast<int *>(lds + (size_t)w.ktot_pad * WLD + (size_t)w.waves * w.region) + (wave & 3);
    const bool paired = w.waves == 8 && w.token != 0;
    if (t < 4) reinterpret_cast<int *>(lds + (size_t)w.ktot_pad * WLD + (size_t)w.waves * w.region)[t] = 0;

    // ---- W^T image, once per block: Wt[koff + k][j] = W(n0 + j, k), zero where j >= ncols or k >= K ----
    // Zero fill first, then only the valid (j, k) pairs: the loaded value goes to LDS as it is.  (Masking it with
    // `valid ? lv : 0` made hipcc sink every load under its own branch with its own s_waitcnt vmcnt(0): 8 serialised HBM
    // round trips per iteration, ~35 us per launch.)
    for (int i = t; i < w.ktot_pad * WLD; i += nthreads) Wt[i] = 0.0f;
    __syncthreads();
    {
        int koff = 0;
        for (int seg = 0; seg < a.nseg; ++seg) {
            const int K = a.K[seg], Kp = (K + 1) & ~1;
            const float *__restrict__ W = a.W[seg];
            const int64_t sj = a.w_layout == 0 ? a.ldw[seg] : 1;   // element strides of W(j, k): one load expression
            const int64_t sk = a.w_layout == 0 ? 1 : a.ldw[seg];  // for both layouts (no per-element branch)
            const int total = K * ncols;
            constexpr int WB = 8;  // independent loads in flight per thread
            for (int idx0 = t; idx0 < total; idx0 += nthreads * WB) {
                float v[WB];
                int dst[WB];
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int idx = idx0 + u * nthreads;
                    const int idc = min(idx, total - 1);
                    int jj, k;
                    if (a.w_layout == 0) { jj = idc / K; k = idc - jj * K; }       // consecutive threads walk k (contiguous in W)
                    else                 { k = idc / ncols; jj = idc - k * ncols; }
                    dst[u] = idx < total ? (koff + k) * WLD + jj : -1;
                    v[u] = W[(int64_t)(n0 + jj) * sj + (int64_t)k * sk];
                }
#pragma unroll
                for (int u = 0; u < WB; ++u)
                    if (dst[u] >= 0) Wt[dst[u]] = v[u];
            }
            koff += Kp;
        }
    }
    __syncthreads();

    const int64_t n_tiles = (a.N + 31) / 32;
    const int XLD = w.xld, OLD = w.old_;
    // this lane's bias entries (its accumulator columns never change): loaded once, not once per tile
    float bcol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int c = min(nt * 32 + (lane & 31), ncols - 1);
        bcol[nt] = a.bias ? a.bias[n0 + c] : 0.0f;
    }
    // Phase skew.  PMC (SQ_WAIT_INST_ANY = exactly 2 x the MFMA time per wave) showed the two waves of a SIMD running
    // in LOCKSTEP: both in the k-loop (alternating on the matrix pipe), then both staging (pipe idle) — 50 % pipe
    // utilisation.  Delaying the second wave of every SIMD (waves 4-7 of an 8-wave block) by about half a tile period
    // puts one wave's staging under the other's MFMAs.  Performance only: nothing depends on the timing.
    if (w.skew > 0 && w.waves == 8 && __builtin_amdgcn_readfirstlane(wave) >= 4) {
        for (int i = 0; i < w.skew; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int64_t tile_stride = (int64_t)gridDim.x * w.waves;
    // Cross-tile prefetch (one segment, whole-K staging, 16-byte-aligned rows): the NEXT full tile's 32 x K block is loaded
    // into registers before this tile's k-loop starts, so its HBM round trip hides under this wave's own MFMAs (products
    // 100 => 100: 740 -> 700 us).  The 64 prefetch registers double as the staging batch of the non-prefetch path — kept
    // apart, the kernel needed 270 VGPRs and spilled (-20 %).
    constexpr int PF = 16;                                   // float4 per lane: 32 rows x 128 columns at most
    const bool pf_on = w.prefetch && a.nseg == 1 && w.ks >= ((a.K[0] + 1) & ~1) && (a.K[0] & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(a.x[0]) & 15) == 0);
    float4 pfv[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) pfv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int pf_l2 = 0;
    while ((1 << pf_l2) < (a.K[0] >> 2)) ++pf_l2;
    const int pf_rpi = 64 >> pf_l2, pf_c4 = lane & ((1 << pf_l2) - 1), pf_rsub = lane >> pf_l2;
    const int pf_nit = pf_rpi >= 32 ? 1 : 32 / pf_rpi;
    const bool pf_lane = pf_c4 < (a.K[0] >> 2) && pf_rsub < 32;
    const int64_t pf_off = (int64_t)min(pf_rsub, 31) * a.K[0] + (pf_lane ? pf_c4 : 0) * 4;
#define GNNMP_PF_LOAD(T)                                                                                         \
    {                                                                                                            \
        const float *xl_ = a.x[0] + (T) * 32 * (int64_t)a.K[0] + pf_off;                                         \
        _Pragma("unroll") for (int u = 0; u < PF; ++u)                                                           \
            pfv[u] = *reinterpret_cast<const float4 *>(xl_ + (int64_t)min(u, pf_nit - 1) * pf_rpi * a.K[0]);     \
    }
    bool pf_have = false;
    {
        const int64_t t0 = (int64_t)blockIdx.x * w.waves + wave;
        if (pf_on && t0 < n_tiles && (t0 + 1) * 32 <= a.N) {
            GNNMP_PF_LOAD(t0)
            pf_have = true;
        }
    }
    for (int64_t tile = (int64_t)blockIdx.x * w.waves + wave; tile < n_tiles; tile += tile_stride) {
        const int64_t m0 = tile * 32;
        const int rows = (int)min<int64_t>(32, a.N - m0);
        f32x16 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        int koff = 0;
        for (int seg = 0; seg < a.nseg; ++seg) {
            const int K = a.K[seg], Kp = (K + 1) & ~1;
            const float *__restrict__ x = a.x[seg] + m0 * K;
            const bool vec = rows == 32 && (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
            // The 32 x K tile goes through the wave's region in k-chunks of w.ks columns (w.ks = K when the whole image
            // fits; smaller chunks buy the LDS for two waves per SIMD — measured 148 -> see LABNOTES.md on the arxiv shape).
            for (int kc0 = 0; kc0 < Kp; kc0 += w.ks) {
                const int kcn = min(w.ks, K - kc0);          // real columns in this chunk (may be odd at the tail)
                const int kcp = min(w.ks, Kp - kc0);         // even number of k-steps' worth
                // ---- stage x[m0 : m0+32][kc0 : kc0+kcn] into the A-operand image xs[row][k] ----
                if (pf_have) {
                    // this tile's block is already in registers: image it, then put the next full tile's loads in flight
                    float *dl = reg + pf_rsub * XLD + pf_c4 * 4;
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        if (pf_lane && u < pf_nit) {
                            float *d = dl + u * pf_rpi * XLD;
                            d[0] = pfv[u].x; d[1] = pfv[u].y; d[2] = pfv[u].z; d[3] = pfv[u].w;
                        }
                    }
                    const int64_t nx = tile + tile_stride;
                    pf_have = nx < n_tiles && (nx + 1) * 32 <= a.N;
                    if (pf_have) GNNMP_PF_LOAD(nx)
                } else if (vec && (kcn & 3) == 0) {
                    // batches of SB independent 16-byte loads per lane, THEN the LDS writes: a plain load->write loop
                    // serialises one HBM round trip per iteration (measured: waves parked 42 % of their cycles)
                    // Lane -> (row, column) by shifts, no integer division: lpr = 2^k >= q lanes cover one row's q float4
                    // (25 of 32 active at K = 100), 64 / lpr rows per load instruction.  (The linear i -> (i / q, i % q)
                    // mapping cost ~40 VALU per element: 2 200 non-MFMA instructions per 200 MFMAs, PMC in profiles/.)
                    constexpr int SB = 16;                   // the whole 32 x 128 chunk in ONE batch: one HBM round trip per tile
                    const int q = kcn >> 2;                  // float4 per row in this chunk (<= 32: ks <= 128)
                    int l2 = 0;
                    while ((1 << l2) < q) ++l2;
                    const int rpi = 64 >> l2;                // rows per instruction
                    const int c4 = lane & ((1 << l2) - 1);
                    const int rsub = lane >> l2;
                    const bool on = c4 < q && rsub < 32;     // (rpi = 64 when q = 1: only the first 32 lanes hold rows)
                    const float *xl = x + (int64_t)min(rsub, 31) * K + kc0 + (c4 < q ? c4 : 0) * 4;
                    float *dl = reg + rsub * XLD + c4 * 4;
                    const int nit = rpi >= 32 ? 1 : 32 / rpi;
                    static_assert(SB == PF, "the staging batch lives in the prefetch registers (idle on this path)");
                    float4 (&v)[PF] = pfv;
                    for (int it0 = 0; it0 < nit; it0 += SB) {
#pragma unroll
                        for (int u = 0; u < SB; ++u) {       // unconditional (clamped) loads: no per-element branch + wait
                            const int it = min(it0 + u, nit - 1);
                            v[u] = *reinterpret_cast<const float4 *>(xl + (int64_t)it * rpi * K);
                        }
#pragma unroll
                        for (int u = 0; u < SB; ++u) {
                            if (on && it0 + u < nit) {
                                float *d = dl + (it0 + u) * rpi * XLD;
                                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                            }
                        }
                    }
                } else {
                    const int total = 32 * kcn;
                    for (int e = lane; e < total; e += 64) {
                        const int row = e / kcn, k = e - row * kcn;
                        reg[row * XLD + k] = row < rows ? x[(int64_t)row * K + kc0 + k] : 0.0f;
                    }
                }
                if (kcn < kcp) {                             // odd K: the last k-step's second column is zero
                    if (lane < 32) reg[lane * XLD + kcn] = 0.0f;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // ---- kcp/2 k-steps, software-pipelined by hand: the operands of k-step kk+2 are read from LDS before
                // the MFMAs of k-step kk issue (hipcc leaves the plain loop read -> wait -> 2 MFMA -> read -> wait -> 2 MFMA)
                const float *xa = reg + (lane & 31) * XLD + (lane >> 5);
                const float *wb = Wt + (koff + kc0 + (lane >> 5)) * WLD + (lane & 31);
                // Matrix-pipe token.  PMC showed the two waves of a SIMD in LOCKSTEP (SQ_WAIT_INST_ANY = exactly twice the
                // MFMA time per wave: both alternate on the pipe during their k-loops, then both stage while the pipe idles:
                // 50 % utilisation).  Taking a per-SIMD token around the k-loop serialises the two k-loops, so each runs at
                // the full issue rate while the partner stages / stores: the pair falls into alternation by itself.
                if (paired) {
                    if (lane == 0) {
                        while (atomicCAS(tok, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // Three operand sets in rotation (no moves): the reads of k-step s + 2 are issued before the MFMAs of k-step s,
                // i.e. every LDS read has two MFMA groups (~512 cycles) to return — the other waves of the CU write their x
                // images and output tiles through the same LDS, and one group of slack was not always enough.
                // sched_barrier(0) pins that order: left alone, hipcc's scheduler sinks each read group to just before its
                // first use and folds the register sets into one (ISA: ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma three times
                // per iteration).
                const int nsteps = kcp >> 1;
#define GNNMP_RD(A, B, STEP)                                                        \
    {                                                                               \
        const int st_ = min((STEP), nsteps - 1) * 2;                                \
        const float *wk_ = wb + st_ * WLD;                                          \
        A = xa[st_];                                                                \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) B[nt] = wk_[nt * 32];     \
    }
#define GNNMP_MM(A, B)                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                   \
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B[nt], acc[nt], 0, 0, 0);                     \
    __builtin_amdgcn_sched_barrier(0);
                float a0, a1, a2;
                float b0[NT], b1[NT], b2[NT];
                GNNMP_RD(a0, b0, 0)
                GNNMP_RD(a1, b1, 1)
                int st = 0;
                for (; st + 3 <= nsteps; st += 3) {
                    GNNMP_RD(a2, b2, st + 2)
                    GNNMP_MM(a0, b0)
                    GNNMP_RD(a0, b0, st + 3)
                    GNNMP_MM(a1, b1)
                    GNNMP_RD(a1, b1, st + 4)
                    GNNMP_MM(a2, b2)
                }
                if (st < nsteps) { GNNMP_MM(a0, b0) }          // set 0 holds step st, set 1 step st + 1
                if (st + 1 < nsteps) { GNNMP_MM(a1, b1) }
#undef GNNMP_RD
#undef GNNMP_MM
                if (paired) {
                    if (lane == 0) atomicExch(tok, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            koff += Kp;
        }
        // ---- epilogue: bias + act, then whole rows through the region ----
        // The region is sized for the x image; the output tile goes through it in passes of w.tp column tiles.
        const int colb = lane & 31, rowb = 4 * (lane >> 5);
        const bool vec_ok = (a.Dout & 3) == 0 && (n0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
        for (int c_lo = 0; c_lo < ncols; c_lo += w.tp * 32) {
            const int c_hi = min(ncols, c_lo + w.tp * 32);
            const int pc = c_hi - c_lo;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = nt * 32 + colb;
                if (c >= c_lo && c < c_hi) {
                    const float b = bcol[nt];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[nt][r];
                        if (a.bias) v = v + b;
                        if (a.act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
                        reg[((r & 3) + 8 * (r >> 2) + rowb) * OLD + (c - c_lo)] = v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float *o = a.out + m0 * a.Dout + n0 + c_lo;
            if (vec_ok && (pc & 3) == 0) {
                const int c4n = pc >> 2;                      // <= 32 float4 per row: same shift mapping as the staging
                int l2 = 0;
                while ((1 << l2) < c4n) ++l2;
                const int rpi = 64 >> l2;
                const int c4 = lane & ((1 << l2) - 1);
                const int rsub = lane >> l2;
                if (c4 < c4n) {
                    for (int row = rsub; row < rows; row += rpi) {
                        const float4 v = *reinterpret_cast<const float4 *>(reg + row * OLD + c4 * 4);
                        *reinterpret_cast<float4 *>(o + (int64_t)row * a.Dout + c4 * 4) = v;
                    }
                }
            } else {
                const int total = rows * pc;
                for (int i = lane; i < total; i += 64) {
                    const int row = i / pc, c = i - row * pc;
                    o[(int64_t)row * a.Dout + c] = reg[row * OLD + c];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

template <int NT>
static int launch_wlds(const DenseWArgs &w, size_t lds_bytes, int col_tiles, int64_t n_row_tiles, hipStream_t stream) {
    GNNMP_LDS_OPTIN("dense_wlds_kernel", &dense_wlds_kernel<NT>);
    const int cus = device_cus();
    int64_t gx = (n_row_tiles + w.waves - 1) / w.waves;
    if (gx > cus) gx = cus;  // one persistent block per CU (the LDS image allows no more)
    dim3 grid((unsigned)gx, (unsigned)col_tiles);
    dense_wlds_kernel<NT><<<grid, 64 * w.waves, lds_bytes, stream>>>(w);
    GNNMP_LAUNCH_CHECK("dense_wlds_kernel");
    return GNNMP_OK;
}
// ---- narrow outputs (a classifier head: Dense(128 => 2) after the pool) ---------------------------------------------------
// Dout <= 8: no matrix core has anything to do (a 16-wide tile would be 7/8 padding and the whole product is a few MFLOP); the
// MFMA kernels above spend their time staging W.  Eight lanes per row: lane l of the group reads the row's float4s l, l + 8, ...
// (128 contiguous bytes per group and step), multiplies them into Dout running dot products against W read through L1 (every
// group reads the same few KB), the eight partial sums meet in a DPP butterfly, lane 0 adds the bias and stores.
template <int NOUT>
__global__ void __launch_bounds__(256) dense_narrow_kernel(const float *__restrict__ x1, const float *__restrict__ W1, int K1,
                                                           int64_t sj1, int64_t sk1, const float *__restrict__ x2,
                                                           const float *__restrict__ W2, int K2, int64_t sj2, int64_t sk2,
                                                           const float *__restrict__ bias, int act, float *__restrict__ out,
                                                           int64_t N, int Dout) {
    const int lane8 = threadIdx.x & 7;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int64_t r = row < N ? row : N - 1;
    float acc[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) acc[j] = 0.0f;
    auto segment = [&](const float *x, const float *W, int K, int64_t sj, int64_t sk) {
        const float *xr = x + r * K;
        for (int k = 4 * lane8; k < K; k += 32) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + k);
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                if (j < Dout) {
                    const float *w = W + (int64_t)j * sj + (int64_t)k * sk;
                    acc[j] = acc[j] + v.x * w[0];
                    acc[j] = acc[j] + v.y * w[sk];
                    acc[j] = acc[j] + v.z * w[2 * sk];
                    acc[j] = acc[j] + v.w * w[3 * sk];
                }
            }
        }
    };
    segment(x1, W1, K1, sj1, sk1);
    if (K2 > 0) segment(x2, W2, K2, sj2, sk2);
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        float v = acc[j];
        v = v + __shfl_xor(v, 1, 64);
        v = v + __shfl_xor(v, 2, 64);
        v = v + __shfl_xor(v, 4, 64);
        acc[j] = v;
    }
    if (lane8 == 0 && row < N) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            if (j < Dout) {
                float v = acc[j];
                if (bias) v = v + bias[j];
                if (act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
                out[row * Dout + j] = v;
            }
        }
    }
}

// Returns GNNMP_OK if it launched, 1 if the shape is not one this kernel takes.
static int dense_narrow_try(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2, const float *W2,
                            int64_t D2, int64_t ldw2, int w_layout, const float *bias, int act, float *out, int64_t N,
                            int64_t Dout, hipStream_t stream) {
    if (Dout > 8 || (D1 & 3) || (D2 & 3) || D1 > 4096 || D2 > 4096) return 1;
    if ((reinterpret_cast<uintptr_t>(x1) & 15) || (D2 > 0 && (reinterpret_cast<uintptr_t>(x2) & 15))) return 1;
    const int64_t sj1 = w_layout == 0 ? ldw1 : 1, sk1 = w_layout == 0 ? 1 : ldw1;
    const int64_t sj2 = w_layout == 0 ? ldw2 : 1, sk2 = w_layout == 0 ? 1 : ldw2;
    const unsigned nb = (unsigned)((N * 8 + 255) / 256);
    if (Dout <= 2)
        dense_narrow_kernel<2><<<nb, 256, 0, stream>>>(x1, W1, (int)D1, sj1, sk1, x2, W2, (int)D2, sj2, sk2, bias, act, out, N, (int)Dout);
    else if (Dout <= 4)
        dense_narrow_kernel<4><<<nb, 256, 0, stream>>>(x1, W1, (int)D1, sj1, sk1, x2, W2, (int)D2, sj2, sk2, bias, act, out, N, (int)Dout);
    else
        dense_narrow_kernel<8><<<nb, 256, 0, stream>>>(x1, W1, (int)D1, sj1, sk1, x2, W2, (int)D2, sj2, sk2, bias, act, out, N, (int)Dout);
    GNNMP_LAUNCH_CHECK("dense_narrow_kernel");
    return GNNMP_OK;
}

int dense_split_try(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2, const float *W2, int64_t D2,
                    int64_t ldw2, int w_layout, const float *bias, int act, float *out, int64_t N, int64_t Dout,
                    hipStream_t stream);   // dense_split.hip
int dense_t16_try(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2, const float *W2, int64_t D2,
                  int64_t ldw2, int w_layout, const float *bias, int act, float *out, int64_t N, int64_t Dout,
                  hipStream_t stream);   // dense_t16.hip
}  // namespace gnnmp

using namespace gnnmp;

extern "C" int gnnmp_dense_f32(const float *x1, const float *W1, int64_t D1, int64_t ldw1,
                               const float *x2, const float *W2, int64_t D2, int64_t ldw2,
                               int w_layout, const float *bias, int act, float *out, int64_t N,
                               int64_t Dout, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || Dout <= 0 || D1 <= 0 || D2 < 0 || Dout > (1 << 20) || D1 > (1 << 20) || D2 > (1 << 20))
        return fail(GNNMP_EINVAL, "dense: bad size");
    if (w_layout != 0 && w_layout != 1) return fail(GNNMP_EINVAL, "dense: bad w_layout %d", w_layout);
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "dense: bad act %d", act);
    if (N == 0) return GNNMP_OK;
    if (!x1 || !W1 || !out || (D2 > 0 && (!x2 || !W2))) return fail(GNNMP_EINVAL, "dense: null pointer");
    {
        // round 3: the split-bf16 core (three exact bf16 planes per operand, six bf16 MFMAs per product: fp32-class accuracy at
        // 2.7x the fp32-MFMA rate) for every shape whose W image fits LDS
        const int rc = dense_split_try(x1, W1, D1, ldw1, x2, W2, D2, ldw2, w_layout, bias, act, out, N, Dout, stream);
        if (rc != 1) return rc;
    }
    {
        // the shapes of the hot path (K a multiple of 4, <= 128 per segment): operands straight from HBM, 16x16x4 MFMAs
        const int rc = dense_t16_try(x1, W1, D1, ldw1, x2, W2, D2, ldw2, w_layout, bias, act, out, N, Dout, stream);
        if (rc != 1) return rc;
    }
    if (knob(KNOB_DENSE_GENERIC) == 0) {
        const int rc = dense_narrow_try(x1, W1, D1, ldw1, x2, W2, D2, ldw2, w_layout, bias, act, out, N, Dout, stream);
        if (rc != 1) return rc;
    }
    DenseArgs a;
    a.x[0] = x1; a.W[0] = W1; a.K[0] = (int)D1; a.ldw[0] = (int)ldw1;
    a.x[1] = x2; a.W[1] = W2; a.K[1] = (int)D2; a.ldw[1] = (int)ldw2;
    a.nseg = D2 > 0 ? 2 : 1;
    a.w_layout = w_layout;
    a.bias = bias;
    a.act = act;
    a.out = out;
    a.N = N;
    a.Dout = (int)Dout;
    // W-resident kernel when W^T (for one column tile) plus the wave regions fit the 160 KB LDS.  The column tile is 128
    // wide unless that leaves room for only 4 waves (K1 + K2 around 256): then 64-wide tiles — x is staged twice, but 8
    // waves (two per SIMD) overlap one wave's staging with the other's MFMAs (GraphConv 128 => 128: see LABNOTES.md).
    {
        const int k0p = ((int)D1 + 1) & ~1, k1p = ((int)D2 + 1) & ~1;
        const int ktot = k0p + (D2 > 0 ? k1p : 0);
        const int kmax = std::max(k0p, k1p);
        const size_t budget = 160 * 1024 - 64;   // 4 pipe tokens live after the wave regions
        struct Cfg { int tw, waves, ks, xld, old_, tp; size_t wbytes, region; };
        auto size_for = [&](int tw) -> Cfg {
            Cfg c{};
            c.tw = tw;
            const int full = (int)(Dout / tw), rem = (int)(Dout % tw);
            const int nt_max = full > 0 ? tw / 32 : (rem + 31) / 32;
            const int ncols_max = full > 0 ? tw : rem;
            c.wbytes = (size_t)ktot * (size_t)(nt_max * 32 + 1) * sizeof(float);
            // Wave regions: prefer 8 waves per CU.  If the whole 32 x K image does not leave room for 8 regions, stage x
            // in k-chunks (ks columns at a time); only if even 48-column chunks do not fit, fall back to 4 waves.
            for (int wv : {8, 4}) {  // fewer than one wave per SIMD cannot feed the matrix pipe: K-chunked kernel instead
                if (c.wbytes >= budget) break;
                const int cols_fit = (int)((budget - c.wbytes) / ((size_t)wv * 32 * sizeof(float)));  // floats per region row
                int kfit = ((cols_fit - 1) & ~3);               // leave the +1 (odd leading dimension)
                kfit = std::min(kfit, (kmax + 3) & ~3);
                kfit = std::min(kfit, 128);                      // <= 32 float4 per staged row (shift-mapped staging)
                if (kfit >= ((kmax + 3) & ~3) || kfit >= 48) {   // (24-column chunks measured slower than 4 waves x 44)
                    const int nkc = (kmax + kfit - 1) / kfit;     // balanced chunks, multiple of 4
                    c.ks = (((kmax + nkc - 1) / nkc) + 3) & ~3;
                    c.waves = wv;
                    break;
                }
            }
            c.xld = c.ks + 1;                                   // odd: conflict-free A-operand reads
            // the wave region is sized for the x image (>= one 32-column output tile); the output tile passes through it
            // whole if it fits, else tp column tiles at a time
            const int region_cols = (std::max(c.xld, 32) + 3) & ~3;
            c.tp = 4;
            c.old_ = (ncols_max + 3) & ~3;
            if (c.old_ > region_cols) {
                c.tp = region_cols / 32;
                c.old_ = c.tp * 32;
            }
            c.region = (size_t)32 * (size_t)region_cols;
            if (c.waves > 0 && c.wbytes + (size_t)c.waves * c.region * sizeof(float) > budget) c.waves = 0;
            return c;
        };
        Cfg c = size_for(128);
        if (c.waves < 8 && Dout >= 128) {
            const Cfg c64 = size_for(64);
            if (c64.waves == 8) c = c64;
        }
        if (c.waves > 0 && N >= 256 && knob(KNOB_DENSE_GENERIC) != 1) {
            DenseWArgs w;
            w.d = a;
            w.waves = c.waves;
            w.xld = c.xld;
            w.old_ = c.old_;
            w.tp = c.tp;
            w.ks = c.ks;
            w.skew = knob(KNOB_DENSE_PREFETCH) & 15;          // experiment knob (slot 7): low 4 bits = s_sleep(127) count,
            w.token = (knob(KNOB_DENSE_PREFETCH) >> 4) & 1;   //                           bit 4 = matrix-pipe token
            w.prefetch = ((knob(KNOB_DENSE_PREFETCH) >> 5) & 1) ^ 1;   //                  bit 5 = cross-tile prefetch OFF
            w.region = (int)c.region;
            w.ktot_pad = ktot;
            const int64_t n_row_tiles = (N + 31) / 32;
            const size_t lds_bytes = c.wbytes + (size_t)c.waves * c.region * sizeof(float) + 16;   // + 4 pipe tokens
            const int full = (int)(Dout / c.tw), rem = (int)(Dout % c.tw);
            if (full > 0) {
                w.n0 = 0;
                int rc = c.tw == 128 ? launch_wlds<4>(w, lds_bytes, full, n_row_tiles, stream)
                                     : launch_wlds<2>(w, lds_bytes, full, n_row_tiles, stream);
                if (rc) return rc;
            }
            if (rem > 0) {
                w.n0 = full * c.tw;
                const int nt = (rem + 31) / 32;
                int rc = GNNMP_OK;
                switch (nt) {
                    case 1: rc = launch_wlds<1>(w, lds_bytes, 1, n_row_tiles, stream); break;
                    case 2: rc = launch_wlds<2>(w, lds_bytes, 1, n_row_tiles, stream); break;
                    case 3: rc = launch_wlds<3>(w, lds_bytes, 1, n_row_tiles, stream); break;
                    default: rc = launch_wlds<4>(w, lds_bytes, 1, n_row_tiles, stream); break;
                }
                if (rc) return rc;
            }
            return GNNMP_OK;
        }
    }
    dim3 grid((unsigned)((N + BM - 1) / BM), (unsigned)((Dout + BN - 1) / BN));
    dense_mfma_kernel<<<grid, 256, 0, stream>>>(a);
    GNNMP_LAUNCH_CHECK("dense_mfma_kernel");
    return GNNMP_OK;
}
