// dense_backward.hip — adjoints of the dense part of the layer bodies ("next"):
//   y = σ.(W * x .+ b)   =>   Δz = Δy .* σ'(z),   ΔW = Δz * x',   Δb = sum(Δz, dims = 2),   Δx = W' * Δz
// Δx is the FORWARD kernel with the weight read transposed (gnnmp_dense_f32(Δz, W, w_layout = 1)).  New here:
//   act_grad_kernel   Δz = Δy .* (y > 0)            (NNlib: relu'(x) = x > 0; y > 0 <=> x > 0)
//   colsum_*          Δb, deterministic two-stage column sums
//   dense_gradw_*     ΔW[o][k] = Σ_n Δz[n][o] * x[n][k]: a GEMM whose reduction dimension is N (millions).  Both MFMA
//                     operands of v_mfma_f32_32x32x2_f32 are then read straight from HBM in their natural row-major
//                     layout (lane l holds Δz[n + l/32][o0 + l%32] and x[n + l/32][k0 + l%32]: 128-byte segments), no LDS.
//                     Each block reduces a contiguous slab of rows into a private partial ΔW; a second kernel folds the
//                     partials in slab order (no atomics: run-to-run identical).
#include <algorithm>

#include "common.h"

namespace gnnmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) act_grad_kernel(const float *dy, const float *y, int act, float *dz,
                                                       int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = dy[i];
    dz[i] = (act == GNNMP_ACT_RELU) ? (y[i] > 0.0f ? g : 0.0f) : g;
}

// reduce_nodes' pullback (GNNlib/src/utils.jl:12-16 = scatter over the graph indicator: Δx_i = Δpool[g(i)], ./ the member graph's node
// count for mean) and the relu' of the layer that produced x, in one pass: dz[i][:] = (y[i][:] > 0) ? Δpool[g(i)][:] * inv[g(i)] : 0.
// Same operations as gnnmp_mul_rows_f32 + gnnmp_gather_f32 + gnnmp_act_grad_f32 (bit-identical), one read of y and one write of dz
// instead of three passes over (N, D).
template <int V>
__global__ void __launch_bounds__(256) pool_grad_act_kernel(const float *__restrict__ dpool, const void *__restrict__ gi, int idx_bytes, int base,
                                                            const float *__restrict__ inv, const float *__restrict__ y, int act,
                                                            float *__restrict__ dz, int64_t N, int64_t G, int D) {
    const int per = D / V;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * per) return;
    const int64_t i = t / per;
    const int c = (int)(t - i * per) * V;
    int64_t g = load_index(gi, i, idx_bytes, base);
    const bool ok = g >= 0 && g < G;                       // (an indicator outside 1..G: the row gets zeros, like a gather's bounds guard)
    if (!ok) g = 0;
    const float sc = inv ? inv[g] : 1.0f;
    float v[V], yy[V];
    Vec<V>::load(dpool + g * D + c, v);
#pragma unroll
    for (int q = 0; q < V; ++q) yy[q] = 1.0f;
    if (act == GNNMP_ACT_RELU) Vec<V>::load(y + i * D + c, yy);
#pragma unroll
    for (int q = 0; q < V; ++q) {
        float r = inv ? sc * v[q] : v[q];                   // mul_rows: a .* b with the 1-channel factor first
        r = ok ? r : 0.0f;
        v[q] = (act == GNNMP_ACT_RELU) ? (yy[q] > 0.0f ? r : 0.0f) : r;
    }
    Vec<V>::store(dz + i * D + c, v);
}

// stage 1: block b sums rows [b*R, (b+1)*R) of x[N][D] for every column -> part[b][D]
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float *x, int64_t N, int D, int64_t R,
                                                             float *part) {
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int64_t r1 = min(N, r0 + R);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.0f;
        int64_t r = r0;
        for (; r + 8 <= r1; r += 8) {          // 8 independent loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(r + u) * D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + v[u];
        }
        for (; r < r1; ++r) acc = acc + x[r * D + d];
        part[(int64_t)blockIdx.x * D + d] = acc;
    }
}
// out[g][i] = Σ_{p in group g} part[p][i], parts added in order, 8 loads in flight; blockIdx.y = group of `per` parts.
// Called twice (slabs -> FOLD_GROUPS -> 1) when there are many slabs: a serial walk over 2048 partials cost 0.5 ms.
constexpr int FOLD_GROUPS = 32;
__global__ void __launch_bounds__(256) fold_partials_kernel(const float *part, int nparts, int per, int64_t len,
                                                            float *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const int p0 = (int)blockIdx.y * per, p1 = min(nparts, p0 + per);
    float acc = 0.0f;
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(p + u) * len + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; p < p1; ++p) acc = acc + part[(int64_t)p * len + i];
    out[(int64_t)blockIdx.y * len + i] = acc;
}
// folds part[nparts][len] into out[len]; `scratch` holds FOLD_GROUPS * len floats
static int fold_partials(const float *part, int nparts, int64_t len, float *out, float *scratch, hipStream_t stream) {
    const unsigned nb = (unsigned)((len + 255) / 256);
    if (nparts <= 2 * FOLD_GROUPS) {
        fold_partials_kernel<<<dim3(nb, 1), 256, 0, stream>>>(part, nparts, nparts, len, out);
    } else {
        const int per = (nparts + FOLD_GROUPS - 1) / FOLD_GROUPS;
        const int groups = (nparts + per - 1) / per;
        fold_partials_kernel<<<dim3(nb, (unsigned)groups), 256, 0, stream>>>(part, nparts, per, len, scratch);
        fold_partials_kernel<<<dim3(nb, 1), 256, 0, stream>>>(scratch, groups, groups, len, out);
    }
    GNNMP_LAUNCH_CHECK("fold_partials_kernel");
    return GNNMP_OK;
}

struct GradWArgs {
    const float *dz;   // [N][Dout]
    const float *x;    // [N][K]
    float *part;       // [slabs][Dout][K]
    int64_t N;
    int64_t rows_per_slab;
    int Dout, K;
};

// block = 4 waves; wave w owns output row tile (blockIdx.y * 4 + w) (32 rows of ΔW = 32 columns of Δz) and the four
// column tiles blockIdx.z * 4 .. + 3 (128 columns of x).  RP row pairs are loaded before their MFMAs are issued.
template <int RP>
__global__ void __launch_bounds__(256) dense_gradw_kernel(const GradWArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int o0 = ((int)blockIdx.y * 4 + wave) * 32;          // ΔW rows (Δz columns) of this wave
    const int k0 = (int)blockIdx.z * 128;                       // ΔW columns (x columns) of this block
    if (o0 >= a.Dout) return;
    const int half = lane >> 5, li = lane & 31;
    const bool o_ok = o0 + li < a.Dout;
    bool k_ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) k_ok[t] = k0 + t * 32 + li < a.K;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int64_t n0 = (int64_t)blockIdx.x * a.rows_per_slab;
    const int64_t n1 = min(a.N, n0 + a.rows_per_slab);
    // Lanes whose ΔW row / column does not exist read a clamped (valid) column instead: their accumulator rows /
    // columns are never stored, so nothing needs masking in the main loop — and nothing may be masked there: a select on
    // the loaded value makes hipcc sink the load under a branch with its own vmcnt(0) (seen in the ISA: 20 serialised
    // round trips per batch, 3.8 ms; 4x slower than this form).
    const float *pa = a.dz + (o_ok ? o0 + li : 0) + (n0 + half) * a.Dout;
    const float *pb = a.x + k0 + (n0 + half) * a.K;
    int koff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) koff[t] = k_ok[t] ? t * 32 + li : 0;
    const int64_t sa = 2 * (int64_t)a.Dout, sb = 2 * (int64_t)a.K;   // one row pair
    int64_t n = n0;
    // Software pipeline, NB batches deep: a wave is latency-bound per batch (~1.2 us from load to MFMA) while a batch's
    // 4*RP MFMAs take 0.1-0.2 us, so NB - 1 further batches are in flight while one is consumed.  The ring lives in
    // registers: the slot index is a compile-time constant in the unrolled loop.  Batches past the end re-read the last
    // full batch (harmless, never consumed).
    constexpr int NB = 4;
    const int64_t nfull = n1 > n0 ? (n1 - n0) / (2 * RP) : 0;     // full batches of RP row pairs in this slab
    if (nfull > 0) {
        float av[NB][RP], bv[NB][RP][4];
        auto load_batch = [&](int slot, int64_t b) {
            const int64_t bc = b < nfull ? b : nfull - 1;
            const float *qa = pa + bc * RP * sa;
            const float *qb = pb + bc * RP * sb;
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                av[slot][p] = qa[p * sa];
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[slot][p][t] = qb[p * sb + koff[t]];
            }
        };
#pragma unroll
        for (int s = 0; s < NB - 1; ++s) load_batch(s, s);
        for (int64_t b0 = 0; b0 < nfull; b0 += NB) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                load_batch((s + NB - 1) % NB, b0 + s + NB - 1);
                if (b0 + s < nfull) {
#pragma unroll
                    for (int p = 0; p < RP; ++p)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][p], bv[s][p][t], acc[t], 0, 0, 0);
                }
            }
        }
        n += nfull * 2 * RP;
        pa += nfull * RP * sa;
        pb += nfull * RP * sb;
    }
    // tail of the slab: fewer than RP row pairs, the last one possibly half empty (N odd)
    for (; n < n1; n += 2) {
        const bool r_ok = n + half < n1;
        const int64_t back = r_ok ? 0 : 1;               // the missing row re-reads the previous one, weight 0
        const float va = pa[-back * (int64_t)a.Dout];
        float vb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) vb[t] = pb[-back * (int64_t)a.K + koff[t]];
        const float am = r_ok ? 1.0f : 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(va * am, vb[t] * am, acc[t], 0, 0, 0);
        pa += sa;
        pb += sb;
    }
    // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float *part = a.part + (int64_t)blockIdx.x * a.Dout * a.K;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = k0 + t * 32 + li;
        if (col >= a.K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = o0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < a.Dout) part[(int64_t)row * a.K + col] = acc[t][r];
        }
    }
}

// ---- ΔW on v_mfma_f32_16x16x4_f32 ------------------------------------------------------------------------------------------
// The 32x32x2 kernel above pads 100 x 100 to 128 x 128 (61 % of its MFMAs useful) and feeds each 64-cycle MFMA with two
// 4-byte loads per lane.  With 16 x 16 tiles 100 x 100 is 7 x 7 tiles (80 % useful), and the reduction index of one MFMA step
// is FOUR rows: lane (i, q) supplies Δz[n + q][o0 + 16 a + i] and x[n + q][k0 + 16 b + i] — both operands still straight from the
// row-major matrices in HBM (64-byte segments of four rows per load instruction), no LDS.  A wave owns up to 8 x 4 tiles
// (128 ΔW rows x 64 ΔW columns, 128 accumulator registers); the waves of a block share the Δz rows and split the columns of
// ΔW; a block reduces one slab of rows into its partial.  NB batches of four rows are in flight in a register ring.
struct GradW16Args {
    const float *dz;   // [N][Dout]
    const float *x;    // [N][K]
    float *part;       // [slabs][part_stride]: the slab's [Dout][K] partial (then, two operands: its [Dout][K2] partial)
    float *part_db;    // [slabs][db_stride] column sums of Δz (Δb), or null
    int64_t N;
    int64_t rows_per_slab;   // multiple of 4
    int Dout, K;
    int tiles_per_wave;      // k-tiles (16 columns of ΔW) per wave, <= TK
    // a SECOND operand sharing the pass over Δz (round 5: ΔW_root and ΔW_agg of graph_conv / sage_conv, conv.jl:102-108, from one read
    // of Δz): ΔW2[o][k] = Σ_n Δz[n][o] x2[n][k].  The waves of a block split the columns of the CONCATENATED [x | x2]; a wave's
    // tiles lie in one operand (the host picks tiles_per_wave so that K % (16 tiles_per_wave) == 0).  K2 = 0: one operand.
    const float *x2;   // [N][K2] or null
    int K2;
    int64_t part_stride, db_stride;
};

template <int TO, int TK, int NB, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) dense_gradw16_kernel(const GradW16Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int o0 = (int)blockIdx.y * 128;
    const int kv0 = ((int)blockIdx.z * (int)(blockDim.x >> 6) + wave) * 16 * a.tiles_per_wave;      // column of [x | x2]
    if (kv0 >= a.K + a.K2) return;
    const bool second = kv0 >= a.K;
    const float *xop = second ? a.x2 : a.x;
    const int Kop = second ? a.K2 : a.K;
    const int k0 = second ? kv0 - a.K : kv0;
    const int nto = min(TO, (a.Dout - o0 + 15) >> 4);
    const int ntk = min(a.tiles_per_wave, (Kop - k0 + 15) >> 4);
    // columns that do not exist read column 0 of the row instead (valid memory); their accumulator rows / columns are never
    // stored.  No select on the loaded values: see the note in dense_gradw_kernel.
    int ocol[TO], kcol[TK];
#pragma unroll
    for (int t = 0; t < TO; ++t) ocol[t] = (o0 + 16 * t + i < a.Dout) ? o0 + 16 * t + i : 0;
#pragma unroll
    for (int t = 0; t < TK; ++t) kcol[t] = (k0 + 16 * t + i < Kop) ? k0 + 16 * t + i : 0;
    f32x4 acc[TO][TK];
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int u = 0; u < TK; ++u) acc[t][u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // Δb rides along: the first wave of the first column block already holds every Δz value of the slab in its A operands
    const bool do_db = a.part_db != nullptr && wave == 0 && blockIdx.z == 0;
    float dbacc[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) dbacc[t] = 0.0f;
    const int64_t n0 = (int64_t)blockIdx.x * a.rows_per_slab;
    const int64_t n1 = min(a.N, n0 + a.rows_per_slab);
    if (n0 < n1) {
        const int64_t nfull = (n1 - n0) >> 2;               // full batches of four rows
        auto mfma_batch = [&](const float (&av)[TO], const float (&bv)[TK]) {
#pragma unroll
            for (int t = 0; t < TO; ++t) {
                if (t < nto) {
#pragma unroll
                    for (int u = 0; u < TK; ++u)
                        if (u < ntk) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
                }
            }
            if (do_db) {
#pragma unroll
                for (int t = 0; t < TO; ++t) dbacc[t] = dbacc[t] + av[t];
            }
        };
        if (nfull > 0) {
            float av[NB][TO], bv[NB][TK];
            // batches past the last full one re-read it (never consumed); tiles past nto / ntk read column 0 (never used)
            auto load_batch = [&](int slot, int64_t b) {
                const int64_t row = n0 + 4 * (b < nfull ? b : nfull - 1) + q;
                const float *pa = a.dz + row * a.Dout;
                const float *pb = xop + row * Kop;
#pragma unroll
                for (int t = 0; t < TO; ++t) av[slot][t] = pa[ocol[t]];
#pragma unroll
                for (int t = 0; t < TK; ++t) bv[slot][t] = pb[kcol[t]];
            };
#pragma unroll
            for (int s = 0; s < NB - 1; ++s) load_batch(s, s);
            for (int64_t b0 = 0; b0 < nfull; b0 += NB) {
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    load_batch((s + NB - 1) % NB, b0 + s + NB - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (b0 + s < nfull) mfma_batch(av[s], bv[s]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (n0 + 4 * nfull < n1) {   // the ragged last batch: rows past the end re-read the last row and are replaced by zeros
            const int64_t row = n0 + 4 * nfull + q;
            const bool in = row < n1;   // a SELECT, not a multiplication by 0: 0 * Inf = NaN would poison whole rows / columns of ΔW and Δb
                                        // when the last row holds a non-finite value; outside the pipelined loop it costs nothing
            const float *pa = a.dz + min(row, n1 - 1) * a.Dout;
            const float *pb = xop + min(row, n1 - 1) * Kop;
            float av[TO], bv[TK];
#pragma unroll
            for (int t = 0; t < TO; ++t) av[t] = in ? pa[ocol[t]] : 0.0f;
#pragma unroll
            for (int t = 0; t < TK; ++t) bv[t] = in ? pb[kcol[t]] : 0.0f;
            mfma_batch(av, bv);
        }
    }
    if (do_db) {   // rows n + q, q = 0..3, of every batch sit in four lane groups: fold them in a fixed order
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            float v = dbacc[t];
            v = v + __shfl_xor(v, 16, 64);
            v = v + __shfl_xor(v, 32, 64);
            if (q == 0 && o0 + 16 * t + i < a.Dout) a.part_db[(int64_t)blockIdx.x * a.db_stride + o0 + 16 * t + i] = v;
        }
    }
    // C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg
    float *part = a.part + (int64_t)blockIdx.x * a.part_stride + (second ? (int64_t)a.Dout * a.K : 0);
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (t >= nto) continue;
#pragma unroll
        for (int u = 0; u < TK; ++u) {
            if (u >= ntk) continue;
            const int col = k0 + 16 * u + i;
            if (col >= Kop) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = o0 + 16 * t + 4 * q + r;
                if (row < a.Dout) part[(int64_t)row * Kop + col] = acc[t][u][r];
            }
        }
    }
}

static int gradw_slabs(int64_t N) {
    const int cus = device_cus();
    // rows per slab floor.  A wave is latency-bound per batch of row pairs (~1.2 us), so a block's time is proportional to
    // its slab length: about three slabs per CU on small inputs (169 343 x 128 x 128: 166 -> 114 us with 256 rows instead
    // of 512; 20 000 rows: 67 -> 24 us with 64), never more than 256 rows (400 000 rows: 426 -> 340 us, 800 000: 651 -> 565
    // us; no difference at 2.4 M, where the slab count is capped by slabs-per-CU anyway).  knob 11 overrides.
    int min_rows = knob(KNOB_GRADW_MIN_ROWS);
    if (min_rows <= 0) {
        min_rows = 64;
        while (min_rows < 256 && (int64_t)min_rows * 3 * cus < N) min_rows <<= 1;
    }
    const int64_t by_rows = (N + min_rows - 1) / min_rows;        // at least min_rows rows per slab
    int per_cu = knob(KNOB_GRADW_SLABS);
    if (per_cu <= 0) per_cu = 4;   // with the 4-deep register pipeline 4 slabs per CU are enough (1.05 vs 1.09 ms with 8 at
                                   // 2.4M x 128 x 128; before the pipeline it took 8 to cover the load latency)
    return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)per_cu * cus, by_rows));
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_act_grad_f32(const float *dy, const float *y, int act, float *dz, int64_t n, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) return fail(GNNMP_EINVAL, "act_grad: negative n");
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "act_grad: bad act %d", act);
    if (n == 0) return GNNMP_OK;
    if (!dy || !dz || (act == GNNMP_ACT_RELU && !y)) return fail(GNNMP_EINVAL, "act_grad: null pointer");
    act_grad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(dy, y, act, dz, n);
    GNNMP_LAUNCH_CHECK("act_grad_kernel");
    return GNNMP_OK;
}

int64_t gnnmp_dense_grad_workspace(int64_t N, int64_t Dout, int64_t K) {
    if (N <= 0 || Dout <= 0 || K <= 0) return 0;
    const int64_t slabs = gradw_slabs(N);
    return (slabs + FOLD_GROUPS) * std::max(Dout * K, Dout) + slabs * Dout;   // ΔW partials + fold scratch, Δb partials
}

int gnnmp_dense_grad_w_f32(const float *dz, const float *x, int64_t N, int64_t Dout, int64_t K, float *dW,
                           float *db, float *workspace, int64_t workspace_floats, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || Dout <= 0 || K <= 0 || Dout > (1 << 16) || K > (1 << 16)) return fail(GNNMP_EINVAL, "dense_grad_w: bad size");
    if (!dW && !db) return fail(GNNMP_EINVAL, "dense_grad_w: nothing to compute");
    if (N == 0) {
        if (dW) GNNMP_HIP(hipMemsetAsync(dW, 0, sizeof(float) * Dout * K, stream));
        if (db) GNNMP_HIP(hipMemsetAsync(db, 0, sizeof(float) * Dout, stream));
        return GNNMP_OK;
    }
    if (!dz || (dW && !x) || !workspace) return fail(GNNMP_EINVAL, "dense_grad_w: null pointer");
    const int slabs = gradw_slabs(N);
    if (workspace_floats < gnnmp_dense_grad_workspace(N, Dout, K))
        return fail(GNNMP_EINVAL, "dense_grad_w: workspace too small (%lld < %lld floats)", (long long)workspace_floats,
                    (long long)gnnmp_dense_grad_workspace(N, Dout, K));
    int64_t rps = (N + slabs - 1) / slabs;
    rps = (rps + 3) & ~(int64_t)3;                               // multiple of 4: the four rows of an MFMA step never straddle slabs
    // a slab length that is a multiple of 64 rows puts every slab's stream at the same offset of the HBM channel
    // interleave when the row size is a power of two too (245 760 x 128: all blocks camp on two channels, 221 us for a
    // job that takes 118 us at 169 343 rows): de-tune it.  Slabs past the end of the input are empty (n0 >= N).
    if ((rps & 63) == 0) rps += 4;
    bool db_done = false;
    if (dW) {
        if (knob(KNOB_GRADW_RP) >= 0) {
            GradW16Args a;
            a.dz = dz;
            a.x = x;
            a.part = workspace;
            a.part_db = db ? workspace + ((int64_t)slabs + FOLD_GROUPS) * std::max(Dout * K, Dout) : nullptr;
            db_done = db != nullptr;
            a.N = N;
            a.rows_per_slab = rps;
            a.Dout = (int)Dout;
            a.K = (int)K;
            a.x2 = nullptr;
            a.K2 = 0;
            a.part_stride = Dout * K;
            a.db_stride = Dout;
            // 16 x 16 tiles of ΔW per wave: 8 x 4 (128 rows x 64 columns, 128 accumulator registers), two waves per SIMD; the
            // waves of a block share the Δz rows and split the columns.  (ONE wave per SIMD holding all of a 7 x 7 or 8 x 7-tile
            // ΔW in AGPRs — equal work for every wave — was slower: 756 vs 723 us at 100 x 100, 889 vs 807 us at 128 x 100; one
            // wave does not cover its own load latency.  Alternating which wave of a block takes the short share of a 4 + 3 column
            // split changed nothing either.)
            const int ktiles = (int)((K + 15) / 16);
            const int kwaves_all = (ktiles + 3) / 4;
            const int waves = std::min(4, kwaves_all);
            const int kblocks = (kwaves_all + waves - 1) / waves;
            a.tiles_per_wave = (ktiles + kblocks * waves - 1) / (kblocks * waves);
            dim3 grid((unsigned)slabs, (unsigned)((Dout + 127) / 128), (unsigned)kblocks);
            if ((Dout + 15) / 16 <= 7)
                dense_gradw16_kernel<7, 4, 6, 2><<<grid, 64 * waves, 0, stream>>>(a);
            else
                dense_gradw16_kernel<8, 4, 4, 2><<<grid, 64 * waves, 0, stream>>>(a);
            GNNMP_LAUNCH_CHECK("dense_gradw16_kernel");
        } else {
        GradWArgs a;
        a.dz = dz;
        a.x = x;
        a.part = workspace;
        a.N = N;
        a.rows_per_slab = rps;
        a.Dout = (int)Dout;
        a.K = (int)K;
        dim3 grid((unsigned)slabs, (unsigned)((Dout + 127) / 128), (unsigned)((K + 127) / 128));
        dense_gradw_kernel<2><<<grid, 256, 0, stream>>>(a);
        GNNMP_LAUNCH_CHECK("dense_gradw_kernel");
        }
        const int64_t len = Dout * K;
        if (int rc = fold_partials(workspace, slabs, len, dW, workspace + (int64_t)slabs * len, stream)) return rc;
    }
    if (db && db_done) {
        const float *pdb = workspace + ((int64_t)slabs + FOLD_GROUPS) * std::max(Dout * K, Dout);
        if (int rc = fold_partials(pdb, slabs, Dout, db, workspace + (int64_t)slabs * Dout * K, stream)) return rc;
    } else if (db) {
        colsum_partial_kernel<<<(unsigned)slabs, 256, 0, stream>>>(dz, N, (int)Dout, rps, workspace);
        GNNMP_LAUNCH_CHECK("colsum_partial_kernel");
        if (int rc = fold_partials(workspace, slabs, Dout, db, workspace + (int64_t)slabs * Dout, stream)) return rc;
    }
    return GNNMP_OK;
}

/* Two weight gradients and the bias gradient from ONE pass over Δz (gnnmp.h): out = [ΔW1 (Dout x K1) | ΔW2 (Dout x K2) | Δb (Dout)] */
int64_t gnnmp_dense_grad_w2_workspace(int64_t N, int64_t Dout, int64_t K1, int64_t K2) {
    if (N <= 0 || Dout <= 0 || K1 <= 0 || K2 <= 0) return 0;
    return ((int64_t)gradw_slabs(N) + FOLD_GROUPS) * (Dout * (K1 + K2) + Dout);
}

int gnnmp_dense_grad_w2_f32(const float *dz, const float *x1, int64_t K1, const float *x2, int64_t K2, int64_t N, int64_t Dout,
                            float *out, float *workspace, int64_t workspace_floats, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || Dout <= 0 || K1 <= 0 || K2 <= 0 || Dout > (1 << 16) || K1 > (1 << 16) || K2 > (1 << 16))
        return fail(GNNMP_EINVAL, "dense_grad_w2: bad size");
    if (!out) return fail(GNNMP_EINVAL, "dense_grad_w2: null output");
    const int64_t L = Dout * (K1 + K2) + Dout;
    if (N == 0) {
        GNNMP_HIP(hipMemsetAsync(out, 0, sizeof(float) * L, stream));
        return GNNMP_OK;
    }
    if (!dz || !x1 || !x2 || !workspace) return fail(GNNMP_EINVAL, "dense_grad_w2: null pointer");
    // a wave's column tiles must lie in ONE operand: K1 a multiple of 16 tiles_per_wave
    if (K1 % 16 != 0) return fail(GNNMP_EUNSUPPORTED, "dense_grad_w2: K1 = %lld is not a multiple of 16 (use gnnmp_dense_grad_w_f32 twice)", (long long)K1);
    if (workspace_floats < gnnmp_dense_grad_w2_workspace(N, Dout, K1, K2))
        return fail(GNNMP_EINVAL, "dense_grad_w2: workspace too small (%lld < %lld floats)", (long long)workspace_floats,
                    (long long)gnnmp_dense_grad_w2_workspace(N, Dout, K1, K2));
    const int slabs = gradw_slabs(N);
    int64_t rps = (N + slabs - 1) / slabs;
    rps = (rps + 3) & ~(int64_t)3;
    if ((rps & 63) == 0) rps += 4;          // (the channel-interleave de-tuning of gnnmp_dense_grad_w_f32: same slabs, same bits)
    GradW16Args a;
    a.dz = dz;
    a.x = x1;
    a.x2 = x2;
    a.K = (int)K1;
    a.K2 = (int)K2;
    a.part = workspace;
    a.part_db = workspace + Dout * (K1 + K2);
    a.part_stride = a.db_stride = L;
    a.N = N;
    a.rows_per_slab = rps;
    a.Dout = (int)Dout;
    const int kt1 = (int)(K1 / 16), kt2 = (int)((K2 + 15) / 16);
    int tpw = 4;
    while (tpw > 1 && kt1 % tpw != 0) tpw >>= 1;
    a.tiles_per_wave = tpw;
    const int kwaves_all = kt1 / tpw + (kt2 + tpw - 1) / tpw;
    const int waves = std::min(4, kwaves_all);
    const int kblocks = (kwaves_all + waves - 1) / waves;
    dim3 grid((unsigned)slabs, (unsigned)((Dout + 127) / 128), (unsigned)kblocks);
    if ((Dout + 15) / 16 <= 7)
        dense_gradw16_kernel<7, 4, 6, 2><<<grid, 64 * waves, 0, stream>>>(a);
    else
        dense_gradw16_kernel<8, 4, 4, 2><<<grid, 64 * waves, 0, stream>>>(a);
    GNNMP_LAUNCH_CHECK("dense_gradw16_kernel (two operands)");
    return fold_partials(workspace, slabs, L, out, workspace + (int64_t)slabs * L, stream);
}

/* Δx of GlobalPool(+ | mean) followed by the activation's derivative of the layer that FED the pool, in one pass (gnnmp.h) */
int gnnmp_pool_grad_act_f32(const float *dpool, const void *graph_indicator, int idx_bytes, int index_base, const float *inv_count,
                            const float *y, int act, float *dz, int64_t N, int64_t G, int64_t D, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "pool_grad_act: idx_bytes must be 4 or 8");
    if (act != GNNMP_ACT_IDENTITY && act != GNNMP_ACT_RELU) return fail(GNNMP_EINVAL, "pool_grad_act: bad act %d", act);
    if (N < 0 || G < 0 || D <= 0) return fail(GNNMP_EINVAL, "pool_grad_act: bad size");
    if (G == 0 && N > 0) return fail(GNNMP_EINVAL, "pool_grad_act: %lld rows but no graphs (the kernel reads dpool[0] for a row whose indicator is out of range)", (long long)N);
    if (N == 0) return GNNMP_OK;
    if (!dpool || !graph_indicator || !dz || (act == GNNMP_ACT_RELU && !y)) return fail(GNNMP_EINVAL, "pool_grad_act: null pointer");
    const int64_t n4 = (D % 4 == 0 && ((reinterpret_cast<uintptr_t>(dpool) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
                           ? D / 4 : 0;
    const int64_t items = n4 ? N * n4 : N * D;
    if (n4)
        pool_grad_act_kernel<4><<<(unsigned)((items + 255) / 256), 256, 0, stream>>>(dpool, graph_indicator, idx_bytes, index_base, inv_count, y, act,
                                                                                   dz, N, G, (int)D);
    else
        pool_grad_act_kernel<1><<<(unsigned)((items + 255) / 256), 256, 0, stream>>>(dpool, graph_indicator, idx_bytes, index_base, inv_count, y, act,
                                                                                   dz, N, G, (int)D);
    GNNMP_LAUNCH_CHECK("pool_grad_act_kernel");
    return GNNMP_OK;
}

}  // extern "C"
