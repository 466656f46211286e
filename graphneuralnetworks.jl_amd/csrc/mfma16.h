// mfma16.h — the fp32 contraction core shared by the dense kernel (dense_t16.hip) and the fused aggregate-then-transform
// kernels (fused_conv.hip): out^T = W · x^T on v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 32-cycle issue).
//
// The product is formed TRANSPOSED: the A operand is a 16-feature block of W, the B operand 16 nodes of x.  Two things
// follow from that choice.
//   * C/D layout (col = lane & 15, row = 4 (lane >> 4) + reg): lane (n, q) ends up with output features 16c + 4q .. + 3 of
//     node n — FOUR CONSECUTIVE FLOATS OF ONE OUTPUT ROW: the epilogue is one 16-byte global store per accumulator, no LDS
//     round trip to re-layout the tile.
//   * B layout (lane (n, q) supplies x[n][k-slot q]): the contraction index may be permuted freely (a sum over k), so k-slot q
//     of MFMA step (j, i) is defined as k = 16 j + 4 q + i.  Lane (n, q) then needs x[n][16 j + 4 q .. + 3] for the four steps
//     of block j: ONE 16-byte load straight from the row-major feature matrix in HBM into the operand register — no LDS
//     staging of x at all.  W is laid out to match, once per block, as the LDS image img[kq][f] = float4(W[f][4 kq .. + 3])
//     (kq = 4 j + q), so that the A operands of the same four steps are one conflict-free ds_read_b128.
// The summation order over k therefore differs from a k-ordered loop (and from BLAS): tolerance, not bits, like every dense
// product of this library.
//
// K of a segment must be a multiple of 4 (16-byte rows); kq = K / 4 float4 per row, nfull = kq / 4 full blocks and a tail of
// rem = kq % 4 float4.  rem = 1 (K = 100): every lane loads the row's last float4 and lane (n, q) contributes component q —
// one MFMA step instead of four.  rem = 2, 3: a block whose lanes q >= rem supply explicit zeros.
#pragma once
#include "common.h"

namespace gnnmp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// rows of the W image a segment of K floats occupies (multiple of 4: a block never reads another segment's rows)
__host__ __device__ inline int t16_img_rows(int K) { return ((K / 4) + 3) & ~3; }

// Fill the W image of one segment: img[(kq0 + kq) * DP + f] = (W(n0 + f, 4 kq + i))_i for f < ncols, kq < K / 4; everything else
// (padding rows, padding columns) zero.  W(j, k) at W[j * sj + k * sk].  All threads of the block call this; caller barriers.
__device__ __forceinline__ void t16_fill_image(f32x4 *img, int kq0, int DP, const float *__restrict__ W, int64_t sj, int64_t sk,
                                               int K, int n0, int ncols, int tid, int nthreads) {
    const int kqn = K >> 2, rows = t16_img_rows(K);
    const bool vec = sk == 1 && (sj & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    for (int idx = tid; idx < rows * DP; idx += nthreads) {
        // consecutive threads take consecutive features: the LDS writes are contiguous (conflict-free); the global reads are
        // 16 bytes from 64 different rows of W — uncoalesced, but W is a few tens of KB and L2-resident after the first block
        const int kq = idx / DP, f = idx - kq * DP;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (f < ncols && kq < kqn) {
            const float *p = W + (int64_t)(n0 + f) * sj + (int64_t)(4 * kq) * sk;
            if (vec) {
                const float4 t = *reinterpret_cast<const float4 *>(p);
                v = f32x4{t.x, t.y, t.z, t.w};
            } else {
                v = f32x4{p[0], p[sk], p[2 * sk], p[3 * sk]};
            }
        }
        img[(kq0 + kq) * DP + f] = v;
    }
}

// the four MFMA steps of one full block j for every column block, in groups of CG = 4 column blocks: the group's A operands
// (one ds_read_b128 per column block) are read first, then the steps run i-major so that consecutive MFMAs never share an
// accumulator (a dependent 16x16x4 pair costs 40 cycles instead of 32).  Four column blocks at a time keep the operand
// registers at 16 (all NCB at once: 32 — with two row buffers and the accumulators that spilled).
template <int NCB>
__device__ __forceinline__ void t16_block(f32x4 (&acc)[NCB], const f32x4 *__restrict__ wr, const float4 xv) {
    constexpr int CG = 4;
#pragma unroll
    for (int c0 = 0; c0 < NCB; c0 += CG) {
        f32x4 w[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (c0 + c < NCB) w[c] = wr[16 * (c0 + c)];
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][0], xv.x, acc[c0 + c]);
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][1], xv.y, acc[c0 + c]);
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][2], xv.z, acc[c0 + c]);
#pragma unroll
        for (int c = 0; c < CG; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][3], xv.w, acc[c0 + c]);
    }
}
// the single MFMA step of a rem = 1 tail: lane (n, q) contributes component q of the row's last float4
template <int NCB>
__device__ __forceinline__ void t16_tail1(f32x4 (&acc)[NCB], const f32x4 *__restrict__ img_row, int n, int q, const float4 xv) {
    const float xs = q == 0 ? xv.x : (q == 1 ? xv.y : (q == 2 ? xv.z : xv.w));
    const float *wr = reinterpret_cast<const float *>(img_row + n) + q;
    float w[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) w[c] = wr[64 * c];
#pragma unroll
    for (int c = 0; c < NCB; ++c) acc[c] = mfma16(w[c], xs, acc[c]);
}

// Compile-time K (KQ = K / 4): put one node row's operands in flight — MAXB 16-byte loads.
//   xload(kcol) -> float4: this lane's node's x[kcol .. kcol + 3] (from HBM for the dense kernel, from the wave's LDS tile for
//   the fused kernels).  q = lane >> 4.
template <int MAXB, int KQ, class XLoad>
__device__ __forceinline__ void t16_load(float4 (&xv)[MAXB], int q, XLoad xload) {
    constexpr int nfull = KQ >> 2, rem = KQ & 3;
    static_assert(MAXB >= nfull + (rem ? 1 : 0), "MAXB too small for this K");
#pragma unroll
    for (int j = 0; j < MAXB; ++j) {
        if (j < nfull) {
            xv[j] = xload(16 * j + 4 * q);
        } else if (j == nfull && rem == 1) {
            xv[j] = xload(16 * j);                       // the row's last float4, the same for the four lanes of a node
        } else if (j == nfull && rem > 1) {
            xv[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (q < rem) xv[j] = xload(16 * j + 4 * q);  // lanes past the end of the row keep explicit zeros
        }
    }
}
// ... and run the segment's blocks on them, as a software pipeline over GROUPS of up to four column blocks: while the MFMAs of
// group S run (16 instructions, 512 cycles of the matrix pipe), the A operands of group S + 1 — the next column blocks of this
// k-block, or the first ones of the next k-block — are already being read from LDS into the other half of a two-deep
// register buffer.  Left to itself hipcc issues each group's ds_reads immediately before the MFMAs that need them: every
// k-block then starts with an exposed LDS round trip (PMC: matrix pipe 82 % busy with NO memory traffic at all).  The
// sched_barriers pin the order (and keep hipcc from hoisting ALL reads to the top: 49 x 4 registers, 100-190 spills).
template <int NCB, int KQ>
struct T16Steps {
    static constexpr int nfull = KQ >> 2, rem = KQ & 3;
    static constexpr int NG = (NCB + 3) / 4;                       // groups per k-block
    static constexpr int nblocks = nfull + (rem ? 1 : 0);
    static constexpr int total = nblocks * NG;
};
// read group S's A operands: full blocks (and rem > 1 tails) one ds_read_b128 per column block; a rem = 1 tail one scalar
template <int NCB, int KQ, int S>
__device__ __forceinline__ void t16_group_read(f32x4 (&w)[4], const f32x4 *__restrict__ img, int kq0, int n, int q) {
    using St = T16Steps<NCB, KQ>;
    constexpr int DP = NCB * 16;
    constexpr int j = S / St::NG, c0 = (S % St::NG) * 4;
    if constexpr (j < St::nfull || St::rem > 1) {
        const f32x4 *wr = img + (kq0 + 4 * j + q) * DP + n + 16 * c0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < NCB) w[c] = wr[16 * c];
    } else {
        const float *wr = reinterpret_cast<const float *>(img + (kq0 + 4 * j) * DP + n + 16 * c0) + q;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < NCB) w[c][0] = wr[64 * c];
    }
}
template <int NCB, int KQ, int S, int MAXB>
__device__ __forceinline__ void t16_group_mfma(f32x4 (&acc)[NCB], const f32x4 (&w)[4], const float4 (&xv)[MAXB], int q) {
    using St = T16Steps<NCB, KQ>;
    constexpr int j = S / St::NG, c0 = (S % St::NG) * 4;
    if constexpr (j < St::nfull || St::rem > 1) {
        // i-major: consecutive MFMAs never share an accumulator (a dependent 16x16x4 pair costs 40 cycles instead of 32)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][0], xv[j].x, acc[c0 + c]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][1], xv[j].y, acc[c0 + c]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][2], xv[j].z, acc[c0 + c]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][3], xv[j].w, acc[c0 + c]);
    } else {
        const float xs = q == 0 ? xv[j].x : (q == 1 ? xv[j].y : (q == 2 ? xv[j].z : xv[j].w));
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < NCB) acc[c0 + c] = mfma16(w[c][0], xs, acc[c0 + c]);
    }
}
template <int NCB, int MAXB, int KQ, int S>
__device__ __forceinline__ void t16_pipeline(f32x4 (&acc)[NCB], const f32x4 *__restrict__ img, int kq0, int n, int q,
                                             const float4 (&xv)[MAXB], f32x4 (&wa)[4], f32x4 (&wb)[4]) {
    using St = T16Steps<NCB, KQ>;
    if constexpr (S < St::total) {
        // wa holds group S; fetch group S + 1 into wb, then compute S from wa; roles swap in the next step
        if constexpr (S + 1 < St::total) t16_group_read<NCB, KQ, S + 1>(wb, img, kq0, n, q);
        __builtin_amdgcn_sched_barrier(0);
        t16_group_mfma<NCB, KQ, S, MAXB>(acc, wa, xv, q);
        __builtin_amdgcn_sched_barrier(0);
        t16_pipeline<NCB, MAXB, KQ, S + 1>(acc, img, kq0, n, q, xv, wb, wa);
    }
}
template <int NCB, int MAXB, int KQ>
__device__ __forceinline__ void t16_compute(f32x4 (&acc)[NCB], const f32x4 *__restrict__ img, int kq0, int n, int q,
                                            const float4 (&xv)[MAXB]) {
    f32x4 wa[4], wb[4];
    t16_group_read<NCB, KQ, 0>(wa, img, kq0, n, q);
    t16_pipeline<NCB, MAXB, KQ, 0>(acc, img, kq0, n, q, xv, wa, wb);
}

// Run-time K: a loop over the blocks with the next block's load in flight during the current block's MFMAs.
template <int NCB, class XLoad>
__device__ __forceinline__ void t16_segment_rt(f32x4 (&acc)[NCB], const f32x4 *__restrict__ img, int kq0, int kq, int n, int q,
                                               XLoad xload) {
    constexpr int DP = NCB * 16;
    const int nfull = kq >> 2, rem = kq & 3;
    const int nb = nfull + (rem ? 1 : 0);
    if (nb == 0) return;
    // block b's operand: full blocks and rem > 1 tails load column 16 b + 4 q (zeros past the row's end), a rem = 1 tail the
    // row's last float4 in every lane
    auto fetch = [&](int b) -> float4 {
        if (b < nfull) return xload(16 * b + 4 * q);
        if (rem == 1) return xload(16 * b);
        return q < rem ? xload(16 * b + 4 * q) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    float4 cur = fetch(0);
    for (int b = 0; b < nb; ++b) {
        const float4 nxt = fetch(min(b + 1, nb - 1));
        if (b < nfull || rem > 1)
            t16_block<NCB>(acc, img + (kq0 + 4 * b + q) * DP + n, cur);
        else
            t16_tail1<NCB>(acc, img + (kq0 + 4 * b) * DP, n, q, cur);
        cur = nxt;
    }
}

// Epilogue of one 16-node tile: + bias, activation, one 16-byte store per accumulator.  bias4: LDS image of the bias, f32x4
// per 4 features (zero past Dout).  ncols = valid columns of this column tile; requires Dout % 4 == 0 and a 16-byte aligned out.
template <int NCB>
__device__ __forceinline__ void t16_store(const f32x4 (&acc)[NCB], const f32x4 *__restrict__ bias4, bool has_bias, int act,
                                          float *__restrict__ out_row, bool row_ok, int ncols, int q) {
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        f32x4 v = acc[c];
        if (has_bias) v = v + bias4[4 * c + q];
        if (act == GNNMP_ACT_RELU) {   // NNlib.relu = ifelse(x < 0, 0, x) (NaN-preserving)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.0f ? 0.0f : v[r];
        }
        const int col = 16 * c + 4 * q;
        if (row_ok && col < ncols) *reinterpret_cast<float4 *>(out_row + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace gnnmp
