// msplit.h — the fp32 contraction core of round 3: out^T = W · x^T on v_mfma_f32_32x32x16_bf16 with both operands split
// EXACTLY into three bf16 planes, six plane products per k-block, fp32 accumulation inside the matrix core.
//
// Why.  gfx950 has no TF32-class path: its f32-input MFMA (v_mfma_f32_16x16x4_f32, mfma16.h) runs at the fp32 VECTOR rate,
// 1/16 of the bf16 rate, and every dense product of the layer bodies (weight * x, W1*xi .+ W2*m, W*vcat(xi, m), dense_x:
// GNNlib/src/layers/conv.jl:36-40,70,106,136,281) was bound by it (profiles/README.md, rounds 1-2: 74-79 % of a 157 TF pipe).
// An fp32 number is exactly the sum of three bf16 numbers (8 + 8 + 8 significant bits):
//     x = x0 + x1 + x2,   x0 = trunc_bf16(x),  x1 = trunc_bf16(x - x0),  x2 = x - x0 - x1   (both subtractions exact)
// so  x*w = sum_{a,b} xa*wb  exactly, and every plane product xa*wb is exact in the matrix core (8 x 8 bits).  Keeping the
// six terms with a + b <= 2 drops x1*w2 + x2*w1 + x2*w2 < 2^-21 |x w| — the size of ONE fp32 rounding — so the result is in
// the error class of an fp32 fma chain (measured against float64: tests/test_dense_split.py; the chain itself is
// 0.75-1.5e-7 * sum|a b|, cdna_hip_programming.md §3), at 6/16 of the fp32-MFMA time.  Like every dense product of this
// library it is a tolerance op, not a bit-exact one: BLAS on the reference side has its own order.
//
// Non-finite inputs.  The split of +-Inf has x1 = Inf - Inf = NaN, so any Inf / NaN operand turns the whole output row into
// NaN — never into a wrong finite number.  The kernels test their accumulators for NaN (one ballot per tile) and recompute
// such tiles with plain fp32 fma loops (cold path): Inf * w = +-Inf, the -Inf of an empty max-aggregation, NaN propagation
// all behave like the reference's product.
//
// Layout (transposed product, like mfma16.h: the epilogue is 16-byte stores of whole output-row pieces, the x operand is
// 16-byte loads of whole input-row pieces, nothing is staged through LDS except W):
//   wave tile = 32 nodes; lane (n = lane & 31, h = lane >> 5).
//   k-block kb = 16 positions of the CONCATENATED contraction index c (segment 1 then segment 2, each a multiple of 4 long —
//   `vcat(xi, m)` is never built): lane (n, h) supplies element e = 0..7 <-> c = 16 kb + 8 (e >> 2) + 4 h + (e & 3) — two 16-byte
//   loads of row n at c = 16 kb + 4 h and + 8: ONE load instruction reads 32 contiguous bytes of each row (lanes h = 0, 1), i.e.
//   every 32-byte sector is requested by exactly one instruction.
//   A operand: lane (f, h) supplies W(n0 + 32 cb + f, c) for the same c: one ds_read_b128 per plane from the LDS image
//       img[plane][(2 kb + h) * DP + col]  (16-byte units of 8 bf16 in the same element order; consecutive lanes -> consecutive
//       units: conflict-free).
//   C/D (32x32): lane (n, h), register r = 4 a + b holds feature 32 cb + 8 a + 4 h + b of node n: four consecutive floats of
//   one output row per a — one 16-byte store.
//   The hardware's own k <-> (h, e) assignment does not matter: A and B use the same one, and a sum over k is permutation-free.
#pragma once
#include "common.h"

namespace gnnmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the three planes of eight values, packed as the MFMA wants them (element e in half e & 1 of dword e >> 1)
struct Split8 {
    u32x4 p0, p1, p2;
};

// bits of the three planes of one value: the plane is the TOP 16 bits of each word
__device__ __forceinline__ void split3(const float v, uint32_t &b0, uint32_t &b1, uint32_t &b2) {
    const uint32_t u = __float_as_uint(v);
    const float r = v - __uint_as_float(u & 0xffff0000u);     // exact: the low 16 mantissa bits of v
    const uint32_t ur = __float_as_uint(r);
    const float l = r - __uint_as_float(ur & 0xffff0000u);    // exact: at most 8 significant bits left
    b0 = u;
    b1 = ur;
    b2 = __float_as_uint(l);
}
// the high halves of two words in one: (b & 0xffff0000) | (a >> 16) — one v_perm_b32 (bytes 0-3 = second operand)
__device__ __forceinline__ uint32_t pack_hi(const uint32_t a, const uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ Split8 split8(const float4 lo, const float4 hi) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t b0[8], b1[8], b2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(v[e], b0[e], b1[e], b2[e]);
    Split8 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s.p0[i] = pack_hi(b0[2 * i], b0[2 * i + 1]);
        s.p1[i] = pack_hi(b1[2 * i], b1[2 * i + 1]);
        s.p2[i] = pack_hi(b2[2 * i], b2[2 * i + 1]);
    }
    return s;
}

// ---- W side -------------------------------------------------------------------------------------------------------------
// the (up to) two weight blocks of a contraction: W(j, c) = c < K[0] ? W[0][j * sj[0] + c * sk[0]] : W[1][j * sj[1] + (c - K[0]) * sk[1]]
struct WCat {
    const float *W[2];
    int64_t sj[2], sk[2];
    int K[2];
};
__host__ __device__ inline int split_nkb(int kcat) { return (kcat + 15) >> 4; }
// bytes of the three-plane image of a DP-column tile
__host__ __device__ inline size_t split_img_bytes(int kcat, int DP) { return (size_t)3 * split_nkb(kcat) * 2 * DP * 16; }

__device__ __forceinline__ float wcat_at(const WCat &w, int j, int c) {
    const int s = c >= w.K[0];
    return w.W[s][(int64_t)j * w.sj[s] + (int64_t)(c - (s ? w.K[0] : 0)) * w.sk[s]];
}

// Build the image of columns n0 .. n0 + ncols - 1 (zero beyond, and for c >= K[0] + K[1]).  All threads of the block; the caller
// barriers.  Consecutive threads take consecutive columns: contiguous 16-byte LDS writes; the global reads are 32 bytes of 64
// different rows of W — uncoalesced, but W is at most a few hundred KB and L2-resident after the first block.
__device__ __forceinline__ void split_fill_image(u32x4 *img, int nkb, int DP, const WCat &w, int n0, int ncols, int tid, int nthreads) {
    const int kcat = w.K[0] + w.K[1];
    const int units = nkb * 2 * DP;
    const bool vec = w.sk[0] == 1 && (w.sj[0] & 3) == 0 && (reinterpret_cast<uintptr_t>(w.W[0]) & 15) == 0 &&
                     (w.K[1] == 0 || (w.sk[1] == 1 && (w.sj[1] & 3) == 0 && (reinterpret_cast<uintptr_t>(w.W[1]) & 15) == 0));
    // four units per thread and round: their eight 16-byte loads are in flight together (one unit at a time made the fill a chain of
    // ~12 dependent L2 round trips per block: ~10 us of every launch of the chain kernels)
    constexpr int UB = 4;
    for (int idx0 = tid; idx0 < units; idx0 += UB * nthreads) {
        float4 q[UB][2];
#pragma unroll
        for (int b = 0; b < UB; ++b) {
            const int idx = idx0 + b * nthreads;
            const int col = idx % DP, kbh = idx / DP;
            const int c0 = (kbh >> 1) * 16 + (kbh & 1) * 4;         // 16 kb + 4 h; the second float4 sits 8 positions on
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = c0 + 8 * u;
                q[b][u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (idx < units && col < ncols && c < kcat) {        // K[0], K[1] multiples of 4: a float4 never straddles
                    if (vec) {
                        const int s = c >= w.K[0];
                        q[b][u] = *reinterpret_cast<const float4 *>(w.W[s] + (int64_t)(n0 + col) * w.sj[s] + (c - (s ? w.K[0] : 0)));
                    } else {
                        q[b][u] = make_float4(wcat_at(w, n0 + col, c), wcat_at(w, n0 + col, c + 1), wcat_at(w, n0 + col, c + 2),
                                              wcat_at(w, n0 + col, c + 3));
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < UB; ++b) {
            const int idx = idx0 + b * nthreads;
            if (idx < units) {
                const Split8 s = split8(q[b][0], q[b][1]);
                img[idx] = s.p0;
                img[units + idx] = s.p1;
                img[2 * units + idx] = s.p2;
            }
        }
    }
}

// the bias of the tile's columns as f32x4 per 4 features (zero beyond ncols)
__device__ __forceinline__ void split_fill_bias(float4 *bias4, int DP, const float *bias, int n0, int ncols, int tid, int nthreads) {
    for (int i = tid; i < DP / 4; i += nthreads) {
        float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * i + r < ncols) b[r] = bias[n0 + 4 * i + r];
        }
        bias4[i] = make_float4(b[0], b[1], b[2], b[3]);
    }
}

// ---- one k-block against NCB column blocks ----------------------------------------------------------------------------------
// wa: this lane's A-operand base for k-block kb: img + (2 kb + h) * DP + n;  plane stride `units`.
// A-operand reads run one column block ahead of the MFMAs in a two-deep register buffer (wa0/wa1): left alone, hipcc puts each
// ds_read right before its consumer and every block starts with an exposed LDS round trip (mfma16.h has the same note).
struct SplitA {
    u32x4 w0, w1, w2;
};
__device__ __forceinline__ SplitA split_read_a(const u32x4 *__restrict__ p, int units) {
    SplitA a;
    a.w0 = p[0];
    a.w1 = p[units];
    a.w2 = p[2 * units];
    return a;
}
// the six plane products of one (k-block, column block): smallest terms first
__device__ __forceinline__ f32x16 split_mac(f32x16 acc, const SplitA &a, const Split8 &b) {
    acc = mfma32(a.w2, b.p0, acc);
    acc = mfma32(a.w0, b.p2, acc);
    acc = mfma32(a.w1, b.p1, acc);
    acc = mfma32(a.w1, b.p0, acc);
    acc = mfma32(a.w0, b.p1, acc);
    acc = mfma32(a.w0, b.p0, acc);
    return acc;
}

// all NCB column blocks of k-block kb; `cur` holds column block 0's operands on entry, and on exit column block 0's operands of
// k-block `kb_next` (pass kb_next = kb on the last call: a harmless re-read)
template <int NCB>
__device__ __forceinline__ void split_kblock(f32x16 (&acc)[NCB], const u32x4 *__restrict__ img, int units, int DP, int kb,
                                             int kb_next, int n, int h, const Split8 &b, SplitA &cur) {
    const u32x4 *row = img + (2 * kb + h) * DP + n;
    const u32x4 *row_next = img + (2 * kb_next + h) * DP + n;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const SplitA nxt = split_read_a(cb + 1 < NCB ? row + 32 * (cb + 1) : row_next, units);
        __builtin_amdgcn_sched_barrier(0);
        acc[cb] = split_mac(acc[cb], cur, b);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
}

// ---- epilogue ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 split_out4(const f32x16 &acc, int a, const float4 bias, int act) {
    float v[4] = {acc[4 * a] + bias.x, acc[4 * a + 1] + bias.y, acc[4 * a + 2] + bias.z, acc[4 * a + 3] + bias.w};
    if (act == GNNMP_ACT_RELU) {   // NNlib.relu = ifelse(x < 0, 0, x) (NaN-preserving)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.0f ? 0.0f : v[r];
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
// any NaN among this wave's accumulators?  (wave-uniform)
template <int NCB>
__device__ __forceinline__ bool split_any_nan(const f32x16 (&acc)[NCB]) {
    bool bad = false;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        float s = 0.0f;   // a NaN anywhere makes the sum NaN; Inf - Inf does too, which only sends a tile to the exact path
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[cb][r];
        bad |= (s != s);
    }
    return __builtin_amdgcn_ballot_w64(bad) != 0;
}
// + bias, activation, one 16-byte store per four features.  bias4: LDS image (split_fill_bias).  Dout % 4 == 0, out 16-byte aligned.
template <int NCB>
__device__ __forceinline__ void split_store(const f32x16 (&acc)[NCB], const float4 *__restrict__ bias4, int act,
                                            float *__restrict__ out_row, bool row_ok, int ncols, int h) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = 32 * cb + 8 * a + 4 * h;
            const float4 v = split_out4(acc[cb], a, bias4[col >> 2], act);
            if (row_ok && col < ncols) *reinterpret_cast<float4 *>(out_row + col) = v;
        }
    }
}

// The same with NO predicated store (hipcc wraps every predicated store in an exec-skip branch, and its s_waitcnt insertion then
// cannot count that store as younger than a pending operand load: with 16 conditional stores per tile every tile began by waiting
// for the previous tile's stores to complete — measured 240 us of 610 at 2.4 M x 100 => 128).  The caller passes the row pointer
// of min(row, N - 1): lanes past the last row hold that row's operands, hence bit-identical results, and store them again to the
// same addresses.  FULLCOLS = false: pieces past ncols re-store the lane's first piece (columns 4 h .. 4 h + 3, always valid).
template <int NCB, bool FULLCOLS>
__device__ __forceinline__ void split_store_all(const f32x16 (&acc)[NCB], const float4 *__restrict__ bias4, int act,
                                                float *__restrict__ out_row, int ncols, int h) {
    const float4 first = split_out4(acc[0], 0, bias4[h], act);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int col = 32 * cb + 8 * a + 4 * h;
            float4 v = split_out4(acc[cb], a, bias4[col >> 2], act);
            int colv = col;
            if (!FULLCOLS) {
                const bool ok = col < ncols;
                v = make_float4(ok ? v.x : first.x, ok ? v.y : first.y, ok ? v.z : first.z, ok ? v.w : first.w);
                colv = ok ? col : 4 * h;
            }
            *reinterpret_cast<float4 *>(out_row + colv) = v;
        }
    }
}

// The same through a wave-private 4 KB LDS stage, one 32-column block at a time: the accumulator layout hands a store instruction 32
// rows x 32 bytes (a quarter of a 128-byte line per row: four instructions complete a line), the stage turns that into 8 rows x 128
// contiguous bytes per instruction — whole lines, a quarter of the write requests.  Measured at 2.4 M x 100 => 128: the stores were
// 290 us of a 595 us launch (the same kernel without them: 306 us).  Stage row n = 128 bytes, 16-byte piece 2 a + h XOR-swizzled by
// the row; rows past the end store row N - 1's identical values again (no predicated store, see above); FULLCOLS = false: pieces past
// ncols store the block's last valid piece again, blocks past ncols are skipped (uniform).
__device__ __forceinline__ void split_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int NCB, bool FULLCOLS>
__device__ __forceinline__ void split_store_rows(const f32x16 (&acc)[NCB], const float4 *__restrict__ bias4, int act,
                                                 unsigned char *stg, float *__restrict__ out, int64_t ldo, int row0, int nlast,
                                                 int ncols, int lane) {
    const int n = lane & 31, h = lane >> 5;
    const int rr = lane >> 3, q = lane & 7;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (!FULLCOLS && 32 * cb >= ncols) break;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 v = split_out4(acc[cb], a, bias4[(32 * cb + 8 * a + 4 * h) >> 2], act);
            *reinterpret_cast<float4 *>(stg + n * 128 + (((2 * a + h) ^ (n & 7)) << 4)) = v;
        }
        split_wave_sync();
        const int qc = FULLCOLS ? q : min(q, ((ncols - 32 * cb) >> 2) - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rr + 8 * j;
            const float4 v = *reinterpret_cast<const float4 *>(stg + r * 128 + ((qc ^ (r & 7)) << 4));
            *reinterpret_cast<float4 *>(out + (int64_t)min(row0 + r, nlast) * ldo + 32 * cb + 4 * qc) = v;
        }
        split_wave_sync();      // the next column block overwrites the stage
    }
}

// The exact path of a tile that met a non-finite operand: lane (n, h) recomputes its own outputs with fp32 fma loops in c order.
// xat(c) -> this lane's node's operand element c (fp32); emit(col, value) takes the raw contraction of tile column `col`
// (bias / activation / store are the caller's).  Cold code, out of line, no attempt at speed; operands by value so that nothing of
// the hot path has to live in memory for it.
template <class XAt, class Emit>
__device__ __noinline__ void split_exact_tile(int ncb, const WCat w, int n0, int ncols, int h, XAt xat, Emit emit) {
    const int kcat = w.K[0] + w.K[1];
#pragma unroll 1
    for (int i = 0; i < 16 * ncb; ++i) {
        const int cb = i >> 4, r = i & 15;
        const int col = 32 * cb + 8 * (r >> 2) + 4 * h + (r & 3);
        if (col >= ncols) continue;
        float s = 0.0f;
#pragma unroll 1
        for (int c = 0; c < kcat; ++c) s = fmaf(wcat_at(w, n0 + col, c), xat(c), s);
        emit(col, s);
    }
}

}  // namespace gnnmp
