// dense_wreg.hip — the split-bf16 contraction (msplit.h) for W * vcat(xi, m) with 256 outputs: W lives in REGISTERS, x goes through LDS once.
//   W * vcat(xi, m)     sage_conv  GNNlib/src/layers/conv.jl:281   (SAGEConv(100 => 256): BASELINE.json config 4)
//   W1 * xi .+ W2 * m   graph_conv conv.jl:106                      (GraphConv(100 => 256))
// dense_split.hip keeps the three planes of W^T for a 128-column tile in LDS (160 KB at K = 208): 256 outputs are two column tiles, every
// wave reads and SPLITS each row of x twice (1.55-1.63 ms at 2.4 M x 200 => 256; the six bf16 MFMAs per product need 0.62 ms at 2.4 GHz).
// Here a block is eight waves and wave w keeps output columns 32 w .. 32 w + 31 of all three planes in its own registers for the whole
// launch — 13 k-blocks x 3 planes x 4 VGPRs = 156 of the 256 a wave may have at two waves a SIMD — as the MFMA's A operand.  A 32-row
// tile of [xi | m] is loaded ONCE per block (each wave four rows), split ONCE into its three planes and written to LDS in the B-operand
// layout of msplit.h (41 KB an image; rows padded to 33 units, see wr_unit); all eight waves then read it (3 ds_read_b128 per k-block
// for 6 MFMAs).  Two images; one block barrier per tile.  Epilogue per wave: bias, σ, its 32 columns of 32 rows through a private 4 KB
// stage as whole 128-byte lines.  Non-finite operands: NaN accumulators, the tile is redone by fp32 fma loops afterwards (msplit.h).
// Same products in the same order as dense_split_kernel: the results are bit-identical to it (tests/test_dense_split.py).
//
// Measured on MI355X at 2.4 M x (100 + 100) => 256 (tools/experiments/dense_wreg_ab.py, both kernels interleaved on one box): 1.43-1.50 ms against
// 1.55-1.63 ms.  What the time is made of (s_memtime stamps per wave and phase, round 4):
//   * a tile costs ~7900 shader cycles a SIMD where its 156 MFMAs need 5000: the ~440 other instructions a wave issues per tile (39 LDS
//     reads + waits, the split of its two units of the next tile, the epilogue, addresses) only partly fit in the gaps between MFMAs;
//   * a wave's OWN vector instructions hide between its MFMAs; a SIMD partner's do not: with one wave of a SIMD on the matrix pipe
//     (78 dependent MFMAs, 33.7 cycles each) and the other storing / splitting / loading in the meantime — two barriers a tile, the halves
//     of the block half a tile apart — the partner's ~120-instruction epilogue took 3400 cycles and the whole kernel 1.50 ms;
//   * s_setprio only swaps which half of the block waits at the barrier (older half 4400 + 1500 waiting, younger 6500, or vice versa);
//   * under this load the chip clocks at ~1.65 GHz (7900 cycles in 4.8 us): the matrix pipe alone would need 0.9 ms, not 0.62.
#include <algorithm>
#include <mutex>

#include "msplit.h"

namespace gnnmp {

struct WregArgs {
    const float *x[2];
    WCat w;
    const float *bias;
    int act;
    float *out;
    int64_t N;
    int Dout;               // 256
};

constexpr int WR_WAVES = 8, WR_THREADS = 64 * WR_WAVES, WR_DP = 32 * WR_WAVES;

// 16-byte unit (k-block kb, half hh, row n) of a plane of the tile image
// Rows of 32 units padded to 33: the loader's lanes run along r at a fixed n (33 units = 132 dwords apart: the eight lanes of a
// ds_write_b128 group cover all 32 banks), the MFMA waves' lanes along n at a fixed r (contiguous); and a k-block is a CONSTANT
// 1056 bytes further on, so that the 39 reads of a tile share one address register (an XOR by r costs one per k-block: 26 VGPRs
// that this kernel does not have).
constexpr int WR_ROW = 33;
__device__ __forceinline__ int wr_unit(int kb, int hh, int n) {
    return (2 * kb + hh) * WR_ROW + n;
}

template <int K0C, int K1C>
__global__ void __launch_bounds__(WR_THREADS) dense_wreg_kernel(const WregArgs a) {
    constexpr int NKB = (K0C + K1C + 15) / 16, KCAT = K0C + K1C;
    constexpr bool TWO = K1C > 0;
    constexpr int PLANE = NKB * 2 * WR_ROW;               // units per plane of a tile image
    constexpr int UPW = 4 * 2 * NKB;                      // units a wave loads per tile: 4 rows x 2 NKB
    constexpr int RND = (UPW + 63) / 64;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    u32x4 *bimg = reinterpret_cast<u32x4 *>(lds_raw);                                  // [2][3][PLANE]
    float4 *bias4 = reinterpret_cast<float4 *>(bimg + 2 * 3 * PLANE);                   // [WR_DP / 4]
    unsigned char *stages = reinterpret_cast<unsigned char *>(bias4 + WR_DP / 4);       // [WR_WAVES][4096]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int n0 = 32 * wave;
    split_fill_bias(bias4, WR_DP, a.bias, 0, WR_DP, tid, WR_THREADS);

    // ---- this wave's 32 columns of W, three planes, all k-blocks: resident A operands (lane (f, h): W(n0 + f, 16 kb + 8 (e >> 2) + 4 h + (e & 3)))
    u32x4 wr[NKB][3];
    {
        const bool vec = a.w.sk[0] == 1 && (a.w.sj[0] & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w.W[0]) & 15) == 0 &&
                         (!TWO || (a.w.sk[1] == 1 && (a.w.sj[1] & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w.W[1]) & 15) == 0));
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            float4 q[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = 16 * kb + 4 * h + 8 * u;
                q[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (c < KCAT) {
                    const int s = TWO && c >= K0C;
                    if (vec) q[u] = *reinterpret_cast<const float4 *>(a.w.W[s] + (int64_t)(n0 + n) * a.w.sj[s] + (c - (s ? K0C : 0)));
                    else q[u] = make_float4(wcat_at(a.w, n0 + n, c), wcat_at(a.w, n0 + n, c + 1), wcat_at(a.w, n0 + n, c + 2), wcat_at(a.w, n0 + n, c + 3));
                }
            }
            const Split8 s8 = split8(q[0], q[1]);
            wr[kb][0] = s8.p0; wr[kb][1] = s8.p1; wr[kb][2] = s8.p2;
        }
    }

    const int ntiles = (int)((a.N + 31) >> 5);
    const int nlast = (int)(a.N - 1);
    // the loader's units: wave w loads rows 4 w .. 4 w + 3 of the tile; unit u = nl * 2 NKB + (2 kb + hh), lanes run along the row
    float4 pre[2];                                          // ONE unit in flight per lane: the rounds take turns (see the tile loop)
    auto load_unit = [&](int t, int r) {
        const int u = min(r * 64 + lane, UPW - 1);          // (idle lanes of the last round repeat the last unit)
        const int nl = u / (2 * NKB), kbh = u - nl * (2 * NKB);
        const int64_t row = min(t * 32 + 4 * wave + nl, nlast);
        const int c0 = 16 * (kbh >> 1) + 4 * (kbh & 1);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int ce = min(c0 + 8 * v, KCAT - 4);      // positions past the end re-read the row's last piece: W holds zeros there
            const float *p = (TWO && ce >= K0C) ? a.x[1] + row * K1C + (ce - K0C) : a.x[0] + row * K0C + ce;
            pre[v] = *reinterpret_cast<const float4 *>(p);
        }
    };
    auto write_unit = [&](u32x4 *buf, int r) {             // (the repeated unit: same data to the same address)
        const int u = min(r * 64 + lane, UPW - 1);
        const int nl = u / (2 * NKB), kbh = u - nl * (2 * NKB);
        const Split8 s8 = split8(pre[0], pre[1]);
        const int idx = wr_unit(kbh >> 1, kbh & 1, 4 * wave + nl);
        buf[idx] = s8.p0;
        buf[PLANE + idx] = s8.p1;
        buf[2 * PLANE + idx] = s8.p2;
    };

    // One block barrier per tile.  While a wave runs tile j's 78 MFMAs it splits and writes its four rows of tile j + 1 (loaded during
    // tile j - 1) in the gaps between them — the matrix pipe takes an MFMA every 32 cycles, the wave's own vector and LDS instructions
    // fit in between (a SIMD partner's do not: measured, 25-30 cycles per instruction against a wave that keeps the pipe full) — and
    // then issues the loads of tile j + 2.  Image i lives in buffer i & 1.
    const int nt = ntiles > (int)blockIdx.x ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;   // this block's tiles
    auto tile_of = [&](int j) { return (int)blockIdx.x + min(j, nt - 1) * (int)gridDim.x; };      // (past the end: the last tile again)
    bool any_bad = false;
    if (nt > 0) {
#pragma unroll
        for (int r = 0; r < RND; ++r) {
            load_unit(tile_of(0), r);
            write_unit(bimg, r);
        }
        load_unit(tile_of(1), 0);
    }
    __syncthreads();
    for (int j = 0; j < nt; ++j) {
        const int tile = tile_of(j);
        const u32x4 *buf = bimg + (j & 1) * 3 * PLANE;
        u32x4 *nbuf = bimg + ((j + 1) & 1) * 3 * PLANE;
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            // (no operand prefetch across k-blocks: the registers are W's; the SIMD's other wave covers the LDS latency with its MFMAs)
            Split8 b;
            const int idx = wr_unit(kb, h, n);
            b.p0 = buf[idx]; b.p1 = buf[PLANE + idx]; b.p2 = buf[2 * PLANE + idx];
            SplitA av;
            av.w0 = wr[kb][0]; av.w1 = wr[kb][1]; av.w2 = wr[kb][2];
            acc[0] = split_mac(acc[0], av, b);
            // the wave's rows of image j + 1, one unit (two float4 of x) per lane at a time through the same eight registers
            // (one round of units when the concatenated K is at most 128 — 8 k-blocks, 64 units a wave — else two)
            static_assert((RND == 2 && NKB >= 9) || (RND == 1 && NKB >= 4), "one or two rounds of units per wave and tile");
            if (kb == 0) write_unit(nbuf, 0);
            if constexpr (RND == 2) {
                if (kb == 1) load_unit(tile_of(j + 1), 1);
                if (kb == NKB - 4) write_unit(nbuf, 1);
            }
            if (kb == NKB - 2) load_unit(tile_of(j + 2), 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);              // 3 ds_read, then 6 x (1 MFMA, 3 VALU), ...
#pragma unroll
            for (int mm = 0; mm < 6; ++mm) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                if (mm == 2 || mm == 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        any_bad |= split_any_nan<1>(acc);
        split_store_rows<1, true>(acc, bias4 + 8 * wave, a.act, stages + wave * 4096, a.out + n0, a.Dout, tile * 32, nlast, 32, lane);
        __syncthreads();                                    // the other buffer is complete; this one is free
    }
    if (!any_bad) return;
    // tiles whose stored rows hold a NaN (non-finite operands) are redone with plain fp32 fma loops: this wave's 32 columns
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's stores have landed
    for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
        const int row = t * 32 + n;
        const int rowc = min(row, nlast);
        float *out_row = a.out + (int64_t)rowc * a.Dout + n0;
        bool bad = false;
        for (int c = 4 * h; c < 32; c += 8) {
            const float4 v = *reinterpret_cast<const float4 *>(out_row + c);
            bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
        }
        if (__builtin_amdgcn_ballot_w64(bad) == 0) continue;
        const float *p0 = a.x[0] + (int64_t)rowc * K0C;
        const float *p1 = TWO ? a.x[1] + (int64_t)rowc * K1C - K0C : nullptr;
        const float *bias = a.bias;
        const int act = a.act;
        const bool row_ok = row <= nlast;
        split_exact_tile(1, a.w, n0, 32, h,
                         [=](int c) { return (!TWO || c < K0C) ? p0[c] : p1[c]; },
                         [=](int col, float sv) {
                             float v = sv + (bias ? bias[n0 + col] : 0.0f);
                             if (act == GNNMP_ACT_RELU) v = v < 0.0f ? 0.0f : v;
                             if (row_ok) out_row[col] = v;
                         });
    }
}

template <int K0C, int K1C>
static int launch_wreg(const WregArgs &a, hipStream_t stream) {
    constexpr int NKB = (K0C + K1C + 15) / 16;
    const size_t lds = (size_t)2 * 3 * NKB * 2 * WR_ROW * 16 + (size_t)WR_DP * 4 + (size_t)WR_WAVES * 4096;
    GNNMP_LDS_OPTIN("dense_wreg_kernel", &dense_wreg_kernel<K0C, K1C>);
    const int64_t ntiles = (a.N + 31) / 32;
    const int64_t gx = std::min<int64_t>(device_cus(), ntiles);
    dense_wreg_kernel<K0C, K1C><<<(unsigned)gx, WR_THREADS, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("dense_wreg_kernel");
    return GNNMP_OK;
}

// Returns GNNMP_OK if it launched, 1 if the shape is not one this kernel takes (dense_split_try goes on with its own kernels).
int dense_wreg_try(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2, const float *W2, int64_t D2, int64_t ldw2,
                   int w_layout, const float *bias, int act, float *out, int64_t N, int64_t Dout, hipStream_t stream) {
    if (knob(KNOB_VARIANT) & 64) return 1;                    // knob 19 bit 6: never (A/B runs)
    if (Dout != WR_DP || N < ((knob(KNOB_VARIANT) & 512) ? 4096 : 32768) || N > (int64_t)INT32_MAX - 64) return 1;      // (knob 19 bit 9: tests)      // (row numbers of a tile are 32-bit)
    const bool two = D2 > 0;
    if ((reinterpret_cast<uintptr_t>(x1) & 15) || (reinterpret_cast<uintptr_t>(out) & 127)) return 1;
    if (two && (reinterpret_cast<uintptr_t>(x2) & 15)) return 1;
    WregArgs a;
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
    // Instantiated for the layer widths of the reference's examples and benchmarks (64, 100, 128 per segment; one segment up to 256): the
    // W planes of a wave's 32 columns take 12 VGPRs per 16 positions of the concatenated K — 96 (K = 128) to 192 (K = 256) of the 256 a
    // wave has at two waves a SIMD.  K = 228 and 256 (100 + 128, 128 + 128, 256) were compiled too and spill 22-34 registers: left to
    // dense_split's LDS-resident W, like every other shape (return 1).  Measured at N = 2.4 M, => 256 (tools/experiments/dense_wreg_ab.py, one box,
    // microseconds, this kernel / dense_split): 100+100 1439 / 1578, 64+64 1006 / 1200, 64+100 1326 / 1626, 128+64 1514 / 1701, 200 1572 /
    // 1670 — and one segment of K <= 128 the other way round (64: 699 / 679, 100: 922 / 871, 128: 988 / 969: not instantiated); at
    // N = 5 000 the eight-wave blocks are too few (31 / 18): from 32 768 rows on.  Bit-identical to dense_split on every shape.
#define GNNMP_WREG_CASE(K0, K1) if (D1 == K0 && D2 == K1) return launch_wreg<K0, K1>(a, stream)
    GNNMP_WREG_CASE(100, 100);      // SAGEConv(100 => 256), GraphConv(100 => 256): BASELINE config 4
    GNNMP_WREG_CASE(64, 64);
    GNNMP_WREG_CASE(64, 100);
    GNNMP_WREG_CASE(100, 64);
    GNNMP_WREG_CASE(64, 128);
    GNNMP_WREG_CASE(128, 64);
    GNNMP_WREG_CASE(200, 0);        // one segment: only where x is split twice by dense_split's two column passes AND K is large
#undef GNNMP_WREG_CASE
    return 1;
}

}  // namespace gnnmp
