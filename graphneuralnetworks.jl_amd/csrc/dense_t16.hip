// dense_t16.hip — the dense feature contractions of the layer bodies (see dense.hip for the list of reference call sites) for
// the shapes the hot path actually has: every segment's K a multiple of 4 and at most 128, any Dout that is a multiple of 4.
// One persistent block per CU (1024 threads; 768 when eight column blocks need more than 128 VGPRs); W^T lives in LDS as the
// operand image of mfma16.h; each WAVE owns 16-node tiles:
//   7-8 16-byte loads per lane straight from the row-major feature matrix into MFMA operand registers (no LDS staging of x),
//   K/4 x NCB v_mfma_f32_16x16x4_f32 with the A operands coming from the image by ds_read_b128, bias + activation on the
//   accumulators, one 16-byte store per accumulator straight to the output row (no LDS staging of the result either).
// Per 16-node tile at 100 => 100: 7 loads, 49 LDS reads, 175 MFMAs, 7 stores, ~60 VALU — against ~1 100 non-MFMA instructions
// per 200 MFMAs in the LDS-staged 32x32x2 kernel this replaces on these shapes (profiles/README.md, round 1).  Three or four
// waves per SIMD, the next tile's rows in flight during the current tile's MFMAs, a two-deep register pipeline of the A-operand
// reads; no barrier after the image.  Measured: matrix pipe 74-79 % busy at the 2.25 GHz the chip holds under this load
// (1 380 W of the 1 400 W socket limit), 89-126 TF by shape (profiles/README.md, round 2).
#include <algorithm>

#include "mfma16.h"

namespace gnnmp {

struct T16Args {
    const float *x[2];
    const float *W[2];
    int K[2];
    int64_t sj[2], sk[2];   // element strides of W(j, k)
    int nseg;
    const float *bias;
    int act;
    float *out;
    int64_t N;
    int Dout;
    int waves;              // waves per block
    int dbg;                // experiment bits: 1 = no stores, 2 = no x loads (same row for every tile)
};

// threads per block the register budget allows: four waves per SIMD (<= 128 VGPRs) up to seven column blocks; with eight, the
// accumulators (32) + two row buffers (2 x 28-32) + the two-deep A-operand buffer (32) need ~140: three waves per SIMD
template <int NCB>
constexpr int t16_max_threads() { return NCB >= 8 ? 768 : 1024; }

template <int NCB, int MAXB, int KQ1, int KQ2>
__global__ void __launch_bounds__(t16_max_threads<NCB>()) dense_t16_kernel(const T16Args a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int DP = NCB * 16;
    f32x4 *img = reinterpret_cast<f32x4 *>(lds_raw);
    const int tid = threadIdx.x, nthreads = a.waves * 64;
    const int n0 = (int)blockIdx.y * DP;
    const int ncols = min(DP, a.Dout - n0);
    const int rows0 = t16_img_rows(KQ1 >= 0 ? 4 * KQ1 : a.K[0]);
    const int rows1 = a.nseg > 1 ? t16_img_rows(KQ2 >= 0 ? 4 * KQ2 : a.K[1]) : 0;
    f32x4 *bias4 = img + (rows0 + rows1) * DP;
    t16_fill_image(img, 0, DP, a.W[0], a.sj[0], a.sk[0], a.K[0], n0, ncols, tid, nthreads);
    if (a.nseg > 1) t16_fill_image(img, rows0, DP, a.W[1], a.sj[1], a.sk[1], a.K[1], n0, ncols, tid, nthreads);
    for (int i = tid; i < DP / 4; i += nthreads) {
        f32x4 b = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * i + r < ncols) b[r] = a.bias[n0 + 4 * i + r];
        }
        bias4[i] = b;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t ntiles = (a.N + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * a.waves;
    const bool has_bias = a.bias != nullptr;
    // wave-major hand-out: round r gives tile r * stride + wave * gridDim.x + block, so the last, partial round spreads over
    // every CU (block-major put all of it on the first blocks: 48 against 36 tiles per CU on the arxiv shape)
    int64_t tile = (int64_t)wave * gridDim.x + blockIdx.x;
    if (tile >= ntiles) return;
    if constexpr (KQ1 >= 0) {
        // Compile-time K.  A software pipeline over "phases" (tile, segment): while a phase's MFMAs run, the NEXT phase's rows
        // are already in flight (same tile's second segment, or the next tile's first), so a wave never waits for HBM after
        // its first tile; rows past the end re-read the last row and are not stored.
        constexpr bool TWO = KQ2 > 0;
        float4 xa[MAXB], xb[MAXB];
        auto row_of = [&](int64_t t) { return (a.dbg & 2) ? (int64_t)n : min(t * 16 + n, a.N - 1); };
        {
            const float *xr = a.x[0] + row_of(tile) * (4 * KQ1);
            t16_load<MAXB, KQ1>(xa, q, [&](int kcol) { return *reinterpret_cast<const float4 *>(xr + kcol); });
        }
        for (; tile < ntiles; tile += stride) {
            const int64_t row = tile * 16 + n;
            const int64_t nrow = row_of(min(tile + stride, ntiles - 1));
            f32x4 acc[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (TWO) {
                const float *x2r = a.x[1] + row_of(tile) * (4 * KQ2);
                t16_load<MAXB, KQ2>(xb, q, [&](int kcol) { return *reinterpret_cast<const float4 *>(x2r + kcol); });
                t16_compute<NCB, MAXB, KQ1>(acc, img, 0, n, q, xa);
                const float *x1r = a.x[0] + nrow * (4 * KQ1);
                t16_load<MAXB, KQ1>(xa, q, [&](int kcol) { return *reinterpret_cast<const float4 *>(x1r + kcol); });
                t16_compute<NCB, MAXB, KQ2>(acc, img, rows0, n, q, xb);
            } else {
                const float *x1r = a.x[0] + nrow * (4 * KQ1);
                t16_load<MAXB, KQ1>(xb, q, [&](int kcol) { return *reinterpret_cast<const float4 *>(x1r + kcol); });
                t16_compute<NCB, MAXB, KQ1>(acc, img, 0, n, q, xa);
#pragma unroll
                for (int j = 0; j < MAXB; ++j) xa[j] = xb[j];
            }
            t16_store<NCB>(acc, bias4, has_bias, a.act, a.out + row * a.Dout + n0, row < a.N && !((a.dbg & 1) && acc[0][0] != 12345.f), ncols, q);
        }
    } else {
        for (; tile < ntiles; tile += stride) {
            const int64_t row = tile * 16 + n;
            const int64_t rl = min(row, a.N - 1);
            f32x4 acc[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            {
                const float *xr = a.x[0] + rl * a.K[0];
                t16_segment_rt<NCB>(acc, img, 0, a.K[0] >> 2, n, q,
                                    [&](int kcol) { return *reinterpret_cast<const float4 *>(xr + kcol); });
            }
            if (a.nseg > 1) {
                const float *xr = a.x[1] + rl * a.K[1];
                t16_segment_rt<NCB>(acc, img, rows0, a.K[1] >> 2, n, q,
                                    [&](int kcol) { return *reinterpret_cast<const float4 *>(xr + kcol); });
            }
            t16_store<NCB>(acc, bias4, has_bias, a.act, a.out + row * a.Dout + n0, row < a.N, ncols, q);
        }
    }
}

template <int NCB, int MAXB, int KQ1, int KQ2>
static int launch_t16(const T16Args &a0, hipStream_t stream) {
    T16Args a = a0;
    constexpr int DP = NCB * 16;
    const int rows = t16_img_rows(a.K[0]) + (a.nseg > 1 ? t16_img_rows(a.K[1]) : 0);
    const size_t lds = (size_t)rows * DP * 16 + (size_t)DP * 4;
    GNNMP_LDS_OPTIN("dense_t16_kernel", &dense_t16_kernel<NCB, MAXB, KQ1, KQ2>);
    const int cus = device_cus();
    const int64_t ntiles = (a.N + 15) / 16;
    // waves per block: 16 (four per SIMD) on large inputs; on small ones fewer, so that every CU gets a block
    constexpr int max_waves = t16_max_threads<NCB>() / 64;
    int waves = (int)std::min<int64_t>(max_waves, std::max<int64_t>(4, (ntiles + cus - 1) / cus));
    const int kw = knob(KNOB_DENSE_T16_WAVES);
    if (kw >= 1 && kw <= max_waves) waves = kw;
    a.waves = waves;
    a.dbg = knob(KNOB_T16_DEBUG);
    const int64_t gx = std::min<int64_t>(cus, (ntiles + waves - 1) / waves);
    dim3 grid((unsigned)gx, (unsigned)((a.Dout + DP - 1) / DP));
    dense_t16_kernel<NCB, MAXB, KQ1, KQ2><<<grid, 64 * waves, lds, stream>>>(a);
    GNNMP_LAUNCH_CHECK("dense_t16_kernel");
    return GNNMP_OK;
}

// Returns GNNMP_OK if it launched, 1 if the shape is not one this kernel takes (the caller falls back to dense.hip's kernels).
int dense_t16_try(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2, const float *W2, int64_t D2,
                  int64_t ldw2, int w_layout, const float *bias, int act, float *out, int64_t N, int64_t Dout,
                  hipStream_t stream) {
    if (knob(KNOB_DENSE_GENERIC) != 0) return 1;
    const int nseg = D2 > 0 ? 2 : 1;
    if ((D1 & 3) || (D2 & 3) || D1 > 128 || D2 > 128 || (Dout & 3) || Dout < 4) return 1;
    if ((reinterpret_cast<uintptr_t>(x1) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return 1;
    if (nseg > 1 && (reinterpret_cast<uintptr_t>(x2) & 15)) return 1;
    if (N < 16) return 1;
    T16Args a;
    a.x[0] = x1; a.W[0] = W1; a.K[0] = (int)D1;
    a.x[1] = x2; a.W[1] = W2; a.K[1] = (int)D2;
    a.sj[0] = w_layout == 0 ? ldw1 : 1; a.sk[0] = w_layout == 0 ? 1 : ldw1;
    a.sj[1] = w_layout == 0 ? ldw2 : 1; a.sk[1] = w_layout == 0 ? 1 : ldw2;
    a.nseg = nseg;
    a.bias = bias;
    a.act = act;
    a.out = out;
    a.N = N;
    a.Dout = (int)Dout;
    a.waves = 16;
    // column blocks of 16 per column tile: as few padded columns as the shape allows (100 -> 7 x 16 = 112, not 128)
    const int cb = (int)((Dout + 15) / 16);
    const int kq1 = (int)D1 / 4, kq2 = (int)D2 / 4;
    if (cb > 8 || cb == 8) {
        // 128-column tiles (grid.y of them): the shapes of the configs first, K known at compile time
        if (kq1 == 25 && kq2 == 0) return launch_t16<8, 7, 25, 0>(a, stream);     // GATConv dense_x 100 => 128
        if (kq1 == 32 && kq2 == 0) return launch_t16<8, 8, 32, 0>(a, stream);     // arxiv 128 => 128
        if (kq1 == 25 && kq2 == 25) return launch_t16<8, 7, 25, 25>(a, stream);   // SAGEConv 100 + 100 => 256
        if (kq1 == 32 && kq2 == 32) return launch_t16<8, 8, 32, 32>(a, stream);   // GraphConv 128 + 128 => 128
        if (kq1 == 4 && kq2 == 4) return launch_t16<8, 1, 4, 4>(a, stream);       // GraphConv 16 + 16 => 128
        return launch_t16<8, 8, -1, -1>(a, stream);
    }
    if (cb == 7) {
        if (kq1 == 25 && kq2 == 0) return launch_t16<7, 7, 25, 0>(a, stream);     // GCNConv 100 => 100
        return launch_t16<7, 8, -1, -1>(a, stream);
    }
    if (cb > 4) return launch_t16<6, 8, -1, -1>(a, stream);
    if (cb > 2) return launch_t16<4, 8, -1, -1>(a, stream);
    return launch_t16<2, 8, -1, -1>(a, stream);
}

}  // namespace gnnmp
