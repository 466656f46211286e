// dense.hip — the dense feature contractions of the four layer bodies, the only MFMA work on the path:
//   weight * x                      GNNlib/src/layers/conv.jl:39,69   (gcn_conv)
//   weight1 * xi .+ weight2 * m     :106                              (graph_conv)
//   weight * vcat(xi, m)            :281                              (sage_conv; never materialises the vcat)
//   dense_x(x)                      :127                              (gat_conv)
// plus `.+ bias` and σ fused in the epilogue (:71,:107,:147,:281).
//
// Shape: tall-skinny, out[N][Dout] with N ~ 1e5..1e7 and K, Dout ~ 16..256.  fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain in k order; 157 TF peak = the fp32 vector rate, but it leaves the VALU
// free and needs one VGPR per operand).  Block = 4 waves = 128 rows x up-to-128 output columns; each wave owns 32 rows
// and up to four 32x32 accumulator tiles; x and W^T chunks of 32 k-values are staged in LDS (padded leading dimensions:
// conflict-free ds_read_b32 for both MFMA operands).
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
    dim3 grid((unsigned)((N + BM - 1) / BM), (unsigned)((Dout + BN - 1) / BN));
    dense_mfma_kernel<<<grid, 256, 0, stream>>>(a);
    GNNMP_LAUNCH_CHECK("dense_mfma_kernel");
    return GNNMP_OK;
}
