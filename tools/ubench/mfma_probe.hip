// mfma_probe.hip — how busy can v_mfma_f32_16x16x4_f32 keep the matrix pipe from 1 / 2 / 3 / 4 waves per SIMD, (a) from registers
// only, (b) with the dense kernel's A-operand pattern (one ds_read_b128 per 4 MFMAs), (c) with accumulator groups of 4 / 7 / 8?
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/mfma_probe.hip -o tools/ubench/libmfma_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ void __launch_bounds__(1024) probe(float *out, int iters, int waves) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 *img = reinterpret_cast<const f32x4 *>(lds) + (lane & 15) + 128 * (lane >> 4);
    float xv = 1.0f + lane * 0.001f;
    f32x4 w[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) w[c] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int c = 0; c < NACC; ++c) w[c] = img[16 * c + ((it & 7) << 9)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][i], xv, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + tid] = s;
}

extern "C" int mfma_probe(int nacc, int use_lds, int waves, int iters, int blocks, float *out, hipStream_t stream) {
    dim3 grid(blocks), blk(64 * waves);
    size_t lds = 65536;
#define L(N, B)                                                                                      \
    {                                                                                                \
        (void)hipFuncSetAttribute((const void *)probe<N, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        probe<N, B><<<grid, blk, lds, stream>>>(out, iters, waves);                                  \
    }
    if (nacc == 4 && !use_lds) L(4, false)
    else if (nacc == 4) L(4, true)
    else if (nacc == 7 && !use_lds) L(7, false)
    else if (nacc == 7) L(7, true)
    else if (nacc == 8 && !use_lds) L(8, false)
    else L(8, true)
    return (int)hipGetLastError();
}
